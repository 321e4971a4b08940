/*
 * md_geom.h -- scalar float32 building blocks of the step() path: ray/shape intersection, OBB
 * overlap tests, lane Frenet transforms, kinematic bicycle sub-step, IDM/PID formulas.
 *
 * Same role as md_math.h: ONE spelling of each formula, compiled both into the HIP kernels and
 * into the gcc-built CPU oracle (oracle/md_oracle.c) with -ffp-contract=off, so that the two agree
 * to the last bit.  What differs between oracle and kernels is everything around these formulas:
 * the oracle is plain brute-force loops; the kernels parallelise over waves/lanes, stage tables
 * in LDS, cull through the static grid and angular sectors, and fuse phases.  The formulas are
 * pinned to the reference by tests/golden (generated from the reference's own Python).
 *
 * Each function cites the reference code it restates (paths relative to /root/reference/metadrive).
 */
#ifndef MD_GEOM_H
#define MD_GEOM_H

#include "md_math.h"
#include "mdstep.h"

#define MD_MISS 2.0f /* "no hit" sentinel, > any valid fraction in [0,1] */

/* ------------------------------------------------------------------------------------------
 * Lidar beam vs oriented box.  Replaces one Bullet rayTestClosest against a BulletBoxShape
 * (component/sensors/distance_detector.py:58; shape base_vehicle.py:588).  The ray is given in the
 * BOX frame: origin (ox,oy), displacement over the whole range (dx,dy); returns the entry
 * fraction t in (0,1] or MD_MISS.  A ray that starts inside the box reports no hit (Bullet's
 * convex cast returns no hit for an initially-penetrating ray; see DESIGN.md "Bullet quirks").
 * -----------------------------------------------------------------------------------------*/
MD_HD float md_ray_box(float ox, float oy, float dx, float dy, float hl, float hw) {
    float tmin = -3.0e38f, tmax = 3.0e38f;
    if (dx != 0.0f) {
        float inv = 1.0f / dx;
        float t1 = (-hl - ox) * inv;
        float t2 = (hl - ox) * inv;
        float lo = t1 < t2 ? t1 : t2;
        float hi = t1 < t2 ? t2 : t1;
        tmin = lo;
        tmax = hi;
    } else if (ox < -hl || ox > hl) {
        return MD_MISS;
    }
    if (dy != 0.0f) {
        float inv = 1.0f / dy;
        float t1 = (-hw - oy) * inv;
        float t2 = (hw - oy) * inv;
        float lo = t1 < t2 ? t1 : t2;
        float hi = t1 < t2 ? t2 : t1;
        tmin = lo > tmin ? lo : tmin;
        tmax = hi < tmax ? hi : tmax;
    } else if (oy < -hw || oy > hw) {
        return MD_MISS;
    }
    if (tmax < tmin) return MD_MISS;
    if (tmin <= 0.0f) return MD_MISS; /* origin inside (or box behind) */
    if (tmin > 1.0f) return MD_MISS;
    return tmin;
}

/* Lidar beam vs circle (cones, warning triangles, pedestrians are Bullet cylinders seen from a
 * horizontal ray; static_object/traffic_object.py:57,100, traffic_participants/pedestrian.py:30).
 * (px,py) = origin - centre, (dx,dy) = displacement over the whole range. */
MD_HD float md_ray_circle(float px, float py, float dx, float dy, float r) {
    float a = dx * dx + dy * dy;
    float b = px * dx + py * dy;
    float cc = px * px + py * py - r * r;
    if (cc <= 0.0f) return MD_MISS; /* origin inside */
    if (b >= 0.0f) return MD_MISS;  /* moving away */
    float disc = b * b - a * cc;
    if (disc < 0.0f) return MD_MISS;
    float t = (-b - md_sqrt(disc)) / a;
    if (t <= 0.0f || t > 1.0f) return MD_MISS;
    return t;
}

/* Ray vs convex quad by Cyrus-Beck clipping: entry fraction in (0,1] or MD_MISS. */
MD_HD float md_ray_quad(float ox, float oy, float dx, float dy, const float* q) {
    float t_in = -3.0e38f, t_out = 3.0e38f;
    for (int i = 0; i < 4; ++i) {
        int j = (i + 1) & 3;
        float ex = q[2 * j] - q[2 * i], ey = q[2 * j + 1] - q[2 * i + 1];
        /* inward normal of a CCW polygon edge: (-ey, ex) */
        float num = (-ey) * (ox - q[2 * i]) + ex * (oy - q[2 * i + 1]); /* >=0 inside */
        float den = (-ey) * dx + ex * dy;
        if (den == 0.0f) {
            if (num < 0.0f) return MD_MISS;
        } else {
            float t = -num / den;
            if (den > 0.0f) { if (t > t_in) t_in = t; }
            else { if (t < t_out) t_out = t; }
        }
    }
    if (t_in > t_out) return MD_MISS;
    if (t_in <= 0.0f || t_in > 1.0f) return MD_MISS;
    return t_in;
}

/* One beam against one mover record. (dirx,diry) is the world-frame displacement of the beam. */
MD_HD float md_ray_shape(float ox, float oy, float dirx, float diry, float cx, float cy, float c, float s, float hl,
                         float hw, int kind) {
    float px = ox - cx, py = oy - cy;
    if (kind == MD_KIND_CONE || kind == MD_KIND_WARNING || kind == MD_KIND_PEDESTRIAN) {
        return md_ray_circle(px, py, dirx, diry, hl);
    }
    /* rotate into the box frame */
    float lox = px * c + py * s;
    float loy = py * c - px * s;
    float ldx = dirx * c + diry * s;
    float ldy = diry * c - dirx * s;
    return md_ray_box(lox, loy, ldx, ldy, hl, hw);
}

/* ------------------------------------------------------------------------------------------
 * Overlap tests (2-D restatement of Bullet contactTest between chassis box and other bodies:
 * base_vehicle.py:704-746, engine/core/collision_callback.py:5-42).
 * -----------------------------------------------------------------------------------------*/
MD_HD int md_obb_obb(float ax, float ay, float ac, float as, float ahl, float ahw, float bx, float by, float bc,
                     float bs, float bhl, float bhw) {
    float tx = bx - ax, ty = by - ay;
    /* |R| entries between the two frames */
    float cc = md_fabs(ac * bc + as * bs);  /* a.u . b.u */
    float cs = md_fabs(ac * bs - as * bc);  /* cross       */
    /* axes of A */
    float ta_u = md_fabs(tx * ac + ty * as);
    if (ta_u > ahl + bhl * cc + bhw * cs) return 0;
    float ta_v = md_fabs(ty * ac - tx * as);
    if (ta_v > ahw + bhl * cs + bhw * cc) return 0;
    /* axes of B */
    float tb_u = md_fabs(tx * bc + ty * bs);
    if (tb_u > bhl + ahl * cc + ahw * cs) return 0;
    float tb_v = md_fabs(ty * bc - tx * bs);
    if (tb_v > bhw + ahl * cs + ahw * cc) return 0;
    return 1;
}

MD_HD int md_obb_circle(float ax, float ay, float ac, float as, float ahl, float ahw, float bx, float by, float r) {
    float tx = bx - ax, ty = by - ay;
    float lx = md_fabs(tx * ac + ty * as) - ahl;
    float ly = md_fabs(ty * ac - tx * as) - ahw;
    if (lx < 0.0f) lx = 0.0f;
    if (ly < 0.0f) ly = 0.0f;
    return (lx * lx + ly * ly) <= r * r;
}

/* chassis OBB vs convex quad q[8] (x0,y0..x3,y3, CCW).  Lane-line ghost boxes
 * (block/base_block.py:470-519) and side-walk strips (pgblock/pg_block.py:294-332) are stored as quads. */
MD_HD int md_obb_quad(float ax, float ay, float ac, float as, float ahl, float ahw, const float* q) {
    /* project the quad on the two chassis axes */
    float umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f;
    for (int i = 0; i < 4; ++i) {
        float dx = q[2 * i] - ax, dy = q[2 * i + 1] - ay;
        float u = dx * ac + dy * as;
        float v = dy * ac - dx * as;
        umin = u < umin ? u : umin;
        umax = u > umax ? u : umax;
        vmin = v < vmin ? v : vmin;
        vmax = v > vmax ? v : vmax;
    }
    if (umin > ahl || umax < -ahl || vmin > ahw || vmax < -ahw) return 0;
    /* chassis corners */
    float ex = ac * ahl, ey = as * ahl; /* half-length vector */
    float fx = -as * ahw, fy = ac * ahw; /* half-width vector */
    float cxs[4], cys[4];
    cxs[0] = ax + ex + fx; cys[0] = ay + ey + fy;
    cxs[1] = ax - ex + fx; cys[1] = ay - ey + fy;
    cxs[2] = ax - ex - fx; cys[2] = ay - ey - fy;
    cxs[3] = ax + ex - fx; cys[3] = ay + ey - fy;
    /* project the chassis on each quad edge normal; separated if all corners are outside one edge */
    for (int i = 0; i < 4; ++i) {
        int j = (i + 1) & 3;
        float edx = q[2 * j] - q[2 * i], edy = q[2 * j + 1] - q[2 * i + 1];
        int all_out = 1;
        for (int k = 0; k < 4; ++k) {
            float cr = edx * (cys[k] - q[2 * i + 1]) - edy * (cxs[k] - q[2 * i]);
            if (cr >= 0.0f) { all_out = 0; }
        }
        if (all_out) return 0;
    }
    return 1;
}

/* point in convex polygon (CCW), boundary inclusive.  Stands in for Bullet's vertical rayTestAll
 * against the lane's BulletConvexHullShape (utils/pg/utils.py:174, block/base_block.py:431-468). */
MD_HD int md_point_in_hull(float x, float y, const float* xy, int n) {
    for (int i = 0; i < n; ++i) {
        int j = (i + 1 == n) ? 0 : i + 1;
        float ex = xy[2 * j] - xy[2 * i], ey = xy[2 * j + 1] - xy[2 * i + 1];
        float cr = ex * (y - xy[2 * i + 1]) - ey * (x - xy[2 * i]);
        if (cr < 0.0f) return 0;
    }
    return n >= 3;
}

/* hull vertices of a lane: inline copy for 4-vertex hulls, else the CSR table */
MD_HD const float* md_lane_hull(const MdLane* L, const float* hull_xy) {
    return (L->hull_n == 4) ? L->hull4 : hull_xy + 2 * (size_t)L->hull_off;
}

/* ------------------------------------------------------------------------------------------
 * Lane Frenet math: StraightLane.local_coordinates (lane/straight_lane.py:69-74),
 * CircularLane.local_coordinates (lane/circular_lane.py:71-121), heading_theta_at
 * (straight_lane.py:63-64, circular_lane.py:63-66), AbstractLane.distance (abs_lane.py:76-82).
 * -----------------------------------------------------------------------------------------*/
MD_HD void md_lane_local(const MdLane* L, float x, float y, float* s_out, float* lat_out) {
    if (L->type == 0) {
        float dx = x - L->ax, dy = y - L->ay;
        *s_out = dx * L->bx + dy * L->by;
        *lat_out = dx * L->by - dy * L->bx;
        return;
    }
    float dx = x - L->ax, dy = y - L->ay;
    float radius = L->bx, start_phase = L->by;
    float abs_phase = md_atan2(dy, dx);
    float d_start = md_fabs(md_wrap_to_pi(abs_phase - start_phase));
    float d_end = md_fabs(md_wrap_to_pi(abs_phase - L->end_phase_w));
    int clockwise = L->dirsign < 0.0f;
    float s;
    if (d_start > d_end) {
        float diff = clockwise ? (L->end_phase - abs_phase) : (abs_phase - L->end_phase);
        s = md_wrap_to_pi(diff) * radius + L->length;
    } else {
        float diff = clockwise ? (start_phase - abs_phase) : (abs_phase - start_phase);
        s = md_wrap_to_pi(diff) * radius;
    }
    *s_out = s;
    *lat_out = L->dirsign * (md_norm(dx, dy) - radius);
}

/* Sign test of ray_localization's heading filter (utils/pg/utils.py:181-184: cos(lane heading at the vehicle's
 * longitudinal, vehicle heading) > 0): the lane's tangent at the point dotted with the vehicle's heading (c, s).
 * Straight lane: its unit direction.  Circular lane: the tangent at the point's own phase, dirsign * (-(y - cy),
 * x - cx), left un-normalised -- only the sign is used -- so the filter needs neither the heading angle nor a sincos. */
MD_HD float md_lane_heading_dot(const MdLane* L, float x, float y, float c, float s) {
    if (L->type == 0) return L->bx * c + L->by * s;
    return L->dirsign * ((x - L->ax) * s - (y - L->ay) * c);
}

MD_HD float md_lane_heading_at(const MdLane* L, float s) {
    if (L->type == 0) return L->heading;
    float phi = L->dirsign * s / L->bx + L->by;
    return phi + MD_HALF_PI_F * L->dirsign;
}

/* centre-line point at longitudinal s: StraightLane.position (straight_lane.py:57-58) /
 * CircularLane.position (circular_lane.py:55-58) with lateral = 0 */
MD_HD void md_lane_position(const MdLane* L, float s, float* x, float* y) {
    if (L->type == 0) {
        *x = L->ax + s * L->bx;
        *y = L->ay + s * L->by;
        return;
    }
    float phi = L->dirsign * s / L->bx + L->by;
    float sn, cs;
    md_sincos(phi, &sn, &cs);
    *x = L->ax + L->bx * cs;
    *y = L->ay + L->bx * sn;
}

MD_HD float md_lane_distance(const MdLane* L, float s, float lat) {
    float a = s - L->length;
    float b = 0.0f - s;
    return md_fabs(lat) + (a > 0.0f ? a : 0.0f) + (b > 0.0f ? b : 0.0f);
}

/* AbstractLane.is_previous_lane_of (abs_lane.py:84-89): end of A within 0.1 m of start of B */
MD_HD int md_lane_is_previous_of(const MdLane* A, const MdLane* B) {
    return md_norm(A->ex - B->sx, A->ey - B->sy) < 0.1f;
}

/* ------------------------------------------------------------------------------------------
 * Kinematic bicycle sub-step.  Stand-in for one Bullet doPhysics(0.02) of the btRaycastVehicle
 * (engine/core/engine_core.py:350-352) under the action set by BaseVehicle._set_action /
 * _apply_throttle_brake (component/vehicle/base_vehicle.py:447-484).  Kinematics as in the
 * reference's own (dead) stand-ins component/vehicle_model/kinematics.py:148-158 and
 * bicycle_model.py:17-51:  beta = atan(lr/(lf+lr) tan(delta)), pdot = v (cos,sin)(psi+beta),
 * psidot = v sin(beta)/lr.  Longitudinal: engine accel while throttle>=0 and speed below max,
 * constant idle-brake drag, brake decel capped by tyre friction; with enable_reverse a negative throttle is a negative
 * engine force instead of the brake (agents only: the traffic's vehicle config keeps it off).
 * Lateral grip: a raycast-vehicle wheel cannot push sideways harder than frictionSlip x its load
 * (wheel_friction in the vehicle config, pg_space.py:226-272), i.e. the lateral acceleration v^2 sin(beta)/lr is
 * bounded by wheel_friction * g: beyond it the slip angle is cut back (the car understeers).  Without the bound a
 * full-lock command at 30 km/h yaws 15 degrees in one 0.1 s step -- and the reference's own PID steering
 * (PID_controller.py, heading gains 1.7 / 0.01 / 3.5 per step) limit-cycles at the step rate on such a plant,
 * which the Bullet vehicle it was tuned on does not do.  For the same reason the yaw RATE has inertia here: it moves
 * towards the kinematic value with the angular acceleration that the front axle's grip can give the chassis
 * (wheel_friction g lf / (2 k^2) ~ 2.3 rad/s^2 for the default car), starting from the rate of the previous step
 * (recovered from last_heading_dir, no extra state).  With both, the PID settles; without, it oscillates +-0.5 of
 * full lock at the step rate (tests/test_oracle_behaviour.py::test_traffic_steering_settles).
 * -----------------------------------------------------------------------------------------*/
typedef struct MdBicycle {
    float acc, dec;   /* engine acceleration / opposing deceleration for this step's action */
    float sb, cb;     /* sin / cos of the slip angle atan(lr/(lf+lr) tan(delta)), grip-limited */
    float sb_over_lr; /* sin(beta) / lr                                                      */
    float yaw_slew;   /* largest change of the yaw increment between two sub-steps: alpha_max dt^2, alpha_max =
                       * wheel_friction g lf / (2 k^2), k^2 = (L^2 + W^2) / 12 the box's radius of gyration squared:
                       * the front axle cannot push harder than its grip, so the yaw rate builds up over a few steps */
    float yaw_cap;    /* |yaw rate| <= |v| yaw_cap: sin of the full-lock slip angle / lr     */
    int rev;          /* reversing: the (negative) engine force applies whatever the speed   */
} MdBicycle;

/* Everything that depends on the action and the speed at the start of the step (constant over the decision_repeat
 * sub-steps), except the engine cut-off which looks at the current speed (base_vehicle.py:474). */
MD_HD void md_bicycle_prepare(float steer, float throttle, float speed, float hl, float hw, float dt, int reverse,
                              const MdParam* P, MdBicycle* b) {
    b->acc = (throttle > 0.0f) ? P->accel_gain * throttle : 0.0f;
    b->rev = 0;
    if (throttle >= 0.0f) b->dec = P->roll_decel; /* setBrake(2.0): idle drag when the engine is idle / cut */
    else if (reverse) { /* enable_reverse: negative engine force, brake released (base_vehicle.py:479-481) */
        b->acc = P->accel_gain * throttle;
        b->dec = 0.0f;
        b->rev = 1;
    } else b->dec = md_min(-throttle * P->brake_gain, P->fric_decel);
    float delta = steer * P->max_steer;
    float sd, cd;
    md_sincos(delta, &sd, &cd);
    float tan_d = sd / cd;
    float beta = md_atan(P->lr / (P->lf + P->lr) * tan_d);
    float sb, cb;
    md_sincos(beta, &sb, &cb);
    float grip = P->fric_decel * P->lr / md_max(speed * speed, 1.0e-3f); /* largest |sin(beta)| the tyres can hold */
    if (md_fabs(sb) > grip) {
        sb = (sb > 0.0f) ? grip : -grip;
        cb = md_sqrt(1.0f - sb * sb);
    }
    b->sb = sb;
    b->cb = cb;
    b->sb_over_lr = sb / P->lr;
    b->yaw_slew = P->fric_decel * P->lf * 1.5f / (hl * hl + hw * hw) * dt * dt;
    b->yaw_cap = 0.5f / P->lr;
}

/* (cp, sp) = cos / sin of the direction of travel psi + beta, carried from sub-step to sub-step: the heading turns
 * by d = vm sin(beta)/lr dt (at most ~0.16 rad) per sub-step, and the pair is ROTATED by d with the series
 * sin d = d (1 - d^2/6 + d^4/120), cos d = 1 - d^2/2 + d^4/24 - d^6/720 (truncation below 1e-9) instead of being
 * re-evaluated with the full sincos: one sincos per step instead of five.  psi itself accumulates exactly as
 * before, and the pose's own (cos psi, sin psi) is taken from it once at the end of the step. */
MD_HD void md_bicycle_substep(float* x, float* y, float* psi, float* v, float* cp_io, float* sp_io, float* yaw_io,
                              float throttle, const MdBicycle* b, const MdParam* P, float dt) {
    float vv = *v;
    float speed_kmh = md_fabs(vv) * 3.6f;
    float acc = 0.0f, dec = 0.0f;
    if (b->rev || (throttle > 0.0f && !(speed_kmh > P->max_speed_kmh))) {
        acc = b->acc; /* engine force on 4 wheels; Bullet applies no brake impulse then */
    } else {
        dec = b->dec;
    }
    /* drag/brake opposes motion and never reverses it within a sub-step */
    float vnew = vv + acc * dt;
    float dv = dec * dt;
    if (vnew > 0.0f) {
        vnew = vnew - dv;
        if (vnew < 0.0f) vnew = 0.0f;
    } else if (vnew < 0.0f) {
        vnew = vnew + dv;
        if (vnew > 0.0f) vnew = 0.0f;
    }
    float sp = *sp_io, cp = *cp_io;
    float vm = 0.5f * (vv + vnew);
    *x = *x + vm * cp * dt;
    *y = *y + vm * sp * dt;
    /* yaw: the kinematic rate v sin(beta) / lr is approached with the angular acceleration the front axle's grip
     * allows (b->yaw_slew per sub-step), and never exceeds what the wheels can trace at the present speed */
    float d = *yaw_io + md_clip(vm * b->sb_over_lr * dt - *yaw_io, -b->yaw_slew, b->yaw_slew);
    float d_max = md_fabs(vm) * b->yaw_cap * dt;
    d = md_clip(d, -d_max, d_max);
    *yaw_io = d;
    *psi = md_wrap_to_pi(*psi + d);
    float d2 = d * d;
    float sd = d * (1.0f - d2 * (0.16666667f - d2 * 0.0083333333f));
    float cd = 1.0f - d2 * (0.5f - d2 * (0.041666667f - d2 * 0.0013888889f));
    *cp_io = cp * cd - sp * sd;
    *sp_io = sp * cd + cp * sd;
    *v = vnew;
}

/* ------------------------------------------------------------------------------------------
 * IDM longitudinal model + PID lateral (policy/idm_policy.py:293-320, PID_controller.py:10-17).
 * Units follow the reference: speeds in km/h throughout.
 * -----------------------------------------------------------------------------------------*/
MD_HD float md_not_zero(float x, float eps) { /* utils/math.py:107-113 */
    if (md_fabs(x) > eps) return x;
    if (x > 0.0f) return eps;
    return -eps;
}

MD_HD float md_pow10(float b) { /* b^10, DELTA = 10.0 (idm_policy.py:199) */
    float b2 = b * b;
    float b4 = b2 * b2;
    float b8 = b4 * b4;
    return b8 * b2;
}

/* desired_gap (idm_policy.py:313-320), projected=True: dv = (v_ego - v_front) . heading, km/h */
MD_HD float md_idm_desired_gap(float ego_speed_kmh, float dv_kmh) {
    const float d0 = 10.0f, tau = 1.5f;
    const float ab = 5.0f; /* -ACC_FACTOR * DEACC_FACTOR = -1.0 * -5 */
    return d0 + ego_speed_kmh * tau + ego_speed_kmh * dv_kmh / (2.0f * md_sqrt(ab));
}

/* acceleration (idm_policy.py:303-311) */
MD_HD float md_idm_acceleration(float ego_speed_kmh, float target_speed_kmh, int has_front, float dist_to_front,
                                float dv_kmh) {
    float ts = md_not_zero(target_speed_kmh, 0.0f);
    float ratio = md_max(ego_speed_kmh, 0.0f) / ts;
    float acc = 1.0f * (1.0f - md_pow10(ratio));
    if (has_front) {
        float sd = md_idm_desired_gap(ego_speed_kmh, dv_kmh) / md_not_zero(dist_to_front, 1e-2f);
        acc -= 1.0f * (sd * sd);
    }
    return acc;
}

/* PIDController.get_result (PID_controller.py:10-17); state = {p,i,d} errors */
MD_HD float md_pid(float* p_err, float* i_err, float* d_err, float kp, float ki, float kd, float err) {
    *i_err = *i_err + err;
    *d_err = err - *p_err;
    *p_err = err;
    return -kp * (*p_err) - ki * (*i_err) - kd * (*d_err);
}

/* BaseVehicle.heading_diff (base_vehicle.py:528-552): cos between heading and the lane's lateral
 * direction, mapped to [0,1] */
MD_HD float md_heading_diff(const MdLane* L, float x, float y, float hc, float hs) {
    float lx, ly;
    if (L->type == 0) {
        /* get_vertical_vector(end-start)[1] = (v.y, -v.x)/|v| */
        float vx = L->ex - L->sx, vy = L->ey - L->sy;
        float n = md_norm(vx, vy);
        lx = vy / n;
        ly = -vx / n;
    } else if (L->dirsign > 0.0f) { /* counter-clockwise: position - centre */
        lx = x - L->ax;
        ly = y - L->ay;
    } else {
        lx = L->ax - x;
        ly = L->ay - y;
    }
    float ln = md_norm(lx, ly);
    float fn = md_norm(hc, hs);
    if (!(ln * fn != 0.0f)) return 0.0f;
    float cosv = (hc * lx + hs * ly) / (ln * fn);
    return md_clip(cosv, -1.0f, 1.0f) / 2.0f + 0.5f;
}

/* One half of the navigation vector: NodeNetworkNavigation._get_info_for_checkpoint
 * (navigation_module/node_network_navigation.py:243-292).  ref = reference lane (lane 0 of the
 * target road); later_middle = (n_cur/2 - 0.5) * w_cur. */
MD_HD void md_navi_for_checkpoint(const MdLane* ref, float later_middle, float x, float y, float hc, float hs,
                                  float n_cur_lanes, float cur_width, float radius_max, float angle_max_deg,
                                  float* out5) {
    const float NAVI_POINT_DIST = 50.0f;
    float cpx = ref->ex + later_middle * ref->elx;
    float cpy = ref->ey + later_middle * ref->ely;
    float dx = cpx - x, dy = cpy - y;
    float dn = md_norm(dx, dy);
    if (dn > NAVI_POINT_DIST) {
        dx = dx / dn * NAVI_POINT_DIST;
        dy = dy / dn * NAVI_POINT_DIST;
    }
    /* BaseVehicle.convert_to_local_coordinates (base_vehicle.py:986-988): (forward, left) */
    float fwd = dx * hc + dy * hs;
    float left = dy * hc - dx * hs;
    out5[0] = md_clip((fwd / NAVI_POINT_DIST + 1.0f) / 2.0f, 0.0f, 1.0f);
    out5[1] = md_clip((left / NAVI_POINT_DIST + 1.0f) / 2.0f, 0.0f, 1.0f);
    float bend = 0.0f, dir = 0.0f, angle = 0.0f;
    if (ref->type == 1) {
        bend = ref->bx / (radius_max + n_cur_lanes * cur_width);
        dir = -ref->dirsign;
        angle = ref->angle;
    }
    out5[2] = md_clip(bend, 0.0f, 1.0f);
    out5[3] = md_clip((dir + 1.0f) / 2.0f, 0.0f, 1.0f);
    out5[4] = md_clip((angle * 57.29577951308232f / angle_max_deg + 1.0f) / 2.0f, 0.0f, 1.0f);
}

/* step energy in mL (base_vehicle.py:255-271): 3.25 * e^(0.01 v_kmh) * distance_km / 100 * 1000 */
MD_HD float md_step_energy(float speed_kmh, float dist_m) {
    float dist_km = dist_m / 1000.0f;
    float e = md_exp(0.01f * speed_kmh);
    float t = 3.25f * e;
    t = t * dist_km;
    t = t / 100.0f;
    return t * 1000.0f;
}

MD_HD float md_probe_eval(int op, float a, float b) {
    float s, c;
    switch (op) {
        case 0: md_sincos(a, &s, &c); return s;
        case 1: md_sincos(a, &s, &c); return c;
        case 2: return md_atan2(a, b);
        case 3: return md_acos(a);
        case 4: return md_exp(a);
        case 5: return a / b;
        case 6: return md_sqrt(a);
        case 7: return md_wrap_to_pi(a);
        case 8: return md_asin(a);
        case 9: return md_norm(a, b);
        case 10: return md_step_energy(a, b);
        default: return 0.0f;
    }
}

#endif /* MD_GEOM_H */

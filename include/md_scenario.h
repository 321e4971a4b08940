/*
 * md_scenario.h -- scalar step logic of SCENARIO mode (MdConfig.traffic_mode 4): the ScenarioEnv path of the
 * reference -- an agent following the SDC's recorded route, traffic replayed from recorded tracks or driven
 * reactively along its own recorded path.  Same role as md_entity.h: one spelling of every formula, compiled into
 * the HIP kernels and into the gcc-built CPU oracle.  What the GPU does differently (polyline projection with one
 * LANE per segment and a wavefront arg-min, one wave per reactive vehicle, ballot-ordered spawns) is written in
 * metadrive_ped_amd/csrc/mdstep.hip; the oracle uses the serial forms below.
 *
 * Reference (paths relative to /root/reference/metadrive):
 *   utils/interpolating_line.py:12-71,176-228,267-291   InterpolatingLine (segments, local_coordinates, get_point ...)
 *   component/lane/point_lane.py:16-131                  PointLane
 *   component/navigation_module/trajectory_navigation.py:96-186   TrajectoryNavigation
 *   envs/scenario_env.py:128-357                         ScenarioEnv done / cost / reward
 *   policy/idm_policy.py:133-171,426-493                 single-lane front search, TrajectoryIDMPolicy
 *   manager/scenario_traffic_manager.py:67-76,89-146,171-236   IDM batches, replay / spawn / removal
 *   policy/replay_policy.py:43-67, scenario/parse_object_state.py:24-75   ReplayTrafficParticipantPolicy
 */
#ifndef MD_SCENARIO_H
#define MD_SCENARIO_H

#include "md_entity.h"

#define MD_TRAJ_DISCRETE_LEN 2.0f     /* TrajectoryNavigation.DISCRETE_LEN          */
#define MD_TRAJ_NUM_WAY_POINT 10      /* TrajectoryNavigation.NUM_WAY_POINT         */
#define MD_TRAJ_NAVI_POINT_DIST 30.0f /* TrajectoryNavigation.NAVI_POINT_DIST       */
#define MD_TRAJ_NAVI_DIM 22           /* NUM_WAY_POINT * CHECK_POINT_INFO_DIM + 2   */
#define MD_TIDM_NORMAL_SPEED 40.0f    /* TrajectoryIDMPolicy.NORMAL_SPEED (km/h)    */
#define MD_TIDM_MAX_DIST 20.0f        /* TrajectoryIDMPolicy.IDM_MAX_DIST           */
#define MD_TIDM_DEST_RADIUS 2.0f      /* TrajectoryIDMPolicy.DEST_REGION_RADIUS     */
#define MD_TIDM_BATCH 5               /* ScenarioTrafficManager.IDM_ACT_BATCH_SIZE  */

/* observation layout of scenario mode (obs/state_obs.py:64-151 with TrajectoryNavigation):
 *   [side cloud n_side | 2 border dims] + heading_diff, speed, steering, 2 last actions, yaw rate
 *   + [lane-line cloud | lateral] + navi 22 + lidar */
MD_HD int md_sc_obs_navi(const MdConfig* c) { return md_obs_ll(c) + (c->n_lane_line > 0 ? c->n_lane_line : 1); }
MD_HD int md_sc_obs_lidar(const MdConfig* c) { return md_sc_obs_navi(c) + MD_TRAJ_NAVI_DIM; }

typedef struct MdPoly {
    const MdSeg* segs;
    int n;          /* >= 1 segments */
    float length;   /* sum of the segment lengths */
} MdPoly;

MD_HD MdPoly md_poly_of(const MdWorld* w, size_t n_global) {
    MdPoly p;
    const int a = w->poly_off[n_global], b = w->poly_off[n_global + 1];
    p.segs = w->segs + a;
    p.n = b - a;
    p.length = (p.n > 0) ? p.segs[p.n - 1].cum + p.segs[p.n - 1].len : 0.0f;
    return p;
}

/* InterpolatingLine.min_lineseg_dist for ONE segment (interpolating_line.py:267-291): a = start, b = end, d = the
 * unit tangent; clamped parallel distance h, perpendicular distance c, result hypot(h, c). */
MD_HD float md_seg_dist(const MdSeg* g, float px, float py) {
    float s = (g->sx - px) * g->dx + (g->sy - py) * g->dy;
    float t = (px - g->ex) * g->dx + (py - g->ey) * g->dy;
    float h = md_max(md_max(s, t), 0.0f);
    float c = (px - g->sx) * g->dy - (py - g->sy) * g->dx;
    return md_norm(h, c);
}

/* local coordinates w.r.t. segment idx (interpolating_line.py:56-66): longitudinal accumulates the lengths of the
 * segments before it; lateral direction = get_vertical_vector(end - start)[1] = (dy, -dx): positive to the RIGHT */
MD_HD void md_poly_local_at(const MdPoly* p, int idx, float px, float py, float* lng, float* lat) {
    const MdSeg* g = &p->segs[idx];
    float ddx = px - g->sx, ddy = py - g->sy;
    *lng = g->cum + (ddx * g->dx + ddy * g->dy);
    *lat = ddx * g->dy - ddy * g->dx;
}

/* InterpolatingLine.local_coordinates, serial form: np.argmin over the segments' distances (first minimum) */
MD_HD int md_poly_local(const MdPoly* p, float px, float py, float* lng, float* lat) {
    int best = 0;
    float bd = 3.0e38f;
    for (int i = 0; i < p->n; ++i) {
        float d = md_seg_dist(&p->segs[i], px, py);
        if (d < bd) {
            bd = d;
            best = i;
        }
    }
    md_poly_local_at(p, best, px, py, lng, lat);
    return best;
}

/* get_heading_theta (interpolating_line.py:190-203): the first segment whose accumulated end lies beyond `s` */
MD_HD int md_poly_seg_heading(const MdPoly* p, float s) {
    for (int i = 0; i < p->n; ++i)
        if (p->segs[i].cum + p->segs[i].len > s) return i;
    return p->n - 1;
}

/* segment() / get_point() (interpolating_line.py:176-188,205-214): the first segment with end + 0.1 >= s */
MD_HD int md_poly_seg_at(const MdPoly* p, float s) {
    for (int i = 0; i < p->n; ++i)
        if (p->segs[i].cum + p->segs[i].len + 0.1f >= s) return i;
    return p->n - 1;
}

MD_HD void md_poly_position(const MdPoly* p, float s, float lateral, float* x, float* y) {
    const MdSeg* g = &p->segs[md_poly_seg_at(p, s)];
    float along = s - g->cum;
    *x = g->sx + along * g->dx + lateral * g->dy;
    *y = g->sy + along * g->dy - lateral * g->dx;
}

/* point strictly inside a simple polygon (even-odd rule): stands in for shapely's Polygon.contains behind
 * AbstractLane.point_on_lane (component/lane/abs_lane.py:109-114); shapely is a third-party package that is not in
 * /root/reference, so this predicate is parity-unpinned (the polygon itself, PointLane.auto_generate_polygon, is
 * pinned by tests/golden/scenario.json). */
MD_HD int md_point_in_polygon(const float* xy, int n, float px, float py) {
    int inside = 0;
    for (int i = 0, j = n - 1; i < n; j = i++) {
        float xi = xy[2 * i], yi = xy[2 * i + 1], xj = xy[2 * j], yj = xy[2 * j + 1];
        if (((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi) + xi)) inside = !inside;
    }
    return inside;
}

/* one polygon edge's contribution to the crossing number (the lane-parallel form sums these over the lanes) */
MD_HD int md_polygon_edge_crosses(const float* xy, int n, int i, float px, float py) {
    int j = (i == 0) ? n - 1 : i - 1;
    float xi = xy[2 * i], yi = xy[2 * i + 1], xj = xy[2 * j], yj = xy[2 * j + 1];
    return ((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi) + xi);
}

/* ------------------------------------------------------------------------------------------
 * Agent: TrajectoryNavigation.update_localization (trajectory_navigation.py:105-146) given the projection of the
 * agent on the reference trajectory, + the state observation, + ScenarioEnv reward / cost / done.
 * The agent's MdPid holds the navigation's memory: hp = current_longitude (last_current_long[1]).
 * -----------------------------------------------------------------------------------------*/
typedef struct MdTrajLoc {
    float lng, lat;       /* local_coordinates of the agent on the reference trajectory */
    float heading_at;     /* heading_theta_at(lng)                                      */
    float lat_dx, lat_dy; /* lateral_direction(lng) (heading_diff)                      */
} MdTrajLoc;

MD_HD void md_traj_locate(const MdPoly* p, float px, float py, MdTrajLoc* o) {
    md_poly_local(p, px, py, &o->lng, &o->lat);
    o->heading_at = p->segs[md_poly_seg_heading(p, o->lng)].heading;
    const MdSeg* g = &p->segs[md_poly_seg_at(p, o->lng)];
    o->lat_dx = g->dy;
    o->lat_dy = -g->dx;
}

/* the 22 navigation dims.  ckpt = the env's checkpoint table (n_ckpt >= 1 points). */
MD_HD int md_traj_next_idx(const MdTrajLoc* L, int n_ckpt) {
    int next_idx = (int)(L->lng / MD_TRAJ_DISCRETE_LEN) + 1;   /* int() truncates towards zero, like the C cast */
    if (next_idx < 0) next_idx = 0;
    if (next_idx > n_ckpt - 1) next_idx = n_ckpt - 1;
    return next_idx;
}

/* way point k (0 .. MD_TRAJ_NUM_WAY_POINT - 2; ckpts[1:]: the first of the ten is skipped): the value BOTH of its slots
 * receive -- trajectory_navigation.py:134: `self._navi_info[start:end], lanes_heading = [a, b]` unpacks the two-element list */
MD_HD float md_traj_navi_point(const float* ckpt, int n_ckpt, int next_idx, int k, float px, float py, float hc, float hs) {
    int idx = next_idx + 1 + k;
    if (idx > n_ckpt - 1) idx = n_ckpt - 1;   /* padded with the last checkpoint */
    float dx = ckpt[2 * idx] - px, dy = ckpt[2 * idx + 1] - py;
    float dn = md_norm(dx, dy);
    if (dn > MD_TRAJ_NAVI_POINT_DIST) {
        dx = dx / dn * MD_TRAJ_NAVI_POINT_DIST;
        dy = dy / dn * MD_TRAJ_NAVI_POINT_DIST;
    }
    float fwd = dx * hc + dy * hs;   /* convert_to_local_coordinates: (forward, left), SURVEY 8a-5 */
    return md_clip((fwd / MD_TRAJ_NAVI_POINT_DIST + 1.0f) / 2.0f, 0.0f, 1.0f);
}

/* with_points = 0: the way points (dims 0..17) are written by the caller (the kernel: one lane per way point) */
MD_HD void md_traj_navi(const float* ckpt, int n_ckpt, const MdTrajLoc* L, float px, float py, float hc, float hs,
                        float heading, float max_lateral_dist, float* out22, int with_points) {
    const int next_idx = md_traj_next_idx(L, n_ckpt);
    for (int i = with_points ? 0 : 2 * (MD_TRAJ_NUM_WAY_POINT - 1); i < MD_TRAJ_NAVI_DIM; ++i) out22[i] = 0.0f;
    if (with_points)
        for (int k = 0; k < MD_TRAJ_NUM_WAY_POINT - 1; ++k) {
            float a = md_traj_navi_point(ckpt, n_ckpt, next_idx, k, px, py, hc, hs);
            out22[2 * k] = a;
            out22[2 * k + 1] = a;
        }
    out22[18] = md_clip((L->lat / max_lateral_dist + 1.0f) / 2.0f, 0.0f, 1.0f);
    out22[19] = md_clip((md_wrap_to_pi(L->heading_at - heading) / MD_PI_F + 1.0f) / 2.0f, 0.0f, 1.0f);
}

/* Observation (state + navi), reward, cost, done of the agent in slot a.  `side_fill`: with the side detector on and
 * no road-line bodies in the scene every beam reports "nothing" (1.0). */
MD_HD void md_scenario_observe_at(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int a, int env_just_reset,
                                  const MdTrajLoc* Lp, float ref_length, int with_points) {
    const int n = a;
    float* obs = s->obs + (size_t)a * c->obs_dim;
    float* info = s->step_info + (size_t)a * 8;
    const MdShape* sh = &s->shape[n];
    MdDyn* d = &s->dyn[n];
    MdNav* nav = &s->nav[n];
    const int just_reset = env_just_reset;
    const MdTrajLoc L = *Lp;
    struct { float length; } ref;
    ref.length = ref_length;
    const float long_last = s->pid[n].hp;   /* navigation.last_longitude */
    s->pid[n].hp = L.lng;
    const float route_completion = L.lng / ref.length;

    const int o_base = md_obs_base(c), o_mid = md_obs_mid(c), o_ll = md_obs_ll(c), o_navi = md_sc_obs_navi(c);
    const MdParam* P = &s->param[n];
    const float speed_kmh = md_fabs(d->speed) * 3.6f;
    if (o_base) {
        obs[0] = md_clip(2.0f * sh->hl / 10.0f, 0.0f, 1.0f);
        obs[1] = md_clip(2.0f * sh->hw / 2.5f, 0.0f, 1.0f);
    }
    if (c->n_side > 0) {
        for (int i = 0; i < c->n_side; ++i) obs[o_base + i] = 1.0f;   /* no ContinuousLaneLine bodies: md_line_detector overwrites these when the scene has road lines */
    } else {
        /* dist_to_left_side / dist_to_right_side (base_vehicle.py:491-499) with TrajectoryNavigation: lane width =
         * the route's width 2 (get_idm_route), lateral range = 2 * width (trajectory_navigation.py:148-152) */
        float tl = L.lat + 1.0f, tr = 4.0f - tl;
        obs[o_base + 0] = md_clip(tl / c->total_width, 0.0f, 1.0f);
        obs[o_base + 1] = md_clip(tr / c->total_width, 0.0f, 1.0f);
    }
    /* heading_diff with a PointLane (base_vehicle.py:537-552) */
    {
        float ln = md_norm(L.lat_dx, L.lat_dy), fn = md_norm(sh->c, sh->s);
        float hd = 0.0f;
        if (ln * fn != 0.0f) hd = md_clip((sh->c * L.lat_dx + sh->s * L.lat_dy) / (ln * fn), -1.0f, 1.0f) / 2.0f + 0.5f;
        obs[o_mid + 0] = hd;
    }
    obs[o_mid + 1] = md_clip((speed_kmh + 1.0f) / (P->max_speed_kmh + 1.0f), 0.0f, 1.0f);
    obs[o_mid + 2] = md_clip((d->steering / 60.0f + 1.0f) / 2.0f, 0.0f, 1.0f);
    obs[o_mid + 3] = md_clip((s->action[2 * n] + 1.0f) / 2.0f, 0.0f, 1.0f);
    obs[o_mid + 4] = md_clip((s->action[2 * n + 1] + 1.0f) / 2.0f, 0.0f, 1.0f);
    {
        float cosb = (sh->c * d->last_c + sh->s * d->last_s) / (md_norm(sh->c, sh->s) * md_norm(d->last_c, d->last_s));
        obs[o_mid + 5] = md_clip(md_acos(md_clip(cosb, 0.0f, 1.0f)) / 0.1f, 0.0f, 1.0f);
    }
    if (c->n_lane_line <= 0) obs[o_ll] = md_clip((L.lat * 2.0f / c->max_lane_width + 1.0f) / 2.0f, 0.0f, 1.0f);
    else
        for (int i = 0; i < c->n_lane_line; ++i) obs[o_ll + i] = 1.0f;
    {
        const int k0 = w->ckpt_off[e], k1 = w->ckpt_off[e + 1];
        md_traj_navi(w->ckpt_xy + 2 * (size_t)k0, k1 - k0, &L, sh->cx, sh->cy, sh->c, sh->s, d->heading, c->max_lateral_dist,
                     obs + o_navi, with_points);
    }

    /* ---- flags from the contact phase ---- */
    uint32_t fl = s->flags[n] & (MD_FL_CRASH_VEHICLE | MD_FL_CRASH_OBJECT | MD_FL_CRASH_HUMAN | MD_FL_CRASH_BUILDING |
                                 MD_FL_CRASH_SIDEWALK | MD_FL_ON_WHITE_CONT | MD_FL_ON_YELLOW_CONT | MD_FL_ON_BROKEN |
                                 MD_FL_ON_CROSSWALK);
    const int on_line = (fl & (MD_FL_ON_YELLOW_CONT | MD_FL_CRASH_SIDEWALK | MD_FL_ON_WHITE_CONT)) != 0;
    /* _is_arrive_destination / _is_out_of_road (scenario_env.py:380-401) */
    const int arrive = (route_completion > 0.95f) || (ref.length < 2.0f);
    int out_of_road;
    if (c->relax_out_of_road_done) out_of_road = md_fabs(L.lat) > c->max_lateral_dist;
    else {
        out_of_road = on_line;
        if (c->out_of_route_done) out_of_road = out_of_road || (md_fabs(L.lat) > 10.0f);
    }
    const int out_of_road_done = out_of_road || (route_completion < -0.1f);   /* done_info[OUT_OF_ROAD] (:141) */
    if (arrive) fl |= MD_FL_ARRIVE_DEST;
    if (out_of_road_done) fl |= MD_FL_OUT_OF_ROAD;

    /* ---- reward (scenario_env.py:220-297) ---- */
    float reward = 0.0f;
    reward += c->driving_reward * (L.lng - long_last);
    const float lateral_penalty = -(md_fabs(L.lat) / c->max_lateral_dist) * c->lateral_penalty;
    reward += lateral_penalty;
    const float heading_diff = md_wrap_to_pi(md_fabs(d->heading - L.heading_at)) / MD_PI_F;
    reward += -heading_diff * c->heading_penalty;
    {
        float steering = md_fabs(s->action[2 * n]);
        float allowed = 1.0f / md_max(md_fabs(d->speed), 1.0e-2f);
        float overflowed = md_min(allowed - steering, 0.0f);
        reward += overflowed * c->steering_range_penalty;
    }
    if (c->no_negative_reward) reward = md_max(reward, 0.0f);
    if (fl & MD_FL_CRASH_VEHICLE) reward = -c->crash_vehicle_penalty;
    if (fl & MD_FL_CRASH_OBJECT) reward = -c->crash_object_penalty;
    if (fl & MD_FL_CRASH_HUMAN) reward = -c->crash_human_penalty;
    if (on_line) reward = -c->on_lane_line_penalty;
    float step_reward = reward;
    if (arrive) reward = c->success_reward;
    else if (out_of_road) reward = -c->out_of_road_penalty;

    /* ---- cost (scenario_env.py:199-218) ---- */
    float cost = 0.0f;
    if (out_of_road) cost += c->out_of_road_cost;
    if (fl & MD_FL_CRASH_VEHICLE) cost += c->crash_vehicle_cost;
    if (fl & MD_FL_CRASH_OBJECT) cost += c->crash_object_cost;
    if (fl & MD_FL_CRASH_HUMAN) cost += c->crash_human_cost;

    /* ---- done (scenario_env.py:128-197) ---- */
    if (!just_reset) nav->steps += 1;
    int max_step = (c->horizon > 0) && (nav->steps >= c->horizon);
    int done = 0;
    if (arrive) done = 1;
    else if (out_of_road_done) done = 1;
    else if ((fl & MD_FL_CRASH_HUMAN) && c->crash_human_done) done = 1;
    else if ((fl & MD_FL_CRASH_VEHICLE) && c->crash_vehicle_done) done = 1;
    else if ((fl & MD_FL_CRASH_OBJECT) && c->crash_object_done) done = 1;
    else if ((fl & MD_FL_CRASH_BUILDING) && c->crash_object_done) done = 1;
    else if (max_step) {
        if (c->truncate_as_terminate) done = 1;
    } else if (c->allowed_more_steps > 0 &&   /* this scene's own length: track_meta[slot 0][1] (<= c->scenario_length, the batch's) */
               nav->steps >= w->track_meta[4 * ((size_t)e * c->cap) + 1] + c->allowed_more_steps) {
        if (c->truncate_as_terminate) done = 1;
        max_step = 1;
    }
    if (max_step) fl |= MD_FL_MAX_STEP;
    if (just_reset) {
        reward = 0.0f;
        cost = 0.0f;
        step_reward = 0.0f;
        nav->done = 0;
    } else {
        nav->done = nav->done || done;
        if (nav->done) fl |= MD_FL_TERMINATED;
        if (max_step) fl |= MD_FL_TRUNCATED;
    }
    s->flags[n] = fl;
    if (s->done_out) ((uint32_t*)s->done_out)[a] = md_done_word(fl);
    s->reward[a] = reward;
    s->cost[a] = cost;
    float step_energy = just_reset ? 0.0f : md_step_energy(speed_kmh, md_norm(d->last_x - sh->cx, d->last_y - sh->cy));
    s->pid[n].energy += step_energy;
    info[0] = step_reward;
    info[1] = md_fabs(d->speed);
    info[2] = step_energy;
    info[3] = s->pid[n].energy;
    info[4] = just_reset ? 0.0f : info[4] + reward;
    info[5] = just_reset ? 0.0f : info[5] + cost;
    info[6] = route_completion;
    info[7] = (float)nav->steps;
    if (c->auto_reset && !just_reset && (fl & (MD_FL_TERMINATED | MD_FL_TRUNCATED))) s->need_reset[0] = 1;
}

/* serial form (oracle): project the agent on its reference trajectory, then observe */
MD_HD void md_scenario_observe(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int a, int env_just_reset) {
    const MdPoly ref = md_poly_of(w, (size_t)e * c->cap + a);
    MdTrajLoc L;
    md_traj_locate(&ref, s->shape[a].cx, s->shape[a].cy, &L);
    md_scenario_observe_at(w, s, c, e, a, env_just_reset, &L, ref.length, 1);
}

/* ------------------------------------------------------------------------------------------
 * The route a reactive vehicle follows: the slot's static polyline (its track's first valid run, host-built), or -- after a
 * spawn at any other frame -- the route md_build_route cut at the spawn frame (MdState.route_*; `s` is the env view).
 * -----------------------------------------------------------------------------------------*/
typedef struct MdRoute {
    MdPoly poly;          /* poly.length is NOT filled (md_route_length) */
    const float* verts;   /* outline polygon */
    int n_verts;
    const float* aux;     /* [8] end point, outline bounding box; NULL = derive */
} MdRoute;

MD_HD int md_route_is_dynamic(const MdState* s, int slot) { return s->route_n && s->route_n[4 * slot] > 0; }

MD_HD MdRoute md_route_of(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int slot) {
    MdRoute r;
    if (md_route_is_dynamic(s, slot)) {
        r.poly.segs = s->route_segs + (size_t)slot * c->route_seg_cap;
        r.poly.n = s->route_n[4 * slot];
        r.verts = s->route_verts + 2 * (size_t)slot * c->route_vert_cap;
        r.n_verts = s->route_n[4 * slot + 1];
        r.aux = s->route_aux + 8 * (size_t)slot;
    } else {
        const size_t ng = (size_t)e * c->cap + slot;
        const int a = w->poly_off[ng], b = w->poly_off[ng + 1];
        r.poly.segs = w->segs + a;
        r.poly.n = b - a;
        r.verts = w->polyv + 2 * (size_t)w->polyv_off[ng];
        r.n_verts = w->polyv_off[ng + 1] - w->polyv_off[ng];
        r.aux = w->poly_aux ? w->poly_aux + 8 * ng : 0;
    }
    r.poly.length = 0.0f;
    return r;
}

MD_HD float md_route_length(const MdPoly* p) { return (p->n > 0) ? p->segs[p->n - 1].cum + p->segs[p->n - 1].len : 0.0f; }

/* the valid run [t0, t1) of the track in slot n_global that contains frame k (get_max_valid_indicis gives [k, t1)); 0 if none */
MD_HD int md_track_run_end(const MdWorld* w, size_t n_global, int k) {
    if (!w->run_off || !w->runs) return 0;
    for (int r = w->run_off[n_global]; r < w->run_off[n_global + 1]; ++r)
        if (w->runs[2 * r] <= k && k < w->runs[2 * r + 1]) return w->runs[2 * r + 1];
    return 0;
}

/* PointLane(points, width 2) for a route cut at a spawn frame (get_idm_route): InterpolatingLine._get_properties
 * (utils/interpolating_line.py:104-146: from a kept point, the next kept one is the first farther than 1 m, else the last point;
 * pieces shorter than 1e-6 are dropped; a path that never moves is one 0.1 m piece along +x), PointLane.auto_generate_polygon
 * (component/lane/point_lane.py:58-106: the strip sampled every metre, one metre beyond both ends), and the aux record
 * (MdWorld.poly_aux).  Geometry in double like the host's numpy (+, -, *, /, sqrt are correctly rounded on both the CPU and the GPU:
 * the oracle and the kernel agree bit for bit), stored as float; the heading is md_atan2 of the stored direction.
 * Points: x = xy[i * stride], y = xy[i * stride + 1], i < n_pts.
 * The element functions below are what both forms are made of: md_build_route walks them serially (oracle); the kernel runs the
 * chain with lanes = candidate points, the pieces with lanes = pieces, the running length on one lane (a sum in a fixed order)
 * and the outline with lanes = vertices. */

/* from kept point i: the first point farther than 1 m (squared distance > 1), else the last point */
MD_HD int md_route_far(const float* xy, size_t stride, int i, int q) {
    const double ux = (double)xy[(size_t)i * stride] - (double)xy[(size_t)q * stride];
    const double uy = (double)xy[(size_t)i * stride + 1] - (double)xy[(size_t)q * stride + 1];
    return ux * ux + uy * uy > 1.0;
}

/* the piece from point i to point j, everything but `cum`; returns its length in double, negative when the piece is dropped */
MD_HD double md_route_piece(const float* xy, size_t stride, int i, int j, MdSeg* g) {
    const double xi = xy[(size_t)i * stride], yi = xy[(size_t)i * stride + 1];
    const double ex = xy[(size_t)j * stride], ey = xy[(size_t)j * stride + 1];
    const double ddx = ex - xi, ddy = ey - yi;
    const double L = __builtin_sqrt(ddx * ddx + ddy * ddy);
    if (L < 1e-6) return -1.0;
    g->sx = (float)xi;
    g->sy = (float)yi;
    g->ex = (float)ex;
    g->ey = (float)ey;
    g->dx = (float)(ddx / L);
    g->dy = (float)(ddy / L);
    g->len = (float)L;
    g->heading = md_atan2(g->dy, g->dx);
    g->cum = 0.0f;
    g->spare[0] = g->spare[1] = g->spare[2] = 0.0f;
    return L;
}

/* a path that never moved: one 0.1 m piece along +x (its lateral direction is hard-wired to (0, 1), interpolating_line.py:137-144) */
MD_HD void md_route_still_piece(const float* xy, MdSeg* g) {
    g->sx = xy[0];
    g->sy = xy[1];
    g->ex = (float)((double)xy[0] + 0.1);
    g->ey = xy[1];
    g->dx = 1.0f;
    g->dy = 0.0f;
    g->len = 0.1f;
    g->heading = 0.0f;
    g->cum = 0.0f;
    g->spare[0] = g->spare[1] = g->spare[2] = 0.0f;
}

/* number of outline samples: len(arange(0, length + 1, 1)) = ceil(length + 1) */
MD_HD int md_route_n_long(double length) {
    int n_long = (int)length + 1;
    if ((double)(n_long - 1) < length) n_long += 1;
    if ((double)(n_long - 1) >= length + 1.0) n_long -= 1;
    return n_long;
}

/* Vertex o of the outline, 0 <= o < 2 * (n_long + 2): the right side (lateral -1) with the samples ascending, then the left side
 * descending; the first and the last sample of a side are flanked by a point one metre further along the end piece's direction.  A
 * sample's piece = the first whose accumulated end + 0.1 reaches it, else the last (InterpolatingLine.get_point), found by
 * bisection (the ends ascend); the geometry is recomputed in double from the stored floats -- the host keeps doubles, the
 * difference is below the float resolution of the result. */
MD_HD void md_route_outline_vertex(const MdSeg* segs, int ns, int never_moved, int n_long, int o, float* fx, float* fy) {
    const int per_side = n_long + 2;
    const int side = o >= per_side ? 1 : 0;
    const int r = o - side * per_side;
    int t = r - 1;
    if (t < 0) t = 0;
    if (t > n_long - 1) t = n_long - 1;
    const int li = side == 0 ? t : n_long - 1 - t;
    const double sv = (double)li, lat = side == 0 ? -1.0 : 1.0;
    int lo = 0, hi = ns - 1;   /* first piece with end + 0.1 >= sv, else the last */
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((double)segs[mid].cum + (double)segs[mid].len + 0.1 >= sv) hi = mid;
        else lo = mid + 1;
    }
    const MdSeg* g = &segs[lo];
    const double along = sv - (double)g->cum;
    const double lx = never_moved ? 0.0 : (double)g->dy, ly = never_moved ? 1.0 : -(double)g->dx;
    double x = (double)g->sx + along * (double)g->dx + lat * lx;
    double y = (double)g->sy + along * (double)g->dy + lat * ly;
    /* the flanking points: before the route's start (-d0) at sample 0, beyond its end (+d1) at the last sample */
    const int extra = (r == 0) || (r == per_side - 1);
    if (extra) {
        if (li == 0) {
            x -= (double)segs[0].dx;
            y -= (double)segs[0].dy;
        } else {
            x += (double)segs[ns - 1].dx;
            y += (double)segs[ns - 1].dy;
        }
    }
    *fx = (float)x;
    *fy = (float)y;
}

/* aux[0..1]: PointLane.end = position(length, 0) in the float arithmetic of md_poly_position */
MD_HD void md_route_end_point(const MdSeg* segs, int ns, float* aux) {
    const float flen = segs[ns - 1].cum + segs[ns - 1].len;
    int ie = ns - 1;
    for (int q = 0; q < ns; ++q)
        if (segs[q].cum + segs[q].len + 0.1f >= flen) {
            ie = q;
            break;
        }
    aux[0] = segs[ie].sx + (flen - segs[ie].cum) * segs[ie].dx;
    aux[1] = segs[ie].sy + (flen - segs[ie].cum) * segs[ie].dy;
}

/* serial form.  counts[0] = pieces, counts[1] = vertices; returns 0, or -1 when the buffers are too small (counts[0] = 0). */
MD_HD int md_build_route(const float* xy, size_t stride, int n_pts, MdSeg* segs, int seg_cap, float* verts, int vert_cap,
                         float* aux, int32_t* counts) {
    int ns = 0;
    double cum = 0.0;
    int i = 0;
    counts[0] = 0;
    counts[1] = 0;
    while (i < n_pts - 1) {
        int j = n_pts - 1;
        for (int q = i + 1; q < n_pts; ++q)
            if (md_route_far(xy, stride, i, q)) {
                j = q;
                break;
            }
        if (ns >= seg_cap) return -1;
        const double L = md_route_piece(xy, stride, i, j, &segs[ns]);
        if (!(L < 0.0)) {
            segs[ns].cum = (float)cum;
            cum += L;
            ++ns;
        }
        i = j;
    }
    int never_moved = 0;
    if (ns == 0) {
        if (seg_cap < 1 || n_pts < 1) return -1;
        never_moved = 1;
        md_route_still_piece(xy, &segs[ns++]);
        cum = 0.1;
    }
    const int n_long = md_route_n_long(cum);
    const int nv = 2 * (n_long + 2);
    if (nv > vert_cap) return -1;
    float bx0 = 3.0e38f, by0 = 3.0e38f, bx1 = -3.0e38f, by1 = -3.0e38f;
    for (int o = 0; o < nv; ++o) {
        float fx, fy;
        md_route_outline_vertex(segs, ns, never_moved, n_long, o, &fx, &fy);
        verts[2 * o] = fx;
        verts[2 * o + 1] = fy;
        bx0 = md_min(bx0, fx);
        by0 = md_min(by0, fy);
        bx1 = md_max(bx1, fx);
        by1 = md_max(by1, fy);
    }
    md_route_end_point(segs, ns, aux);
    aux[2] = bx0;
    aux[3] = by0;
    aux[4] = bx1;
    aux[5] = by1;
    aux[6] = aux[7] = 0.0f;
    counts[0] = ns;
    counts[1] = nv;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * TrajectoryIDMPolicy (policy/idm_policy.py:426-493) for the vehicle in `slot`, decided BEFORE the integration of
 * episode step k (ScenarioTrafficManager.before_step, scenario_traffic_manager.py:67-76).
 * Slot state: MdNav.ck0 = MD_SC_IDM, MdNav.timer = policy_index; MdPid: hp / hi / hd = heading PID (1.2, 0.1, 3.5),
 * lp = last_action[1]; the route is the slot's polyline.
 * -----------------------------------------------------------------------------------------*/
/* one candidate of get_find_front_back_objs_single_lane (idm_policy.py:150-169): returns its longitudinal gap or a
 * negative number when the object does not count.  poly_xy / n_v: the route's outline polygon. */
MD_HD float md_tidm_front_gap(const MdPoly* route, const float* poly_xy, int n_v, float cur_long, float px, float py,
                              const MdShape* o) {
    if (!md_present(o->flags)) return -1.0f;
    if (md_norm(o->cx - px, o->cy - py) > MD_TIDM_MAX_DIST) return -1.0f;
    /* bounding_box (base_object.py:533-542): the four corners; the object counts when any of them is on the lane */
    float ex = o->c * o->hl, ey = o->s * o->hl, fx = -o->s * o->hw, fy = o->c * o->hw;
    int on = md_point_in_polygon(poly_xy, n_v, o->cx + ex + fx, o->cy + ey + fy) ||
             md_point_in_polygon(poly_xy, n_v, o->cx + ex - fx, o->cy + ey - fy) ||
             md_point_in_polygon(poly_xy, n_v, o->cx - ex - fx, o->cy - ey - fy) ||
             md_point_in_polygon(poly_xy, n_v, o->cx - ex + fx, o->cy - ey + fy);
    if (!on) return -1.0f;
    float lg, lt;
    md_poly_local(route, o->cx, o->cy, &lg, &lt);
    return lg - cur_long;
}

/* steering + acceleration once the front object is known (front < 0: none).  Writes action / PID / last action. */
/* `lane_heading` = route.heading_theta_at(own longitudinal + 1) (idm_policy.py:462-468): the heading of the first piece that ends
 * beyond that point (md_poly_seg_heading) -- looked up by the caller, serially in the oracle, one lane per piece in the kernel */
MD_HD void md_tidm_decide(const MdPoly* route, const MdState* s, int slot, int do_speed_control, int front, float front_dist,
                          float lane_heading) {
    MdShape* sh = &s->shape[slot];
    MdDyn* d = &s->dyn[slot];
    MdPid* pid = &s->pid[slot];
    float acc = pid->lp;   /* last_action[-1] */
    if (do_speed_control) {
        float speed_kmh = md_fabs(d->speed) * 3.6f;
        float dv = 0.0f;
        if (front >= 0) {
            const MdShape* fo = &s->shape[front];
            float fv = s->dyn[front].speed;
            float evx = d->speed * sh->c * 3.6f, evy = d->speed * sh->s * 3.6f;
            float fvx = fv * fo->c * 3.6f, fvy = fv * fo->s * 3.6f;
            dv = (evx - fvx) * sh->c + (evy - fvy) * sh->s;
        }
        acc = md_idm_acceleration(speed_kmh, MD_TIDM_NORMAL_SPEED, front >= 0, front_dist, dv);
    }
    (void)route;
    float steering = md_pid(&pid->hp, &pid->hi, &pid->hd, 1.2f, 0.1f, 3.5f, -md_wrap_to_pi(lane_heading - d->heading));
    pid->lp = acc;
    s->action[2 * slot] = steering;
    s->action[2 * slot + 1] = acc;
}

/* serial form (oracle): arrival check, front search over the env's movers (ascending slot: equal gaps keep the
 * lowest slot), decision */
MD_HD void md_tidm_vehicle(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int slot, int k) {
    MdNav* nav = &s->nav[slot];
    const MdShape* sh = &s->shape[slot];
    const MdRoute rt = md_route_of(w, s, c, e, slot);
    MdPoly route = rt.poly;
    route.length = md_route_length(&route);
    float end_x, end_y;
    md_poly_position(&route, route.length, 0.0f, &end_x, &end_y);   /* PointLane.end = position(length, 0) */
    if (md_norm(sh->cx - end_x, sh->cy - end_y) < MD_TIDM_DEST_RADIUS) {
        nav->ck0 = MD_SC_ARRIVED;   /* no action this step: the vehicle rolls on with its previous one */
        return;
    }
    const int do_speed_control = (k % MD_TIDM_BATCH) == nav->timer;
    int front = -1;
    float front_dist = MD_TIDM_MAX_DIST;
    float cur_long, tmp;
    md_poly_local(&route, sh->cx, sh->cy, &cur_long, &tmp);
    if (do_speed_control) {
        const float* pv = rt.verts;
        const int n_v = rt.n_verts;
        for (int j = 0; j < c->cap; ++j) {
            if (j == slot) continue;
            float g = md_tidm_front_gap(&route, pv, n_v, cur_long, sh->cx, sh->cy, &s->shape[j]);
            if (g > 0.0f && g < front_dist) {
                front_dist = g;
                front = j;
            }
        }
    }
    md_tidm_decide(&route, s, slot, do_speed_control, front, front_dist,
                   route.segs[md_poly_seg_heading(&route, cur_long + 1.0f)].heading);
}

/* ------------------------------------------------------------------------------------------
 * agent_policy = ReplayEgoCarPolicy: the agent replays the SDC track.  In the reference (manager/agent_manager.py:164-187) a replayed
 * agent gets before_step([0, 0]), rolls through the physics step, has its after_step (localisation, contact test) and is THEN put on
 * frame k (position, heading, velocity).  Here it is put on frame k in place of the integration: what after_step computes sees the
 * recorded pose instead of the pose Bullet rolled to from frame k-1 (centimetres apart, and Bullet's to begin with -- unpinned like
 * every trajectory); the observation and the lidar see frame k as in the reference.  A frame that is not valid,
 * or past the data, leaves the agent where it is (policy.act returns None).
 * -----------------------------------------------------------------------------------------*/
MD_HD void md_scenario_replay_ego(const MdState* s, const MdConfig* c, int a, int k) {
    MdShape* sh = &s->shape[a];
    MdDyn* d = &s->dyn[a];
    s->action[2 * a] = 0.0f;
    s->action[2 * a + 1] = 0.0f;
    d->last_x = sh->cx;      /* BaseVehicle.before_step: last_position / last_heading_dir */
    d->last_y = sh->cy;
    d->last_c = sh->c;
    d->last_s = sh->s;
    d->steering = 0.0f;
    d->throttle = 0.0f;
    if (k >= c->track_len) return;
    const size_t at = (size_t)k * (size_t)c->n_envs * (size_t)c->cap + (size_t)a;
    const MdShape fr = s->track_shape[at];
    if (!(fr.flags & MD_F_ALIVE)) return;
    sh->cx = fr.cx;
    sh->cy = fr.cy;
    sh->c = fr.c;
    sh->s = fr.s;
    d->heading = s->track_dyn[2 * at];
    d->speed = s->track_dyn[2 * at + 1];
}

/* ------------------------------------------------------------------------------------------
 * Traffic lifecycle at the END of episode step k (ScenarioTrafficManager.after_step / after_reset,
 * scenario_traffic_manager.py:89-146,171-296).  Per track slot (1 .. cap-1), in slot order:
 *   REPLAY   pose / velocity from frame k; removed when the frame is not valid or the data are over (cones and
 *            barriers have no policy: they stay where they were spawned)
 *   ARRIVED  removed
 *   ABSENT   spawned when frame k is valid and the filters pass: replay, or -- reactive_traffic, a moving track
 *            starting behind the ego within 15 m sideways and heading its way, longer than 5 m -- TrajectoryIDMPolicy
 *            at rest on its own path with policy_index = idm_policy_count % 5 (counter: MdState.next_agent_id[0]).
 * The IDM route of a track spawned at the start of its first valid run is host-built (MdWorld.segs); spawned at any other
 * frame it is cut at that frame and built here (md_build_route) -- without MdWorld.runs / MdState.route_* such a spawn is replayed.
 * The decision for one slot given the running IDM count; returns 1 when an IDM policy was created.
 * -----------------------------------------------------------------------------------------*/
MD_HD int md_scenario_slot_after_step(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int j, int k, int idm_count,
                                      int dry) {   /* dry: change nothing, only tell whether an IDM policy would be created */
    MdShape* sh = &s->shape[j];
    MdDyn* d = &s->dyn[j];
    MdNav* nav = &s->nav[j];
    const size_t ng = (size_t)e * c->cap + j;
    const int32_t* tm = w->track_meta + 4 * ng;
    const int in_data = k < c->track_len;
    const size_t at = (size_t)(in_data ? k : 0) * (size_t)c->n_envs * (size_t)c->cap + (size_t)j;
    const MdShape fr = s->track_shape[at];
    const int valid = in_data && (fr.flags & MD_F_ALIVE);
    const int kind = md_kind_of(fr.flags);
    if (dry && nav->ck0 != MD_SC_ABSENT) return 0;
    if (nav->ck0 == MD_SC_REPLAY) {
        const int k_now = md_kind_of(sh->flags);
        if (k_now == MD_KIND_CONE || k_now == MD_KIND_WARNING || k_now == MD_KIND_BARRIER) return 0;   /* static object */
        if (!valid) {
            sh->flags &= ~MD_F_ALIVE;
            nav->ck0 = MD_SC_ABSENT;
            return 0;
        }
        d->last_x = sh->cx;
        d->last_y = sh->cy;
        d->last_c = sh->c;
        d->last_s = sh->s;
        *sh = fr;
        if (kind == MD_KIND_VEHICLE) sh->flags = fr.flags | MD_F_STATIC;   /* kinematic: never integrated */
        d->heading = s->track_dyn[2 * at];
        d->speed = s->track_dyn[2 * at + 1];
        return 0;
    }
    if (nav->ck0 == MD_SC_ARRIVED) {
        sh->flags &= ~MD_F_ALIVE;
        nav->ck0 = MD_SC_ABSENT;
        return 0;
    }
    if (nav->ck0 != MD_SC_ABSENT) return 0;   /* IDM: drives */
    if (!valid || (tm[2] & MD_TM_NEVER)) return 0;
    int idm = 0, late_pts = 0;
    if (kind == MD_KIND_VEHICLE) {
        const int moving = (tm[2] & MD_TM_MOVING) != 0;
        if (c->no_static_vehicles && !moving) return 0;
        const MdShape* ego = &s->shape[0];
        float rx = fr.cx - ego->cx, ry = fr.cy - ego->cy;
        float heading_dist = rx * ego->c + ry * ego->s;    /* convert_to_local_coordinates: (forward, left) */
        float side_dist = ry * ego->c - rx * ego->s;
        if (!c->ego_replay && c->filter_overlapping_car && md_fabs(heading_dist) < 8.0f && md_fabs(side_dist) < 2.0f) return 0;
        const float fr_heading = s->track_dyn[2 * at];
        const int heading_ok = md_fabs(md_wrap_to_pi(s->dyn[0].heading - fr_heading)) < MD_HALF_PI_F;
        const int idm_ok = heading_dist < -1.0f && md_fabs(side_dist) < 15.0f && heading_ok;
        idm = c->reactive_traffic && moving && idm_ok;
        if (idm) {
            if (k == tm[0]) {
                idm = (tm[2] & MD_TM_LENGTH_OK) != 0;    /* the first run from its start: the host-built route */
            } else {
                /* any other frame (a later run, a spawn the overlap filter held back, a respawn after an arrival): the route is
                 * cut at this frame -- get_max_valid_indicis(track, k), IDM_CREATE_MIN_LENGTH on ITS two ends */
                idm = 0;
                if (s->route_n && c->route_seg_cap > 0) {
                    const int t1 = md_track_run_end(w, ng, k);
                    if (t1 > k) {
                        const MdShape last = s->track_shape[(size_t)(t1 - 1) * (size_t)c->n_envs * (size_t)c->cap + (size_t)j];
                        idm = md_norm(fr.cx - last.cx, fr.cy - last.cy) > 5.0f;
                        late_pts = t1 - k;
                    }
                }
            }
        }
    }
    if (dry) return idm;
    if (s->route_n) {
        s->route_n[4 * j] = 0;
        s->route_n[4 * j + 1] = 0;
        s->route_n[4 * j + 2] = k;
        s->route_n[4 * j + 3] = (idm && late_pts > 0) ? late_pts : 0;   /* > 0: md_scenario_build_pending has a route to build */
    }
    *sh = fr;
    d->heading = s->track_dyn[2 * at];
    d->last_x = fr.cx;
    d->last_y = fr.cy;
    d->last_c = fr.c;
    d->last_s = fr.s;
    d->steering = 0.0f;
    d->throttle = 0.0f;
    s->action[2 * j] = 0.0f;
    s->action[2 * j + 1] = 0.0f;
    s->flags[j] = 0;
    MdPid* pid = &s->pid[j];
    pid->hp = pid->hi = pid->hd = 0.0f;
    pid->lp = pid->li = pid->ld = 0.0f;
    pid->target_speed = MD_TIDM_NORMAL_SPEED;
    if (idm) {
        d->speed = 0.0f;   /* spawn_object(vehicle_class, position, heading): at rest */
        nav->ck0 = MD_SC_IDM;
        nav->timer = idm_count % MD_TIDM_BATCH;
        return 1;
    }
    d->speed = s->track_dyn[2 * at + 1];
    if (kind == MD_KIND_VEHICLE) sh->flags = fr.flags | MD_F_STATIC;
    nav->ck0 = MD_SC_REPLAY;
    return 0;
}

/* The route of a slot that md_scenario_slot_after_step just gave a reactive policy at a frame other than its first run's start
 * (route_n[4j+3] = number of frames left in the run): PointLane(positions[k : k + that]) into the slot's route buffers.  `xy` /
 * `stride`: the positions (NULL = the frames in MdState.track_shape; the kernel hands in a staged copy).  When the buffers are too
 * small the slot keeps following its static polyline (route_n[4j] stays 0). */
MD_HD void md_scenario_build_pending(const MdState* s, const MdConfig* c, int j, const float* xy, size_t stride) {
    if (!s->route_n || s->route_n[4 * j + 3] <= 0) return;
    const int k = s->route_n[4 * j + 2];
    const int n_pts = s->route_n[4 * j + 3] < c->route_seg_cap ? s->route_n[4 * j + 3] : c->route_seg_cap;   /* (the host sizes the cap
                                                                                        by the longest run: never cut) */
    if (!xy) {
        xy = &s->track_shape[(size_t)k * (size_t)c->n_envs * (size_t)c->cap + (size_t)j].cx;
        stride = (size_t)c->n_envs * (size_t)c->cap * (sizeof(MdShape) / sizeof(float));
    }
    int32_t counts[2];
    md_build_route(xy, stride, n_pts, s->route_segs + (size_t)j * c->route_seg_cap, c->route_seg_cap,
                   s->route_verts + 2 * (size_t)j * c->route_vert_cap, c->route_vert_cap, s->route_aux + 8 * (size_t)j, counts);
    s->route_n[4 * j] = counts[0];
    s->route_n[4 * j + 1] = counts[1];
    s->route_n[4 * j + 3] = 0;
}

/* serial form (oracle) */
MD_HD void md_scenario_after_step_env(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int k) {
    for (int j = c->agents_per_env; j < c->cap; ++j) {
        s->next_agent_id[0] += md_scenario_slot_after_step(w, s, c, e, j, k, s->next_agent_id[0], 0);
        md_scenario_build_pending(s, c, j, 0, 0);
    }
}

#endif /* MD_SCENARIO_H */

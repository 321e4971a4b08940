/* md_integrator_alt.c -- TEST INFRASTRUCTURE.  A SECOND, independent spelling of the per-step vehicle integration
 * (SURVEY 8 row a-1), so that "HIP kernel == oracle" is not one header compiled twice for this row.
 *
 * The product's integrator is include/md_entity.h: md_integrate_mover + include/md_geom.h: md_bicycle_prepare /
 * md_bicycle_substep (float32, its own sincos / atan, the travel direction carried by a rotation series).  This file
 * includes NEITHER: it reads the record layouts from include/mdstep.h (the ABI) and is written from the model itself --
 *
 *   kinematic bicycle (the reference's own kinematic vehicle, component/vehicle_model/kinematics.py:148-158):
 *       beta = atan(lr / (lf + lr) tan(delta)),  (x, y) += v (cos, sin)(psi + beta) dt,  psi += v sin(beta) / lr dt
 *   with what the Bullet raycast vehicle adds and DESIGN.md section 4 documents:
 *       engine   acc = 4 F_engine / m * throttle while throttle > 0 and |v| <= max_speed (base_vehicle.py:474-478)
 *       brake    dec = min(-throttle * 4 F_brake / (m dt), mu g)  (throttle < 0), rolling drag 2.0 when idle (:473);
 *                it opposes the motion and never reverses it inside a sub-step
 *       reverse  enable_reverse and throttle < 0: negative engine force, no brake (:479-481)
 *       grip     |sin(beta)| <= mu g lr / v^2           (the tyres cannot hold more lateral acceleration)
 *       yaw      the yaw increment per sub-step moves toward v sin(beta)/lr dt by at most alpha_max dt^2,
 *                alpha_max = mu g lf / (2 k^2), k^2 = (L^2 + W^2)/12, and is capped by |v| dt / (2 lr)
 *       carry-in the yaw increment of the previous step ~ sin(heading change of that step) / substeps
 * -- in DOUBLE precision with libm's sin / cos / atan / fmod and the travel direction evaluated as cos / sin of the
 * accumulated angle.  Agreement with the float32 product path is therefore to a tolerance (stated in the tests:
 * 2e-4 m, 2e-5 rad, 2e-5 m/s per step from identical start states), never bit for bit: a wrong formula, a swapped
 * term or a missing clamp in either spelling shows up as a difference orders of magnitude above it.
 *
 * Only tests/ may load this library.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "mdstep.h"

#define EXPORT __attribute__((visibility("default")))
#define ALT_PI 3.14159265358979323846

static int alt_drives(int f) {
    if (!(f & MD_F_ALIVE)) return 0;
    if ((f & MD_KIND_MASK) != MD_KIND_VEHICLE) return 0;
    return !(f & (MD_F_STATIC | MD_F_PENDING));
}

static double alt_unit(double a) { /* safe_clip_for_small_array(a, -1, 1) (utils/math.py:16-26) */
    if (a != a) return 0.0;
    if (a > 3.0e38) return 1.0;
    if (a < -3.0e38) return -1.0;
    return a < -1.0 ? -1.0 : (a > 1.0 ? 1.0 : a);
}

static double alt_wrap(double a) { /* to (-pi, pi] */
    a = fmod(a + ALT_PI, 2.0 * ALT_PI);
    if (a <= 0.0) a += 2.0 * ALT_PI;
    return a - ALT_PI;
}

static double alt_clamp(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* One env-step (c->substeps sub-steps of c->dt) of every driving, not just-spawned vehicle among slots [0, n).
 * out6[6 * j] = x, y, psi, v, cos psi, sin psi after the step (untouched for slots that do not drive). */
EXPORT void alt_integrate(const MdShape* shape, const MdDyn* dyn, const MdParam* param, const float* action, int n,
                          const MdConfig* c, double* out6) {
    for (int j = 0; j < n; ++j) {
        const int f = shape[j].flags;
        if (!alt_drives(f) || (f & MD_F_SPAWNED)) continue;
        const MdParam* P = &param[j];
        const double steer = alt_unit(action[2 * j]), thr = alt_unit(action[2 * j + 1]);
        const double dt = c->dt;
        const int reverse = c->enable_reverse && (f & MD_F_AGENT) && thr < 0.0;
        double x = shape[j].cx, y = shape[j].cy, psi = dyn[j].heading, v = dyn[j].speed;
        /* yaw increment carried in: sin(heading change over the previous step) / substeps */
        double yaw = ((double)shape[j].s * dyn[j].last_c - (double)shape[j].c * dyn[j].last_s) / (double)c->substeps;

        /* what depends on the action and the speed at the start of the step */
        const double mu_g = P->fric_decel;
        const double delta = steer * P->max_steer;
        double beta = atan(P->lr / ((double)P->lf + P->lr) * tan(delta));
        double sb = sin(beta);
        const double v2 = v * v > 1.0e-3 ? v * v : 1.0e-3;
        const double grip = mu_g * P->lr / v2;
        if (fabs(sb) > grip) {
            sb = sb > 0.0 ? grip : -grip;
            beta = asin(sb);
        }
        const double L = 2.0 * shape[j].hl, W = 2.0 * shape[j].hw;
        const double k2 = (L * L + W * W) / 12.0;
        const double alpha_max = mu_g * P->lf / (2.0 * k2);
        const double yaw_slew = alpha_max * dt * dt;
        double th = psi + beta; /* direction of travel */

        for (int k = 0; k < c->substeps; ++k) {
            double acc = 0.0, dec = 0.0;
            const int engine_on = thr > 0.0 && !(fabs(v) * 3.6 > P->max_speed_kmh);
            if (reverse) acc = P->accel_gain * thr;
            else if (engine_on) acc = P->accel_gain * thr;
            else if (thr >= 0.0) dec = P->roll_decel;
            else {
                dec = -thr * P->brake_gain;
                if (dec > mu_g) dec = mu_g;
            }
            double vn = v + acc * dt;
            if (vn > 0.0) {
                vn -= dec * dt;
                if (vn < 0.0) vn = 0.0;
            } else if (vn < 0.0) {
                vn += dec * dt;
                if (vn > 0.0) vn = 0.0;
            }
            const double vm = 0.5 * (v + vn);
            x += vm * cos(th) * dt;
            y += vm * sin(th) * dt;
            const double want = vm * sb / P->lr * dt;
            double d = yaw + alt_clamp(want - yaw, -yaw_slew, yaw_slew);
            const double d_max = fabs(vm) * dt / (2.0 * P->lr);
            d = alt_clamp(d, -d_max, d_max);
            yaw = d;
            psi = alt_wrap(psi + d);
            th += d;
            v = vn;
        }
        out6[6 * j + 0] = x;
        out6[6 * j + 1] = y;
        out6[6 * j + 2] = psi;
        out6[6 * j + 3] = v;
        out6[6 * j + 4] = cos(psi);
        out6[6 * j + 5] = sin(psi);
    }
}

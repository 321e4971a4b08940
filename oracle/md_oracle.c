/*
 * md_oracle.c -- CPU ORACLE for the batched MetaDrive step() path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load the library built from it (oracle/Makefile ->
 * oracle/_build/libmdoracle.so).  The product package (metadrive_ped_amd) never imports it and
 * fails loudly when its HIP library is missing.
 *
 * What it is: a plain scalar restatement, in C, of the reference's per-step algorithm
 * (zhuhaozh/metadrive_ped, MetaDrive v0.4.2.2) for N independent environments -- brute force,
 * single thread per call (ref_step_mt adds an OpenMP-free pthread fan-out over envs for the CPU
 * baseline), no culling, no grid, no fusion.  Every function names the reference code it follows
 * (paths relative to /root/reference/metadrive).  The arithmetic kernels (ray/box, SAT, Frenet,
 * IDM formulas) come from include/md_geom.h + md_math.h, shared with the HIP build so that the
 * comparison is bit-exact; see the headers for why.
 *
 * Parity pinning: tests/golden/ (JSON fixtures) hold outputs of the reference's own Python (lanes, navi,
 * obs normalisation, reward/done, IDM/PID, lidar mask, RNG streams, PG topology per seed) generated
 * in the build container by oracle/gen/gen_golden.py; tests/test_oracle_golden.py checks this file
 * against them.  Parts whose numbers come out of Bullet in the reference (vehicle trajectories,
 * rayTest hit fractions, contact/sweep booleans, hull hits) are PARITY UNPINNED vs the reference:
 * panda3d==1.10.13 (setup.py:54) is a third-party wheel absent from /root/reference; for those this
 * restatement follows the published geometry definitions (block/base_block.py:431-519,
 * pgblock/pg_block.py:259-332, base_vehicle.py:585-598) and the analytic intersector of
 * tests/test_component/test_detector_mask.py:125-147.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "md_scenario.h"

#define EXPORT __attribute__((visibility("default")))

#define kind_of md_kind_of
#define is_circle_kind md_is_circle_kind
#define present md_present
#define drives md_drives

/* ------------------------------------------------------------------------------------------
 * Lidar: perceive() component/sensors/distance_detector.py:27-85 with Lidar's mask
 * (constants.py:242-244); beam i direction = heading + 2*pi*i/B (utils/math.py:76-81,
 * distance_detector.py:177-180); own chassis excluded (distance_detector.py:129,60-71).
 * -----------------------------------------------------------------------------------------*/
static void lidar_agent_det(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int a, float* out,
                            unsigned long long det[2]) {
    int base = e * c->cap;
    const MdShape* me = &s->shape[base + a];
    for (int i = 0; i < c->n_beams; ++i) out[i] = 1.0f;
    det[0] = det[1] = 0ull;
    if (!present(me->flags)) return;
    for (int i = 0; i < c->n_beams; ++i) {
        float bc = w->beam_cs[2 * i], bs = w->beam_cs[2 * i + 1];
        /* rotate the beam by the ego heading, scale by range */
        float dirx = (bc * me->c - bs * me->s) * c->lidar_range;
        float diry = (bs * me->c + bc * me->s) * c->lidar_range;
        float best = 1.0f;
        int best_j = -1; /* the body the beam hits first (detected_objects; equal fractions -> lowest slot) */
        for (int j = 0; j < c->cap; ++j) {
            if (j == a) continue;
            const MdShape* o = &s->shape[base + j];
            if (!present(o->flags)) continue;
            float t = md_ray_shape(me->cx, me->cy, dirx, diry, o->cx, o->cy, o->c, o->s, o->hl, o->hw, kind_of(o->flags));
            if (t < best) {
                best = t;
                best_j = j;
            }
        }
        out[i] = best;
        if (best_j >= 0 && best_j < 128) det[best_j >> 6] |= 1ull << (best_j & 63);
    }
}

static void lidar_agent(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int a, float* out) {
    unsigned long long det[2];
    lidar_agent_det(w, s, c, e, a, out, det);
}

EXPORT int ref_lidar(const MdWorld* w, const MdState* s, const MdConfig* c, float* out, int out_stride, int out_offset) {
    for (int e = 0; e < c->n_envs; ++e)
        for (int a = 0; a < c->agents_per_env; ++a)
            lidar_agent(w, s, c, e, a, out + (size_t)(e * c->agents_per_env + a) * out_stride + out_offset);
    return MD_OK;
}

/* Side / lane-line detector: DistanceDetector.perceive vs static line boxes
 * (component/sensors/distance_detector.py:118-160,194-209). */
EXPORT int ref_line_detector(const MdWorld* w, const MdState* s, const MdConfig* c, const float* beam_cs, int n_beams,
                             float range, uint32_t kind_mask, float* out, int out_stride, int out_offset) {
    for (int e = 0; e < c->n_envs; ++e) {
        int m = w->env_map[e];
        for (int a = 0; a < c->agents_per_env; ++a) {
            const MdShape* me = &s->shape[e * c->cap + a];
            float* o = out + (size_t)(e * c->agents_per_env + a) * out_stride + out_offset;
            for (int i = 0; i < n_beams; ++i) {
                float best = 1.0f;
                if (present(me->flags)) {
                    float bc = beam_cs[2 * i], bs = beam_cs[2 * i + 1];
                    float dirx = (bc * me->c - bs * me->s) * range;
                    float diry = (bs * me->c + bc * me->s) * range;
                    for (int q = w->quad_off[m]; q < w->quad_off[m + 1]; ++q) {
                        if (!((kind_mask >> w->quad_kind[q]) & 1u)) continue;
                        float t = md_ray_quad(me->cx, me->cy, dirx, diry, w->quads + 8 * (size_t)q);
                        if (t < best) best = t;
                    }
                }
                o[i] = best;
            }
        }
    }
    return MD_OK;
}

/* ------------------------------------------------------------------------------------------
 * Dynamics: before_step + 5 x doPhysics (base_vehicle.py:211-232,447-484; engine_core.py:350-352)
 * -----------------------------------------------------------------------------------------*/
EXPORT int ref_integrate(const MdWorld* w, const MdState* s, const MdConfig* c) {
    (void)w;
    for (int e = 0; e < c->n_envs; ++e) {
        MdState v = md_env_view(s, c, e); /* env-local slots: the replay mode tells agents from traffic by slot */
        for (int j = 0; j < c->cap; ++j) md_advance_mover(&v, c, j);
    }
    return MD_OK;
}

/* ------------------------------------------------------------------------------------------
 * Localisation: ray_localization (utils/pg/utils.py:151-203), _get_current_lane
 * (node_network_navigation.py:219-241), _update_current_lane (:294-304),
 * _update_target_checkpoints (:181-201).
 * -----------------------------------------------------------------------------------------*/
static void localize_mover(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int n) {
    (void)c;
    MdShape* sh = &s->shape[n];
    if (!drives(sh->flags)) return;
    MdNav* nav = &s->nav[n];
    int m = w->env_map[e];
    int l0 = w->lane_off[m], l1 = w->lane_off[m + 1];
    const MdLane* lanes = w->lanes + l0;
    int nl = l1 - l0;
    const int32_t* rroads = s->route_roads + (size_t)n * MD_ROUTE_LEN;
    const int32_t* rnodes = s->route_nodes + (size_t)n * MD_ROUTE_LEN;
    int cur_road = rroads[nav->ck0];
    int has_next = (nav->ck1 != nav->ck0);
    int next_road = has_next ? rroads[nav->ck1] : -1;

    int on_lane = 0;
    int best_any = -1, best_cur = -1, best_next = -1;
    float d_any = 3.0e38f, d_cur = 3.0e38f, d_next = 3.0e38f;
    for (int l = 0; l < nl; ++l) {
        const MdLane* L = &lanes[l];
        if (sh->cx < L->x0 || sh->cx > L->x1 || sh->cy < L->y0 || sh->cy > L->y1) continue;
        if (!md_point_in_hull(sh->cx, sh->cy, md_lane_hull(L, w->hull_xy), L->hull_n)) continue;
        on_lane = 1;
        float ls, llat;
        md_lane_local(L, sh->cx, sh->cy, &ls, &llat);
        if (!(md_lane_heading_dot(L, sh->cx, sh->cy, sh->c, sh->s) > 0.0f)) continue;
        float dist = md_lane_distance(L, ls, llat);
        if (dist < d_any) { d_any = dist; best_any = l; }
        if (L->road == cur_road && dist < d_cur) { d_cur = dist; best_cur = l; }
        if (has_next && L->road == next_road && dist < d_next) { d_next = dist; best_next = l; }
    }
    int lane = -1;
    if (best_cur >= 0) lane = best_cur;
    else if (!has_next) lane = best_any;
    else if (best_next >= 0) lane = best_next;
    else lane = best_any;
    uint32_t fl = s->flags[n] & ~(uint32_t)MD_FL_ON_LANE;
    if (on_lane) fl |= MD_FL_ON_LANE;
    s->flags[n] = fl;
    if (lane < 0) lane = nav->lane; /* keep the previous lane (node_network_navigation.py:297-298) */
    nav->lane = lane;
    if (lane < 0) return;
    /* _update_target_checkpoints */
    if (nav->ck0 == nav->ck1) return;
    float ls, llat;
    md_lane_local(&lanes[lane], sh->cx, sh->cy, &ls, &llat);
    if (!(ls < 5.0f)) return; /* CKPT_UPDATE_RANGE (base_navigation.py:23) */
    int start_node = w->roads[w->road_off[m] + lanes[lane].road].start_node;
    int k = nav->route_len;
    int idx = -1;
    for (int j = nav->ck1; j < k - 1; ++j) { /* checkpoints[ck1:-1] */
        if (rnodes[j] == start_node) { idx = j; break; }
    }
    if (idx < 0) return;
    nav->ck0 = idx;
    nav->ck1 = (idx + 1 == k - 1) ? idx : idx + 1;
    nav->road0 = rroads[nav->ck0]; /* the cached ids of current_ref_lanes / next_ref_lanes' roads follow the cursors */
    nav->road1 = rroads[nav->ck1];
}

EXPORT int ref_localize(const MdWorld* w, const MdState* s, const MdConfig* c) {
    for (int e = 0; e < c->n_envs; ++e)
        for (int j = 0; j < c->cap; ++j) localize_mover(w, s, c, e, e * c->cap + j);
    return MD_OK;
}

/* ------------------------------------------------------------------------------------------
 * Contacts: BaseVehicle._state_check (base_vehicle.py:700-767) + collision_callback.
 * -----------------------------------------------------------------------------------------*/
static void contacts_mover(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int slot) {
    int base = e * c->cap;
    int n = base + slot;
    const MdShape* me = &s->shape[n];
    if (!drives(me->flags)) return;
    if (!(me->flags & MD_F_AGENT)) {
        /* traffic: BaseVehicle._state_check runs for every vehicle in the reference, but nothing ever reads a
         * traffic vehicle's crash / line flags (IDM, rewards, infos and removal only use on_lane): not computed */
        s->flags[n] &= MD_FL_ON_LANE;
        return;
    }
    uint32_t keep = s->flags[n] & (MD_FL_ON_LANE);
    uint32_t fl = 0;
    for (int j = 0; j < c->cap; ++j) {
        if (j == slot) continue;
        const MdShape* o = &s->shape[base + j];
        if (!present(o->flags)) continue;
        int k = kind_of(o->flags);
        int hit;
        if (is_circle_kind(k)) hit = md_obb_circle(me->cx, me->cy, me->c, me->s, me->hl, me->hw, o->cx, o->cy, o->hl);
        else hit = md_obb_obb(me->cx, me->cy, me->c, me->s, me->hl, me->hw, o->cx, o->cy, o->c, o->s, o->hl, o->hw);
        if (!hit) continue;
        if (k == MD_KIND_VEHICLE) fl |= MD_FL_CRASH_VEHICLE;
        else if (k == MD_KIND_CONE || k == MD_KIND_WARNING || k == MD_KIND_BARRIER) fl |= MD_FL_CRASH_OBJECT;
        else if (k == MD_KIND_PEDESTRIAN || k == MD_KIND_CYCLIST) fl |= MD_FL_CRASH_HUMAN;
        else if (k == MD_KIND_BUILDING) fl |= MD_FL_CRASH_BUILDING; /* base_vehicle.py:737-738, collision_callback.py:40-41 */
    }
    int m = w->env_map[e];
    for (int q = w->quad_off[m]; q < w->quad_off[m + 1]; ++q) {
        if (!md_obb_quad(me->cx, me->cy, me->c, me->s, me->hl, me->hw, w->quads + 8 * (size_t)q)) continue;
        switch (w->quad_kind[q]) {
            case MD_Q_LINE_WHITE_CONT: fl |= MD_FL_ON_WHITE_CONT; break;
            case MD_Q_LINE_YELLOW_CONT: fl |= MD_FL_ON_YELLOW_CONT; break;
            case MD_Q_LINE_BROKEN: fl |= MD_FL_ON_BROKEN; break;
            case MD_Q_SIDEWALK: fl |= MD_FL_CRASH_SIDEWALK; break;
            case MD_Q_CROSSWALK: fl |= MD_FL_ON_CROSSWALK; break;
            default: break;
        }
    }
    s->flags[n] = keep | fl;
}

EXPORT int ref_contacts(const MdWorld* w, const MdState* s, const MdConfig* c) {
    for (int e = 0; e < c->n_envs; ++e)
        for (int j = 0; j < c->cap; ++j) contacts_mover(w, s, c, e, j);
    return MD_OK;
}

/* PGTrafficManager.after_step (manager/traffic_manager.py:94-122), trigger mode: traffic that left
 * every lane hull is removed. */
EXPORT int ref_traffic_after_step(const MdWorld* w, const MdState* s, const MdConfig* c) {
    for (int n = 0; n < c->n_envs * c->cap; ++n) {
        MdShape* sh = &s->shape[n];
        if (!drives(sh->flags) || (sh->flags & MD_F_AGENT)) continue;
        if (!(s->flags[n] & MD_FL_ON_LANE)) sh->flags &= ~MD_F_ALIVE;
    }
    if (c->traffic_mode == 1 || c->traffic_mode == 2)
        for (int e = 0; e < c->n_envs; ++e) {
            MdState v = md_env_view(s, c, e);
            md_traffic_respawn_env(w, w->lanes + w->lane_off[w->env_map[e]], &v, c, w->env_map[e]);
        }
    return MD_OK;
}

/* ------------------------------------------------------------------------------------------
 * Observation + reward + cost + done (obs/state_obs.py:64-151; node_network_navigation.py:130-179,
 * 243-292; base_vehicle.py:491-499; envs/metadrive_env.py:128-279; envs/base_env.py:586-623).
 * -----------------------------------------------------------------------------------------*/
EXPORT int ref_observe(const MdWorld* w, const MdState* s, const MdConfig* c) {
    for (int e = 0; e < c->n_envs; ++e)
        for (int a = 0; a < c->agents_per_env; ++a) {
            MdState v = md_env_view(s, c, e);
            md_observe_agent(w->lanes + w->lane_off[w->env_map[e]], w->roads + w->road_off[w->env_map[e]], &v, c, a, 0);
        }
    return MD_OK;
}

/* ------------------------------------------------------------------------------------------
 * IDM: PGTrafficManager.before_step trigger (manager/traffic_manager.py:74-92) + IDMPolicy.act
 * (policy/idm_policy.py:235-402) with FrontBackObjects (policy/idm_policy.py:82-132).
 * Objects are visited in slot order (the reference iterates a Python set; order there is
 * arbitrary, here it is canonical).  The bare `except:` fallback (idm_policy.py:254-260) is
 * modelled by the `fail` paths.
 * -----------------------------------------------------------------------------------------*/
static void idm_env(const MdWorld* w, const MdState* s, const MdConfig* c, int e) {
    int base = e * c->cap;
    int m = w->env_map[e];
    const MdLane* lanes = w->lanes + w->lane_off[m];
    /* trigger (traffic_manager.py:80-88): the pending block with the smallest order fires when an
     * agent's current road is its trigger road */
    int min_order = 0x7fffffff;
    for (int j = 0; j < c->cap; ++j) {
        const MdShape* o = &s->shape[base + j];
        if ((o->flags & MD_F_PENDING) && (o->flags & MD_F_ALIVE) && s->nav[base + j].trigger_order < min_order)
            min_order = s->nav[base + j].trigger_order;
    }
    if (min_order != 0x7fffffff) {
        int trig_road = -1;
        for (int j = 0; j < c->cap; ++j) {
            const MdShape* o = &s->shape[base + j];
            if ((o->flags & MD_F_PENDING) && (o->flags & MD_F_ALIVE) && s->nav[base + j].trigger_order == min_order) {
                trig_road = s->nav[base + j].trigger_road;
                break;
            }
        }
        int fire = 0;
        for (int a = 0; a < c->agents_per_env; ++a) {
            const MdShape* ag = &s->shape[base + a];
            if (!drives(ag->flags)) continue;
            int al = s->nav[base + a].lane;
            if (al >= 0 && lanes[al].road == trig_road) fire = 1;
        }
        if (fire)
            for (int j = 0; j < c->cap; ++j) {
                MdShape* o = &s->shape[base + j];
                if ((o->flags & MD_F_PENDING) && (o->flags & MD_F_ALIVE) && s->nav[base + j].trigger_order == min_order)
                    o->flags &= ~MD_F_PENDING;
            }
    }
    /* Decisions use the state at the start of the step for every vehicle: actions are written to
     * s->action, poses are not touched here, so slot order does not matter. */
    /* agent_policy = IDMPolicy: the agents take their action from the same policy (agent_manager.py:37-70) */
    for (int j = c->agent_idm ? 0 : c->agents_per_env; j < c->cap; ++j) {
        const MdShape* o = &s->shape[base + j];
        if (!drives(o->flags) || ((o->flags & MD_F_AGENT) && !c->agent_idm)) continue;
        MdState v = md_env_view(s, c, e);
        md_idm_vehicle(w, &v, c, e, j);
    }
}

EXPORT int ref_idm(const MdWorld* w, const MdState* s, const MdConfig* c) {
    for (int e = 0; e < c->n_envs; ++e) idm_env(w, s, c, e);
    return MD_OK;
}

/* ------------------------------------------------------------------------------------------
 * Whole step for a range of envs: BaseEnv.step (envs/base_env.py:426-463,586-623).
 * -----------------------------------------------------------------------------------------*/
static void step_env(const MdWorld* w, const MdState* s, const MdConfig* c, int e) {
    int base = e * c->cap;
    int just_reset = 0;
    if (s->agent_action && !c->agent_idm) /* this step's agent actions come from the caller's own buffer */
        for (int a = 0; a < c->agents_per_env; ++a) {
            s->action[2 * (base + a)] = s->agent_action[2 * (e * c->agents_per_env + a)];
            s->action[2 * (base + a) + 1] = s->agent_action[2 * (e * c->agents_per_env + a) + 1];
        }
    if (s->need_reset[e]) {
        memcpy(&s->shape[base], &s->shape0[base], sizeof(MdShape) * c->cap);
        memcpy(&s->dyn[base], &s->dyn0[base], sizeof(MdDyn) * c->cap);
        memcpy(&s->nav[base], &s->nav0[base], sizeof(MdNav) * c->cap);
        memcpy(&s->pid[base], &s->pid0[base], sizeof(MdPid) * c->cap);
        for (int j = 0; j < c->cap; ++j) {
            s->flags[base + j] = 0;
            s->action[2 * (base + j)] = 0.0f;
            s->action[2 * (base + j) + 1] = 0.0f;
        }
        if (c->is_multi_agent || c->traffic_mode == 1 || c->traffic_mode == 2) { /* routes are rewritten by respawns */
            memcpy(&s->route_nodes[(size_t)base * MD_ROUTE_LEN], &s->route_nodes0[(size_t)base * MD_ROUTE_LEN], sizeof(int32_t) * MD_ROUTE_LEN * c->cap);
            memcpy(&s->route_roads[(size_t)base * MD_ROUTE_LEN], &s->route_roads0[(size_t)base * MD_ROUTE_LEN], sizeof(int32_t) * MD_ROUTE_LEN * c->cap);
            memcpy(&s->final_lane[base], &s->final_lane0[base], sizeof(int32_t) * c->cap);
        }
        if (c->is_multi_agent && c->random_agent_model && s->param0) /* respawns draw new vehicle classes */
            memcpy(&s->param[base], &s->param0[base], sizeof(MdParam) * c->cap);
        if (c->is_multi_agent) {
            s->env_steps[e] = 0;
            /* names agent0 .. agent{n-1} are taken by the agents present at reset (all slots, or with num_agents = -1
             * one per spawn point: the other slots start free) */
            int n0 = 0;
            for (int j = 0; j < c->agents_per_env; ++j) n0 += (s->shape0[base + j].flags & MD_F_ALIVE) ? 1 : 0;
            s->next_agent_id[e] = n0;
            for (int j = 0; j < c->cap; ++j) s->agent_id[base + j] = j;
        }
        s->need_reset[e] = 0;
        just_reset = 1;
    }
    if (c->is_multi_agent && !just_reset) {
        MdState v = md_env_view(s, c, e);
        md_lifecycle_env(w, &v, c, w->env_map[e]);
    }
    /* Single-agent envs plan the traffic one step ahead: IDMPolicy.act of step t+1 sees exactly the state the
     * end of step t leaves (traffic_manager.before_step runs before anything moves), so taking the decision at
     * the end of step t gives the same trajectories; the HIP kernel uses that to overlap it with the agent's
     * observation.  Multi-agent envs keep the reference order (respawns at the start of a step come first). */
    /* With IDM-driven agents the reference order is kept as well: the observation reports the action applied in THIS
     * step, which planning ahead would overwrite with the next one before the observation is assembled. */
    const int plan_ahead = !c->is_multi_agent && !c->agent_idm;
    if (!just_reset) {
        if (!plan_ahead) idm_env(w, s, c, e);
        MdState v = md_env_view(s, c, e);
        for (int j = 0; j < c->cap; ++j) md_advance_mover(&v, c, j);
    }
    for (int j = 0; j < c->cap; ++j) localize_mover(w, s, c, e, base + j);
    for (int j = 0; j < c->cap; ++j) contacts_mover(w, s, c, e, j);
    for (int j = 0; j < c->cap; ++j) {
        MdShape* sh = &s->shape[base + j];
        if (!drives(sh->flags) || (sh->flags & MD_F_AGENT)) continue;
        if (!(s->flags[base + j] & MD_FL_ON_LANE)) sh->flags &= ~MD_F_ALIVE;
    }
    if (c->traffic_mode == 1 || c->traffic_mode == 2) {
        MdState v = md_env_view(s, c, e);
        md_traffic_respawn_env(w, w->lanes + w->lane_off[w->env_map[e]], &v, c, w->env_map[e]);
    }
    if (plan_ahead) idm_env(w, s, c, e);
    for (int a = 0; a < c->agents_per_env; ++a) {
        MdState v = md_env_view(s, c, e);
        md_observe_agent(w->lanes + w->lane_off[w->env_map[e]], w->roads + w->road_off[w->env_map[e]], &v, c, a, just_reset);
        if (c->n_beams > 0) {
            float* row = s->obs + (size_t)(e * c->agents_per_env + a) * c->obs_dim;
            unsigned long long det[2];
            lidar_agent_det(w, s, c, e, a, row + md_obs_lidar(c), det);
            if (s->detected) {
                s->detected[2 * (size_t)(e * c->agents_per_env + a)] = det[0];
                s->detected[2 * (size_t)(e * c->agents_per_env + a) + 1] = det[1];
            }
            if (c->num_others > 0)
                md_others_block(w->lanes + w->lane_off[w->env_map[e]], w->roads + w->road_off[w->env_map[e]], &v, c, a,
                                det[0], det[1], row + md_obs_others(c));
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Scenario mode (traffic_mode 4): one ScenarioEnv step (envs/scenario_env.py + manager/scenario_traffic_manager.py),
 * in the reference's order: reactive traffic decides (before_step), everything driven moves, the traffic manager's
 * after_step replays / removes / spawns at frame k, then the agent's contacts, navigation, observation, lidar.
 * -----------------------------------------------------------------------------------------*/
static void step_env_scenario(const MdWorld* w, const MdState* s, const MdConfig* c, int e) {
    const int base = e * c->cap;
    int just_reset = 0;
    if (s->agent_action)
        for (int a = 0; a < c->agents_per_env; ++a) {
            s->action[2 * (base + a)] = s->agent_action[2 * (e * c->agents_per_env + a)];
            s->action[2 * (base + a) + 1] = s->agent_action[2 * (e * c->agents_per_env + a) + 1];
        }
    if (s->need_reset[e]) {
        memcpy(&s->shape[base], &s->shape0[base], sizeof(MdShape) * c->cap);
        memcpy(&s->dyn[base], &s->dyn0[base], sizeof(MdDyn) * c->cap);
        memcpy(&s->nav[base], &s->nav0[base], sizeof(MdNav) * c->cap);
        memcpy(&s->pid[base], &s->pid0[base], sizeof(MdPid) * c->cap);
        for (int j = 0; j < c->cap; ++j) {
            s->flags[base + j] = 0;
            s->action[2 * (base + j)] = 0.0f;
            s->action[2 * (base + j) + 1] = 0.0f;
        }
        s->next_agent_id[e] = 0;   /* idm_policy_count (scenario_traffic_manager.py:87) */
        if (s->route_n) memset(&s->route_n[4 * (size_t)base], 0, sizeof(int32_t) * 4 * c->cap);   /* no cut routes left */
        s->need_reset[e] = 0;
        just_reset = 1;
    }
    MdState v = md_env_view(s, c, e);
    const int k = just_reset ? 0 : v.nav[0].steps + 1;   /* engine.episode_step of this step */
    if (!just_reset) {
        for (int j = c->agents_per_env; j < c->cap; ++j)
            if (v.nav[j].ck0 == MD_SC_IDM && md_present(v.shape[j].flags)) md_tidm_vehicle(w, &v, c, e, j, k);
        for (int j = 0; j < c->cap; ++j) {
            if (c->ego_replay && j < c->agents_per_env) md_scenario_replay_ego(&v, c, j, k);
            else md_integrate_mover(&v, c, j);
        }
    }
    /* the agent's contact flags come from BaseVehicle.after_step (_state_check: a contact test at the bodies' present poses), which the
     * agent manager runs BEFORE the traffic manager's after_step (same priority, registered first: envs/scenario_env.py:118-126):
     * replayed bodies are still at frame k-1, bodies removed / spawned in this step are still / not yet there */
    for (int a = 0; a < c->agents_per_env; ++a) {
        s->flags[base + a] = 0;
        contacts_mover(w, s, c, e, a);
    }
    md_scenario_after_step_env(w, &v, c, e, k);
    for (int a = 0; a < c->agents_per_env; ++a) {
        md_scenario_observe(w, &v, c, e, a, just_reset);
        if (c->n_beams > 0) {
            float* row = s->obs + (size_t)(e * c->agents_per_env + a) * c->obs_dim;
            lidar_agent(w, s, c, e, a, row + md_sc_obs_lidar(c));
        }
    }
}

/* md_build_route on a point list (tests: against the reference's PointLane) */
EXPORT int ref_build_route(const float* xy, int n_pts, MdSeg* segs, int seg_cap, float* verts, int vert_cap, float* aux,
                           int32_t* counts) {
    return md_build_route(xy, 2, n_pts, segs, seg_cap, verts, vert_cap, aux, counts);
}

EXPORT int ref_lifecycle(const MdWorld* w, const MdState* s, const MdConfig* c) {
    for (int e = 0; e < c->n_envs; ++e) {
        MdState v = md_env_view(s, c, e);
        md_lifecycle_env(w, &v, c, w->env_map[e]);
    }
    return MD_OK;
}

static void step_any(const MdWorld* w, const MdState* s, const MdConfig* c, int e) {
    if (c->traffic_mode == 4) step_env_scenario(w, s, c, e);
    else step_env(w, s, c, e);
}

EXPORT int ref_step(const MdWorld* w, const MdState* s, const MdConfig* c) {
    for (int e = 0; e < c->n_envs; ++e) step_any(w, s, c, e);
    return MD_OK;
}

typedef struct {
    const MdWorld* w;
    const MdState* s;
    const MdConfig* c;
    int e0, e1;
} StepJob;

static void* step_worker(void* p) {
    StepJob* j = (StepJob*)p;
    for (int e = j->e0; e < j->e1; ++e) step_any(j->w, j->s, j->c, e);
    return 0;
}

/* CPU baseline with all host cores: envs are independent, so a static partition is exact. */
EXPORT int ref_step_mt(const MdWorld* w, const MdState* s, const MdConfig* c, int n_threads) {
    if (n_threads <= 1) return ref_step(w, s, c);
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    StepJob jobs[256];
    int per = (c->n_envs + n_threads - 1) / n_threads;
    int used = 0;
    for (int t = 0; t < n_threads; ++t) {
        int e0 = t * per, e1 = e0 + per;
        if (e0 >= c->n_envs) break;
        if (e1 > c->n_envs) e1 = c->n_envs;
        jobs[t].w = w; jobs[t].s = s; jobs[t].c = c; jobs[t].e0 = e0; jobs[t].e1 = e1;
        pthread_create(&th[t], 0, step_worker, &jobs[t]);
        used++;
    }
    for (int t = 0; t < used; ++t) pthread_join(th[t], 0);
    return MD_OK;
}

/* Scalar probes of the shared formulas, for the golden-vector tests (tests/test_oracle_golden.py). */
EXPORT void ref_lane_local(const MdLane* L, float x, float y, float* out2) { md_lane_local(L, x, y, &out2[0], &out2[1]); }
EXPORT float ref_lane_heading_at(const MdLane* L, float s) { return md_lane_heading_at(L, s); }
EXPORT float ref_heading_diff(const MdLane* L, float x, float y, float hc, float hs) { return md_heading_diff(L, x, y, hc, hs); }
EXPORT void ref_navi(const MdLane* ref, float later_middle, float x, float y, float hc, float hs, float n_cur, float w_cur,
                     float* out5) {
    md_navi_for_checkpoint(ref, later_middle, x, y, hc, hs, n_cur, w_cur, 60.0f, 135.0f, out5);
}
EXPORT float ref_idm_acc(float v, float target, int has_front, float dist, float dv) {
    return md_idm_acceleration(v, target, has_front, dist, dv);
}
EXPORT float ref_pid(float* st3, float kp, float ki, float kd, float err) { return md_pid(&st3[0], &st3[1], &st3[2], kp, ki, kd, err); }
EXPORT float ref_wrap_to_pi(float x) { return md_wrap_to_pi(x); }
EXPORT void ref_sincos(float x, float* out2) { md_sincos(x, &out2[0], &out2[1]); }
EXPORT float ref_atan2(float y, float x) { return md_atan2(y, x); }
EXPORT float ref_acos(float x) { return md_acos(x); }
EXPORT float ref_sanitize(float a) { return md_sanitize(a); }
EXPORT float ref_ray_shape(float ox, float oy, float dx, float dy, const MdShape* o) {
    return md_ray_shape(ox, oy, dx, dy, o->cx, o->cy, o->c, o->s, o->hl, o->hw, kind_of(o->flags));
}
EXPORT int ref_obb_obb(const MdShape* a, const MdShape* b) {
    return md_obb_obb(a->cx, a->cy, a->c, a->s, a->hl, a->hw, b->cx, b->cy, b->c, b->s, b->hl, b->hw);
}
EXPORT int ref_obb_quad(const MdShape* a, const float* q) { return md_obb_quad(a->cx, a->cy, a->c, a->s, a->hl, a->hw, q); }
/* st5 = x, y, psi, v, yaw increment per sub-step carried in from the step before */
EXPORT void ref_bicycle(float* st4, float steer, float thr, const MdParam* P, float dt, int n) {
    MdBicycle b;
    md_bicycle_prepare(steer, thr, st4[3], 2.2575f, 0.926f, dt, 0, P, &b);
    float c0, s0;
    md_sincos(st4[2], &s0, &c0);
    float cp = c0 * b.cb - s0 * b.sb, sp = s0 * b.cb + c0 * b.sb;
    for (int i = 0; i < n; ++i) md_bicycle_substep(&st4[0], &st4[1], &st4[2], &st4[3], &cp, &sp, &st4[4], thr, &b, P, dt);
}
EXPORT void ref_probe_math(int op, const float* a, const float* b, float* out, int n) {
    for (int i = 0; i < n; ++i) out[i] = md_probe_eval(op, a[i], b[i]);
}
EXPORT float ref_idm_gap(float v_kmh, float dv_kmh) { return md_idm_desired_gap(v_kmh, dv_kmh); }
/* IDMPolicy.steering_control (policy/idm_policy.py:293-301); pid6 = heading p,i,d then lateral p,i,d */
EXPORT float ref_idm_steer(const MdLane* L, float x, float y, float heading, float* pid6) {
    float s_, lat;
    md_lane_local(L, x, y, &s_, &lat);
    float lane_heading = md_lane_heading_at(L, s_ + 1.0f);
    float st = md_pid(&pid6[0], &pid6[1], &pid6[2], 1.7f, 0.01f, 3.5f, -md_wrap_to_pi(lane_heading - heading));
    st += md_pid(&pid6[3], &pid6[4], &pid6[5], 0.3f, 0.002f, 0.05f, -lat);
    return st;
}
/* FrontBackObjects.get_find_front_back_objs on a hand-made scene: out = front[3], back[3]; dist = front_d[3], back_d[3] */
EXPORT void ref_front_back(const MdLane* lanes, MdShape* shapes, MdNav* navs, int cap, int self_slot, const int* ids3,
                           float px, float py, int* out6, float* dist6) {
    MdState s;
    memset(&s, 0, sizeof s);
    s.shape = shapes;
    s.nav = navs;
    MdConfig c;
    memset(&c, 0, sizeof c);
    c.cap = cap;
    MdIdmPlan p;
    p.success = 1; p.use_ref = 1; p.fail = 0;
    p.ids[0] = ids3[0]; p.ids[1] = ids3[1]; p.ids[2] = ids3[2];
    FrontBack fb;
    md_find_front_back(&s, &c, lanes, self_slot, &p, px, py, &fb);
    for (int i = 0; i < 3; ++i) {
        out6[i] = fb.front[i]; out6[3 + i] = fb.back[i];
        dist6[i] = fb.front_d[i]; dist6[3 + i] = fb.back_d[i];
    }
}
/* IDMPolicy.act (policy/idm_policy.py:235-267) of ONE vehicle on a hand-made scene: one map, one env of `cap` slots.
 * node_adj_off / node_adj: the road graph (MdWorld layout, n_nodes + 1 offsets).  In / out: navs (target lane, timer,
 * rand cursor), pids (target speed, PID state), actions[2 * slot ..] = (steering, acceleration). */
EXPORT void ref_idm_vehicle(const MdLane* lanes, int n_lanes, const MdRoad* roads, int n_roads, const int32_t* node_adj_off,
                            const int32_t* node_adj, MdShape* shapes, MdDyn* dyns, MdNav* navs, MdPid* pids, float* actions,
                            int32_t* route_roads, const int32_t* idm_rand, int cap, int slot, int enable_lane_change) {
    int32_t env_map = 0, lane_off[2] = {0, n_lanes}, road_off[2] = {0, n_roads}, node_off[2] = {0, 0};
    MdWorld w;
    memset(&w, 0, sizeof w);
    w.n_maps = 1; w.n_envs = 1;
    w.env_map = &env_map; w.lane_off = lane_off; w.lanes = lanes; w.road_off = road_off; w.roads = roads;
    w.node_off = node_off; w.node_adj_off = node_adj_off; w.node_adj = node_adj;
    MdState s;
    memset(&s, 0, sizeof s);
    s.shape = shapes; s.dyn = dyns; s.nav = navs; s.pid = pids; s.action = actions; s.route_roads = route_roads;
    s.idm_rand = idm_rand;
    MdConfig c;
    memset(&c, 0, sizeof c);
    c.n_envs = 1; c.cap = cap; c.agents_per_env = 1; c.enable_idm_lane_change = enable_lane_change;
    md_idm_vehicle(&w, &s, &c, 0, slot);
}
/* scenario-mode probes (tests/test_scenario.py): polyline of n segments */
EXPORT void ref_poly_local(const MdSeg* segs, int n, float px, float py, float* out4) {
    MdPoly p;
    p.segs = segs; p.n = n; p.length = segs[n - 1].cum + segs[n - 1].len;
    MdTrajLoc L;
    md_traj_locate(&p, px, py, &L);
    out4[0] = L.lng; out4[1] = L.lat; out4[2] = L.heading_at; out4[3] = p.length;
}
EXPORT void ref_poly_position(const MdSeg* segs, int n, float s_, float lateral, float* out2) {
    MdPoly p;
    p.segs = segs; p.n = n; p.length = segs[n - 1].cum + segs[n - 1].len;
    md_poly_position(&p, s_, lateral, &out2[0], &out2[1]);
}
EXPORT void ref_traj_navi(const MdSeg* segs, int n, const float* ckpt, int n_ckpt, float px, float py, float heading,
                          float max_lateral_dist, float* out22) {
    MdPoly p;
    p.segs = segs; p.n = n; p.length = segs[n - 1].cum + segs[n - 1].len;
    MdTrajLoc L;
    md_traj_locate(&p, px, py, &L);
    float sn, cs;
    md_sincos(heading, &sn, &cs);
    md_traj_navi(ckpt, n_ckpt, &L, px, py, cs, sn, heading, max_lateral_dist, out22, 1);
}
/* md_scenario_observe of every env's agent on the state as it stands (flags as given: no contact phase) */
EXPORT int ref_scenario_observe(const MdWorld* w, const MdState* s, const MdConfig* c) {
    for (int e = 0; e < c->n_envs; ++e) {
        MdState v = md_env_view(s, c, e);
        md_scenario_observe(w, &v, c, e, 0, 0);
    }
    return MD_OK;
}
/* probe: the "others" block of agent a of env e for a given detected set (Lidar.get_surrounding_vehicles_info) */
EXPORT int ref_others_block(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int a, uint64_t det_lo, uint64_t det_hi,
                            float* out) {
    MdState v = md_env_view(s, c, e);
    const int m = w->env_map[e];
    md_others_block(w->lanes + w->lane_off[m], w->roads + w->road_off[m], &v, c, a, det_lo, det_hi, out);
    return MD_OK;
}

/* TrajectoryIDMPolicy.act of one slot at episode step k */
EXPORT int ref_tidm_vehicle(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int slot, int k) {
    MdState v = md_env_view(s, c, e);
    md_tidm_vehicle(w, &v, c, e, slot, k);
    return MD_OK;
}
/* ScenarioTrafficManager.after_step of env e at episode step k alone (tests: against the reference's spawn_vehicle) */
EXPORT int ref_scenario_after_step(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int k) {
    MdState v = md_env_view(s, c, e);
    md_scenario_after_step_env(w, &v, c, e, k);
    return 0;
}

EXPORT int ref_point_in_polygon(const float* xy, int n, float px, float py) { return md_point_in_polygon(xy, n, px, py); }
EXPORT int ref_abi(int32_t* sizes, int n) {
    int32_t v[11] = {sizeof(MdShape), sizeof(MdDyn), sizeof(MdParam), sizeof(MdNav), sizeof(MdPid), sizeof(MdLane),
                     sizeof(MdRoad), sizeof(MdGrid), sizeof(MdWorld), sizeof(MdState), sizeof(MdConfig)};
    for (int i = 0; i < n && i < 11; ++i) sizes[i] = v[i];
    return MD_ABI_VERSION;
}

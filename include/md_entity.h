/*
 * md_entity.h -- per-entity (one vehicle / one agent) scalar step logic: action sanitising +
 * kinematic integration, observation/reward/done assembly, IDM decision.  These routines are
 * inherently serial per entity; the HIP kernels run one GPU thread per entity over them and the
 * CPU oracle loops over entities, so both share this one spelling (same rationale as md_geom.h).
 * The phases whose GPU form is genuinely different -- lidar (wave per agent-sector with ballot
 * culling), lane localisation and contact tests (wave per vehicle over grid cells), traffic
 * trigger (wave min-reduce) -- are written twice: brute force in oracle/md_oracle.c, wave-parallel
 * in metadrive_ped_amd/csrc/mdstep.hip.
 *
 * Indexing convention: every routine here receives an ENV-LOCAL view of MdState (md_env_view): the
 * pointers are already advanced to the env's first mover / first agent, so slot j is s->shape[j],
 * agent a's observation row is s->obs[a * obs_dim], and the env's reset flag is s->need_reset[0].
 * The oracle makes the view by pointer offset into the global arrays; the fused HIP kernel points
 * it at the env's LDS-staged copies.
 *
 * Reference citations are relative to /root/reference/metadrive.
 */
#ifndef MD_ENTITY_H
#define MD_ENTITY_H

#include "md_geom.h"

/* env-local view of the global state arrays (see header comment) */
MD_HD MdState md_env_view(const MdState* g, const MdConfig* c, int e) {
    MdState v = *g;
    const size_t b = (size_t)e * (size_t)c->cap;
    const size_t a = (size_t)e * (size_t)c->agents_per_env;
    v.shape = g->shape + b;
    v.dyn = g->dyn ? g->dyn + b : 0;
    v.param = g->param ? g->param + b : 0;
    v.nav = g->nav ? g->nav + b : 0;
    v.pid = g->pid ? g->pid + b : 0;
    v.action = g->action ? g->action + 2 * b : 0;
    v.route_nodes = g->route_nodes ? g->route_nodes + b * MD_ROUTE_LEN : 0;
    v.route_roads = g->route_roads ? g->route_roads + b * MD_ROUTE_LEN : 0;
    v.final_lane = g->final_lane ? g->final_lane + b : 0;
    v.idm_rand = g->idm_rand ? g->idm_rand + b * MD_IDM_RAND : 0;
    v.flags = g->flags ? g->flags + b : 0;
    v.obs = g->obs ? g->obs + a * (size_t)c->obs_dim : 0;
    v.reward = g->reward ? g->reward + a : 0;
    v.cost = g->cost ? g->cost + a : 0;
    v.step_info = g->step_info ? g->step_info + a * 8 : 0;
    v.need_reset = g->need_reset ? g->need_reset + e : 0;
    v.shape0 = g->shape0 ? g->shape0 + b : 0;
    v.dyn0 = g->dyn0 ? g->dyn0 + b : 0;
    v.nav0 = g->nav0 ? g->nav0 + b : 0;
    v.pid0 = g->pid0 ? g->pid0 + b : 0;
    v.param0 = g->param0 ? g->param0 + b : 0;
    v.done_out = g->done_out ? g->done_out + 4 * a : 0;
    v.idle_ring = g->idle_ring ? g->idle_ring + a * MD_IDLE_WINDOW : 0;
    v.route_nodes0 = g->route_nodes0 ? g->route_nodes0 + b * MD_ROUTE_LEN : 0;
    v.route_roads0 = g->route_roads0 ? g->route_roads0 + b * MD_ROUTE_LEN : 0;
    v.final_lane0 = g->final_lane0 ? g->final_lane0 + b : 0;
    v.rng = g->rng ? g->rng + e : 0;
    v.env_steps = g->env_steps ? g->env_steps + e : 0;
    v.agent_id = g->agent_id ? g->agent_id + b : 0;
    v.next_agent_id = g->next_agent_id ? g->next_agent_id + e : 0;
    v.detected = g->detected ? g->detected + (size_t)e * c->agents_per_env * 2 : 0;
    v.agent_action = g->agent_action ? g->agent_action + (size_t)e * c->agents_per_env * 2 : 0;
    v.track_shape = g->track_shape ? g->track_shape + b : 0;   /* + t * n_envs * cap per frame */
    v.track_dyn = g->track_dyn ? g->track_dyn + 2 * b : 0;
    v.route_n = g->route_n ? g->route_n + 4 * b : 0;
    v.route_segs = g->route_segs ? g->route_segs + b * (size_t)c->route_seg_cap : 0;
    v.route_verts = g->route_verts ? g->route_verts + 2 * b * (size_t)c->route_vert_cap : 0;
    v.route_aux = g->route_aux ? g->route_aux + 8 * b : 0;
    return v;
}

/* observation layout offsets (obs/state_obs.py:64-151) */
MD_HD int md_obs_base(const MdConfig* c) { return c->random_agent_model ? 2 : 0; }               /* [length, width] first */
MD_HD int md_obs_mid(const MdConfig* c) { return md_obs_base(c) + (c->n_side > 0 ? c->n_side : 2); } /* heading_diff ... */
MD_HD int md_obs_ll(const MdConfig* c) { return md_obs_mid(c) + 6; }                             /* lane-line block  */
MD_HD int md_obs_navi(const MdConfig* c) { return md_obs_ll(c) + (c->n_lane_line > 0 ? c->n_lane_line : 1); }
/* "others" block: num_others nearest detected vehicles x 4 dims (+4 with add_others_navi), between navi and cloud */
/* TollGateStateObservation (envs/marl_envs/marl_tollgate.py:62-74) drops the navigation dims; TollGateObservation (:77-110)
 * appends [in toll block, stayed long enough] after the lidar cloud */
MD_HD int md_is_tollgate(const MdConfig* c) { return c->is_multi_agent && c->ma_kind == MD_MA_TOLLGATE; }
MD_HD int md_navi_dims(const MdConfig* c) { return md_is_tollgate(c) ? 0 : 10; }
MD_HD int md_obs_tail(const MdConfig* c) { return md_is_tollgate(c) ? 2 : 0; }
MD_HD int md_obs_others(const MdConfig* c) { return md_obs_navi(c) + md_navi_dims(c); }
MD_HD int md_others_width(const MdConfig* c) { return c->add_others_navi ? 8 : 4; }
MD_HD int md_obs_lidar(const MdConfig* c) { return md_obs_others(c) + (c->num_others > 0 ? c->num_others * md_others_width(c) : 0); }

/* MdState.done_out word of an agent: terminated | truncated << 8 | flag word << 16 */
MD_HD uint32_t md_done_word(uint32_t fl) {
    return ((fl & MD_FL_TERMINATED) ? 1u : 0u) | ((fl & MD_FL_TRUNCATED) ? 0x100u : 0u) | ((fl & 0xFFFFu) << 16);
}
MD_HD int md_kind_of(int flags) { return flags & MD_KIND_MASK; }
MD_HD int md_is_circle_kind(int k) { return k == MD_KIND_CONE || k == MD_KIND_WARNING || k == MD_KIND_PEDESTRIAN; }
/* present: has a body in the world (seen by lidar, can be hit).  Traffic spawned for a block that
 * is not triggered yet is present but parked (manager/traffic_manager.py:230-277 spawns every
 * vehicle at reset; only triggered ones enter _traffic_vehicles and act, :80-91). */
MD_HD int md_present(int flags) { return (flags & MD_F_ALIVE) && md_kind_of(flags) != MD_KIND_NONE; }
/* drives: is stepped this frame (agent, or triggered traffic) */
MD_HD int md_drives(int flags) {
    return md_present(flags) && md_kind_of(flags) == MD_KIND_VEHICLE && !(flags & (MD_F_STATIC | MD_F_PENDING));
}

/* walks: a traffic participant (pedestrian, cyclist) that the user spawned: a kinematic body moving with the
 * world-frame velocity the user set (Pedestrian.set_velocity -> setLinearVelocity, traffic_participants/
 * pedestrian.py:76-103, base_object.py:310-328), seen by lidar and contacts, not driven by any policy. */
MD_HD int md_walks(int flags) {
    const int k = md_kind_of(flags);
    return (flags & MD_F_ALIVE) && !(flags & MD_F_STATIC) && (k == MD_KIND_PEDESTRIAN || k == MD_KIND_CYCLIST);
}
/* moves: the slot's pose can change in a step (what the fused kernel has to write back) */
MD_HD int md_moves(int flags) { return md_drives(flags) || md_walks(flags); }

MD_HD float md_sanitize(float a) { /* utils/math.py:16-26 safe_clip_for_small_array(.., -1, 1) */
    if (a != a) return 0.0f;
    if (a > 3.0e38f) return 1.0f;
    if (a < -3.0e38f) return -1.0f;
    return md_clip(a, -1.0f, 1.0f);
}

/* traffic_mode 3: a non-agent slot takes its pose from the recorded track at the episode step that is being
 * computed (the first agent's step counter + 1).  Vehicles become kinematic bodies (STATIC: seen by lidar and
 * contacts, never driven, localised or removed). */
MD_HD void md_replay_mover(const MdState* s, const MdConfig* c, int n) {
    int t = s->nav[0].steps + 1;
    if (t >= c->track_len) t = c->track_len - 1;
    if (t < 0) return;
    const size_t at = (size_t)t * (size_t)c->n_envs * (size_t)c->cap + (size_t)n;
    const MdShape r = s->track_shape[at];
    MdShape* sh = &s->shape[n];
    MdDyn* d = &s->dyn[n];
    d->last_x = sh->cx;
    d->last_y = sh->cy;
    d->last_c = sh->c;
    d->last_s = sh->s;
    *sh = r;
    if (md_kind_of(r.flags) == MD_KIND_VEHICLE) sh->flags = r.flags | MD_F_STATIC;
    d->heading = s->track_dyn[2 * at];
    d->speed = s->track_dyn[2 * at + 1];
}

MD_HD void md_integrate_mover(const MdState* s, const MdConfig* c, int n);

/* What the step does with slot n between "actions are known" and "poses are new": replay mode moves the
 * non-agent slots along their tracks, everything else integrates.  (Apart from md_integrate_mover so that the
 * trigger-mode kernel does not carry the replay code: it costs 4 % there.) */
MD_HD void md_walk_mover(const MdState* s, const MdConfig* c, int n);

MD_HD void md_advance_mover(const MdState* s, const MdConfig* c, int n) {
    if (c->traffic_mode == 3 && n >= c->agents_per_env) md_replay_mover(s, c, n);
    else {
        md_walk_mover(s, c, n);
        md_integrate_mover(s, c, n);
    }
}

/* participant: MdDyn.steering / .throttle hold its world-frame velocity (vx, vy), m/s */
MD_HD void md_walk_mover(const MdState* s, const MdConfig* c, int n) {
    MdShape* sh = &s->shape[n];
    if (!md_walks(sh->flags)) return;
    MdDyn* d = &s->dyn[n];
    const float step_dt = c->dt * (float)c->substeps;
    d->last_x = sh->cx;
    d->last_y = sh->cy;
    sh->cx = sh->cx + d->steering * step_dt;
    sh->cy = sh->cy + d->throttle * step_dt;
}

MD_HD void md_integrate_mover(const MdState* s, const MdConfig* c, int n) {
    MdShape* sh = &s->shape[n];
    if (!md_drives(sh->flags) || (sh->flags & MD_F_SPAWNED)) return;
    MdDyn* d = &s->dyn[n];
    float steer = md_sanitize(s->action[2 * n]);
    float thr = md_sanitize(s->action[2 * n + 1]);
    s->action[2 * n] = steer;
    s->action[2 * n + 1] = thr;
    /* yaw increment per sub-step at the end of the previous step ~ its mean: sin(heading change) / substeps */
    float yaw = (sh->s * d->last_c - sh->c * d->last_s) / (float)c->substeps;
    d->last_x = sh->cx;
    d->last_y = sh->cy;
    d->last_c = sh->c;
    d->last_s = sh->s;
    d->steering = steer;
    d->throttle = thr;
    float x = sh->cx, y = sh->cy, psi = d->heading, v = d->speed;
    MdBicycle bike;
    md_bicycle_prepare(steer, thr, v, sh->hl, sh->hw, c->dt, c->enable_reverse && (sh->flags & MD_F_AGENT), &s->param[n], &bike);
    float c0, s0;
    md_sincos(psi, &s0, &c0);
    float cp = c0 * bike.cb - s0 * bike.sb, sp = s0 * bike.cb + c0 * bike.sb; /* travel direction psi + beta */
    for (int k = 0; k < c->substeps; ++k) md_bicycle_substep(&x, &y, &psi, &v, &cp, &sp, &yaw, thr, &bike, &s->param[n], c->dt);
    sh->cx = x;
    sh->cy = y;
    d->heading = psi;
    d->speed = v;
    md_sincos(psi, &sh->s, &sh->c);
}


/* The observation needs nine mutually independent geometric evaluations (Frenet transforms against
 * five different lanes, two checkpoint projections, heading difference, yaw rate / energy).  They
 * are split out as "tasks" so that the HIP kernel can run them on nine lanes of a wave at once and
 * combine on one lane, while the oracle simply loops over them: same code, same bits. */
#define MD_OBS_TASKS 9

typedef struct MdObsCtx {
    const MdLane *lane, *ref0, *ref_last, *next0, *fin, *rl;
    float cur_w, cur_n, positive_road;
    int valid;
    int cur_block; /* navigation.current_road.block_ID() */
} MdObsCtx;

MD_HD void md_observe_ctx(const MdLane* lanes, const MdRoad* roads, const MdState* s, int n, MdObsCtx* k) {
    const MdShape* sh = &s->shape[n];
    const MdNav* nav = &s->nav[n];
    k->valid = md_drives(sh->flags) && nav->lane >= 0;
    k->lane = k->ref0 = k->ref_last = k->next0 = k->fin = k->rl = lanes;
    k->cur_w = k->cur_n = k->positive_road = 0.0f;
    k->cur_block = 0;
    if (!k->valid) return;
    const MdRoad* cur_road = &roads[nav->road0];
    k->cur_block = cur_road->block_kind;
    int has_next = nav->ck1 != nav->ck0;
    const MdRoad* next_road = has_next ? &roads[nav->road1] : cur_road;
    k->lane = &lanes[nav->lane];
    k->ref0 = &lanes[cur_road->first_lane];
    k->ref_last = &lanes[cur_road->first_lane + cur_road->n_lanes - 1];
    k->next0 = &lanes[next_road->first_lane];
    k->fin = &lanes[s->final_lane[n]];
    k->cur_w = k->lane->width; /* get_current_lane_width = current_lane.width */
    k->cur_n = (float)cur_road->n_lanes;
    /* reward lane (metadrive_env.py:248-254) */
    k->positive_road = 1.0f;
    if (k->lane->road == nav->road0) k->rl = k->lane;
    else {
        k->rl = k->ref0;
        k->positive_road = cur_road->negative ? -1.0f : 1.0f;
    }
}

MD_HD void md_observe_task(int task, const MdObsCtx* k, const MdState* s, const MdConfig* c, int n, float* out5) {
    const MdShape* sh = &s->shape[n];
    const MdDyn* d = &s->dyn[n];
    float x = sh->cx, y = sh->cy;
    out5[0] = out5[1] = out5[2] = out5[3] = out5[4] = 0.0f;
    if (!k->valid) return;
    /* Grouped so that lanes of a wave running different tasks share code paths: five tasks are the
     * same Frenet transform on different (lane, point) pairs, two are the same checkpoint projection. */
    if (task == 0 || task == 2 || task == 5 || task == 6 || task == 7) {
        const MdLane* L = (task == 0) ? k->ref0 : ((task == 2) ? k->lane : ((task == 5) ? k->fin : k->rl));
        float px = (task == 6) ? d->last_x : x;
        float py = (task == 6) ? d->last_y : y;
        md_lane_local(L, px, py, &out5[0], &out5[1]);
    } else if (task == 3 || task == 4) {
        md_navi_for_checkpoint((task == 3) ? k->ref0 : k->next0, (k->cur_n / 2.0f - 0.5f) * k->cur_w, x, y, sh->c, sh->s,
                               k->cur_n, k->cur_w, c->curve_radius_max, c->curve_angle_max, out5);
    } else if (task == 1) {
        out5[0] = md_heading_diff(k->ref_last, x, y, sh->c, sh->s); /* obs[2] */
    } else if (task == 8) {
        float cosb = (sh->c * d->last_c + sh->s * d->last_s) / (md_norm(sh->c, sh->s) * md_norm(d->last_c, d->last_s));
        float beta = md_acos(md_clip(cosb, 0.0f, 1.0f));
        out5[0] = md_clip(beta / 0.1f, 0.0f, 1.0f); /* obs[7] yaw rate */
        out5[1] = md_step_energy(md_fabs(d->speed) * 3.6f, md_norm(d->last_x - x, d->last_y - y));
    }
}

MD_HD void md_observe_combine(const MdObsCtx* k, const MdState* s, const MdConfig* c, int a, int env_just_reset,
                              const float (*r)[5]) {
    int n = a;
    /* a vehicle (re)spawned at the start of this step only reports its first observation
     * (multi_agent_metadrive.py:190-212: new_obs, reward 0, not terminated) */
    int just_reset = env_just_reset || ((s->shape[n].flags & MD_F_SPAWNED) != 0);
    s->shape[n].flags &= ~MD_F_SPAWNED;
    int ai = a; /* env-local view: agent a of this env */
    float* obs = s->obs + (size_t)ai * c->obs_dim;
    float* info = s->step_info + (size_t)ai * 8;
    MdDyn* d = &s->dyn[n];
    MdNav* nav = &s->nav[n];
    const int o_mid = md_obs_mid(c), o_ll = md_obs_ll(c), o_navi = md_obs_navi(c);
    const int toll = md_is_tollgate(c);
    if (!k->valid) {
        if (s->done_out) ((uint32_t*)s->done_out)[ai] = 0u;
        for (int i = 0; i < md_obs_lidar(c); ++i) obs[i] = 0.0f;
        for (int i = c->obs_dim - md_obs_tail(c); i < c->obs_dim; ++i) obs[i] = 0.0f;
        s->reward[ai] = 0.0f;
        s->cost[ai] = 0.0f;
        for (int i = 0; i < 8; ++i) info[i] = 0.0f;
        return;
    }
    float cur_w = k->cur_w, cur_n = k->cur_n;

    /* dist to left/right of the route (base_vehicle.py:491-499) */
    float lat0 = r[0][1];
    float to_left = lat0 + cur_w / 2.0f;
    float to_right = cur_w * cur_n - to_left;
    uint32_t fl = s->flags[n] & (MD_FL_CRASH_VEHICLE | MD_FL_CRASH_OBJECT | MD_FL_CRASH_HUMAN | MD_FL_CRASH_BUILDING |
                                 MD_FL_CRASH_SIDEWALK | MD_FL_ON_WHITE_CONT | MD_FL_ON_YELLOW_CONT | MD_FL_ON_BROKEN |
                                 MD_FL_ON_CROSSWALK | MD_FL_ON_LANE);
    if (to_right < 0.0f || to_left < 0.0f) fl |= MD_FL_OUT_OF_ROUTE;

    /* ---- state obs (obs/state_obs.py:64-151) ---- */
    float speed_kmh = md_fabs(d->speed) * 3.6f;
    const MdParam* P = &s->param[n];
    const int o_base = md_obs_base(c);
    if (o_base) { /* random_agent_model: the vehicle's own size (LENGTH / MAX_LENGTH, WIDTH / MAX_WIDTH; state_obs.py:70-75) */
        obs[0] = md_clip(2.0f * s->shape[n].hl / 10.0f, 0.0f, 1.0f);
        obs[1] = md_clip(2.0f * s->shape[n].hw / 2.5f, 0.0f, 1.0f);
    }
    if (c->n_side <= 0) { /* side detector off: distances to the route's left / right border */
        obs[o_base + 0] = md_clip(to_left / c->total_width, 0.0f, 1.0f);
        obs[o_base + 1] = md_clip(to_right / c->total_width, 0.0f, 1.0f);
    }
    obs[o_mid + 0] = r[1][0];
    obs[o_mid + 1] = md_clip((speed_kmh + 1.0f) / (P->max_speed_kmh + 1.0f), 0.0f, 1.0f);
    obs[o_mid + 2] = md_clip((d->steering / 60.0f + 1.0f) / 2.0f, 0.0f, 1.0f); /* MAX_STEERING = 60 (base_vehicle.py:80) */
    obs[o_mid + 3] = md_clip((s->action[2 * n] + 1.0f) / 2.0f, 0.0f, 1.0f);
    obs[o_mid + 4] = md_clip((s->action[2 * n + 1] + 1.0f) / 2.0f, 0.0f, 1.0f);
    obs[o_mid + 5] = r[8][0];
    float ls = r[2][0], llat = r[2][1];
    if (c->n_lane_line <= 0) obs[o_ll] = md_clip((llat * 2.0f / c->max_lane_width + 1.0f) / 2.0f, 0.0f, 1.0f);
    /* ---- navi (node_network_navigation.py:160-168, 243-292) ---- */
    if (!toll)
        for (int i = 0; i < 5; ++i) {
            obs[o_navi + i] = r[3][i];
            obs[o_navi + 5 + i] = r[4][i];
        }

    /* ---- arrive destination (metadrive_env.py:213-227) ---- */
    float fs = r[5][0], flat = r[5][1];
    int arrive = (k->fin->length - 5.0f < fs) && (fs < k->fin->length + 5.0f) && (cur_w / 2.0f >= flat) &&
                 (flat >= (0.5f - cur_n) * cur_w);
    /* ---- out of road (metadrive_env.py:229-237) ---- */
    int out_of_road = !(fl & MD_FL_ON_LANE);
    if (c->out_of_route_done) out_of_road = out_of_road || (fl & MD_FL_OUT_OF_ROUTE);
    else if (c->on_continuous_line_done)
        out_of_road = out_of_road || (fl & (MD_FL_ON_YELLOW_CONT | MD_FL_ON_WHITE_CONT | MD_FL_CRASH_SIDEWALK));
    if (toll) { /* MultiAgentTollgateEnv._is_out_of_road (marl_tollgate.py:241-247): the sidewalk, or the yellow line with
                 * cross_yellow_line_done (carried by on_continuous_line_done); leaving the lanes alone does not count */
        out_of_road = (fl & MD_FL_CRASH_SIDEWALK) != 0;
        if (c->on_continuous_line_done) out_of_road = out_of_road || (fl & MD_FL_ON_YELLOW_CONT);
    }
    if (c->is_multi_agent && c->ma_kind == MD_MA_PARKING_LOT) /* MultiAgentParkingLotEnv._is_out_of_road (marl_parking_lot.py:257-259) */
        out_of_road = !(fl & MD_FL_ON_LANE) || (fl & (MD_FL_ON_YELLOW_CONT | MD_FL_CRASH_SIDEWALK));
    const int racing = c->is_multi_agent && c->ma_kind == MD_MA_RACING;
    if (racing) /* MultiAgentRacingEnv._is_out_of_road (marl_racing_env.py:354-359): the map is fenced by guardrails; only a vehicle
                 * more than 5 m BEHIND the start of the lane it is on counts (longitudinal on vehicle.lane) */
        out_of_road = ls < -5.0f;
    if (arrive) fl |= MD_FL_ARRIVE_DEST;
    if (out_of_road) fl |= MD_FL_OUT_OF_ROAD;

    /* ---- reward (metadrive_env.py:239-279) ---- */
    float positive_road = k->positive_road;
    float long_last = r[6][0], long_now = r[7][0], lateral_now = r[7][1];
    float lateral_factor = 1.0f;
    if (c->use_lateral_reward) lateral_factor = md_clip(1.0f - 2.0f * md_fabs(lateral_now) / cur_w, 0.0f, 1.0f);
    float reward = 0.0f;
    if (toll) { /* MultiAgentTollgateEnv.reward_function (marl_tollgate.py:194-239): no direction factor; inside the toll
                 * block the speed term gives way to a penalty that REPLACES the reward while lane.speed_limit < speed [km/h] */
        reward += c->driving_reward * (long_now - long_last) * lateral_factor;
        if (k->cur_block == '$') {
            if (k->lane->speed_limit < speed_kmh) reward = -c->overspeed_penalty * speed_kmh / P->max_speed_kmh;
        } else reward += c->speed_reward * (speed_kmh / P->max_speed_kmh);
    } else {
        reward += c->driving_reward * (long_now - long_last) * lateral_factor * positive_road;
        reward += c->speed_reward * (speed_kmh / P->max_speed_kmh) * positive_road;
    }
    float step_reward = reward;
    int idle = 0;
    if (racing && !just_reset && s->idle_ring) {
        /* movement_between_steps[agent].append(abs(longitudinal_now - longitudinal_last)) on a deque(maxlen=100), and
         * _is_idle: 100 entries whose sum (oldest first) is below 0.1 m (marl_racing_env.py:392-398,415) */
        float* ring = s->idle_ring + (size_t)ai * MD_IDLE_WINDOW;
        int cnt = nav->toll_state;
        ring[cnt % MD_IDLE_WINDOW] = md_fabs(long_now - long_last);
        cnt += 1;
        if (cnt >= 2 * MD_IDLE_WINDOW) cnt -= MD_IDLE_WINDOW;   /* keeps (cnt mod window) and cnt >= window */
        nav->toll_state = cnt;
        if (cnt >= MD_IDLE_WINDOW) {
            float sum = 0.0f;
            for (int i = 0; i < MD_IDLE_WINDOW; ++i) sum += ring[(cnt + i) % MD_IDLE_WINDOW];
            idle = sum < 0.1f;
        }
    }
    if (arrive) reward = c->success_reward;
    else if (out_of_road) reward = -c->out_of_road_penalty;
    else if (fl & MD_FL_CRASH_VEHICLE) reward = -c->crash_vehicle_penalty;
    else if (racing) { /* marl_racing_env.py:432-439: the sidewalk (guardrail) and idling are penalised, objects are not */
        if (fl & MD_FL_CRASH_SIDEWALK) reward = -c->crash_sidewalk_penalty;
        else if (idle) reward = -c->idle_penalty;
    } else if (fl & MD_FL_CRASH_OBJECT) reward = -c->crash_object_penalty;

    /* ---- cost (metadrive_env.py:201-211) ---- */
    float cost = 0.0f;
    if (out_of_road) cost = c->out_of_road_cost;
    else if (fl & MD_FL_CRASH_VEHICLE) cost = c->crash_vehicle_cost;
    else if (fl & MD_FL_CRASH_OBJECT) cost = c->crash_object_cost;

    /* ---- done (metadrive_env.py:128-199, base_env.py:586-612) ---- */
    if (!just_reset) nav->steps += 1;
    int max_step = (c->horizon > 0) && (nav->steps >= c->horizon);
    int done = 0;
    if (arrive) done = 1;
    if (out_of_road) done = 1;
    if ((fl & MD_FL_CRASH_VEHICLE) && c->crash_vehicle_done) done = 1;
    if ((fl & MD_FL_CRASH_OBJECT) && c->crash_object_done) done = 1;
    if (fl & MD_FL_CRASH_BUILDING) done = 1;
    if ((fl & MD_FL_CRASH_HUMAN) && c->crash_human_done) done = 1;
    if (c->is_multi_agent && !max_step) {
        /* MultiAgentMetaDrive.done_function (multi_agent_metadrive.py:114-128) */
        int crash = (fl & (MD_FL_CRASH_VEHICLE | MD_FL_CRASH_OBJECT | MD_FL_CRASH_BUILDING | MD_FL_CRASH_SIDEWALK |
                           MD_FL_CRASH_HUMAN)) != 0;
        if (toll) crash = (fl & MD_FL_CRASH_VEHICLE) != 0; /* marl_tollgate.py:254: only a vehicle crash is taken back */
        if (crash && !c->crash_done && !(arrive || out_of_road)) done = 0;
        if (out_of_road && !c->out_of_road_done && !arrive) done = 0;
    }
    if (racing && !max_step) { /* MultiAgentRacingEnv.done_function (marl_racing_env.py:361-383) */
        if (idle) fl |= MD_FL_IDLE;
        if (idle && c->idle_done) done = 1;
        if ((fl & MD_FL_CRASH_SIDEWALK) && c->crash_sidewalk_done) done = 1;
    }
    if (toll && !max_step && nav->toll_entry && nav->toll_exit && nav->toll_exit - nav->toll_entry < c->min_pass_steps) {
        /* left the toll block sooner than min_pass_steps after entering it (marl_tollgate.py:254-259): done, reported as
         * out_of_road -- reward and cost above were taken from _is_out_of_road and do not see it */
        done = 1;
        fl |= MD_FL_OUT_OF_ROAD;
    }
    if (max_step) {
        fl |= MD_FL_MAX_STEP;
        if (c->truncate_as_terminate) done = 1;
    }
    if (just_reset) {
        /* _get_reset_return (base_env.py:560-584): obs only; no reward, nothing terminates */
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
    s->reward[ai] = reward;
    s->cost[ai] = cost;
    if (s->done_out) ((uint32_t*)s->done_out)[ai] = md_done_word(fl);
    if (toll) {
        /* TollGateObservation.observe (marl_tollgate.py:96-110): the counter runs on every observation made inside the block */
        int t = nav->toll_state & 0xffffff, last = (nav->toll_state >> 24) & 0xff;
        const int cur = k->cur_block, in_toll = (cur == '$');
        if (in_toll) t += 1;
        obs[c->obs_dim - 2] = in_toll ? 1.0f : 0.0f;
        obs[c->obs_dim - 1] = (in_toll && t > c->min_pass_steps) ? 1.0f : 0.0f;
        /* StayTimeManager.record (marl_tollgate.py:49-60), called after the step for the agents still active (an agent that
         * finished in this step is recorded here too: nothing reads its slot again); times in the agent's own steps
         * (+1, 0 = none): only their difference is read */
        if (last && last != cur) {
            if (in_toll) nav->toll_entry = nav->steps + 1;
            else if ((cur == 'y' || cur == 'Y') && last == '$') nav->toll_exit = nav->steps + 1;
        }
        last = cur;
        nav->toll_state = t | (last << 24);
    }

    /* ---- info (base_vehicle.py:243-271) ---- */
    float step_energy = just_reset ? 0.0f : r[8][1];
    s->pid[n].energy += step_energy;
    info[0] = step_reward;
    info[1] = md_fabs(d->speed);
    info[2] = step_energy;
    info[3] = s->pid[n].energy;
    info[4] = just_reset ? 0.0f : info[4] + reward;
    info[5] = just_reset ? 0.0f : info[5] + cost; /* SafeMetaDriveEnv total_cost (envs/safe_metadrive_env.py:31-35) */
    (void)llat;
    info[6] = ls;
    info[7] = (float)nav->steps;
    if (c->auto_reset && !c->is_multi_agent && !just_reset && (fl & (MD_FL_TERMINATED | MD_FL_TRUNCATED)))
        s->need_reset[0] = 1;
}

/* serial form (oracle): all tasks, then combine */
MD_HD void md_observe_agent(const MdLane* lanes, const MdRoad* roads, const MdState* s, const MdConfig* c, int a,
                             int just_reset) {
    MdObsCtx k;
    float r[MD_OBS_TASKS][5];
    md_observe_ctx(lanes, roads, s, a, &k);
    for (int t = 0; t < MD_OBS_TASKS; ++t) md_observe_task(t, &k, s, c, a, r[t]);
    md_observe_combine(&k, s, c, a, just_reset, (const float (*)[5])r);
}

/* ------------------------------------------------------------------------------------------
 * IDM traffic policy: IDMPolicy.act (policy/idm_policy.py:235-402) with FrontBackObjects
 * (policy/idm_policy.py:82-132).  Split in three stages so that the object scan in the middle can
 * run either serially (oracle, md_find_front_back below) or lane-parallel with wavefront
 * min-reductions (HIP kernel):
 *   md_idm_plan    move_to_next_road (:269-291) + which lanes to scan
 *   front/back     per-object evaluators md_idm_is_candidate / md_fb_same_lane_gap / md_fb_neighbour
 *   md_idm_decide  lane_change_policy (:330-402), steering_control (:293-301), acceleration (:303-320)
 * The bare `except:` fallback (idm_policy.py:254-260) is modelled by the `fail` paths.
 * -----------------------------------------------------------------------------------------*/
typedef struct {
    int front[3], back[3];
    float front_d[3], back_d[3];
    int exist[3];
} FrontBack;

typedef struct {
    int success;  /* move_to_next_road() result                                   */
    int use_ref;  /* scan left/right neighbours too (lane_change_policy path)     */
    int fail;     /* no routing target lane: fallback branch                      */
    int ids[3];   /* left, current, right lane ids to scan (-1 = absent)          */
} MdIdmPlan;

#define IDM_MAX_LONG_DIST 30.0f
#define IDM_SAFE_LANE_CHANGE 15.0f
#define IDM_LANE_CHANGE_FREQ 50
#define IDM_LANE_CHANGE_SPEED_INC 10.0f
#define IDM_NORMAL_SPEED 30.0f
#define IDM_CREEP_SPEED 5.0f
#define IDM_MAX_SPEED 100.0f

MD_HD int md_road_connects(const MdWorld* w, int m, int from_node, int to_node) {
    /* BaseRoadNetwork.has_connection (road_network/base_road_network.py:106-113): graph[from][to] exists */
    int nb = w->node_off[m];
    for (int k = w->node_adj_off[nb + from_node]; k < w->node_adj_off[nb + from_node + 1]; ++k)
        if (w->node_adj[2 * k] == to_node) return 1;
    return 0;
}

/* lidar.get_surrounding_objects(v): bodies touching the r = 50 m ghost cylinder (sensors/lidar.py:170-186) */
MD_HD int md_idm_is_candidate(const MdShape* o, float px, float py) {
    if (!md_present(o->flags)) return 0;
    if (md_is_circle_kind(md_kind_of(o->flags))) {
        float dx = o->cx - px, dy = o->cy - py, rr = 50.0f + o->hl;
        return (dx * dx + dy * dy) <= rr * rr;
    }
    return md_obb_circle(o->cx, o->cy, o->c, o->s, o->hl, o->hw, px, py, 50.0f);
}

/* A pedestrian or cyclist among the surrounding objects: BaseTrafficParticipant has no `.lane`
 * (component/traffic_participants/base_traffic_participant.py:12-32), so FrontBackObjects.get_find_front_back_objs
 * raises AttributeError at `obj.lane is lane` (policy/idm_policy.py:110) as soon as its object loop reaches it -- and it
 * always does, the routing target lane is never None there -- before lane_change_policy has touched target_speed or
 * overtake_timer.  IDMPolicy.act catches everything (:254-260): front object None, distance 5, steering lane = the
 * routing target lane, for THAT vehicle in THAT step.  (move_to_next_road has already run: its effects stay.) */
MD_HD int md_is_participant_kind(int k) { return k == MD_KIND_PEDESTRIAN || k == MD_KIND_CYCLIST; }

MD_HD int md_idm_sees_participant(const MdState* s, const MdConfig* c, int self_slot, float px, float py) {
    for (int j = 0; j < c->cap; ++j) {
        if (j == self_slot) continue;
        const MdShape* o = &s->shape[j];
        if (md_is_participant_kind(md_kind_of(o->flags)) && md_idm_is_candidate(o, px, py)) return 1;
    }
    return 0;
}

/* obj.lane: vehicles carry their localised lane, props the lane they were placed on */
MD_HD int md_obj_lane_of(const MdState* s, int j) {
    const MdShape* o = &s->shape[j];
    return (md_kind_of(o->flags) == MD_KIND_VEHICLE && !(o->flags & MD_F_STATIC)) ? s->nav[j].lane : o->aux;
}

/* same-lane object: signed gap along the lane (idm_policy.py:111-121) */
MD_HD float md_fb_same_lane_gap(const MdLane* L, float cur_long, const MdShape* o) {
    float os, ot;
    md_lane_local(L, o->cx, o->cy, &os, &ot);
    return os - cur_long;
}

/* object on a connected lane (idm_policy.py:123-132): returns 1 = front candidate, 2 = back candidate,
 * 0 = neither; *lg = its distance. */
MD_HD int md_fb_neighbour(const MdLane* L, const MdLane* OL, float cur_long, float left_long, const MdShape* o,
                          int need_front, int need_back, float* lg) {
    if (need_front && md_lane_is_previous_of(L, OL)) {
        float os, ot;
        md_lane_local(OL, o->cx, o->cy, &os, &ot);
        *lg = os + left_long;
        return 1;
    }
    if (need_back && md_lane_is_previous_of(OL, L)) {
        float os, ot;
        md_lane_local(OL, o->cx, o->cy, &os, &ot);
        *lg = OL->length - os + cur_long;
        return 2;
    }
    return 0;
}

/* Stage A */
MD_HD void md_idm_plan(const MdWorld* w, const MdLane* lanes, const MdRoad* roads, const MdState* s, const MdConfig* c,
                        int m, int slot, MdIdmPlan* p) {
    MdNav* nav = &s->nav[slot];
    const MdRoad* cur_road = &roads[nav->road0];
    int success;
    int veh_lane = nav->lane;
    int in_cur = (veh_lane >= 0) && (lanes[veh_lane].road == nav->road0);
    if (nav->target_lane < 0) {
        nav->target_lane = veh_lane;
        success = in_cur;
    } else if (lanes[nav->target_lane].road != nav->road0) {
        success = 0;
        const MdLane* T = &lanes[nav->target_lane];
        int t_end = roads[T->road].end_node;
        for (int k = 0; k < cur_road->n_lanes; ++k) {
            const MdLane* L = &lanes[cur_road->first_lane + k];
            if (md_lane_is_previous_of(T, L) || md_road_connects(w, m, t_end, cur_road->end_node)) {
                nav->target_lane = cur_road->first_lane + k;
                success = 1;
                break;
            }
        }
    } else if (in_cur && nav->target_lane != veh_lane) {
        nav->target_lane = veh_lane;
        nav->timer = s->idm_rand[(size_t)slot * MD_IDM_RAND + (nav->rand_cursor % MD_IDM_RAND)];
        nav->rand_cursor += 1;
        success = 1;
    } else {
        success = 1;
    }
    p->success = success;
    p->fail = (nav->target_lane < 0);
    p->use_ref = success && c->enable_idm_lane_change;
    p->ids[0] = p->ids[2] = -1;
    p->ids[1] = nav->target_lane;
    if (!p->fail && p->use_ref) {
        int idx = lanes[nav->target_lane].idx;
        if (idx > 0) p->ids[0] = cur_road->first_lane + idx - 1;
        if (idx + 1 < cur_road->n_lanes) p->ids[2] = cur_road->first_lane + idx + 1;
    }
}

/* Stage B, serial form.  The reference walks a Python set (arbitrary order) and lets an object on the
 * successor / predecessor lane compete only while no same-lane object has been found yet
 * (idm_policy.py:110-130).  Canonical, order-independent form: same-lane objects first;
 * connected-lane objects only when the same lane offered none.  Ties -> lowest slot. */
MD_HD void md_find_front_back(const MdState* s, const MdConfig* c, const MdLane* lanes, int self_slot,
                              const MdIdmPlan* p, float px, float py, FrontBack* fb) {
    for (int i = 0; i < 3; ++i) {
        fb->front[i] = fb->back[i] = -1;
        fb->exist[i] = p->ids[i] >= 0;
        fb->front_d[i] = fb->back_d[i] = IDM_MAX_LONG_DIST;
        if (p->ids[i] < 0) continue;
        const MdLane* L = &lanes[p->ids[i]];
        float cur_long, tmp;
        md_lane_local(L, px, py, &cur_long, &tmp);
        float left_long = L->length - cur_long;
        int found_front = 0, found_back = 0;
        for (int j = 0; j < c->cap; ++j) {
            if (j == self_slot) continue;
            const MdShape* o = &s->shape[j];
            if (!md_idm_is_candidate(o, px, py)) continue;
            if (md_obj_lane_of(s, j) != p->ids[i]) continue;
            float lg = md_fb_same_lane_gap(L, cur_long, o);
            if (fb->front_d[i] > lg && lg > 0.0f) {
                fb->front_d[i] = lg;
                fb->front[i] = j;
                found_front = 1;
            }
            if (lg < 0.0f && md_fabs(lg) < fb->back_d[i]) {
                fb->back_d[i] = md_fabs(lg);
                fb->back[i] = j;
                found_back = 1;
            }
        }
        if (found_front && found_back) continue;
        for (int j = 0; j < c->cap; ++j) {
            if (j == self_slot) continue;
            const MdShape* o = &s->shape[j];
            if (!md_idm_is_candidate(o, px, py)) continue;
            int ol = md_obj_lane_of(s, j);
            if (ol < 0 || ol == p->ids[i]) continue;
            float lg;
            int cls = md_fb_neighbour(L, &lanes[ol], cur_long, left_long, o, !found_front, !found_back, &lg);
            if (cls == 1) {
                if (fb->front_d[i] > lg && lg > 0.0f) {
                    fb->front_d[i] = lg;
                    fb->front[i] = j;
                }
            } else if (cls == 2) {
                if (fb->back_d[i] > lg) {
                    fb->back_d[i] = lg;
                    fb->back[i] = j;
                }
            }
        }
    }
}

/* Stage C */
MD_HD void md_idm_decide(const MdLane* lanes, const MdRoad* roads, const MdState* s, int slot, const MdIdmPlan* p,
                         const FrontBack* fbp) {
    MdShape* sh = &s->shape[slot];
    MdNav* nav = &s->nav[slot];
    MdPid* pid = &s->pid[slot];
    MdDyn* d = &s->dyn[slot];
    const MdRoad* cur_road = &roads[nav->road0];
    int has_next = nav->ck1 != nav->ck0;
    const MdRoad* next_road = has_next ? &roads[nav->road1] : 0;
    float px = sh->cx, py = sh->cy;
    float speed_kmh = md_fabs(d->speed) * 3.6f;
    FrontBack fb = *fbp;

    int front_obj = -1;
    float front_dist = 5.0f;
    int steer_lane = nav->target_lane;
    int fail = p->fail;
    if (!fail) {
        if (p->use_ref) {
            /* ---- lane_change_policy (idm_policy.py:330-402) ---- */
            int ncur = cur_road->n_lanes;
            int avail_lo = 0, avail_hi = ncur - 1;
            int lane_num_diff = has_next ? (ncur - next_road->n_lanes) : 0;
            int tidx = lanes[nav->target_lane].idx;
            int decided = 0;
            if (lane_num_diff > 0) {
                if (md_lane_is_previous_of(&lanes[cur_road->first_lane], &lanes[next_road->first_lane])) {
                    avail_lo = 0;
                    avail_hi = next_road->n_lanes - 1;
                } else {
                    avail_lo = lane_num_diff;
                    avail_hi = ncur - 1;
                }
                if (tidx < avail_lo || tidx > avail_hi) {
                    if (tidx > avail_hi) { /* change to left */
                        if (!fb.exist[0]) fail = 1;
                        else if (fb.back_d[0] < IDM_SAFE_LANE_CHANGE || fb.front_d[0] < 5.0f) {
                            pid->target_speed = IDM_CREEP_SPEED;
                            front_obj = fb.front[1]; front_dist = fb.front_d[1]; steer_lane = nav->target_lane;
                        } else {
                            pid->target_speed = IDM_NORMAL_SPEED;
                            front_obj = fb.front[0]; front_dist = fb.front_d[0];
                            steer_lane = cur_road->first_lane + tidx - 1;
                        }
                    } else { /* change to right */
                        if (!fb.exist[2]) fail = 1;
                        else if (fb.back_d[2] < IDM_SAFE_LANE_CHANGE || fb.front_d[2] < 5.0f) {
                            pid->target_speed = IDM_CREEP_SPEED;
                            front_obj = fb.front[1]; front_dist = fb.front_d[1]; steer_lane = nav->target_lane;
                        } else {
                            pid->target_speed = IDM_NORMAL_SPEED;
                            front_obj = fb.front[2]; front_dist = fb.front_d[2];
                            steer_lane = cur_road->first_lane + tidx + 1;
                        }
                    }
                    decided = 1;
                }
            }
            if (!decided && !fail) {
                int overtake = 0;
                if (md_fabs(speed_kmh - IDM_NORMAL_SPEED) > 3.0f && fb.front[1] >= 0) {
                    float fsp = md_fabs(s->dyn[fb.front[1]].speed) * 3.6f;
                    if (md_fabs(fsp - IDM_NORMAL_SPEED) > 3.0f && nav->timer > IDM_LANE_CHANGE_FREQ) overtake = 1;
                }
                if (overtake) {
                    /* speeds of neighbours; -1 encodes None */
                    float right_sp = -1.0f, left_sp = -1.0f;
                    if (fb.front[2] >= 0) right_sp = md_fabs(s->dyn[fb.front[2]].speed) * 3.6f;
                    else if (fb.exist[2] && fb.front_d[2] > IDM_SAFE_LANE_CHANGE && fb.back_d[2] > IDM_SAFE_LANE_CHANGE)
                        right_sp = IDM_MAX_SPEED;
                    float front_sp = md_fabs(s->dyn[fb.front[1]].speed) * 3.6f;
                    if (fb.front[0] >= 0) left_sp = md_fabs(s->dyn[fb.front[0]].speed) * 3.6f;
                    else if (fb.exist[0] && fb.front_d[0] > IDM_SAFE_LANE_CHANGE && fb.back_d[0] > IDM_SAFE_LANE_CHANGE)
                        left_sp = IDM_MAX_SPEED;
                    if (left_sp >= 0.0f && left_sp - front_sp > IDM_LANE_CHANGE_SPEED_INC) {
                        int ex = tidx - 1;
                        if (ex >= avail_lo && ex <= avail_hi) {
                            front_obj = fb.front[0]; front_dist = fb.front_d[0];
                            steer_lane = cur_road->first_lane + ex;
                            decided = 1;
                        }
                    }
                    if (!decided && right_sp >= 0.0f && right_sp - front_sp > IDM_LANE_CHANGE_SPEED_INC) {
                        int ex = tidx + 1;
                        if (ex >= avail_lo && ex <= avail_hi) {
                            front_obj = fb.front[2]; front_dist = fb.front_d[2];
                            steer_lane = cur_road->first_lane + ex;
                            decided = 1;
                        }
                    }
                }
                if (!decided) {
                    pid->target_speed = IDM_NORMAL_SPEED;
                    nav->timer += 1;
                    front_obj = fb.front[1]; front_dist = fb.front_d[1]; steer_lane = nav->target_lane;
                }
            }
        } else {
            front_obj = fb.front[1];
            front_dist = fb.front_d[1];
            steer_lane = nav->target_lane;
        }
    }
    if (fail) { /* bare except fallback (idm_policy.py:254-260) */
        front_obj = -1;
        front_dist = 5.0f;
        steer_lane = nav->target_lane;
    }
    if (steer_lane < 0) { /* never localised: coast straight */
        s->action[2 * slot] = 0.0f;
        s->action[2 * slot + 1] = 0.0f;
        return;
    }
    /* ---- steering_control (idm_policy.py:293-301) ---- */
    const MdLane* TL = &lanes[steer_lane];
    float tl_s, tl_lat;
    md_lane_local(TL, px, py, &tl_s, &tl_lat);
    float lane_heading = md_lane_heading_at(TL, tl_s + 1.0f);
    float steering = md_pid(&pid->hp, &pid->hi, &pid->hd, 1.7f, 0.01f, 3.5f, -md_wrap_to_pi(lane_heading - d->heading));
    steering += md_pid(&pid->lp, &pid->li, &pid->ld, 0.3f, 0.002f, 0.05f, -tl_lat);
    /* ---- acceleration (idm_policy.py:303-320) ---- */
    float dv = 0.0f;
    if (front_obj >= 0) {
        const MdShape* fo = &s->shape[front_obj];
        float fv = (md_kind_of(fo->flags) == MD_KIND_VEHICLE) ? s->dyn[front_obj].speed : 0.0f;
        float evx = d->speed * sh->c * 3.6f, evy = d->speed * sh->s * 3.6f;
        float fvx = fv * fo->c * 3.6f, fvy = fv * fo->s * 3.6f;
        dv = (evx - fvx) * sh->c + (evy - fvy) * sh->s;
    }
    float acc = md_idm_acceleration(speed_kmh, pid->target_speed, front_obj >= 0, front_dist, dv);
    s->action[2 * slot] = steering;
    s->action[2 * slot + 1] = acc;
}

/* All three stages for one vehicle, serial. */
MD_HD void md_idm_vehicle(const MdWorld* w, const MdState* s, const MdConfig* c, int e, int slot) {
    MdIdmPlan plan;
    FrontBack fb;
    const MdLane* lanes = w->lanes + w->lane_off[w->env_map[e]];
    const MdRoad* roads = w->roads + w->road_off[w->env_map[e]];
    md_idm_plan(w, lanes, roads, s, c, w->env_map[e], slot, &plan);
    for (int i = 0; i < 3; ++i) {
        fb.front[i] = fb.back[i] = -1;
        fb.exist[i] = 0;
        fb.front_d[i] = fb.back_d[i] = IDM_MAX_LONG_DIST;
    }
    if (!plan.fail && md_idm_sees_participant(s, c, slot, s->shape[slot].cx, s->shape[slot].cy)) plan.fail = 1;
    if (!plan.fail) md_find_front_back(s, c, lanes, slot, &plan, s->shape[slot].cx, s->shape[slot].cy, &fb);
    md_idm_decide(lanes, roads, s, slot, &plan, &fb);
}

/* ------------------------------------------------------------------------------------------
 * Multi-agent lifecycle of one env at the START of a step (serial; one thread per env on the GPU).
 * References in include/mdstep.h at md_lifecycle.  Random draws (which free spawn place, which
 * destination) come from a per-env xorshift32 stream: the reference uses an UNSEEDED RandomState for
 * both (spawn_manager.py:217-220, multi_agent_metadrive.py:199), so any stream is admissible and
 * this one is reproducible.
 * -----------------------------------------------------------------------------------------*/
MD_HD uint32_t md_rng_next(uint32_t* st) {
    uint32_t x = *st;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    *st = x;
    return x;
}

/* ------------------------------------------------------------------------------------------
 * "Others" block of LidarStateObservation: Lidar.get_surrounding_vehicles_info
 * (component/sensors/lidar.py:93-138).  det_lo/det_hi = bit set of the slots some lidar beam of agent `a`
 * hit first (detected_objects); vehicles among them (broken-down ones included, they are BaseVehicles;
 * cones / tripods / barriers are not) are ranked by centre distance -- ties to the lowest slot, the reference
 * sorts a Python set -- and the nearest num_others are described in the ego frame:
 *   [ (fwd/R+1)/2, (left/R+1)/2, (dv_fwd/vmax+1)/2, (dv_left/vmax+1)/2 ]   R = lidar distance, speeds in km/h,
 *   vmax = the ego's max_speed_km_h; with add_others_navi also the two checkpoints of that vehicle
 *   (BaseNavigation.get_checkpoints, base_navigation.py:145-152), clipped to R, same projection.
 * Missing vehicles are zero-filled.  Frame: (forward, left), the convert_to_local_coordinates derivation of
 * SURVEY 8a-5 (a Panda3D transform: unpinned).  Velocity = speed along the heading (the kinematic model's
 * slip angle is ignored here).
 * -----------------------------------------------------------------------------------------*/
MD_HD void md_others_project(float dx, float dy, float hc, float hs, float scale, float* o0, float* o1) {
    float fwd = dx * hc + dy * hs;
    float left = -dx * hs + dy * hc;
    *o0 = md_clip((fwd / scale + 1.0f) / 2.0f, 0.0f, 1.0f);
    *o1 = md_clip((left / scale + 1.0f) / 2.0f, 0.0f, 1.0f);
}

MD_HD void md_others_ckpt(const MdLane* lanes, const MdRoad* roads, const MdState* s, int j, float* c1x, float* c1y,
                          float* c2x, float* c2y) {
    const MdNav* nav = &s->nav[j];
    const MdRoad* r1 = &roads[nav->road0];
    const MdRoad* r2 = (nav->ck1 != nav->ck0) ? &roads[nav->road1] : r1;
    /* later_middle uses the CURRENT lane's width and the current road's lane count (base_navigation.py:147) */
    float wdt = (nav->lane >= 0) ? lanes[nav->lane].width : lanes[r1->first_lane].width;
    float lm = ((float)r1->n_lanes / 2.0f - 0.5f) * wdt;
    const MdLane* L1 = &lanes[r1->first_lane];
    const MdLane* L2 = &lanes[r2->first_lane];
    *c1x = L1->ex + lm * L1->elx;
    *c1y = L1->ey + lm * L1->ely;
    *c2x = L2->ex + lm * L2->elx;
    *c2y = L2->ey + lm * L2->ely;
}

MD_HD void md_others_block(const MdLane* lanes, const MdRoad* roads, const MdState* s, const MdConfig* c, int a,
                           unsigned long long det_lo, unsigned long long det_hi, float* out) {
    const int wd = md_others_width(c);
    const MdShape* me = &s->shape[a];
    const float R = c->lidar_range;
    const float vmax = s->param[a].max_speed_kmh;
    const float ego_v = s->dyn[a].speed * 3.6f;
    /* keep vehicles only */
    for (int j = 0; j < c->cap && j < 128; ++j) {
        if (md_kind_of(s->shape[j].flags) == MD_KIND_VEHICLE) continue;
        if (j < 64) det_lo &= ~(1ull << j);
        else det_hi &= ~(1ull << (j - 64));
    }
    for (int k = 0; k < c->num_others; ++k) {
        float* o = out + k * wd;
        int best = -1;
        float bd = 3.0e38f;
        for (int j = 0; j < c->cap && j < 128; ++j) {
            unsigned long long m = (j < 64) ? det_lo : det_hi;
            if (!((m >> (j & 63)) & 1ull)) continue;
            float d = md_norm(me->cx - s->shape[j].cx, me->cy - s->shape[j].cy);
            if (d < bd) {
                bd = d;
                best = j;
            }
        }
        if (best < 0 || !md_present(me->flags)) {
            for (int i = 0; i < wd; ++i) o[i] = 0.0f;
            continue;
        }
        if (best < 64) det_lo &= ~(1ull << best);
        else det_hi &= ~(1ull << (best - 64));
        const MdShape* v = &s->shape[best];
        md_others_project(v->cx - me->cx, v->cy - me->cy, me->c, me->s, R, &o[0], &o[1]);
        float ov = s->dyn[best].speed * 3.6f;
        md_others_project(ov * v->c - ego_v * me->c, ov * v->s - ego_v * me->s, me->c, me->s, vmax, &o[2], &o[3]);
        if (c->add_others_navi) {
            float c1x = me->cx, c1y = me->cy, c2x = me->cx, c2y = me->cy;
            if (s->nav[best].route_len >= 2 && s->nav[best].road0 >= 0)
                md_others_ckpt(lanes, roads, s, best, &c1x, &c1y, &c2x, &c2y);
            float pts[4] = {c1x, c1y, c2x, c2y};
            for (int q = 0; q < 2; ++q) {
                float dx = pts[2 * q] - me->cx, dy = pts[2 * q + 1] - me->cy;
                float nd = md_norm(dx, dy);
                if (nd > R) { /* _project_to_vehicle_system, lidar.py:85-91 */
                    dx = dx / nd * R;
                    dy = dy / nd * R;
                }
                md_others_project(dx, dy, me->c, me->s, R, &o[4 + 2 * q], &o[5 + 2 * q]);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Traffic modes "respawn" / "hybrid": PGTrafficManager.after_step (manager/traffic_manager.py:94-122).
 * A traffic vehicle that left every lane is removed and, in these two modes, a NEW vehicle of the same
 * type starts on a random respawn lane at longitude U[0, length/2) with a fresh IDMPolicy
 * (policy/idm_policy.py:225-233: overtake_timer = randint(0, LANE_CHANGE_FREQ), PIDs zeroed) and a
 * fresh route (navigation.reset).  The slot is reused.  Serial per env, ascending slot order (the
 * reference walks its _traffic_vehicles list).  Draws come from the env's xorshift32 stream; the
 * reference draws from the manager's RandomState whose position depends on Bullet-placed spawns, so
 * the stream is unpinned either way.
 * The respawn lanes and their routes are MdWorld.spawn_* with n_dest == 1 (spawn_place unused).
 * -----------------------------------------------------------------------------------------*/
MD_HD int md_traffic_wants_respawn(int flags) {
    return md_kind_of(flags) == MD_KIND_VEHICLE && !(flags & (MD_F_ALIVE | MD_F_PENDING | MD_F_AGENT | MD_F_STATIC));
}

MD_HD void md_traffic_respawn_env(const MdWorld* w, const MdLane* lanes, const MdState* s, const MdConfig* c, int m) {
    if ((c->traffic_mode != 1 && c->traffic_mode != 2) || !w->spawn_off) return;
    const int p0 = w->spawn_off[m], np_ = w->spawn_off[m + 1] - p0;
    if (np_ <= 0) return;
    for (int slot = c->agents_per_env; slot < c->cap; ++slot) {
        MdShape* sh = &s->shape[slot];
        if (!md_traffic_wants_respawn(sh->flags)) continue;
        const int p = p0 + (int)(md_rng_next(s->rng) % (uint32_t)np_);
        const MdLane* L = &lanes[w->spawn_lane[p]];
        const float u = (float)(md_rng_next(s->rng) >> 8) * (1.0f / 16777216.0f);
        const float lng = u * L->length * 0.5f;
        float x, y;
        md_lane_position(L, lng, &x, &y);
        const float h = md_wrap_to_pi(md_lane_heading_at(L, lng));
        sh->cx = x;
        sh->cy = y;
        md_sincos(h, &sh->s, &sh->c);
        sh->flags = MD_KIND_VEHICLE | MD_F_ALIVE;
        sh->aux = -1;
        MdDyn* d = &s->dyn[slot];
        d->heading = h;
        d->speed = 0.0f;
        d->steering = 0.0f;
        d->throttle = 0.0f;
        d->last_x = x;
        d->last_y = y;
        d->last_c = sh->c;
        d->last_s = sh->s;
        MdNav* nav = &s->nav[slot];
        nav->lane = w->spawn_lane[p];
        nav->route_len = w->spawn_route_meta[2 * (size_t)p];
        nav->ck0 = 0;
        nav->ck1 = (nav->route_len <= 2) ? 0 : 1;
        nav->target_lane = -1;
        nav->timer = (int)(md_rng_next(s->rng) % (uint32_t)IDM_LANE_CHANGE_FREQ);
        nav->trigger_road = -1;
        nav->trigger_order = 0;
        nav->steps = 0;
        nav->done = 0;
        s->final_lane[slot] = w->spawn_route_meta[2 * (size_t)p + 1];
        const int32_t* rt = w->spawn_route + (size_t)p * 2 * MD_ROUTE_LEN;
        for (int k = 0; k < MD_ROUTE_LEN; ++k) {
            s->route_nodes[(size_t)slot * MD_ROUTE_LEN + k] = rt[k];
            s->route_roads[(size_t)slot * MD_ROUTE_LEN + k] = rt[MD_ROUTE_LEN + k];
        }
        nav->road0 = rt[MD_ROUTE_LEN + nav->ck0];
        nav->road1 = rt[MD_ROUTE_LEN + nav->ck1];
        MdPid* pid = &s->pid[slot];
        pid->hp = pid->hi = pid->hd = 0.0f;
        pid->lp = pid->li = pid->ld = 0.0f;
        pid->target_speed = IDM_NORMAL_SPEED;
        pid->energy = 0.0f;
        s->flags[slot] = 0;
        s->action[2 * slot] = 0.0f;
        s->action[2 * slot + 1] = 0.0f;
    }
}

/* a (re)spawned agent of a multi-agent env with random_agent_model: one of the vehicle classes, uniformly
 * (VehicleAgentManager._create_agents -> random_vehicle_type, manager/agent_manager.py:37-43) */
MD_HD void md_draw_vehicle_class(const MdWorld* w, const MdState* s, int slot) {
    const float* v = w->vclass + 12 * (size_t)(md_rng_next(s->rng) % (uint32_t)w->n_vclass);
    float* prm = (float*)&s->param[slot];
    for (int i = 0; i < 8; ++i) prm[i] = v[i];
    s->shape[slot].hl = v[8];
    s->shape[slot].hw = v[9];
}

MD_HD int md_popcount32(uint32_t v) {
    int n = 0;
    for (; v; v &= v - 1) ++n;
    return n;
}
MD_HD int md_kth_set_bit(uint32_t v, int k) { /* index of the k-th (0-based) set bit, ascending */
    while (k-- > 0) v &= v - 1;
    int i = 0;
    while (!((v >> i) & 1u)) ++i;
    return i;
}

#define MD_RESPAWN_HALF_LEN 4.0f  /* RESPAWN_REGION_LONGITUDE / 2 (spawn_manager.py:28) */
#define MD_RESPAWN_HALF_WID 1.5f  /* RESPAWN_REGION_LATERAL / 2 */

/* agent_policy = IDMPolicy in a multi-agent env: a (re)spawned agent gets a fresh IDMPolicy (manager/agent_manager.py:37-52:
 * add_policy(obj.id, policy_cls, obj, self.generate_seed())) -- overtake_timer = randint(0, LANE_CHANGE_FREQ) (idm_policy.py:229), clean
 * PID states, NORMAL_SPEED, no routing target lane.  The reference seeds that policy from the agent manager's stream; the respawn
 * draws here come from the env's xorshift32 stream like the place and the destination. */
MD_HD void md_agent_idm_init(const MdState* s, const MdConfig* c, int slot) {
    if (!c->agent_idm) return;
    MdNav* nav = &s->nav[slot];
    nav->timer = (int)(md_rng_next(s->rng) % (uint32_t)IDM_LANE_CHANGE_FREQ);
    nav->target_lane = -1;
    nav->rand_cursor = 0;
    MdPid* p = &s->pid[slot];
    p->hp = p->hi = p->hd = p->lp = p->li = p->ld = 0.0f;
    p->target_speed = IDM_NORMAL_SPEED;
}

MD_HD void md_lifecycle_env(const MdWorld* w, const MdState* s, const MdConfig* c, int m) {
    const int A = c->agents_per_env;
    s->env_steps[0] += 1;
    int active = 0, dying = 0;
    /* MultiAgentParkingLotEnv (marl_parking_lot.py:47-95): a vehicle entering from outside is given a free parking space as its
     * destination and holds it until it is done (ParkingLotSpawnManager.get_parking_space / after_vehicle_done); the set of
     * free spaces is therefore what no ACTIVE agent holds -- nav.toll_entry = space + 1 -- and needs no state of its own */
    const int parking = c->ma_kind == MD_MA_PARKING_LOT;
    uint32_t reserved = 0;
    for (int a = 0; a < A; ++a) {
        MdShape* sh = &s->shape[a];
        MdNav* nav = &s->nav[a];
        if (!(sh->flags & MD_F_ALIVE)) continue;
        if (!(sh->flags & MD_F_STATIC)) {
            /* active: did it finish at the previous step? (_after_vehicle_done / _finish) */
            uint32_t fl = s->flags[a];
            if (nav->done || (fl & MD_FL_TRUNCATED)) {
                if ((fl & MD_FL_ARRIVE_DEST) || c->delay_done <= 0) {
                    sh->flags &= ~MD_F_ALIVE;
                    continue;
                }
                sh->flags |= MD_F_STATIC;
                nav->timer = c->delay_done;
                s->dyn[a].speed = 0.0f;
            } else {
                active++;
                if (parking && nav->toll_entry > 0) reserved |= 1u << (nav->toll_entry - 1);
                continue;
            }
        }
        /* dying: countdown (VehicleAgentManager.before_step) */
        nav->timer -= 1;
        if (nav->timer <= 0) sh->flags &= ~MD_F_ALIVE;
        else dying++;
    }
    const int horizon_open = !(c->horizon > 0 && s->env_steps[0] >= c->horizon);
    if (c->allow_respawn && horizon_open && w->spawn_off) {
        const int p0 = w->spawn_off[m], np_ = w->spawn_off[m + 1] - p0;
        uint32_t used = 0; /* spawn_places_used, reset every step */
        const int n_in = parking ? np_ - c->n_parking : 0; /* places 0 .. n_in-1: the entrances; then one per parking space */
        uint32_t avail = parking ? (((1u << c->n_parking) - 1u) & ~reserved) : 0u;
        while (active + dying < A) {
            /* safe places: not used this step and no vehicle chassis inside the 8 m x 3 m region */
            int safe[32];
            int n_safe = 0;
            for (int p = 0; p < np_ && p < 32; ++p) {
                if ((used >> p) & 1u) continue;
                if (parking && p < n_in && avail == 0u) continue; /* no space to send it to: the entrance stays shut (:103-105) */
                const float* pl = w->spawn_place + 8 * (size_t)(p0 + p);
                int hit = 0;
                for (int j = 0; j < c->cap && !hit; ++j) {
                    const MdShape* o = &s->shape[j];
                    if (!md_present(o->flags) || md_kind_of(o->flags) != MD_KIND_VEHICLE) continue;
                    hit = md_obb_obb(pl[0], pl[1], pl[2], pl[3], MD_RESPAWN_HALF_LEN, MD_RESPAWN_HALF_WID, o->cx, o->cy, o->c,
                                     o->s, o->hl, o->hw);
                }
                if (!hit) {
                    safe[n_safe++] = p;
                    used |= 1u << p; /* get_available_respawn_places marks every returned place as used */
                }
            }
            if (n_safe == 0) break;
            /* one vehicle per call of _respawn_single_vehicle; places found safe stay "used" this step */
            const int p = safe[md_rng_next(s->rng) % (uint32_t)n_safe];
            int slot = -1;
            for (int a = 0; a < A; ++a)
                if (!(s->shape[a].flags & MD_F_ALIVE)) { slot = a; break; }
            if (slot < 0) break;
            const float* pl = w->spawn_place + 8 * (size_t)(p0 + p);
            int dest = 0, space = 0;
            if (!parking) dest = (int)(md_rng_next(s->rng) % (uint32_t)w->n_dest);
            else if (p < n_in) { /* update_destination_for (:80-88): from an entrance to a free space, drawn uniformly */
                dest = md_kth_set_bit(avail, (int)(md_rng_next(s->rng) % (uint32_t)md_popcount32(avail)));
                avail &= ~(1u << dest);
                space = dest + 1;
            } else dest = c->n_parking + (int)(md_rng_next(s->rng) % (uint32_t)(w->n_dest - c->n_parking)); /* out through an entrance */
            const size_t ri = ((size_t)(p0 + p) * w->n_dest + dest);
            const int32_t* rt = w->spawn_route + ri * 2 * MD_ROUTE_LEN;
            MdShape* sh = &s->shape[slot];
            sh->cx = pl[0];
            sh->cy = pl[1];
            sh->c = pl[2];
            sh->s = pl[3];
            sh->flags = MD_KIND_VEHICLE | MD_F_ALIVE | MD_F_AGENT | MD_F_SPAWNED;
            sh->aux = -1;
            MdDyn* d = &s->dyn[slot];
            d->heading = pl[4];
            d->speed = 0.0f;
            d->steering = 0.0f;
            d->throttle = 0.0f;
            d->last_x = pl[0];
            d->last_y = pl[1];
            d->last_c = pl[2];
            d->last_s = pl[3];
            MdNav* nav = &s->nav[slot];
            nav->lane = w->spawn_lane[p0 + p];
            nav->route_len = w->spawn_route_meta[2 * ri];
            nav->ck0 = 0;
            nav->ck1 = (nav->route_len <= 2) ? 0 : 1;
            nav->target_lane = -1;
            nav->timer = 0;
            nav->steps = 0;
            nav->done = 0;
            nav->toll_state = nav->toll_entry = nav->toll_exit = 0;
            nav->toll_entry = space;
            md_agent_idm_init(s, c, slot);
            if (c->random_agent_model && w->n_vclass > 0) md_draw_vehicle_class(w, s, slot);
            s->final_lane[slot] = w->spawn_route_meta[2 * ri + 1];
            for (int k = 0; k < MD_ROUTE_LEN; ++k) {
                s->route_nodes[(size_t)slot * MD_ROUTE_LEN + k] = rt[k];
                s->route_roads[(size_t)slot * MD_ROUTE_LEN + k] = rt[MD_ROUTE_LEN + k];
            }
            nav->road0 = rt[MD_ROUTE_LEN + nav->ck0];
            nav->road1 = rt[MD_ROUTE_LEN + nav->ck1];
            s->pid[slot].energy = 0.0f;
            s->flags[slot] = 0;
            s->action[2 * slot] = 0.0f;
            s->action[2 * slot + 1] = 0.0f;
            s->agent_id[slot] = s->next_agent_id[0];
            s->next_agent_id[0] += 1;
            active++;
        }
    }
    /* episode over: nobody left and nobody can come back */
    if (c->auto_reset && active == 0 && !(c->allow_respawn && horizon_open)) s->need_reset[0] = 1;
}

#endif /* MD_ENTITY_H */

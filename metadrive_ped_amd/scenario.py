"""Host side of SCENARIO mode: scenario descriptions (the reference's unified format, metadrive/scenario/
scenario_description.py) -> the tables and reset state of a batch of ScenarioEnv scenes.

What the reference does per episode (envs/scenario_env.py, manager/scenario_map_manager.py:49-75,
manager/scenario_traffic_manager.py, manager/scenario_data_manager.py) and what becomes of it here:

  ScenarioMapManager.update_route   the SDC's recorded track -> PointLane reference trajectory, the agent's spawn pose /
                                    velocity = the track's first frame            -> slot 0, polyline 0, checkpoint table
  ScenarioTrafficManager            every other track -> a mover slot: frames [T] of pose / validity (replay), its own
                                    path as a polyline + outline polygon (TrajectoryIDMPolicy route), static / length tests
  TrajectoryNavigation.set_route    checkpoints every 2 m along the reference trajectory

PolyLine restates utils/interpolating_line.py:12-71 (segment construction) in float64; the device tables are float32.
Road lines and road boundaries of a description's `map_features` ARE bodies (line pieces in the scene's own static map: the
side / lane-line detectors see them, crossing them raises the line flags; pinned by tests/golden/scenario_lines.json).  A
description without such features -- e.g. the lanes-only maps scenario_export.py writes -- simply has none: the side detector
then reports "nothing" and no line flag is raised.  Sidewalk / crosswalk polygons are not built.
"""
import copy
import math
import os

import numpy as np

from metadrive_ped_amd import abi
from metadrive_ped_amd.mapgen.tables import beam_table

# ScenarioEnv's own defaults (envs/scenario_env.py:21-95); keys not listed keep BaseEnv's
SCENARIO_DEFAULT_CONFIG = dict(
    data_directory=None,          # a ScenarioNet dataset folder (scenario_data.py); None: descriptions are handed in / synthetic
    start_scenario_index=0, num_scenarios=3, sequential_seed=False,
    no_traffic=False, no_static_vehicles=False, no_light=False, reactive_traffic=False, filter_overlapping_car=True,
    even_sample_vehicle_class=True, default_vehicle_in_traffic=False, static_traffic_object=True,
    success_reward=5.0, out_of_road_penalty=5.0, on_lane_line_penalty=1.0, crash_vehicle_penalty=1.0,
    crash_object_penalty=1.0, crash_human_penalty=1.0, driving_reward=1.0, steering_range_penalty=0.5,
    heading_penalty=1.0, lateral_penalty=0.5, max_lateral_dist=4.0, no_negative_reward=True,
    crash_vehicle_cost=1.0, crash_object_cost=1.0, out_of_road_cost=1.0, crash_human_cost=1.0,
    out_of_route_done=False, crash_vehicle_done=False, crash_object_done=False, crash_human_done=False,
    relax_out_of_road_done=True, allowed_more_steps=None,
    map_region_size=512,          # envs/scenario_env.py:40: line bodies exist within +-256 m of the origin
)
SCENARIO_VEHICLE_CONFIG = dict(lidar=dict(num_lasers=120, distance=50), lane_line_detector=dict(num_lasers=0, distance=50),
                               side_detector=dict(num_lasers=12, distance=50))
_ONLY_SCENARIO_KEYS = ("data_directory", "start_scenario_index", "sequential_seed", "no_traffic", "no_static_vehicles", "no_light",
                       "reactive_traffic", "filter_overlapping_car", "even_sample_vehicle_class",
                       "default_vehicle_in_traffic", "on_lane_line_penalty", "crash_human_penalty",
                       "steering_range_penalty", "heading_penalty", "lateral_penalty", "max_lateral_dist",
                       "no_negative_reward", "crash_human_cost", "relax_out_of_road_done", "allowed_more_steps")

STATIC_THRESHOLD = 3.0        # ScenarioTrafficManager.STATIC_THRESHOLD
IDM_CREATE_MIN_LENGTH = 5.0   # ScenarioTrafficManager.IDM_CREATE_MIN_LENGTH
MIN_VALID_FRAME_LEN = 20      # ScenarioTrafficManager.MIN_VALID_FRAME_LEN
ROUTE_WIDTH = 2.0             # get_idm_route(traj_points, width=2)


def make_scenario_config(user=None):
    """BaseEnv defaults + ScenarioEnv's (scenario_env.py:21-95) + the batch keys; `user` laid over them."""
    from metadrive_ped_amd.config import make_config
    user = copy.deepcopy(dict(user or {}))
    sc = copy.deepcopy(SCENARIO_DEFAULT_CONFIG)
    # agent_policy = ReplayEgoCarPolicy (policy/replay_policy.py:70-82; the reference's own scenario benchmark runs with it,
    # tests/benchmark_FPS/benchmark_waymo.py): the agent replays the SDC track, step()'s actions are ignored
    pol = user.get("agent_policy", "EnvInputPolicy")
    pol = pol if isinstance(pol, str) else getattr(pol, "__name__", repr(pol))
    ego_replay = pol == "ReplayEgoCarPolicy"
    if ego_replay:
        user.pop("agent_policy")
    own = {}
    for k in list(user):
        if k in _ONLY_SCENARIO_KEYS:
            own[k] = user.pop(k)
    for k in _ONLY_SCENARIO_KEYS:
        own.setdefault(k, sc[k])
    base = {k: v for k, v in sc.items() if k not in _ONLY_SCENARIO_KEYS}
    vc = copy.deepcopy(SCENARIO_VEHICLE_CONFIG)
    for k, v in (user.pop("vehicle_config", None) or {}).items():
        if isinstance(v, dict) and k in vc:
            vc[k].update(v)
        else:
            vc[k] = v
    base.update(user)
    base["vehicle_config"] = vc
    base.setdefault("traffic_density", 0.0)
    cfg = make_config(base)
    cfg.update(own)
    cfg["scenario_mode"] = True
    if ego_replay:
        cfg["agent_policy"] = "ReplayEgoCarPolicy"
    if cfg["agent_policy"] == "IDMPolicy":
        raise NotImplementedError("agent_policy=IDMPolicy needs a road network: not in BatchedScenarioEnv (EnvInputPolicy, ReplayEgoCarPolicy)")
    return cfg


class PolyLine:
    """InterpolatingLine (utils/interpolating_line.py): consecutive points are merged until a piece is longer than 1 m;
    pieces shorter than 1e-6 are dropped; a track that never moves becomes one 0.1 m piece along +x."""
    def __init__(self, points):
        pts = np.asarray(points, dtype=np.float64)[..., :2]
        starts, ends = [], []
        i, n = 0, len(pts)
        while i < n - 1:
            j = n - 1
            for q in range(i + 1, n):
                if math.hypot(*(pts[i] - pts[q])) > 1:
                    j = q
                    break
            if math.hypot(*(pts[i] - pts[j])) >= 1e-6:
                starts.append(pts[i])
                ends.append(pts[j])
            i = j
        static = not starts
        if static:
            starts, ends = [pts[0]], [pts[0] + np.array([0.1, 0.0])]
        self.start = np.asarray(starts)
        self.end = np.asarray(ends)
        d = self.end - self.start
        self.seg_len = np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2)       # utils/math.py norm: sqrt(x**2 + y**2)
        self.direction = d / self.seg_len[:, None]
        self.heading = np.arctan2(d[:, 1], d[:, 0])
        self.cum = np.concatenate([[0.0], np.cumsum(self.seg_len)[:-1]])
        self.length = float(sum(self.seg_len.tolist()))           # the reference's left-to-right sum (int(length / 1.5) reads it)
        # lateral_direction = get_vertical_vector(end - start)[1] = (dy, -dx); the never-moving one-piece line has (0, 1)
        # hard-wired (interpolating_line.py:118-131) -- the device derives (dy, -dx) from the direction in every case,
        # which only differs for that degenerate line (a parked SDC: the episode ends at once, route length < 2)
        self.lateral = np.stack([self.direction[:, 1], -self.direction[:, 0]], 1)
        if static:
            self.lateral = np.array([[0.0, 1.0]])

    def _seg_at(self, s):
        """segment() / get_point(): the first piece whose accumulated end + 0.1 reaches s, else the last"""
        acc = 0.0
        for i, L in enumerate(self.seg_len):
            acc += L
            if acc + 0.1 >= s:
                return i
        return len(self.seg_len) - 1

    def position(self, s, lateral=0.0):
        i = self._seg_at(s)
        d = self.direction[i]
        return self.start[i] + (s - self.cum[i]) * d + lateral * self.lateral[i]

    def positions(self, s, lateral=0.0):
        """position() for an array of longitudinals (same piece rule: first accumulated end + 0.1 >= s, else the last)"""
        s = np.asarray(s, dtype=np.float64)
        ends = np.cumsum(self.seg_len)      # the reference accumulates in this order too
        i = np.minimum(np.searchsorted(ends + 0.1, s, side="left"), len(ends) - 1)
        return self.start[i] + (s - self.cum[i])[:, None] * self.direction[i] + lateral * self.lateral[i]

    def heading_at(self, s):
        acc = 0.0
        for i, L in enumerate(self.seg_len):
            acc += L
            if acc > s:
                return float(self.heading[i])
        return float(self.heading[-1])

    def local_coordinates(self, p):
        p = np.asarray(p, dtype=np.float64)
        sgn = ((self.start - p) * self.direction).sum(1)
        t = ((p - self.end) * self.direction).sum(1)
        h = np.maximum.reduce([sgn, t, np.zeros(len(sgn))])
        dpa = p - self.start
        c = dpa[:, 0] * self.direction[:, 1] - dpa[:, 1] * self.direction[:, 0]
        i = int(np.argmin(np.hypot(h, c)))
        d = self.direction[i]
        return float(self.cum[i] + dpa[i] @ d), float(dpa[i] @ self.lateral[i])

    def records(self):
        r = np.zeros(len(self.seg_len), dtype=abi.SEG_DT)
        r["sx"], r["sy"] = self.start[:, 0], self.start[:, 1]
        r["ex"], r["ey"] = self.end[:, 0], self.end[:, 1]
        r["dx"], r["dy"] = self.direction[:, 0], self.direction[:, 1]
        r["len"], r["heading"], r["cum"] = self.seg_len, self.heading, self.cum
        return r

    def checkpoints(self):
        """TrajectoryNavigation.discretize_reference_trajectory (trajectory_navigation.py:96-103)"""
        num = int(self.length / 2.0)
        return np.concatenate([self.positions(np.arange(num) * 2.0), self.position(self.length, 0.0)[None]])

    def outline(self, width=ROUTE_WIDTH):
        """PointLane.auto_generate_polygon (component/lane/point_lane.py:60-106): the strip of the given width sampled
        every metre, prolonged by one metre beyond both ends."""
        h0, h1 = self.heading_at(0.0), self.heading_at(self.length)
        d0 = np.array([math.cos(h0), math.sin(h0)])
        d1 = np.array([math.cos(h1), math.sin(h1)])
        longs = np.arange(0, self.length + 1.0, 1.0)
        out = []
        for side in (0, 1):
            seq = longs if side == 0 else longs[::-1]
            lat = -width / 2 if side == 0 else width / 2
            pts = self.positions(seq, lat)
            for t in range(len(seq)):
                p = pts[t]
                at_start = (t == 0 and side == 0) or (t == len(seq) - 1 and side == 1)
                at_end = (t == 0 and side == 1) or (t == len(seq) - 1 and side == 0)
                if at_start:
                    if side == 1:
                        out.append(p)
                    out.append(p - d0)
                    if side == 0:
                        out.append(p)
                elif at_end:
                    if side == 0:
                        out.append(p)
                    out.append(p + d1)
                    if side == 1:
                        out.append(p)
                else:
                    out.append(p)
        return np.asarray(out)


def _first_run(valid):
    """[t0, t1) of the first run of valid frames, or None"""
    idx = np.nonzero(valid)[0]
    if len(idx) == 0:
        return None
    t0 = int(idx[0])
    t1 = t0
    while t1 < len(valid) and valid[t1]:
        t1 += 1
    return t0, t1


def _all_runs(valid):
    """every maximal run of valid frames [t0, t1): what get_max_valid_indicis (scenario/parse_object_state.py:8-16) returns the
    end of for any frame inside it"""
    v = np.concatenate([[False], np.asarray(valid, bool), [False]])
    d = np.diff(v.astype(np.int8))
    return list(zip(np.nonzero(d == 1)[0].tolist(), np.nonzero(d == -1)[0].tolist()))


_KIND_OF_TYPE = {"VEHICLE": abi.KIND_VEHICLE, "PEDESTRIAN": abi.KIND_PEDESTRIAN, "CYCLIST": abi.KIND_CYCLIST,
                 "TRAFFIC_CONE": abi.KIND_CONE, "TRAFFIC_BARRIER": abi.KIND_BARRIER}


def vehicle_class_for(length, counters):
    """get_vehicle_type with even sampling (scenario_traffic_manager.py:339-361).  `counters`: the per-episode type
    counts; the reference seeds them from the traffic manager's stream, whose position depends on Bullet-side
    spawns: they start at 0 here (unpinned)."""
    if length <= 4:
        return "s"
    if length <= 5.5:
        counters[1] += 1
        return ["l", "m", "s"][counters[1] % 3]
    counters[2] += 1
    return ["l", "xl"][counters[2] % 2]


STRIPE_LENGTH = 1.5          # PGDrivableAreaProperty.STRIPE_LENGTH (constants.py:313)
_CONT_WHITE = ("UNKNOWN_LINE", "ROAD_LINE_SOLID_SINGLE_WHITE", "ROAD_LINE_SOLID_DOUBLE_WHITE",
               "UNKNOWN", "ROAD_EDGE_BOUNDARY", "ROAD_EDGE_MEDIAN")          # is_road_boundary_line -> continuous, grey
_CONT_YELLOW = ("ROAD_LINE_SOLID_SINGLE_YELLOW", "ROAD_LINE_SOLID_DOUBLE_YELLOW", "ROAD_LINE_PASSING_DOUBLE_YELLOW")
_BROKEN = ("ROAD_LINE_BROKEN_SINGLE_WHITE", "ROAD_LINE_BROKEN_SINGLE_YELLOW", "ROAD_LINE_BROKEN_DOUBLE_YELLOW")


def line_pieces(polyline, broken):
    """The stripes ScenarioBlock.construct_continuous_line / construct_broken_line cut a road line into
    (component/scenario_block/scenario_block.py:74-99): [(start, end)] in order."""
    line = PolyLine(polyline)
    if broken:
        n = int(line.length / (2 * STRIPE_LENGTH))
        k = np.arange(n, dtype=np.float64)
        sa = k * STRIPE_LENGTH * 2
        sb = k * STRIPE_LENGTH * 2 + STRIPE_LENGTH
        if n:
            sb[-1] = line.length - STRIPE_LENGTH
    else:
        n = int(line.length / STRIPE_LENGTH)
        k = np.arange(n, dtype=np.float64)
        sa = STRIPE_LENGTH * k
        sb = (k + 1) * STRIPE_LENGTH
        if n:
            sb[-1] = line.length
    if n == 0:
        return []
    pa, pb = line.positions(sa), line.positions(sb)      # get_point for all stripes at once (same piece rule)
    return [(pa[i], pb[i]) for i in range(n)]


def scene_line_quads(map_features, map_region_size):
    """Line bodies of a scenario map (ScenarioBlock.create_in_world, scenario_block.py:45-72): every road line / road
    boundary feature with at least two points, cut into stripes, each a box LANE_LINE_WIDTH / 2 wide; pieces whose
    middle lies outside the map region are not built (block/base_block.py:481).  Sidewalk / crosswalk POLYGONS are not
    turned into bodies here."""
    from metadrive_ped_amd.mapgen.tables import _line_box
    quads, kinds = [], []
    for fid, f in (map_features or {}).items():
        typ = f.get("type")
        if "polyline" not in f or len(f["polyline"]) <= 1:
            continue
        if typ in _BROKEN:
            kind, broken = abi.Q_LINE_BROKEN, True
        elif typ in _CONT_YELLOW:
            kind, broken = abi.Q_LINE_YELLOW_CONT, False
        elif typ in _CONT_WHITE:
            kind, broken = abi.Q_LINE_WHITE_CONT, False
        else:
            continue
        for a, b in line_pieces(np.asarray(f["polyline"], dtype=np.float64)[:, :2], broken):
            q = _line_box(a, b, region=map_region_size)
            if q is not None:
                quads.append(q)
                kinds.append(kind)
    return quads, kinds


def _to_frames(a, T):
    """first T frames of a per-frame array, zero / False padded"""
    a = np.asarray(a)
    if len(a) >= T:
        return a[:T]
    out = np.zeros((T, ) + a.shape[1:], dtype=a.dtype)
    out[:len(a)] = a
    return out


def _build_scene(job):
    """One scenario description -> the per-scene arrays (module-level so that a fork pool can run it)."""
    from metadrive_ped_amd.scene import vehicle_param_record
    from metadrive_ped_amd.rng import get_np_random
    e, sc, cap, T, seed, dt, no_traffic, region = job
    shape0 = np.zeros(cap, dtype=abi.SHAPE_DT)
    shape0["aux"] = -1
    dyn0 = np.zeros(cap, dtype=abi.DYN_DT)
    param = np.zeros(cap, dtype=abi.PARAM_DT)
    param["max_speed_kmh"], param["lf"], param["lr"] = 80.0, 1.0, 1.0
    fshape = np.zeros((T, cap), dtype=abi.SHAPE_DT)
    fshape["aux"] = -1
    fdyn = np.zeros((T, cap, 2), np.float32)
    meta = np.zeros((cap, 4), np.int32)
    meta[:, 2] = abi.TM_NEVER
    runs = [[] for _ in range(cap)]      # vehicles: all valid runs (a route cut at a later spawn frame ends with its run)
    cut_frames, cut_metres = 1, 1.0      # bounds of such a route: frames of the longest run, its path length
    sdc_id = str(sc["metadata"]["sdc_id"])
    order = [sdc_id] + [str(k) for k in sc["tracks"] if str(k) != sdc_id]
    counters = [0, 0, 0]
    polys = [None] * cap
    ck = None
    for j, oid in enumerate(order):
        tr = sc["tracks"][oid] if oid in sc["tracks"] else sc["tracks"][int(oid)]
        st = tr["state"]
        valid = np.asarray(st["valid"]).astype(bool)
        pos = np.asarray(st["position"], dtype=np.float64)[:, :2]
        heading = np.asarray(st["heading"], dtype=np.float64)
        vel = np.asarray(st["velocity"], dtype=np.float64)
        own_len = len(pos)
        if own_len != T:   # a batch of scenes of different lengths: shorter ones end with invalid frames
            valid, pos, heading, vel = (_to_frames(a, T) for a in (valid, pos, heading, vel))
        n = j
        run = _first_run(valid)
        if j == 0:
            # the agent: default vehicle at the SDC's first frame (scenario_map_manager.py:55-75); its route =
            # the whole track up to the first > 100 m jump (parse_full_trajectory, parse_object_state.py:77-90)
            cut = own_len
            for t in range(own_len - 1):
                if math.hypot(*(pos[t] - pos[t + 1])) > 100:
                    cut = t
                    break
            polys[0] = PolyLine(pos[:cut])
            prm, length, width, _ = vehicle_param_record("default", int(get_np_random(seed).randint(0, 2 ** 16)), dt)
            param[n] = prm
            h = float(heading[0])
            sh = shape0[n]
            sh["cx"], sh["cy"], sh["c"], sh["s"] = pos[0, 0], pos[0, 1], math.cos(h), math.sin(h)
            sh["hl"], sh["hw"] = length / 2, width / 2
            sh["flags"] = abi.KIND_VEHICLE | abi.F_ALIVE | abi.F_AGENT
            shape0[n] = sh
            d = dyn0[n]
            d["heading"], d["speed"] = h, float(vel[0, 0] * math.cos(h) + vel[0, 1] * math.sin(h))
            d["last_x"], d["last_y"], d["last_c"], d["last_s"] = pos[0, 0], pos[0, 1], math.cos(h), math.sin(h)
            dyn0[n] = d
            meta[n] = (0, int(sc.get("length", own_len)), abi.TM_NEVER, 0)   # [1] = this scene's current_scenario_length
            ck = polys[0].checkpoints()
            # the SDC's own frames (read only with agent_policy = ReplayEgoCarPolicy)
            f = fshape[:, 0]
            f["cx"][valid], f["cy"][valid] = pos[valid, 0], pos[valid, 1]
            f["c"][valid], f["s"][valid] = np.cos(heading[valid]), np.sin(heading[valid])
            f["hl"][valid], f["hw"][valid] = length / 2, width / 2
            f["flags"][valid] = abi.KIND_VEHICLE | abi.F_ALIVE | abi.F_AGENT
            fshape[:, 0] = f
            fdyn[valid, 0, 0] = heading[valid]
            fdyn[valid, 0, 1] = np.hypot(vel[valid, 0], vel[valid, 1])
            continue
        kind = _KIND_OF_TYPE.get(tr["type"])
        if kind is None or run is None or no_traffic:
            continue
        t0, t1 = run
        flags = 0
        if kind == abi.KIND_VEHICLE:
            vp = pos[valid]
            if float(np.max(np.std(vp, axis=0)[:2])) > STATIC_THRESHOLD:
                flags |= abi.TM_MOVING
            if math.hypot(*(pos[t0] - pos[t1 - 1])) > IDM_CREATE_MIN_LENGTH:
                flags |= abi.TM_LENGTH_OK
            rec_len = float(np.asarray(st["length"])[t0]) if "length" in st else 4.5
            vtype = vehicle_class_for(rec_len, counters)
            prm, length, width, _ = vehicle_param_record(vtype, int(get_np_random(seed * 131 + j).randint(0, 2 ** 16)), dt)
            param[n] = prm
            hl, hw = length / 2, width / 2
            polys[j] = PolyLine(pos[t0:t1])
            runs[j] = _all_runs(valid)
            for a_, b_ in runs[j]:
                cut_frames = max(cut_frames, b_ - a_)
                if b_ - a_ > 1:
                    step = np.diff(pos[a_:b_], axis=0)
                    cut_metres = max(cut_metres, float(np.sqrt((step ** 2).sum(1)).sum()))
        elif kind == abi.KIND_PEDESTRIAN:
            hl = hw = 0.35
        elif kind == abi.KIND_CYCLIST:
            hl, hw = 0.875, 0.2
        elif kind == abi.KIND_CONE:
            hl = hw = 0.2
            if int(valid.sum()) < MIN_VALID_FRAME_LEN:
                flags |= abi.TM_NEVER
        else:
            hl, hw = 0.15, 1.0
            if int(valid.sum()) < MIN_VALID_FRAME_LEN:
                flags |= abi.TM_NEVER
        meta[n] = (t0, t1, flags, 0)
        f = fshape[:, n]
        f["cx"][valid], f["cy"][valid] = pos[valid, 0], pos[valid, 1]
        f["c"][valid], f["s"][valid] = np.cos(heading[valid]), np.sin(heading[valid])
        f["hl"][valid], f["hw"][valid] = hl, hw
        fl = kind | abi.F_ALIVE
        if kind in (abi.KIND_CONE, abi.KIND_BARRIER):
            fl |= abi.F_STATIC
        f["flags"][valid] = fl
        fshape[:, n] = f
        fdyn[valid, n, 0] = heading[valid]
        fdyn[valid, n, 1] = np.hypot(vel[valid, 0], vel[valid, 1])
        # a free slot still carries the class's size, so that the device only rewrites the pose on a spawn
        shape0[n]["hl"], shape0[n]["hw"] = hl, hw
    segs, verts = [], []
    for j in range(cap):
        pl = polys[j]
        segs.append(pl.records() if pl is not None else np.zeros(0, dtype=abi.SEG_DT))
        want_outline = j > 0 and pl is not None and (meta[j, 2] & abi.TM_MOVING) and (meta[j, 2] & abi.TM_LENGTH_OK)
        verts.append(pl.outline() if want_outline else np.zeros((0, 2)))
    from metadrive_ped_amd.mapgen.tables import StaticTables
    static = StaticTables(*scene_line_quads(sc.get("map_features"), region))      # road-line bodies + their grid
    return dict(shape0=shape0, dyn0=dyn0, param=param, fshape=fshape, fdyn=fdyn, meta=meta, order=order, segs=segs, verts=verts, ckpt=ck,
                static=static, runs=runs, cut_frames=cut_frames, cut_metres=cut_metres)



class _World:
    def __init__(self, arrays, n_envs):
        self.arrays = arrays
        self.n_maps = 1
        self.n_envs = n_envs


def _poly_aux(poly_off, segs, polyv_off, polyv):
    """MdWorld.poly_aux: per slot the polyline's end point -- float32 arithmetic in the order of md_poly_position(p, length, 0):
    the first piece with cum + len + 0.1 >= length, then sx + (length - cum) * dx -- and the bounding box of the outline."""
    f = np.float32
    n_slot = len(poly_off) - 1
    out = np.zeros((n_slot, 8), np.float32)
    for k in range(n_slot):
        a, b = int(poly_off[k]), int(poly_off[k + 1])
        if b > a:
            g = segs[a:b]
            length = f(g["cum"][-1]) + f(g["len"][-1])
            ends = (g["cum"].astype(f) + g["len"].astype(f)) + f(0.1)
            hit = np.nonzero(ends >= length)[0]
            ge = g[int(hit[0]) if len(hit) else b - a - 1]
            along = length - f(ge["cum"])
            out[k, 0] = f(ge["sx"]) + along * f(ge["dx"])
            out[k, 1] = f(ge["sy"]) + along * f(ge["dy"])
        va, vb = int(polyv_off[k]), int(polyv_off[k + 1])
        if vb > va:
            v = polyv[va:vb]
            out[k, 2:6] = v[:, 0].min(), v[:, 1].min(), v[:, 0].max(), v[:, 1].max()
        else:
            out[k, 2:6] = 1.0, 1.0, -1.0, -1.0      # empty box: nothing is inside
    return out


def _poly_balls(poly_off, segs):
    """MdWorld.poly_ball / poly_ball_off: a circle around every group of MD_POLY_GROUP consecutive pieces of a slot's polyline
    (centre = middle of the end points' bounding box, radius = farthest end point, in float64; stored as float32 with the radius
    rounded UP past the centre's rounding and a 1e-3 m margin): the projections' exact cull."""
    G = abi.MD_POLY_GROUP
    po = np.asarray(poly_off, np.int64)
    n_g = (np.diff(po) + G - 1) // G
    off = np.zeros(len(po), np.int64)
    off[1:] = np.cumsum(n_g)
    total = int(off[-1])
    out = np.zeros((max(total, 1), 4), np.float32)
    if total == 0:
        return out, off.astype(np.int32)
    # first piece of every group, in order (the pieces of a group are contiguous in `segs`)
    starts = np.repeat(po[:-1], n_g) + G * (np.arange(total) - np.repeat(off[:-1], n_g))
    count = np.minimum(G, np.repeat(po[1:], n_g) - starts)
    gid = np.repeat(np.arange(total), count)             # group of every piece of every polyline (pieces outside polylines: none)
    piece = np.repeat(starts, count) + (np.arange(len(gid)) - np.repeat(np.cumsum(count) - count, count))
    sx, sy = segs["sx"][piece].astype(np.float64), segs["sy"][piece].astype(np.float64)
    ex, ey = segs["ex"][piece].astype(np.float64), segs["ey"][piece].astype(np.float64)
    first = np.cumsum(count) - count
    xmin = np.minimum(np.minimum.reduceat(sx, first), np.minimum.reduceat(ex, first))
    xmax = np.maximum(np.maximum.reduceat(sx, first), np.maximum.reduceat(ex, first))
    ymin = np.minimum(np.minimum.reduceat(sy, first), np.minimum.reduceat(ey, first))
    ymax = np.maximum(np.maximum.reduceat(sy, first), np.maximum.reduceat(ey, first))
    cx, cy = (0.5 * (xmin + xmax)).astype(np.float32), (0.5 * (ymin + ymax)).astype(np.float32)
    cxd, cyd = cx.astype(np.float64)[gid], cy.astype(np.float64)[gid]
    d = np.maximum(np.hypot(sx - cxd, sy - cyd), np.hypot(ex - cxd, ey - cyd))
    r = np.maximum.reduceat(d, first)
    out[:total, 0], out[:total, 1] = cx, cy
    out[:total, 2] = np.nextafter((r + 1.0e-3 + 1.0e-6 * (np.abs(cx) + np.abs(cy))).astype(np.float32), np.float32(np.inf))
    return out, off.astype(np.int32)


class ScenarioHostScene:
    """The HostScene of scenario mode: one scenario description per env (`scenarios[e]` -> env e)."""
    def __init__(self, cfg, scenarios):
        from metadrive_ped_amd.engine import make_md_config
        from metadrive_ped_amd.scene import vehicle_param_record
        from metadrive_ped_amd.rng import get_np_random
        self.cfg = cfg
        E = cfg["num_envs"]
        if len(scenarios) != E:
            raise ValueError("need one scenario per env: got {} for {} envs".format(len(scenarios), E))
        T = max(int(sc["length"]) for sc in scenarios)   # frames of the batch; a shorter scene is over (all invalid) after its own
        n_tracks = max(len(sc["tracks"]) for sc in scenarios)
        cap = cfg["mover_capacity"] or min(abi.MD_MAX_CAP, max(8, (n_tracks + 7) // 8 * 8))
        if n_tracks > cap:
            raise ValueError("a scenario holds {} objects, the mover capacity is {}".format(n_tracks, cap))
        A = 1
        self.E, self.cap, self.A, self.T = E, cap, A, T
        vc = cfg["vehicle_config"]
        self.n_beams = int(vc["lidar"]["num_lasers"]) if vc["lidar"]["distance"] > 0 else 0
        self.n_side = int(vc["side_detector"]["num_lasers"]) if vc["side_detector"]["distance"] > 0 else 0
        self.n_ll = int(vc["lane_line_detector"]["num_lasers"]) if vc["lane_line_detector"]["distance"] > 0 else 0
        self.obs_base = 0
        self.state_dim = (self.n_side or 2) + 6 + (self.n_ll or 1) + 22
        self.num_others, self.add_others_navi, self.others_dim = 0, False, 0
        self.obs_dim = self.state_dim + self.n_beams
        # scene e <-> dataset index start_scenario_index + (env_seed_offset + e) % num_scenarios (scenario_data.scenario_indices):
        # the identity checkpoints and track sets are checked against, and the fallback parameter seed of a description that
        # carries none -- tied to WHICH scenario it is, not to where it sits in the batch
        from metadrive_ped_amd.scenario_data import scenario_indices
        self.seeds = scenario_indices(cfg, E)
        self.scenario_ids = [str(sc.get("id", sc.get("metadata", {}).get("scenario_id", i))) for sc, i in zip(scenarios, self.seeds)]
        self.spawn = None
        self.traffic_respawns = False
        self.scenes, self.map_tables = {}, []
        N = E * cap
        dt = cfg["physics_world_step_size"]

        # vehicle parameters are sampled from a stream seeded by the scenario's OWN seed where it carries one, so that a
        # scene behaves the same in whatever batch (slot, shard) it is loaded
        jobs = [(e, scenarios[e], cap, T, int(scenarios[e]["metadata"].get("seed", self.seeds[e])), dt, bool(cfg["no_traffic"]),
                 float(cfg["map_region_size"])) for e in range(E)]
        from metadrive_ped_amd import hostpool
        built = hostpool.build_all(_build_scene, jobs, workers=int(cfg.get("build_workers", 0)))
        shape0 = np.concatenate([b_["shape0"] for b_ in built])
        dyn0 = np.concatenate([b_["dyn0"] for b_ in built])
        param = np.concatenate([b_["param"] for b_ in built])
        nav0 = np.zeros(N, dtype=abi.NAV_DT)
        nav0["lane"], nav0["target_lane"], nav0["road0"], nav0["road1"] = -1, -1, -1, -1
        pid0 = np.zeros(N, dtype=abi.PID_DT)
        pid0["target_speed"] = 40.0
        fshape = np.concatenate([b_["fshape"] for b_ in built], axis=1)
        fdyn = np.concatenate([b_["fdyn"] for b_ in built], axis=1)
        meta = np.concatenate([b_["meta"] for b_ in built])
        self.track_ids = [b_["order"] for b_ in built]
        segs, poly_off, verts, polyv_off, ckpts, ckpt_off = [], [0], [], [0], [], [0]
        for b_ in built:
            for r in b_["segs"]:
                segs.append(r)
                poly_off.append(poly_off[-1] + len(r))
            for v in b_["verts"]:
                verts.append(v)
                polyv_off.append(polyv_off[-1] + len(v))
            ckpts.append(b_["ckpt"])
            ckpt_off.append(ckpt_off[-1] + len(b_["ckpt"]))
        # static bodies: one map per scene (its road lines + their grid); the lane / road / node tables are placeholders
        from metadrive_ped_amd.mapgen.tables import WorldTables
        self.map_tables = [b_["static"] for b_ in built]
        self.world = WorldTables(self.map_tables, list(range(E)), beam_table(self.n_beams))
        a = self.world.arrays
        a["poly_off"] = np.asarray(poly_off, np.int32)
        a["segs"] = np.concatenate(segs) if sum(len(x) for x in segs) else np.zeros(1, dtype=abi.SEG_DT)
        a["polyv_off"] = np.asarray(polyv_off, np.int32)
        vv = np.concatenate(verts) if sum(len(x) for x in verts) else np.zeros((1, 2))
        a["polyv"] = np.ascontiguousarray(vv, dtype=np.float32)
        a["ckpt_off"] = np.asarray(ckpt_off, np.int32)
        a["ckpt_xy"] = np.ascontiguousarray(np.concatenate(ckpts), dtype=np.float32)
        a["track_meta"] = meta
        run_off, run_list = [0], []
        for b_ in built:
            for r in b_["runs"]:
                run_list.extend(r)
                run_off.append(run_off[-1] + len(r))
        a["run_off"] = np.asarray(run_off, np.int32)
        a["runs"] = np.asarray(run_list if run_list else [(0, 0)], np.int32).reshape(-1, 2)
        # buffers for routes cut at a later spawn frame (MdState.route_*): at most one piece per frame of the longest run, the
        # outline two vertices per metre of its path + the end caps
        self.route_seg_cap = int(max(b_["cut_frames"] for b_ in built))
        self.route_vert_cap = 2 * (int(math.ceil(max(b_["cut_metres"] for b_ in built))) + 3) + 4
        a["poly_aux"] = _poly_aux(a["poly_off"], a["segs"], a["polyv_off"], a["polyv"])
        a["poly_ball"], a["poly_ball_off"] = _poly_balls(a["poly_off"], a["segs"])
        st = {}
        st["shape0"], st["dyn0"], st["nav0"], st["pid0"], st["param"] = shape0, dyn0, nav0, pid0, param
        st["route_nodes"] = np.full((N, abi.MD_ROUTE_LEN), -1, np.int32)
        st["route_roads"] = np.full((N, abi.MD_ROUTE_LEN), -1, np.int32)
        st["final_lane"] = np.zeros(N, np.int32)
        st["idm_rand"] = np.zeros((N, abi.MD_IDM_RAND), np.int32)
        for k in ("shape", "dyn", "nav", "pid"):
            st[k] = st[k + "0"].copy()
        st["action"] = np.zeros((N, 2), np.float32)
        st["flags"] = np.zeros(N, np.uint32)
        st["obs"] = np.zeros((E * A, self.obs_dim), np.float32)
        st["reward"] = np.zeros(E * A, np.float32)
        st["cost"] = np.zeros(E * A, np.float32)
        st["step_info"] = np.zeros((E * A, 8), np.float32)
        st["done_out"] = np.zeros((E * A, 4), np.uint8)
        st["need_reset"] = np.ones(E, np.int32)
        st["next_agent_id"] = np.zeros(E, np.int32)     # ScenarioTrafficManager.idm_policy_count
        if cfg["reactive_traffic"]:
            st["route_n"] = np.zeros((N, 4), np.int32)
            st["route_segs"] = np.zeros((N, self.route_seg_cap), dtype=abi.SEG_DT)
            st["route_verts"] = np.zeros((N, self.route_vert_cap, 2), np.float32)
            st["route_aux"] = np.zeros((N, 8), np.float32)
        self.state = st
        self.tracks = dict(shape=fshape, dyn=fdyn, seeds=list(self.seeds), cap=cap)
        k = make_md_config(dict(cfg, traffic_mode="trigger"), E, A, cap, self.n_beams)
        k.traffic_mode = 4
        k.n_side, k.n_lane_line = self.n_side, self.n_ll
        k.obs_dim = self.obs_dim
        k.track_len = T
        k.scenario_length = T
        for name in ("on_lane_line_penalty", "crash_human_penalty", "steering_range_penalty", "heading_penalty",
                     "lateral_penalty", "max_lateral_dist", "crash_human_cost"):
            setattr(k, name, float(cfg[name]))
        for name in ("no_negative_reward", "relax_out_of_road_done", "reactive_traffic", "filter_overlapping_car",
                     "no_static_vehicles"):
            setattr(k, name, int(bool(cfg[name])))
        k.allowed_more_steps = int(cfg["allowed_more_steps"] or 0)
        k.route_seg_cap, k.route_vert_cap = self.route_seg_cap, self.route_vert_cap
        k.ego_replay = int(cfg["agent_policy"] == "ReplayEgoCarPolicy")
        self.md_config = k
        self.side_beams = beam_table(self.n_side, np.pi / 2) if self.n_side else None
        self.ll_beams = beam_table(self.n_ll, np.pi / 2) if self.n_ll else None

    def clone_state(self):
        return {k: v.copy() for k, v in self.state.items()}


# --------------------------------------------------------------------------------------------------
# Synthetic scenario descriptions (bench / tests): the container holds no ScenarioNet data (nuScenes, Waymo), so scenes
# of the same SHAPE are generated -- a curving multi-lane road, an SDC track along it, vehicles ahead, beside and
# behind it (the ones behind are what ScenarioTrafficManager makes reactive), parked cars, late-appearing and
# vanishing tracks, pedestrians, cones -- in the reference's own description format (scenario_description.py:1-120).
# --------------------------------------------------------------------------------------------------
def _path(rng, n_pts, ds=0.5):
    """a smooth centre line: heading = integral of a slowly varying curvature"""
    amp = rng.uniform(0.004, 0.02)
    wl = rng.uniform(120.0, 300.0)
    ph = rng.uniform(0, 2 * math.pi)
    s = np.arange(n_pts) * ds
    kappa = amp * np.sin(2 * math.pi * s / wl + ph)
    h0 = rng.uniform(-math.pi, math.pi)
    heading = h0 + np.cumsum(kappa) * ds
    x = np.cumsum(np.cos(heading)) * ds + rng.uniform(-500, 500)
    y = np.cumsum(np.sin(heading)) * ds + rng.uniform(-500, 500)
    return s, np.stack([x, y], 1), heading


def _sample(s_axis, xy, heading, s, lateral):
    """pose at arc length s (clamped) with a lateral offset to the right"""
    s = np.clip(s, s_axis[0], s_axis[-1])
    x = np.interp(s, s_axis, xy[:, 0])
    y = np.interp(s, s_axis, xy[:, 1])
    h = np.interp(s, s_axis, heading)
    return x + lateral * np.sin(h), y - lateral * np.cos(h), h


def _track_dict(oid, typ, T, valid, x, y, h, speed, length, width, height):
    v = valid.astype(np.float32)
    pos = np.zeros((T, 3), np.float32)
    pos[:, 0], pos[:, 1] = x * v, y * v
    vel = np.stack([speed * np.cos(h), speed * np.sin(h)], 1).astype(np.float32) * v[:, None]
    return {"type": typ,
            "state": {"position": pos, "heading": (h * v).astype(np.float32), "velocity": vel, "valid": valid.copy(),
                      "length": np.full(T, length, np.float32) * v, "width": np.full(T, width, np.float32) * v,
                      "height": np.full(T, height, np.float32) * v},
            "metadata": {"type": typ, "object_id": str(oid), "track_length": int(T)}}


def synthetic_scenario(seed, T=200, n_vehicles=18, n_parked=3, n_pedestrians=2, n_cones=4):
    rng = np.random.RandomState(seed)
    s_axis, xy, heading = _path(rng, 4000)
    t = np.arange(T) * 0.1
    tracks = {}
    ego_v = rng.uniform(6.0, 11.0)
    ego_s0 = 600.0
    ego_s = ego_s0 + ego_v * t + 0.5 * rng.uniform(-0.15, 0.15) * t * t
    # coordinates relative to the SDC's first position, as the dataset converters deliver them: ScenarioEnv only builds
    # the line bodies whose middle lies within map_region_size / 2 (= 256 m) of the origin (block/base_block.py:481)
    x0_, y0_, _ = _sample(s_axis, xy, heading, np.asarray([ego_s0]), 0.0)
    xy = xy - np.array([float(x0_[0]), float(y0_[0])])
    x, y, h = _sample(s_axis, xy, heading, ego_s, 0.0)
    sp = np.gradient(ego_s, 0.1)
    tracks["0"] = _track_dict("0", "VEHICLE", T, np.ones(T, bool), x, y, h, sp, 4.5, 1.85, 1.5)
    oid = 1
    for i in range(n_vehicles):
        lane = float(rng.choice([-3.5, 0.0, 3.5]))
        behind = i < n_vehicles // 2
        ds0 = rng.uniform(-45.0, -9.0) if behind else rng.uniform(9.0, 70.0)
        if lane == 0.0 and abs(ds0) < 12.0:
            ds0 = math.copysign(12.0, ds0)
        v = max(0.5, ego_v + rng.uniform(-3.0, 3.0))
        s_v = ego_s0 + ds0 + v * t
        x, y, h = _sample(s_axis, xy, heading, s_v, lane)
        valid = np.ones(T, bool)
        r = rng.rand()
        if r < 0.15:
            valid[:int(rng.randint(5, 60 if T > 85 else max(6, T // 3)))] = False          # appears later
        elif r < 0.3:
            valid[int(rng.randint(80 if T > 85 else T // 2, T - 5)):] = False       # vanishes
        length = float(rng.choice([3.9, 4.6, 5.2, 6.0]))
        tracks[str(oid)] = _track_dict(oid, "VEHICLE", T, valid, x, y, h, np.full(T, v), length, 1.9, 1.6)
        oid += 1
    for i in range(n_parked):
        s_p = ego_s0 + rng.uniform(-30.0, 150.0)
        x, y, h = _sample(s_axis, xy, heading, np.full(T, s_p), float(rng.choice([-6.5, 6.5])))
        tracks[str(oid)] = _track_dict(oid, "VEHICLE", T, np.ones(T, bool), x, y, h, np.zeros(T), 4.4, 1.8, 1.5)
        oid += 1
    for i in range(n_pedestrians):
        s_p = ego_s0 + rng.uniform(10.0, 120.0) + rng.uniform(-1.0, 1.0) * t
        x, y, h = _sample(s_axis, xy, heading, s_p, float(rng.choice([-7.5, 7.5])))
        typ = "PEDESTRIAN" if i % 2 == 0 else "CYCLIST"
        tracks[str(oid)] = _track_dict(oid, typ, T, np.ones(T, bool), x, y, h, np.full(T, 1.0), 0.7, 0.7, 1.75)
        oid += 1
    for i in range(n_cones):
        s_p = ego_s0 + 40.0 + 3.0 * i
        x, y, h = _sample(s_axis, xy, heading, np.full(T, s_p), 5.2)
        valid = np.ones(T, bool)
        if i == n_cones - 1:
            valid[10:] = False                                 # a noise object: fewer than MIN_VALID_FRAME_LEN frames
        tracks[str(oid)] = _track_dict(oid, "TRAFFIC_CONE", T, valid, x, y, h, np.zeros(T), 0.4, 0.4, 1.0)
        oid += 1
    # the road itself: three lanes 3.5 m wide around the centre line, solid white edges, broken white separators, a
    # road boundary 0.75 m outside each edge, and (odd seeds) a solid yellow line instead of the right-hand edge
    feats = {}
    s_road = np.arange(ego_s0 - 80.0, ego_s0 + 260.0, 2.0)

    def offset_line(lateral):
        px, py, _ = _sample(s_axis, xy, heading, s_road, lateral)
        return np.stack([px, py, np.zeros_like(px)], 1).astype(np.float32)
    for i, lat in enumerate((-3.5, 0.0, 3.5)):
        feats["lane%d" % i] = {"type": "LANE_SURFACE_STREET", "polyline": offset_line(lat), "entry_lanes": [], "exit_lanes": [],
                               "left_neighbor": [], "right_neighbor": []}
    feats["line_left"] = {"type": "ROAD_LINE_SOLID_SINGLE_WHITE", "polyline": offset_line(-5.25)}
    feats["line_right"] = {"type": "ROAD_LINE_SOLID_SINGLE_YELLOW" if seed % 2 else "ROAD_LINE_SOLID_SINGLE_WHITE",
                           "polyline": offset_line(5.25)}
    feats["sep_left"] = {"type": "ROAD_LINE_BROKEN_SINGLE_WHITE", "polyline": offset_line(-1.75)}
    feats["sep_right"] = {"type": "ROAD_LINE_BROKEN_SINGLE_WHITE", "polyline": offset_line(1.75)}
    feats["edge_left"] = {"type": "ROAD_EDGE_BOUNDARY", "polyline": offset_line(-6.0)}
    feats["edge_right"] = {"type": "ROAD_EDGE_BOUNDARY", "polyline": offset_line(6.0)}
    return {"id": "synthetic-%d" % seed, "version": "metadrive_ped_amd synthetic (MetaDrive v0.4.2.2 scenario format)",
            "length": int(T),
            "metadata": {"ts": t.astype(np.float32), "metadrive_processed": False, "coordinate": "metadrive",
                         "dataset": "synthetic", "seed": int(seed), "sdc_id": "0", "scenario_id": "synthetic-%d" % seed},
            "tracks": tracks, "dynamic_map_states": {}, "map_features": feats}


def synthetic_scenarios(n, seed0=0, **kw):
    return [synthetic_scenario(seed0 + i, **kw) for i in range(n)]

"""Recorded batches -> scenario descriptions in the reference's unified format.

`tracks` is what BatchedEngine.stop_recording() returns (poses of every mover slot per step); this module turns
the episode of env e into the dict layout documented in metadrive/scenario/scenario_description.py:1-120 and checked
by ScenarioDescription.sanity_check (:200-257): first-level keys id / version / length / metadata / tracks /
dynamic_map_states / map_features; per track type + state{position [T,3], heading [T], velocity [T,2], valid [T],
length / width / height [T]} + metadata{type, object_id, track_length}; per lane a centre polyline (1 m spacing),
its outline polygon and the topological neighbours; metadata with ts / metadrive_processed / coordinate.  It is the
role of BaseEnv.export_scenarios + convert_recorded_scenario_exported (envs/base_env.py:775-836,
scenario/utils.py) for the batched engine.  Values where a slot is not valid are zero, as the reference requires.

Nothing here touches the device: inputs are numpy (or torch tensors, converted once).
"""
import math

import numpy as np

from metadrive_ped_amd import abi

VERSION = "metadrive_ped_amd (MetaDrive v0.4.2.2 scenario format)"
DT = 0.1   # decision_repeat * physics_world_step_size

_TYPE_OF_KIND = {abi.KIND_VEHICLE: "VEHICLE", abi.KIND_CONE: "TRAFFIC_CONE", abi.KIND_WARNING: "TRAFFIC_CONE",
                 abi.KIND_BARRIER: "TRAFFIC_BARRIER", abi.KIND_PEDESTRIAN: "PEDESTRIAN", abi.KIND_CYCLIST: "CYCLIST"}
_HEIGHT_OF_KIND = {abi.KIND_VEHICLE: 1.5, abi.KIND_CONE: 1.0, abi.KIND_WARNING: 1.2, abi.KIND_BARRIER: 2.0,
                   abi.KIND_PEDESTRIAN: 1.75, abi.KIND_CYCLIST: 1.75}


def _as_numpy(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def _lane_features(mt):
    """map_features of one map: every lane with its centre polyline (1 m), outline polygon and neighbours."""
    feats = {}
    n = len(mt.lane_objs)
    roads = mt.roads
    for k, lane in enumerate(mt.lane_objs):
        ss = np.append(np.arange(0.0, lane.length, 1.0), lane.length)
        poly = np.asarray([lane.position(float(s), 0.0) for s in ss], dtype=np.float32)
        rec = mt.lanes[k]
        road = roads[rec["road"]]
        idx, n_in = int(rec["idx"]), int(rec["n_in_road"])
        first = int(road["first_lane"])
        entry, exit_ = [], []
        for r2 in range(len(roads)):
            other = roads[r2]
            if other["end_node"] == road["start_node"]:
                entry += ["lane_%d" % (int(other["first_lane"]) + j) for j in range(int(other["n_lanes"]))]
            if other["start_node"] == road["end_node"]:
                exit_ += ["lane_%d" % (int(other["first_lane"]) + j) for j in range(int(other["n_lanes"]))]
        hull = mt.hull_xy[int(rec["hull_off"]):int(rec["hull_off"]) + int(rec["hull_n"])]
        feats["lane_%d" % k] = {
            "type": "LANE_SURFACE_STREET",
            "polyline": poly,
            "polygon": np.asarray(hull, dtype=np.float32).reshape(-1, 2),
            "left_neighbor": ["lane_%d" % (first + idx - 1)] if idx > 0 else [],
            "right_neighbor": ["lane_%d" % (first + idx + 1)] if idx + 1 < n_in else [],
            "entry_lanes": entry,
            "exit_lanes": exit_,
            "width": np.full((len(poly), 2), float(rec["width"]) / 2.0, dtype=np.float32),
        }
    assert len(feats) == n
    feats.update(_line_features(mt))
    return feats


def _line_features(mt, interval=2.0):
    """PGMap.get_boundary_line_vector (component/map/pg_map.py:130-167): the left line of every lane and the right line of the
    last lane of each road, typed by PGLineType / colour (SIDE lines are solid white), sampled every `interval` m at +-
    width / 2 -- the features ScenarioBlock turns into line bodies when the export is loaded as a scenario."""
    from metadrive_ped_amd.mapgen.lanes import COLOR_YELLOW, LINE_BROKEN, LINE_CONTINUOUS, LINE_NONE, LINE_SIDE
    feats = {}
    for k, lane in enumerate(mt.lane_objs):
        rec = mt.lanes[k]
        last = int(rec["idx"]) + 1 == int(rec["n_in_road"])
        decoration = tuple(lane.index[:2]) == ("decoration", "decoration_")
        for side in range(2 if (last or decoration) else 1):
            lt, color = lane.line_types[side], lane.line_colors[side]
            if lt == LINE_NONE:
                continue
            yellow = color == COLOR_YELLOW
            if lt == LINE_CONTINUOUS:
                typ = "ROAD_LINE_SOLID_SINGLE_YELLOW" if yellow else "ROAD_LINE_SOLID_SINGLE_WHITE"
            elif lt == LINE_BROKEN:
                typ = "ROAD_LINE_BROKEN_SINGLE_YELLOW" if yellow else "ROAD_LINE_BROKEN_SINGLE_WHITE"
            elif lt == LINE_SIDE:
                typ = "ROAD_LINE_SOLID_SINGLE_WHITE"
            else:
                typ = "UNKNOWN_LINE"
            lateral = float(lane.width) / 2.0 * (-1.0 if side == 0 else 1.0)
            ss = list(np.arange(0.0, lane.length, interval)) + [lane.length]
            poly = np.asarray([lane.position(float(x), lateral) for x in ss], dtype=np.float32)
            feats["lane_%d_%d" % (k, side)] = {"type": typ, "polyline": poly, "speed_limit_kmh": float(getattr(lane, "speed_limit", 1000.0))}
    return feats


TELEPORT = 10.0   # m between two consecutive frames: more than any mover covers in 0.1 s -> the slot was refilled


def _segments(valid, x, y):
    """[t0, t1) runs of one slot that belong to ONE object: split where the slot is not alive and where its pose
    jumps (a respawned traffic vehicle / a new agent took the slot).  The reference gives each a new object name."""
    T = len(valid)
    out, t0 = [], None
    for t in range(T):
        if valid[t] and t0 is not None and math.hypot(float(x[t]) - float(x[t - 1]), float(y[t]) - float(y[t - 1])) > TELEPORT:
            out.append((t0, t))
            t0 = t
        if valid[t] and t0 is None:
            t0 = t
        if not valid[t] and t0 is not None:
            out.append((t0, t))
            t0 = None
    if t0 is not None:
        out.append((t0, T))
    return out


def _track(oid, seg, k0, shape, dyn, j, A):
    T = len(seg)
    v = seg.astype(np.float32)
    heading = (dyn[:, 0] * v).astype(np.float32)
    speed = dyn[:, 1] * v
    pos = np.zeros((T, 3), np.float32)
    pos[:, 0] = shape["cx"] * v
    pos[:, 1] = shape["cy"] * v
    vel = np.stack([speed * np.cos(heading), speed * np.sin(heading)], 1).astype(np.float32)
    typ = _TYPE_OF_KIND.get(k0, "OTHER")
    md = {"type": typ, "object_id": oid, "track_length": int(T), "kind_code": k0, "slot": int(j)}
    if j < A:
        md["agent_slot"] = int(j)
    return {
        "type": typ,
        "state": {
            "position": pos, "heading": heading, "velocity": vel, "valid": seg.copy(),
            "length": (2.0 * shape["hl"] * v).astype(np.float32),
            "width": (2.0 * shape["hw"] * v).astype(np.float32),
            "height": (np.float32(_HEIGHT_OF_KIND.get(k0, 1.5)) * v).astype(np.float32),
        },
        "metadata": md,
    }


def object_summary(track, object_id):
    """ScenarioDescription.get_object_summary (scenario_description.py:341-375)."""
    st = track["state"]
    xy = st["position"][np.where(st["valid"].astype(int))][..., :2]
    dist = float(sum(np.linalg.norm(xy[i] - xy[i + 1]) for i in range(xy.shape[0] - 1)))
    run = 0
    for v in st["valid"]:
        if v:
            run += 1
        if run > 0 and not v:
            break
    return {"type": track["type"], "object_id": str(object_id), "track_length": int(len(st["position"])),
            "moving_distance": dist, "valid_length": int(sum(st["valid"])), "continuous_valid_length": int(run)}


def update_summaries(scenario):
    """ScenarioDescription.update_summaries (:417-437): metadata['object_summary'] / ['number_summary'] in place."""
    objs = {k: object_summary(t, k) for k, t in scenario["tracks"].items()}
    each, moving_each = {}, {}
    for t in scenario["tracks"].values():
        each[t["type"]] = each.get(t["type"], 0) + 1
    for o in objs.values():
        if o["moving_distance"] > 1:
            moving_each[o["type"]] = moving_each.get(o["type"], 0) + 1
    scenario["metadata"]["object_summary"] = objs
    scenario["metadata"]["number_summary"] = {
        "num_objects": len(objs), "object_types": set(each), "num_objects_each_type": each,
        "num_moving_objects": sum(moving_each.values()), "num_moving_objects_each_type": moving_each,
        "num_traffic_lights": 0, "num_traffic_light_types": set(), "num_traffic_light_each_step": {},
        "num_map_features": len(scenario["map_features"]),
    }
    return scenario


def tracks_to_scenarios(tracks, host, envs=None):
    """-> list of scenario dicts, one per env in `envs` (default: all).  `host` is the HostScene the recording engine
    was built from (maps, seeds, capacity)."""
    shape = _as_numpy(tracks["shape"])
    dyn = _as_numpy(tracks["dyn"]).astype(np.float32)
    T = shape.shape[0]
    E, cap, A = host.E, host.cap, host.A
    if shape.dtype != abi.SHAPE_DT:
        shape = np.ascontiguousarray(shape).view(abi.SHAPE_DT)
    shape = shape.reshape(T, E, cap)
    dyn = dyn.reshape(T, E, cap, 2)
    out = []
    feats_of_map = {}
    env_map = host.world.arrays["env_map"]
    for e in (range(E) if envs is None else envs):
        m = int(env_map[e])
        if m not in feats_of_map:
            feats_of_map[m] = _lane_features(host.map_tables[m])
        sc_tracks = {}
        for j in range(cap):
            fl = shape["flags"][:, e, j].astype(np.int64)
            kind = fl & abi.KIND_MASK
            valid = ((fl & abi.F_ALIVE) != 0) & (kind != 0)
            if not valid.any():
                continue
            x, y = shape["cx"][:, e, j], shape["cy"][:, e, j]
            for n_seg, (t0, t1) in enumerate(_segments(valid, x, y)):
                seg = np.zeros(T, bool)
                seg[t0:t1] = True
                oid = str(j) if n_seg == 0 else "%d_%d" % (j, n_seg)
                sc_tracks[oid] = _track(oid, seg, int(kind[t0]), shape[:, e, j], dyn[:, e, j], j, A)
        seed = int(host.seeds[e])
        out.append(update_summaries({
            "id": "pg-seed%d-env%d" % (seed, e),
            "version": VERSION,
            "length": int(T),
            "metadata": {"ts": (np.arange(T) * DT).astype(np.float32), "metadrive_processed": True,
                         "coordinate": "metadrive", "dataset": "pg", "seed": seed, "sdc_id": "0",
                         "scenario_id": "pg-seed%d" % seed},
            "tracks": sc_tracks,
            "dynamic_map_states": {},
            "map_features": feats_of_map[m],
        }))
    return out


_KIND_OF_TYPE = {"VEHICLE": abi.KIND_VEHICLE, "TRAFFIC_BARRIER": abi.KIND_BARRIER, "PEDESTRIAN": abi.KIND_PEDESTRIAN,
                 "CYCLIST": abi.KIND_CYCLIST}


def scenarios_to_tracks(scenarios, host):
    """The inverse of tracks_to_scenarios for scenario descriptions exported by it: -> tracks (BatchedEngine.set_tracks /
    env.load_tracks layout) for a replay env built with the same scenario assignment (`host`: its HostScene).  One
    scenario per env, in env order; every scenario must carry the seed of the env it is loaded into and as many frames
    as the others.  This is the ScenarioEnv-style path for data recorded here: record -> export_scenarios() -> store ->
    scenarios_to_tracks() -> traffic_mode='replay'.  (TRAFFIC_CONE covers both cones and warning tripods in the
    reference's type system; the exporter keeps the exact kind in the track metadata.)"""
    E, cap, A = host.E, host.cap, host.A
    if len(scenarios) != E:
        raise ValueError("need one scenario per env: got {} for {} envs".format(len(scenarios), E))
    T = int(scenarios[0]["length"])
    shape = np.zeros((T, E, cap), dtype=abi.SHAPE_DT)
    shape["aux"] = -1
    dyn = np.zeros((T, E, cap, 2), np.float32)
    for e, sc in enumerate(scenarios):
        if int(sc["length"]) != T:
            raise ValueError("scenario {} has {} frames, the first one {}".format(sc["id"], sc["length"], T))
        if int(sc["metadata"].get("seed", -1)) != int(host.seeds[e]):
            raise ValueError("scenario {} was recorded on scenario seed {}, env {} runs seed {}".format(
                sc["id"], sc["metadata"].get("seed"), e, host.seeds[e]))
        for oid, tr in sc["tracks"].items():
            md = tr["metadata"]
            j = int(md["slot"]) if "slot" in md else int(str(oid).split("_")[0])
            if not 0 <= j < cap:
                raise ValueError("track {} of scenario {} names slot {}, the env has {}".format(oid, sc["id"], j, cap))
            st = tr["state"]
            v = np.asarray(st["valid"], bool)
            kind = int(md["kind_code"]) if "kind_code" in md else _KIND_OF_TYPE.get(tr["type"], abi.KIND_CONE)
            h = np.asarray(st["heading"], np.float32)
            rec = shape[:, e, j]
            rec["cx"][v], rec["cy"][v] = st["position"][v, 0], st["position"][v, 1]
            rec["c"][v], rec["s"][v] = np.cos(h[v]), np.sin(h[v])
            rec["hl"][v], rec["hw"][v] = np.asarray(st["length"])[v] / 2.0, np.asarray(st["width"])[v] / 2.0
            flags = kind | abi.F_ALIVE | (abi.F_AGENT if j < A else 0)
            if kind not in (abi.KIND_VEHICLE, abi.KIND_PEDESTRIAN, abi.KIND_CYCLIST):
                flags |= abi.F_STATIC
            rec["flags"][v] = flags
            shape[:, e, j] = rec
            dyn[v, e, j, 0] = h[v]
            dyn[v, e, j, 1] = np.hypot(st["velocity"][v, 0], st["velocity"][v, 1])
    return dict(shape=shape.reshape(T, E * cap), dyn=dyn.reshape(T, E * cap, 2), seeds=list(host.seeds), cap=cap)

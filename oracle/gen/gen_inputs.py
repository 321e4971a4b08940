"""INPUTS of the golden-vector generator (TEST INFRASTRUCTURE, build container only).

The scenario sections of gen_golden.py need scenario descriptions to feed the reference's classes (tracks along a curving
road, late-appearing and vanishing objects, parked cars, pedestrians, cones, road lines).  They used to call the product's
own synthetic_scenario(): every edit of that function silently changed what a regeneration would produce, and the committed
fixture could no longer be reproduced from HEAD (round-2 verdict).  This module is the FROZEN input generator: it belongs to
the fixtures, not to the product -- never edit it without regenerating every fixture that names it (tools/check_golden.py
regenerates and compares).  The product's synthetic_scenario() is free to change.

Every fixture also stores the inputs it was computed from (points, poses, configs), so tests never call this module.
"""
import math

import numpy as np

INPUT_VERSION = 1


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


def frozen_scenario(seed, T=200, n_vehicles=18, n_parked=3, n_pedestrians=2, n_cones=4):
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
    return {"id": "synthetic-%d" % seed, "version": "golden-input v1 (MetaDrive v0.4.2.2 scenario format)",
            "length": int(T),
            "metadata": {"ts": t.astype(np.float32), "metadrive_processed": False, "coordinate": "metadrive",
                         "dataset": "synthetic", "seed": int(seed), "sdc_id": "0", "scenario_id": "synthetic-%d" % seed},
            "tracks": tracks, "dynamic_map_states": {}, "map_features": feats}

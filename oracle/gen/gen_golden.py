#!/usr/bin/env python
"""Generate tests/golden/*.json from the reference's own Python.  TEST INFRASTRUCTURE.

Run in the build container only (the reference tree does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen/gen_golden.py [section ...]

Every section imports reference modules through refshim (inert placeholders for the absent
third-party packages; see refshim.py) and records inputs + outputs of reference functions whose
arithmetic is plain numpy/math.  `assert_plain` guarantees no placeholder object leaks into a
recorded value.  The JSON fixtures are data (inputs and expected outputs); no reference source text
is stored.
"""
import json
import math
import os
import sys
from collections import OrderedDict
from unittest.mock import MagicMock

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
import numpy as np  # noqa: E402
import seaborn  # noqa: E402  (placeholder)

# BaseObject.__init__ picks a display colour from seaborn's palette (base_class/base_object.py:154-156);
# the value is cosmetic and drawn from an UNSEEDED RandomState, it never touches a seeded stream.
seaborn.color_palette = lambda *a, **k: [(i / 10., i / 10., i / 10.) for i in range(10)]

GOLDEN = os.path.join(ROOT, "tests", "golden")


def dump(name, obj):
    refshim.assert_plain(obj, name)

    def default(o):
        if isinstance(o, np.ndarray):
            return o.tolist()
        if isinstance(o, np.generic):
            return o.item()
        raise TypeError(type(o))

    path = os.path.join(GOLDEN, name)
    with open(path, "w") as f:
        json.dump(obj, f, default=default, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


# -------------------------------------------------------------------------------------------------
LINE_NAME = {}


def _line_names():
    from metadrive.constants import PGLineType
    return {PGLineType.BROKEN: "broken", PGLineType.CONTINUOUS: "continuous", PGLineType.SIDE: "side",
            PGLineType.NONE: "none", PGLineType.GUARDRAIL: "guardrail"}


def lane_record(lane):
    from metadrive.component.lane.circular_lane import CircularLane
    names = _line_names()
    rec = dict(kind=1 if isinstance(lane, CircularLane) else 0, length=float(lane.length), width=float(lane.width),
               start=[float(lane.start[0]), float(lane.start[1])], end=[float(lane.end[0]), float(lane.end[1])],
               line_types=[names[t] for t in lane.line_types],
               line_colors=["yellow" if c[1] < 1 else "grey" for c in lane.line_colors])
    if rec["kind"] == 1:
        rec.update(center=[float(lane.center[0]), float(lane.center[1])], radius=float(lane.radius),
                   start_phase=float(lane.start_phase), end_phase=float(lane.end_phase), angle=float(lane.angle),
                   clockwise=bool(lane.is_clockwise()))
    else:
        rec.update(heading=float(lane.heading))
    return rec


def cs_dist(p_curve, p_straight):
    from metadrive.component.algorithm.blocks_prob_dist import PGBlockDistConfig
    d = OrderedDict((k, 0.0) for k in PGBlockDistConfig.BLOCK_TYPE_DISTRIBUTION_V2)
    d["Curve"], d["Straight"] = p_curve, p_straight

    class CS(PGBlockDistConfig):
        BLOCK_TYPE_DISTRIBUTION_V2 = dict(d)

    return CS, d


def build_reference_map(seed, lane_num, lane_width, exit_length, method, parameter, dist_cls):
    from metadrive.component.algorithm.BIG import BIG
    from metadrive.component.road_network.node_road_network import NodeRoadNetwork
    net = NodeRoadNetwork()
    big = BIG(lane_num, lane_width, net, MagicMock(), MagicMock(), exit_length=exit_length, random_seed=seed,
              block_dist_config=dist_cls)
    big.generate(method, parameter)
    return big, net


def section_pg_maps():
    """PG topology per seed: BIG + FirstPGBlock/Straight/Curve."""
    dist_cls, d = cs_dist(0.6, 0.4)
    cases = []
    specs = [(seed, 3, 3.5, 50, "block_num", 3) for seed in range(0, 12)]
    specs += [(100 + seed, 2, 3.0, 50, "block_num", 4) for seed in range(0, 4)]
    specs += [(7, 3, 3.5, 50, "block_sequence", "S"), (8, 3, 3.5, 50, "block_sequence", "CCC"),
              (9, 3, 4.0, 30, "block_sequence", "SCS"), (1010, 3, 3.5, 50, "block_sequence", "S")]
    for seed, lane_num, lane_width, exit_length, method, parameter in specs:
        big, net = build_reference_map(seed, lane_num, lane_width, exit_length, method, parameter, dist_cls)
        roads = []
        for f, td in net.graph.items():
            for t, lanes in td.items():
                roads.append(dict(start=f, end=t, lanes=[lane_record(l) for l in lanes]))
        blocks = [dict(name=b.name, config={k: float(v) for k, v in dict(b.get_config()).items()},
                       trials=int(b.number_of_sample_trial),
                       sockets=[[s.positive_road.start_node, s.positive_road.end_node] for s in b.get_socket_list()])
                  for b in big.blocks]
        # spawn lanes per block as PGTrafficManager._create_vehicles_once sees them (traffic_manager.py:238-251)
        spawn = []
        for b in big.blocks[1:]:
            spawn.append([[list(l.index) for l in lanes] for lanes in b.get_intermediate_spawn_lanes()])
        cases.append(dict(seed=seed, lane_num=lane_num, lane_width=lane_width, exit_length=exit_length,
                          method=method, parameter=parameter, dist=list(d.items()), blocks=blocks, roads=roads,
                          spawn_lanes=spawn))
    dump("pg_maps.json", dict(cases=cases))


def section_lanes():
    """Lane Frenet transforms: position / local_coordinates / heading_theta_at / distance / polygon."""
    from metadrive.component.lane.straight_lane import StraightLane
    from metadrive.component.lane.circular_lane import CircularLane
    rng = np.random.RandomState(1234)
    out = []
    lanes = [StraightLane([0, 0], [50, 0], 3.5), StraightLane([3.2, -7.5], [-40.1, 22.3], 3.0)]
    for cw in (True, False):
        for radius, sp, ang in ((25.0, 0.3, 1.2), (60.0, -2.9, 2.3), (10.0, 3.0, 1.0471975512)):
            lanes.append(CircularLane((5.0, -3.0), radius, sp, ang, cw, 3.5))
    for lane in lanes:
        rec = lane_record(lane)
        pts = []
        for _ in range(40):
            s = float(rng.uniform(-8, lane.length + 8))
            lat = float(rng.uniform(-6, 6))
            p = lane.position(s, lat)
            ls, llat = lane.local_coordinates(p)
            pts.append(dict(s=s, lat=lat, p=[float(p[0]), float(p[1])], local=[float(ls), float(llat)],
                            heading=float(lane.heading_theta_at(ls)), dist=float(lane.distance(p))))
        rec["samples"] = pts
        rec["polygon"] = np.asarray(lane.polygon, dtype=float).tolist()
        out.append(rec)
    dump("lanes.json", dict(lanes=out))


def section_utils():
    """Known-answer vectors of the dependency-free helpers."""
    from metadrive.utils.math import safe_clip_for_small_array, wrap_to_pi, clip, not_zero, get_laser_end
    from metadrive.utils.random_utils import get_np_random
    from metadrive.component.vehicle.PID_controller import PIDController
    from metadrive.component.pg_space import ParameterSpace, VehicleParameterSpace, BlockParameterSpace
    out = {}
    arr = [float("nan"), float("inf"), float("-inf"), 1e27, -1e16, 0, 1, 0.3, -0.7]
    out["safe_clip"] = dict(inp=["nan", "inf", "-inf", 1e27, -1e16, 0, 1, 0.3, -0.7],
                            out=[float(x) for x in safe_clip_for_small_array(arr, -1, 1)])
    xs = [float(x) for x in np.linspace(-20, 20, 81)]
    out["wrap_to_pi"] = dict(x=xs, y=[float(wrap_to_pi(x)) for x in xs])
    out["not_zero"] = dict(x=[0.0, 0.005, -0.005, 0.5, -3.0], y=[float(not_zero(x)) for x in [0.0, 0.005, -0.005, 0.5, -3.0]])
    rs = {}
    for seed in (0, 1, 5, 1010, 65535, 123456789):
        r = get_np_random(seed)
        rs[str(seed)] = dict(randint_65536=int(r.randint(0, 65536)), rand=float(r.rand()), randint_1e6=int(r.randint(0, int(1e6))),
                             choice5=int(r.choice(5, p=[0.2, 0.3, 0.3, 0.2, 0.0])))
    out["rng"] = rs
    # parameter sampling rule incl. the BoxSpace(max,min) quirk (pg_space.py:14,226-272)
    ps = {}
    for name in ("DEFAULT_VEHICLE", "S_VEHICLE", "M_VEHICLE", "L_VEHICLE", "XL_VEHICLE", "STATIC_DEFAULT_VEHICLE"):
        space = ParameterSpace(getattr(VehicleParameterSpace, name))
        samples = []
        for seed in (0, 1, 77, 999999):
            space.seed(seed)
            samples.append(dict(seed=seed, values={k: float(v[0]) for k, v in space.sample().items()}))
        ps[name] = samples
    for name in ("STRAIGHT", "CURVE"):
        space = ParameterSpace(getattr(BlockParameterSpace, name))
        samples = []
        for seed in (0, 1, 77, 999999):
            space.seed(seed)
            samples.append(dict(seed=seed, values={k: float(v[0]) for k, v in space.sample().items()}))
        ps[name] = samples
    out["param_space"] = ps
    pid = PIDController(1.7, 0.01, 3.5)
    errs = [0.3, 0.1, -0.2, 0.05, 0.0, 0.7]
    out["pid"] = dict(k=[1.7, 0.01, 3.5], err=errs, out=[float(pid.get_result(e)) for e in errs])
    lr = [2 * np.pi * i / 240 for i in range(240)]
    out["laser_end"] = [dict(i=i, heading=h, end=[float(v) for v in get_laser_end(lr, 50.0, i, h, 3.0, -2.0)])
                        for i, h in ((0, 0.0), (1, 0.3), (60, -1.0), (239, 2.5))]
    dump("utils.json", out)


SECTIONS = OrderedDict(pg_maps=section_pg_maps, lanes=section_lanes, utils=section_utils)

if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    names = sys.argv[1:] or list(SECTIONS)
    for n in names:
        SECTIONS[n]()

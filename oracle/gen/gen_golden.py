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

# BIG backtracking destroys blocks; their destroy() walks Panda3D scene-graph nodes (placeholders here) and
# raises on anything that is not a real NodePath.  It is pure scene-graph cleanup: make it a no-op.
import metadrive.base_class.base_object as _bo  # noqa: E402

_bo.clear_node_list = lambda node_path_list: node_path_list.clear() if hasattr(node_path_list, "clear") else None

GOLDEN = os.environ.get("MD_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # tools/check_golden.py regenerates elsewhere


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



def section_agent_step():
    """Observation(9) + navi + reward + cost + done for posed agents.

    Inputs are POSES and FLAGS (not actions): everything downstream of pose is pure Python in the
    reference.  The fake vehicle exposes exactly the attributes the reference functions read; lanes,
    roads and the road network are the reference's own objects from a BIG-generated map.
    Not pinned here: navi dims 0,1,5,6 go through BaseVehicle.convert_to_local_coordinates, which is a
    Panda3D NodePath transform (placeholder in this container) -- the generator substitutes the
    (forward, left) projection SURVEY 8a-5 derives, and the fixture marks those dims as such.
    """
    import types
    from types import SimpleNamespace
    from metadrive.component.road_network import Road
    from metadrive.component.navigation_module.node_network_navigation import NodeNetworkNavigation
    from metadrive.component.vehicle.base_vehicle import BaseVehicle
    from metadrive.obs.state_obs import StateObservation
    from metadrive.envs.metadrive_env import MetaDriveEnv, METADRIVE_DEFAULT_CONFIG
    from metadrive.utils.math import Vector
    from metadrive.component.map.base_map import BaseMap
    dist_cls, d = cs_dist(0.6, 0.4)
    rng = np.random.RandomState(2024)
    cases = []

    class PosedNavigation(NodeNetworkNavigation):
        # `map` is a property reading the engine singleton (base_navigation.py:137-139); only the two
        # BaseMap class constants are read from it on this path (obs/state_obs.py:91-92,146)
        map = SimpleNamespace(MAX_LANE_NUM=BaseMap.MAX_LANE_NUM, MAX_LANE_WIDTH=BaseMap.MAX_LANE_WIDTH)

    for seed in (0, 3, 5, 8, 11):
        big, net = build_reference_map(seed, 3, 3.5, 50, "block_num", 3, dist_cls)
        net.after_init() if hasattr(net, "after_init") and not net.is_initialized else None
        for f, td in net.graph.items():
            for t, lanes in td.items():
                for i, l in enumerate(lanes):
                    l.index = (f, t, i)
        dest = big.blocks[-1].get_socket_list()[0].positive_road.end_node
        ckpts = net.shortest_path((">", ">>", 0), dest)
        final_lane = net.graph[ckpts[-2]][ckpts[-1]][-1]
        samples = []
        for _ in range(40):
            k = int(rng.randint(0, len(ckpts) - 1))
            cur_lanes = net.graph[ckpts[k]][ckpts[k + 1]]
            last_road = (k + 1 == len(ckpts) - 1)
            idx = [k, k] if last_road else [k, k + 1]
            next_lanes = None if last_road else net.graph[ckpts[k + 1]][ckpts[k + 2]]
            li = int(rng.randint(0, len(cur_lanes)))
            on_other_road = rng.rand() < 0.15
            if on_other_road:
                nr = Road(ckpts[k], ckpts[k + 1])
                neg = -nr
                lane = net.graph[neg.start_node][neg.end_node][li]
            else:
                lane = cur_lanes[li]
            s_ = float(rng.uniform(0.5, lane.length - 0.5))
            lat = float(rng.uniform(-2.6, 2.6))
            pos = lane.position(s_, lat)
            heading = float(lane.heading_theta_at(s_) + rng.uniform(-0.5, 0.5))
            speed = float(rng.uniform(0, 24))
            last_heading = heading - float(rng.uniform(-0.08, 0.08))
            last_pos = (float(pos[0] - math.cos(last_heading) * speed * 0.1), float(pos[1] - math.sin(last_heading) * speed * 0.1))
            act = [float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))]
            flags = dict(crash_vehicle=bool(rng.rand() < 0.15), crash_object=bool(rng.rand() < 0.1),
                         crash_building=bool(rng.rand() < 0.03), crash_human=bool(rng.rand() < 0.05),
                         crash_sidewalk=bool(rng.rand() < 0.1), on_lane=bool(rng.rand() < 0.9),
                         on_yellow_continuous_line=bool(rng.rand() < 0.1), on_white_continuous_line=bool(rng.rand() < 0.1))
            steps, horizon = int(rng.randint(0, 60)), int(rng.choice([0, 50]))

            nav = object.__new__(PosedNavigation)
            nav.checkpoints = ckpts
            nav._target_checkpoints_index = idx
            nav.current_ref_lanes = cur_lanes
            nav.next_ref_lanes = next_lanes
            nav.current_road = Road(ckpts[k], ckpts[k + 1])
            nav.final_lane = final_lane
            nav._current_lane = lane
            hx, hy = math.cos(heading), math.sin(heading)
            veh = SimpleNamespace(
                position=Vector((float(pos[0]), float(pos[1]))), last_position=last_pos, heading=Vector((hx, hy)),
                heading_theta=heading, last_heading_dir=Vector((math.cos(last_heading), math.sin(last_heading))),
                speed_km_h=speed * 3.6, max_speed_km_h=80.0, steering=act[0], MAX_STEERING=BaseVehicle.MAX_STEERING,
                last_current_action=[(0.0, 0.0), (act[0], act[1])], lane=lane, navigation=nav, engine=None,
                config={"side_detector": {"num_lasers": 0, "distance": 50}, "lane_line_detector": {"num_lasers": 0, "distance": 20}},
                out_of_route=False, **flags)
            veh.convert_to_local_coordinates = lambda vec, origin, hx=hx, hy=hy: np.array(
                [(vec[0] - (origin[0] if hasattr(origin, "__len__") else origin)) * hx +
                 (vec[1] - (origin[1] if hasattr(origin, "__len__") else origin)) * hy,
                 (vec[1] - (origin[1] if hasattr(origin, "__len__") else origin)) * hx -
                 (vec[0] - (origin[0] if hasattr(origin, "__len__") else origin)) * hy])
            veh.heading_diff = types.MethodType(BaseVehicle.heading_diff, veh)
            left, right = BaseVehicle._dist_to_route_left_right(veh)
            veh.dist_to_left_side, veh.dist_to_right_side = left, right
            veh.out_of_route = bool(right < 0 or left < 0)
            obs_self = SimpleNamespace(config={"random_agent_model": False}, engine=None)
            state9 = [float(x) for x in StateObservation.vehicle_state(obs_self, veh)]
            n1, _, ck1 = NodeNetworkNavigation._get_info_for_checkpoint(nav, 0, cur_lanes[0], veh)
            n2, _, ck2 = NodeNetworkNavigation._get_info_for_checkpoint(
                nav, 1, next_lanes[0] if next_lanes is not None else cur_lanes[0], veh)
            cfg = dict(METADRIVE_DEFAULT_CONFIG)
            cfg.update(horizon=(horizon or None), truncate_as_terminate=False)
            env = SimpleNamespace(agents={"a": veh}, config=cfg, episode_lengths={"a": steps + 1}, current_seed=seed,
                                  logger=MagicMock())
            env._is_out_of_road = types.MethodType(MetaDriveEnv._is_out_of_road, env)
            env._is_arrive_destination = MetaDriveEnv._is_arrive_destination
            reward, rinfo = MetaDriveEnv.reward_function(env, "a")
            done, dinfo = MetaDriveEnv.done_function(env, "a")
            cost, _ = MetaDriveEnv.cost_function(env, "a")
            samples.append(dict(
                road_k=k, lane=list(lane.index), idx=idx, pos=[float(pos[0]), float(pos[1])], heading=heading,
                last_pos=list(last_pos), last_heading=last_heading, speed=speed, action=act, flags=flags, steps=steps,
                horizon=horizon, state9=state9, navi=[float(x) for x in n1] + [float(x) for x in n2],
                checkpoints=[[float(ck1[0]), float(ck1[1])], [float(ck2[0]), float(ck2[1])]],
                left_right=[float(left), float(right)], out_of_route=veh.out_of_route, reward=float(reward),
                step_reward=float(rinfo["step_reward"]), cost=float(cost), done=bool(done),
                done_info={k_: bool(v) for k_, v in dinfo.items() if k_ != "env_seed"}))
        cases.append(dict(seed=seed, route=ckpts, final_lane=list(final_lane.index), samples=samples))
    dump("agent_step.json", dict(navi_dims_supplied_by_generator=[0, 1, 5, 6], cases=cases))


def section_objects():
    """SafeMetaDrive prop scenes: TrafficObjectManager.reset on BIG maps, with the
    manager's own seeded stream.  spawn_object is replaced by a recorder (it would create Bullet
    bodies); the placement arithmetic, the scene choice and the RNG consumption are the reference's."""
    from types import SimpleNamespace
    import metadrive.manager.object_manager as om
    from metadrive.manager.object_manager import TrafficObjectManager
    from metadrive.component.vehicle.vehicle_type import random_vehicle_type
    from metadrive.utils.random_utils import get_np_random
    dist_cls, d = cs_dist(0.6, 0.4)
    cases = []
    for seed in range(0, 16):
        big, net = build_reference_map(seed, 3, 3.5, 50, "block_num", 3, dist_cls)
        for f, td in net.graph.items():
            for t, lanes in td.items():
                for i, l in enumerate(lanes):
                    l.index = (f, t, i)
        spawned = []
        traffic_rng = get_np_random(seed)

        class Rec(TrafficObjectManager):
            def __init__(self):  # no engine / BaseManager machinery
                self.np_random = get_np_random(seed)
                self.accident_prob = 0.8
                self.accident_lanes = []

            def spawn_object(self, cls, **kw):
                rec = dict(cls=cls if isinstance(cls, str) else cls.__name__)
                if "position" in kw:
                    rec.update(position=[float(kw["position"][0]), float(kw["position"][1])],
                               heading=float(kw["heading_theta"]), lane=list(kw["lane"].index))
                else:
                    rec.update(lane=list(kw["vehicle_config"]["spawn_lane_index"]),
                               longitude=float(kw["vehicle_config"]["spawn_longitude"]))
                spawned.append(rec)
                return MagicMock()

        mgr = Rec()
        map_ = SimpleNamespace(blocks=big.blocks, road_network=net, config={"lane_width": 3.5}, LANE_WIDTH="lane_width")
        tm = SimpleNamespace(random_vehicle_type=lambda: {"SVehicle": "s", "MVehicle": "m", "LVehicle": "l", "XLVehicle": "xl",
                                                          "DefaultVehicle": "default"}[
            random_vehicle_type(traffic_rng, [0.2, 0.3, 0.3, 0.2, 0.0]).__name__])
        fake_engine = SimpleNamespace(current_map=map_, global_config={"static_traffic_object": True}, traffic_manager=tm)
        Rec.engine = property(lambda self: fake_engine)
        old = om.get_engine
        om.get_engine = lambda: fake_engine
        try:
            TrafficObjectManager.reset(mgr)
        finally:
            om.get_engine = old
        cases.append(dict(seed=seed, objects=spawned, accident_lanes=[list(l.index) for l in mgr.accident_lanes],
                          traffic_stream_next=float(traffic_rng.rand()), object_stream_next=float(mgr.np_random.rand())))
    dump("objects.json", dict(accident_prob=0.8, cases=cases))


def section_roundabout():
    """Map of MultiAgentRoundaboutEnv: FirstPGBlock(60 m, 2 lanes) + Roundabout(exit 10,
    inner 30, angle 70) as MARoundaboutMap._generate builds it (marl_inout_roundabout.py:27-60), the spawn
    roads, and the shortest checkpoint paths between every spawn road and every destination."""
    from metadrive.component.pgblock.first_block import FirstPGBlock
    from metadrive.component.pgblock.roundabout import Roundabout
    from metadrive.component.road_network.node_road_network import NodeRoadNetwork
    from metadrive.component.road_network import Road
    from metadrive.envs.marl_envs.marl_inout_roundabout import MARoundaboutConfig
    from metadrive.manager.spawn_manager import SpawnManager
    net = NodeRoadNetwork()
    first = FirstPGBlock(net, 3.5, 2, MagicMock(), MagicMock(), length=60)
    Roundabout.EXIT_PART_LENGTH = 60
    rb = Roundabout(1, first.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    ok = rb.construct_block(MagicMock(), MagicMock(), extra_config={"exit_radius": 10, "inner_radius": 30, "angle": 70})
    roads = []
    for f, td in net.graph.items():
        for t, lanes in td.items():
            roads.append(dict(start=f, end=t, lanes=[lane_record(l) for l in lanes]))
    spawn_roads = [[r.start_node, r.end_node] for r in MARoundaboutConfig["spawn_roads"]]
    routes = []
    for sr in MARoundaboutConfig["spawn_roads"]:
        for er in MARoundaboutConfig["spawn_roads"]:
            dest = (-er).end_node
            path = net.shortest_path((sr.start_node, sr.end_node, 0), dest)
            routes.append(dict(start=[sr.start_node, sr.end_node], dest=dest, path=path))
    dump("roundabout.json", dict(no_cross=bool(ok), roads=roads, spawn_roads=spawn_roads, routes=routes,
                                 max_capacity=int(SpawnManager.max_capacity(MARoundaboutConfig["spawn_roads"], 60, 2)),
                                 config={k: float(v) for k, v in dict(rb.get_config()).items()}))


def section_ma_intersection():
    """Map of MultiAgentIntersectionEnv: FirstPGBlock(60 m, 2 lanes) + InterSection with
    U-turns enabled as MAIntersectionMap._generate builds it (marl_intersection.py:27-70), spawn roads, routes."""
    from metadrive.component.pgblock.first_block import FirstPGBlock
    from metadrive.component.pgblock.intersection import InterSection
    from metadrive.component.road_network.node_road_network import NodeRoadNetwork
    from metadrive.envs.marl_envs.marl_intersection import MAIntersectionConfig
    from metadrive.manager.spawn_manager import SpawnManager
    net = NodeRoadNetwork()
    first = FirstPGBlock(net, 3.5, 2, MagicMock(), MagicMock(), length=60)
    InterSection.EXIT_PART_LENGTH = 60
    blk = InterSection(1, first.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    blk.enable_u_turn(True)
    ok = blk.construct_block(MagicMock(), MagicMock())
    InterSection.EXIT_PART_LENGTH = 35
    roads = []
    for f, td in net.graph.items():
        for t, lanes in td.items():
            roads.append(dict(start=f, end=t, lanes=[lane_record(l) for l in lanes]))
    spawn_roads = [[r.start_node, r.end_node] for r in MAIntersectionConfig["spawn_roads"]]
    routes = []
    for sr in MAIntersectionConfig["spawn_roads"]:
        for er in MAIntersectionConfig["spawn_roads"]:
            dest = (-er).end_node
            path = net.shortest_path((sr.start_node, sr.end_node, 0), dest)
            routes.append(dict(start=[sr.start_node, sr.end_node], dest=dest, path=path))
    dump("ma_intersection.json", dict(no_cross=bool(ok), roads=roads, spawn_roads=spawn_roads, routes=routes,
                                      max_capacity=int(SpawnManager.max_capacity(MAIntersectionConfig["spawn_roads"], 60, 2)),
                                      num_agents=int(MAIntersectionConfig["num_agents"]),
                                      config={k: float(v) for k, v in dict(blk.get_config()).items()}))


def section_ma_tinyinter():
    """Map of MultiAgentTinyInter (envs/marl_envs/tinyinter.py:328-420): MAIntersectionMap._generate (marl_intersection.py:27-70) with
    the tiny env's map_config -- ONE lane of 4 m per direction, arms of 30 m, U-turns off -- for the sampled radius and for
    map_config["radius"] = 50; the destinations MAIntersectionSpawnManager(disable_u_turn=True).update_destination_for can draw for
    every spawn road (:78-85: the spawn roads but the vehicle's own), the routes to them, and the env's default config values."""
    from metadrive.component.pgblock.first_block import FirstPGBlock
    from metadrive.component.pgblock.intersection import InterSection
    from metadrive.component.road_network import Road
    from metadrive.component.road_network.node_road_network import NodeRoadNetwork
    from metadrive.envs.marl_envs.marl_intersection import MAIntersectionConfig
    from metadrive.manager.spawn_manager import SpawnManager
    try:
        from metadrive.envs.marl_envs.tinyinter import MultiAgentTinyInter
        dc = MultiAgentTinyInter.default_config()
        defaults = dict(num_agents=int(dc["num_agents"]), num_RL_agents=int(dc["num_RL_agents"]), success_reward=float(dc["success_reward"]),
                        out_of_road_penalty=float(dc["out_of_road_penalty"]), crash_vehicle_penalty=float(dc["crash_vehicle_penalty"]),
                        crash_object_penalty=float(dc["crash_object_penalty"]), ignore_delay_done=bool(dc["ignore_delay_done"]),
                        exit_length=float(dc["map_config"]["exit_length"]), lane_num=int(dc["map_config"]["lane_num"]),
                        lane_width=float(dc["map_config"]["lane_width"]), delay_done=int(dc["delay_done"]),
                        use_communication_obs=bool(dc["use_communication_obs"]), target_speed=float(dc["target_speed"]))
    except Exception as ex:       # noqa: BLE001
        defaults = dict(unavailable="%s: %s" % (type(ex).__name__, ex))
    maps = []
    for radius in (None, 50.0):
        net = NodeRoadNetwork()
        first = FirstPGBlock(net, 4.0, 1, MagicMock(), MagicMock(), length=30)
        InterSection.EXIT_PART_LENGTH = 30
        kw = dict(radius=radius) if radius else {}
        blk = InterSection(1, first.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False, **kw)
        blk.enable_u_turn(False)
        ok = blk.construct_block(MagicMock(), MagicMock())
        InterSection.EXIT_PART_LENGTH = 35
        roads = []
        for f, td in net.graph.items():
            for t, lanes in td.items():
                roads.append(dict(start=f, end=t, lanes=[lane_record(l) for l in lanes]))
        routes = []
        for sr in MAIntersectionConfig["spawn_roads"]:
            end_roads = [r for r in MAIntersectionConfig["spawn_roads"] if Road(sr.start_node, sr.end_node) != r]   # disable_u_turn
            for er in end_roads:
                dest = (-er).end_node
                routes.append(dict(start=[sr.start_node, sr.end_node], dest=dest,
                                   path=net.shortest_path((sr.start_node, sr.end_node, 0), dest)))
        maps.append(dict(radius=radius, used_radius=float(blk.radius), no_cross=bool(ok), roads=roads, routes=routes,
                         config={k: float(v) for k, v in dict(blk.get_config()).items()}))
    dump("ma_tinyinter.json", dict(maps=maps, spawn_roads=[[r.start_node, r.end_node] for r in MAIntersectionConfig["spawn_roads"]],
                                   max_capacity=int(SpawnManager.max_capacity(MAIntersectionConfig["spawn_roads"], 30, 1)),
                                   defaults=defaults))


def section_ma_racing():
    """MultiAgentRacingEnv (envs/marl_envs/marl_racing_env.py): the hand-built RacingMap as RacingMap._generate builds it (:76-320;
    run on a bare object that carries what the method reads: config, road_network, blocks, engine placeholders), every lane with
    its line types, the sidewalk strips the guardrails generate (PGBlock._generate_sidewalk_from_line, one per lane), the route of
    an agent from the spawn road to the end, and RACING_CONFIG's numbers."""
    from types import SimpleNamespace
    from metadrive.component.road_network.node_road_network import NodeRoadNetwork
    from metadrive.envs.marl_envs.marl_racing_env import RACING_CONFIG, RacingMap
    from metadrive.manager.spawn_manager import SpawnManager
    from metadrive.envs.marl_envs.multi_agent_metadrive import MULTI_AGENT_METADRIVE_DEFAULT_CONFIG as MA
    net = NodeRoadNetwork()
    fake = SimpleNamespace(config={"lane_num": 2, "lane_width": 3.5}, road_network=net, blocks=[],
                           engine=SimpleNamespace(worldNP=MagicMock(), physics_world=MagicMock()))
    RacingMap._generate(fake)
    for f, td in net.graph.items():
        for t, lanes in td.items():
            for i, l in enumerate(lanes):
                l.index = (f, t, i)
    roads = []
    for f, td in net.graph.items():
        for t, lanes in td.items():
            roads.append(dict(start=f, end=t, lanes=[lane_record(l) for l in lanes]))
    # the sidewalk strips: rebuild them through the reference's own method on each block (it only needs `sidewalks`)
    strips = []
    for b in fake.blocks:
        b.sidewalks = {}
        for f, td in b.block_network.graph.items():
            for t, lanes in td.items():
                for l in lanes:
                    for idx, lt in zip([-1, 1], l.line_types):
                        if lt == "guardrail" or str(lt).lower().endswith("guardrail"):
                            b._generate_sidewalk_from_line(l, sidewalk_height=4.0, lateral_direction=idx)
        for key, sw in b.sidewalks.items():
            strips.append(dict(lane=key, polygon=[[float(p[0]), float(p[1])] for p in sw["polygon"]]))
    spawn = [[r.start_node, r.end_node] for r in MA["spawn_roads"]]
    last = fake.blocks[-1].get_socket_list()[0].positive_road.end_node
    path = net.shortest_path((spawn[0][0], spawn[0][1], 0), last)
    cfg = {k: (float(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else v) for k, v in RACING_CONFIG.items()
           if k in ("num_agents", "out_of_road_penalty", "idle_penalty", "success_reward", "crash_sidewalk_penalty", "cross_yellow_line_done",
                    "out_of_road_done", "on_continuous_line_done", "out_of_route_done", "crash_done", "horizon", "idle_done",
                    "crash_sidewalk_done", "crash_vehicle_done", "allow_respawn", "traffic_density")}
    cfg["map_config"] = {k: float(v) for k, v in RACING_CONFIG["map_config"].items()}
    cfg["lidar"] = {k: float(v) for k, v in RACING_CONFIG["vehicle_config"]["lidar"].items() if not isinstance(v, bool)}
    cfg["side_detector"] = {k: float(v) for k, v in RACING_CONFIG["vehicle_config"]["side_detector"].items()}
    dump("ma_racing.json", dict(roads=roads, blocks=[b.ID for b in fake.blocks], sidewalks=strips, spawn_roads=spawn, route=path,
                                capacity_exit_20=int(SpawnManager.max_capacity(MA["spawn_roads"], 20, 2)),
                                capacity_exit_60=int(SpawnManager.max_capacity(MA["spawn_roads"], 60, 2)), config=cfg))


def section_ma_racing_rules():
    """MultiAgentRacingEnv's own rules, called step by step: reward_function, done_function (through
    MultiAgentMetaDrive's and MetaDriveEnv's), _is_out_of_road and _is_idle with its 100-step movement deque
    (envs/marl_envs/marl_racing_env.py:354-441) on a vehicle walked along the reference's own RacingMap lanes through five scripted
    episodes: standing still until idle, scraping the guardrail / another vehicle, falling behind the start of its lane, arriving,
    and running into the horizon while idle.  The env is a bare instance (no engine); what the methods read is laid on it."""
    import types
    from collections import defaultdict, deque
    from types import SimpleNamespace
    from metadrive.component.map.base_map import BaseMap
    from metadrive.component.navigation_module.node_network_navigation import NodeNetworkNavigation
    from metadrive.component.road_network import Road
    from metadrive.component.road_network.node_road_network import NodeRoadNetwork
    from metadrive.envs.marl_envs.marl_racing_env import MultiAgentRacingEnv, RacingMap
    from metadrive.envs.metadrive_env import MetaDriveEnv
    from metadrive.utils.math import Vector
    net = NodeRoadNetwork()
    fake_map = SimpleNamespace(config={"lane_num": 2, "lane_width": 3.5}, road_network=net, blocks=[],
                               engine=SimpleNamespace(worldNP=MagicMock(), physics_world=MagicMock()))
    RacingMap._generate(fake_map)
    for f, td in net.graph.items():
        for t, lanes in td.items():
            for i, l in enumerate(lanes):
                l.index = (f, t, i)
    last = fake_map.blocks[-1].get_socket_list()[0].positive_road.end_node
    ckpts = net.shortest_path((">>", ">>>", 0), last)
    final_lane = net.graph[ckpts[-2]][ckpts[-1]][-1]
    cfg = dict(MultiAgentRacingEnv.default_config())

    class PosedNavigation(NodeNetworkNavigation):
        map = SimpleNamespace(MAX_LANE_NUM=BaseMap.MAX_LANE_NUM, MAX_LANE_WIDTH=BaseMap.MAX_LANE_WIDTH)

    class BareRacing(MultiAgentRacingEnv):       # class attributes stand where the real env has engine-backed properties
        agents = None
        current_seed = 0
        config = None

    def run(script, horizon):
        env = object.__new__(BareRacing)
        env.config = dict(cfg, horizon=horizon)
        env.logger = MagicMock()
        env.movement_between_steps = defaultdict(lambda: deque(maxlen=100))
        env.episode_lengths = {"agent0": 0}
        out = []
        last_pos = None
        for (k, li, s_, lat, speed, crash_vehicle, crash_sidewalk) in script:
            lanes = net.graph[ckpts[k]][ckpts[k + 1]]
            lane = lanes[li]
            pos = lane.position(s_, lat)
            pos = (float(pos[0]), float(pos[1]))
            if last_pos is None:
                last_pos = pos
            heading = float(lane.heading_theta_at(max(0.0, min(s_, lane.length))))
            last_road = (k + 1 == len(ckpts) - 1)
            nav = object.__new__(PosedNavigation)
            nav.checkpoints = ckpts
            nav._target_checkpoints_index = [k, k] if last_road else [k, k + 1]
            nav.current_ref_lanes = lanes
            nav.next_ref_lanes = None if last_road else net.graph[ckpts[k + 1]][ckpts[k + 2]]
            nav.current_road = Road(ckpts[k], ckpts[k + 1])
            nav.final_lane = final_lane
            nav._current_lane = lane
            veh = SimpleNamespace(position=Vector(pos), last_position=last_pos, lane=lane, navigation=nav, speed_km_h=speed * 3.6,
                                  max_speed_km_h=80.0, crash_vehicle=bool(crash_vehicle), crash_sidewalk=bool(crash_sidewalk),
                                  crash_object=False, crash_building=False, crash_human=False, on_lane=True,
                                  on_yellow_continuous_line=False, on_white_continuous_line=False, out_of_route=False)
            env.agents = {"agent0": veh}
            env.episode_lengths["agent0"] += 1
            reward, rinfo = MultiAgentRacingEnv.reward_function(env, "agent0")
            done, dinfo = MultiAgentRacingEnv.done_function(env, "agent0")
            out.append(dict(k=k, lane=list(lane.index), idx=list(nav._target_checkpoints_index), pos=list(pos), last_pos=list(last_pos),
                            heading=heading, speed=speed, crash_vehicle=bool(crash_vehicle), crash_sidewalk=bool(crash_sidewalk),
                            steps=env.episode_lengths["agent0"] - 1, reward=float(reward), step_reward=float(rinfo["step_reward"]),
                            progress=float(rinfo["progress"]), done=bool(done),
                            done_info={k_: bool(v) for k_, v in dinfo.items() if k_ != "env_seed"}))
            last_pos = pos
        return out

    episodes = []
    # A: 30 moving steps (1.5 m each), then 104 standing ones: idle at the 100th still step, done
    sc = [(1, 0, 2.0 + 1.5 * i, 0.3, 15.0, 0, 0) for i in range(30)] + [(1, 0, 2.0 + 1.5 * 29, 0.3, 0.0, 0, 0)] * 104
    episodes.append(dict(name="idle", horizon=3000, steps=run(sc, 3000)))
    # B: the guardrail and another vehicle: penalties, nothing terminal; vehicle crash has priority over the sidewalk
    sc = []
    for i in range(40):
        sc.append((2, i % 2, 5.0 + 2.0 * i, -1.2 if i % 2 == 0 else 1.2, 20.0, 1 if i in (7, 8, 20) else 0, 1 if i in (12, 13, 20, 30) else 0))
    episodes.append(dict(name="scrapes", horizon=3000, steps=run(sc, 3000)))
    # C: sliding back behind the start of one's lane: out of road only beyond 5 m
    sc = [(3, 1, 3.0 - 1.0 * i, 0.0, 1.0, 0, 0) for i in range(12)]
    episodes.append(dict(name="behind_lane_start", horizon=3000, steps=run(sc, 3000)))
    # D: the last road, up to the destination
    kl = len(ckpts) - 2
    L = float(final_lane.length)
    sc = [(kl, 1, L - 30.0 + 2.0 * i, 0.2, 20.0, 0, 0) for i in range(16)]
    episodes.append(dict(name="arrive", horizon=3000, steps=run(sc, 3000)))
    # E: standing still into the horizon: max_step, and the idle verdict is not even reported then (marl_racing_env.py:363-364)
    sc = [(1, 1, 10.0, 0.0, 0.0, 0, 0)] * 103
    episodes.append(dict(name="idle_at_horizon", horizon=100, steps=run(sc, 100)))
    dump("ma_racing_rules.json", dict(route=ckpts, final_lane=list(final_lane.index), episodes=episodes,
                                      config={k_: cfg[k_] for k_ in ("driving_reward", "speed_reward", "success_reward", "out_of_road_penalty",
                                                                    "crash_vehicle_penalty", "crash_sidewalk_penalty", "idle_penalty",
                                                                    "idle_done", "crash_sidewalk_done", "crash_done", "out_of_road_done",
                                                                    "crash_vehicle_done", "truncate_as_terminate")}))


def section_ma_bottleneck():
    """Map of MultiAgentBottleneckEnv: FirstPGBlock(60 m, 4 lanes) + Merge (to 1 lane over 20 m)
    + Split (back to 4, exit 60 m) as MABottleneckMap._generate builds it (marl_bottleneck.py:28-69)."""
    from metadrive.component.pgblock.first_block import FirstPGBlock
    from metadrive.component.pgblock.bottleneck import Merge, Split
    from metadrive.component.road_network.node_road_network import NodeRoadNetwork
    from metadrive.envs.marl_envs.marl_bottleneck import MABottleneckConfig
    from metadrive.manager.spawn_manager import SpawnManager
    net = NodeRoadNetwork()
    first = FirstPGBlock(net, 3.5, 4, MagicMock(), MagicMock(), length=60)
    merge = Merge(1, first.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    ok1 = merge.construct_from_config(dict(lane_num=3, length=20), MagicMock(), MagicMock())
    split = Split(2, merge.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    ok2 = split.construct_from_config({"length": 60, "lane_num": 3}, MagicMock(), MagicMock())
    roads = []
    for f, td in net.graph.items():
        for t, lanes in td.items():
            roads.append(dict(start=f, end=t, lanes=[lane_record(l) for l in lanes]))
    spawn_roads = [[r.start_node, r.end_node] for r in MABottleneckConfig["spawn_roads"]]
    routes = []
    for sr in MABottleneckConfig["spawn_roads"]:
        for er in MABottleneckConfig["spawn_roads"]:
            dest = (-er).end_node
            path = net.shortest_path((sr.start_node, sr.end_node, 0), dest)
            routes.append(dict(start=[sr.start_node, sr.end_node], dest=dest, path=path))
    dump("ma_bottleneck.json", dict(no_cross=bool(ok1 and ok2), roads=roads, spawn_roads=spawn_roads, routes=routes,
                                    max_capacity=int(SpawnManager.max_capacity(MABottleneckConfig["spawn_roads"], 60, 4)),
                                    num_agents=int(MABottleneckConfig["num_agents"]),
                                    merge_config={k: float(v) for k, v in dict(merge.get_config()).items()},
                                    split_config={k: float(v) for k, v in dict(split.get_config()).items()}))


def section_ma_tollgate():
    """MultiAgentTollgateEnv (envs/marl_envs/marl_tollgate.py): the map MATollGateMap._generate builds (FirstPGBlock 70 m x 3
    lanes + Split to 8 lanes + TollGate 10 m + Merge back to 3), its booths, speed limits and block ids; and its own
    reward_function / done_function / TollGateObservation toll dims / StayTimeManager bookkeeping, called on a posed vehicle
    that is walked through the toll block at different speeds (the reference's methods on a bare instance, no engine)."""
    import types
    from types import SimpleNamespace
    from metadrive.component.pgblock.first_block import FirstPGBlock
    from metadrive.component.pgblock.bottleneck import Merge, Split
    from metadrive.component.pgblock.tollgate import TollGate
    import metadrive.component.pgblock.tollgate as tg
    from metadrive.component.buildings.tollgate_building import TollGateBuilding
    from metadrive.component.road_network.node_road_network import NodeRoadNetwork
    from metadrive.component.road_network import Road
    from metadrive.component.vehicle.base_vehicle import BaseVehicle
    from metadrive.envs.marl_envs import marl_tollgate as mt
    from metadrive.envs.marl_envs.multi_agent_metadrive import MultiAgentMetaDrive
    from metadrive.manager.spawn_manager import SpawnManager
    from metadrive.utils.math import Vector
    spawned = []

    def spawn_object(cls, lane=None, position=None, heading_theta=None, **kw):
        spawned.append((cls.__name__, lane, [float(position[0]), float(position[1])], float(heading_theta)))
        return MagicMock()
    tg.get_engine = lambda: MagicMock(spawn_object=spawn_object)
    mc = mt.MATollConfig["map_config"]
    net = NodeRoadNetwork()
    first = FirstPGBlock(net, 3.5, mc["lane_num"], MagicMock(), MagicMock(), length=mc["exit_length"])
    split = Split(1, first.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    ok1 = split.construct_block(MagicMock(), MagicMock(), {"length": 2, "lane_num": mc["toll_lane_num"] - mc["lane_num"],
                                                           "bottle_len": mt.MATollGateMap.BOTTLE_LENGTH})
    toll = TollGate(2, split.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    ok2 = toll.construct_block(MagicMock(), MagicMock(), {"length": mc["toll_length"]})
    merge = Merge(3, toll.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    ok3 = merge.construct_from_config(dict(lane_num=mc["toll_lane_num"] - mc["lane_num"], length=mc["exit_length"],
                                           bottle_len=mt.MATollGateMap.BOTTLE_LENGTH), MagicMock(), MagicMock())
    for f, td in net.graph.items():
        for t, lanes in td.items():
            for i, l in enumerate(lanes):
                l.index = (f, t, i)
    roads = []
    for f, td in net.graph.items():
        for t, lanes in td.items():
            roads.append(dict(start=f, end=t, block_id=Road(f, t).block_ID(), lanes=[lane_record(l) for l in lanes],
                              speed_limit=[float(l.speed_limit) for l in lanes]))
    spawn_roads = [[r.start_node, r.end_node] for r in mt.MATollConfig["spawn_roads"]]
    routes = []
    for sr in mt.MATollConfig["spawn_roads"]:
        for er in mt.MATollConfig["spawn_roads"]:
            dest = (-er).end_node
            routes.append(dict(start=[sr.start_node, sr.end_node], dest=dest,
                               path=net.shortest_path((sr.start_node, sr.end_node, 0), dest)))
    buildings = [dict(cls=c, lane=list(l.index), position=p_, heading=h, width=float(l.width),
                      length=float(TollGateBuilding.BUILDING_LENGTH)) for c, l, p_, h in spawned]

    # ---- the env's own step logic on a posed vehicle walked along the positive route
    class BareEnv(mt.MultiAgentTollgateEnv):     # properties of BaseEnv that read the engine singleton
        current_seed = 0
        agents = property(lambda self: self._agents)
        engine = None

    cfg = dict(MultiAgentMetaDrive.default_config().get_dict())
    for k, v in mt.MATollConfig.items():
        if isinstance(v, dict) and isinstance(cfg.get(k), dict):
            cfg[k] = dict(cfg[k], **v)
        else:
            cfg[k] = v
    ckpts = net.shortest_path((">>", ">>>", 0), (-mt.MATollConfig["spawn_roads"][1]).end_node)
    final_lane = net.graph[ckpts[-2]][ckpts[-1]][-1]
    rng = np.random.RandomState(77)
    episodes = []
    # steps spent inside the 10 m toll block: around the 30-step minimum, one far below, one crawling under the 3 km/h limit
    for ep, n_toll in enumerate((12, 28, 33, 34, 35, 36, 37, 125)):
        env = object.__new__(BareEnv)
        env.config = dict(cfg, horizon=1000, crash_done=bool(ep % 2), out_of_road_done=bool(ep % 3 != 1))
        env.stay_time_manager = mt.StayTimeManager()
        env.logger = MagicMock()
        obs = object.__new__(mt.TollGateObservation)
        obs.in_toll_time = 0
        obs.state_observe = lambda v: np.zeros(0)
        obs.lidar_observe = lambda v: []
        toll_lane = int(rng.choice([0, 2, 4, 6]))            # even lanes are open, odd ones hold a booth
        d_toll = 10.0 / n_toll + 1e-3
        k, s_ = 0, 40.0
        steps = []
        tstep = 0
        lat_off = float(rng.uniform(-0.6, 0.6))
        while tstep < 260 and k < len(ckpts) - 1:
            cur_lanes = net.graph[ckpts[k]][ckpts[k + 1]]
            road = Road(ckpts[k], ckpts[k + 1])
            in_toll = road.block_ID() == TollGate.ID
            li = min(toll_lane if len(cur_lanes) > 3 else 1, len(cur_lanes) - 1)
            lane = cur_lanes[li]
            if s_ > lane.length:
                s_ -= lane.length
                k += 1
                continue
            adv = d_toll if in_toll else float(rng.uniform(3.0, 5.0))
            speed = adv / 0.1
            pos = lane.position(s_, lat_off)
            heading = float(lane.heading_theta_at(s_))
            last_pos = lane.position(s_ - adv, lat_off)
            last_road = (k + 1 == len(ckpts) - 1)
            nav = SimpleNamespace(current_ref_lanes=cur_lanes, current_road=road, final_lane=final_lane, checkpoints=ckpts,
                                  get_current_lane_width=lambda lane=lane: lane.width,
                                  get_current_lane_num=lambda cur_lanes=cur_lanes: len(cur_lanes),
                                  _target_checkpoints_index=[k, k] if last_road else [k, k + 1])
            flags = dict(crash_vehicle=bool(rng.rand() < 0.04), crash_object=bool(rng.rand() < 0.02), crash_building=False,
                         crash_human=False, crash_sidewalk=bool(rng.rand() < 0.02), on_lane=bool(rng.rand() < 0.97),
                         on_yellow_continuous_line=bool(rng.rand() < 0.03), on_white_continuous_line=bool(rng.rand() < 0.1),
                         on_broken_line=False)
            veh = SimpleNamespace(position=Vector((float(pos[0]), float(pos[1]))), last_position=(float(last_pos[0]), float(last_pos[1])),
                                  heading_theta=heading, speed_km_h=speed * 3.6, max_speed_km_h=80.0, lane=lane, navigation=nav,
                                  config=dict(cfg["vehicle_config"]), out_of_route=False, **flags)
            veh.overspeed = bool(lane.speed_limit < veh.speed_km_h)
            env._agents = {"a": veh}
            env.episode_lengths = {"a": tstep + 1}
            done, dinfo = mt.MultiAgentTollgateEnv.done_function(env, "a")
            reward, rinfo = mt.MultiAgentTollgateEnv.reward_function(env, "a")
            o = mt.TollGateObservation.observe(obs, veh)
            env.stay_time_manager.record({"a": veh}, tstep)
            stm = env.stay_time_manager
            steps.append(dict(lane=list(lane.index), idx=nav._target_checkpoints_index, pos=[float(pos[0]), float(pos[1])],
                              heading=heading, last_pos=[float(last_pos[0]), float(last_pos[1])], speed=speed,
                              flags=[k_ for k_, v in flags.items() if v], block_id=road.block_ID(), overspeed=veh.overspeed,
                              done=bool(done), done_info=[k_ for k_, v in dinfo.items() if v is True], reward=float(reward),
                              step_reward=float(rinfo["step_reward"]), toll_obs=[float(x) for x in o[-2:]],
                              in_toll_time=int(obs.in_toll_time), entry=stm.entry_time.get("a"), exit=stm.exit_time.get("a")))
            s_ += adv
            tstep += 1
        episodes.append(dict(crash_done=env.config["crash_done"], out_of_road_done=env.config["out_of_road_done"],
                             toll_lane=toll_lane, steps_in_toll=n_toll, steps=steps))
    keys = ("num_agents", "cross_yellow_line_done", "speed_reward", "overspeed_penalty", "driving_reward", "success_reward",
            "out_of_road_penalty", "crash_vehicle_penalty", "crash_object_penalty", "use_lateral_reward", "delay_done", "horizon")
    dump("ma_tollgate.json", dict(no_cross=bool(ok1 and ok2 and ok3), roads=roads, spawn_roads=spawn_roads, routes=routes,
                                  buildings=buildings, route=ckpts, final_lane=list(final_lane.index),
                                  max_capacity=int(SpawnManager.max_capacity(mt.MATollConfig["spawn_roads"], mc["exit_length"],
                                                                             mc["lane_num"])),
                                  config={k_: cfg[k_] for k_ in keys}, min_pass_steps=int(cfg["vehicle_config"]["min_pass_steps"]),
                                  vehicle_config={k_: dict(cfg["vehicle_config"][k_]) for k_ in ("lidar", "side_detector", "lane_line_detector")},
                                  map_config=dict(mc), episodes=episodes))


def section_ma_parking_lot():
    """Map of MultiAgentParkingLotEnv (envs/marl_envs/marl_parking_lot.py:140-180): FirstPGBlock (20 m, one lane) + ParkingLot
    (4 spaces a side) + TInterSection (t_type 1, exits 10 m); its parking spaces (destination roads), the spawn roads (three
    entrances + the eight spaces, out direction) and the shortest path from every spawn road to every destination."""
    from metadrive.component.pgblock.first_block import FirstPGBlock
    from metadrive.component.pgblock.parking_lot import ParkingLot
    from metadrive.component.pgblock.t_intersection import TInterSection
    from metadrive.component.road_network.node_road_network import NodeRoadNetwork
    from metadrive.envs.marl_envs import marl_parking_lot as mp
    from metadrive.manager.spawn_manager import SpawnManager
    cfg = mp.MAParkingLotConfig
    mc = cfg["map_config"]
    n_space = cfg["parking_space_num"]
    net = NodeRoadNetwork()
    first = FirstPGBlock(net, 3.5, mc["lane_num"], MagicMock(), MagicMock(), length=mc["exit_length"])
    lot = ParkingLot(1, first.get_socket(0), net, 1, ignore_intersection_checking=False)
    ok1 = lot.construct_block(MagicMock(), MagicMock(), {"one_side_vehicle_number": int(n_space / 2)})
    old = TInterSection.EXIT_PART_LENGTH
    TInterSection.EXIT_PART_LENGTH = 10
    try:
        t = TInterSection(2, lot.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
        ok2 = t.construct_block(MagicMock(), MagicMock(), extra_config={"t_type": 1, "change_lane_num": 0})
    finally:
        TInterSection.EXIT_PART_LENGTH = old
    roads = []
    for f, td in net.graph.items():
        for tt, lanes in td.items():
            roads.append(dict(start=f, end=tt, lanes=[lane_record(l) for l in lanes]))
    in_roads = cfg["in_spawn_roads"]
    out_roads = mp.MultiAgentParkingLotEnv._get_out_spawn_roads(n_space)
    dests = [r.end_node for r in lot.dest_roads] + [(-r).end_node for r in in_roads]
    routes = []
    for sr in in_roads + out_roads:
        for d in dests:
            routes.append(dict(start=[sr.start_node, sr.end_node], dest=d,
                               path=net.shortest_path((sr.start_node, sr.end_node, 0), d)))
    dump("ma_parking_lot.json", dict(
        no_cross=bool(ok1 and ok2), roads=roads, in_spawn_roads=[[r.start_node, r.end_node] for r in in_roads],
        out_spawn_roads=[[r.start_node, r.end_node] for r in out_roads],
        parking_space=[[r.start_node, r.end_node] for r in lot.dest_roads],
        in_direction_of_out=[[x.start_node, x.end_node] for x in (ParkingLot.in_direction_parking_space(r) for r in out_roads)],
        routes=routes, num_agents=int(cfg["num_agents"]), parking_space_num=int(n_space), map_config=dict(mc),
        enable_reverse=bool(cfg["vehicle_config"]["enable_reverse"]),
        max_capacity=int(SpawnManager.max_capacity(in_roads + out_roads, mc["exit_length"], mc["lane_num"])),
        lot_config={k: float(v) for k, v in dict(lot.get_config()).items()},
        t_config={k: float(v) for k, v in dict(t.get_config()).items()}))


def section_scenario_lines():
    """Road lines of a scenario map (component/scenario_block/scenario_block.py:45-99): which map-feature types become line
    bodies, continuous or broken, white or yellow, and the stripe pieces ScenarioBlock.construct_continuous_line /
    construct_broken_line cut a polyline into (the Bullet boxes themselves are recorded, not built: their end points)."""
    from metadrive.component.scenario_block.scenario_block import ScenarioBlock
    from metadrive.constants import PGLineColor, PGLineType, PGDrivableAreaProperty
    from metadrive.type import MetaDriveType
    names = ["UNKNOWN_LINE", "ROAD_LINE_BROKEN_SINGLE_WHITE", "ROAD_LINE_SOLID_SINGLE_WHITE", "ROAD_LINE_SOLID_DOUBLE_WHITE",
             "ROAD_LINE_BROKEN_SINGLE_YELLOW", "ROAD_LINE_BROKEN_DOUBLE_YELLOW", "ROAD_LINE_SOLID_SINGLE_YELLOW",
             "ROAD_LINE_SOLID_DOUBLE_YELLOW", "ROAD_LINE_PASSING_DOUBLE_YELLOW", "UNKNOWN", "ROAD_EDGE_BOUNDARY", "ROAD_EDGE_MEDIAN",
             "ROAD_EDGE_SIDEWALK", "CROSSWALK", "LANE_SURFACE_STREET", "LANE_SURFACE_UNSTRUCTURE", "STOP_SIGN", "SPEED_BUMP"]
    types = {n: dict(road_line=bool(MetaDriveType.is_road_line(n)), broken=bool(MetaDriveType.is_broken_line(n)),
                     yellow=bool(MetaDriveType.is_yellow_line(n)), boundary=bool(MetaDriveType.is_road_boundary_line(n)),
                     lane=bool(MetaDriveType.is_lane(n))) for n in names}
    rng = np.random.RandomState(5)
    cases = []
    for ci in range(7):
        n = int(rng.choice([2, 3, 12, 40]))
        step = float(rng.choice([0.4, 1.0, 2.7]))
        ang = np.cumsum(rng.uniform(-0.08, 0.08, n)) + rng.uniform(-3, 3)
        pts = np.concatenate([[[0.0, 0.0]], np.cumsum(np.stack([np.cos(ang), np.sin(ang)], 1) * step, 0)]) + rng.uniform(-50, 50, 2)
        rec = []

        class Bare(ScenarioBlock):
            def _construct_lane_line_segment(self, start, end, color, line_type):
                rec.append([[float(start[0]), float(start[1])], [float(end[0]), float(end[1])],
                            "yellow" if color == PGLineColor.YELLOW else "grey",
                            "broken" if line_type == PGLineType.BROKEN else "continuous"])
                return []
        b = object.__new__(Bare)
        b._node_path_list = []
        out = {}
        for fn, color in (("construct_continuous_line", PGLineColor.GREY), ("construct_broken_line", PGLineColor.YELLOW)):
            rec.clear()
            getattr(Bare, fn)(b, np.asarray(pts), color)
            out[fn] = [list(r) for r in rec]
        cases.append(dict(polyline=[[float(a), float(c)] for a, c in pts], **out))
    dump("scenario_lines.json", dict(types=types, stripe_length=float(PGDrivableAreaProperty.STRIPE_LENGTH),
                                     line_width=float(PGDrivableAreaProperty.LANE_LINE_WIDTH), cases=cases))


def section_others():
    """The "others" block of LidarStateObservation (obs/state_obs.py:172-183 -> Lidar.get_surrounding_vehicles_info,
    component/sensors/lidar.py:93-138): the num_others nearest DETECTED vehicles, four dims each (relative position /
    perceive_distance, relative velocity / max_speed, in the ego's frame), padded with zeros; non-vehicles among the detected
    objects are dropped first (get_surrounding_vehicles, :76-83).  Fake vehicles (a BaseVehicle subclass with plain attributes);
    convert_to_local_coordinates is supplied as (forward, left), as in agent_step."""
    from metadrive.component.sensors.lidar import Lidar
    from metadrive.component.vehicle.base_vehicle import BaseVehicle
    from metadrive.utils.math import Vector

    class FakeVehicle(BaseVehicle):     # plain attributes instead of the Panda-backed properties
        position = None
        velocity_km_h = None
        max_speed_km_h = 80.0
        navigation = None
        heading = None

        def convert_to_local_coordinates(self, vec, origin):
            hx, hy = self.heading
            dx, dy = vec[0] - origin[0], vec[1] - origin[1]
            return np.array([dx * hx + dy * hy, dy * hx - dx * hy])

    def make(pos, heading, speed):
        v = object.__new__(FakeVehicle)
        v.position = Vector((float(pos[0]), float(pos[1])))
        v.heading = (math.cos(heading), math.sin(heading))
        v.velocity_km_h = Vector((speed * 3.6 * math.cos(heading), speed * 3.6 * math.sin(heading)))
        return v
    rng = np.random.RandomState(31)
    lidar = object.__new__(Lidar)
    cases = []
    for ci in range(40):
        ego_h = float(rng.uniform(-3.1, 3.1))
        ego = make(rng.uniform(-50, 50, 2), ego_h, float(rng.uniform(0, 22)))
        n = int(rng.randint(0, 9))
        others, objs = [], []
        for k in range(n):
            d, a = float(rng.uniform(3, 70)), float(rng.uniform(-3.1, 3.1))    # some beyond the 50 m perceive distance
            pos = (ego.position[0] + d * math.cos(a), ego.position[1] + d * math.sin(a))
            h, sp = float(rng.uniform(-3.1, 3.1)), float(rng.uniform(0, 25))
            is_vehicle = bool(rng.rand() < 0.8)
            if is_vehicle:
                objs.append(make(pos, h, sp))
            else:
                objs.append(MagicMock())      # a cone / pedestrian body: not a BaseVehicle
            others.append(dict(pos=[float(pos[0]), float(pos[1])], heading=h, speed=sp, vehicle=is_vehicle))
        num_others = int(rng.choice([1, 2, 4, 6]))
        dist = float(rng.choice([30.0, 50.0]))
        res = Lidar.get_surrounding_vehicles_info(lidar, ego, set(objs), dist, num_others, False)
        cases.append(dict(ego=dict(pos=[float(ego.position[0]), float(ego.position[1])], heading=ego_h,
                                   speed=float(math.hypot(*ego.velocity_km_h) / 3.6)),
                          others=others, num_others=num_others, perceive_distance=dist, info=[float(x) for x in res]))
    dump("others.json", dict(frame_supplied_by_generator="(forward, left)", cases=cases))


def section_ma_bidirection():
    """Map of MultiAgentBidirectionEnv: FirstPGBlock(60 m, 4 lanes) + Merge (to 1 lane over 3 m) +
    Bidirection (one lane shared by both directions, seed 1) + Split (back to 4, exit 60 m) as MABidirectionMap._generate
    builds it (marl_bidirection.py:28-73)."""
    from metadrive.component.pgblock.bidirection import Bidirection
    from metadrive.component.pgblock.bottleneck import Merge, Split
    from metadrive.component.pgblock.first_block import FirstPGBlock
    from metadrive.component.road_network.node_road_network import NodeRoadNetwork
    from metadrive.envs.marl_envs.marl_bidirection import MABidirectionConfig
    from metadrive.manager.spawn_manager import SpawnManager
    net = NodeRoadNetwork()
    first = FirstPGBlock(net, 3.5, 4, MagicMock(), MagicMock(), length=60)
    merge = Merge(1, first.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    ok1 = merge.construct_from_config(dict(lane_num=3, length=3), MagicMock(), MagicMock())
    both = Bidirection(2, merge.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    ok2 = both.construct_block(MagicMock(), MagicMock())
    split = Split(3, both.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    ok3 = split.construct_from_config({"length": 60, "lane_num": 3}, MagicMock(), MagicMock())
    roads = []
    for f, td in net.graph.items():
        for t, lanes in td.items():
            roads.append(dict(start=f, end=t, lanes=[lane_record(l) for l in lanes]))
    spawn_roads = [[r.start_node, r.end_node] for r in MABidirectionConfig["spawn_roads"]]
    routes = []
    for sr in MABidirectionConfig["spawn_roads"]:
        for er in MABidirectionConfig["spawn_roads"]:
            dest = (-er).end_node
            path = net.shortest_path((sr.start_node, sr.end_node, 0), dest)
            routes.append(dict(start=[sr.start_node, sr.end_node], dest=dest, path=path))
    dump("ma_bidirection.json", dict(no_cross=bool(ok1 and ok2 and ok3), roads=roads, spawn_roads=spawn_roads, routes=routes,
                                     max_capacity=int(SpawnManager.max_capacity(MABidirectionConfig["spawn_roads"], 60, 4)),
                                     num_agents=int(MABidirectionConfig["num_agents"]),
                                     bidirection_config={k: float(v) for k, v in dict(both.get_config()).items()}))


def section_idm():
    """IDM longitudinal model, desired gap, PID steering and the front/back object search
    (SURVEY 8a-10): IDMPolicy.acceleration / desired_gap / steering_control and
    FrontBackObjects.get_find_front_back_objs of the reference on fake vehicles."""
    from types import SimpleNamespace
    from metadrive.policy.idm_policy import IDMPolicy, FrontBackObjects
    from metadrive.component.vehicle.PID_controller import PIDController
    from metadrive.component.lane.straight_lane import StraightLane
    from metadrive.component.lane.circular_lane import CircularLane
    rng = np.random.RandomState(77)
    out = {}

    def policy(ego):
        p = object.__new__(IDMPolicy)
        p.control_object = ego
        p.target_speed = IDMPolicy.NORMAL_SPEED
        p.disable_idm_deceleration = False
        p.heading_pid = PIDController(1.7, 0.01, 3.5)
        p.lateral_pid = PIDController(0.3, .002, 0.05)
        return p

    acc = []
    for _ in range(120):
        h = float(rng.uniform(-3.1, 3.1))
        v = float(rng.uniform(0, 20))
        fv = float(rng.uniform(0, 20))
        fh = h + float(rng.uniform(-0.3, 0.3))
        target = float(rng.choice([30.0, 5.0]))
        has_front = bool(rng.rand() < 0.7)
        dist = float(rng.uniform(0.005, 30))
        ego = SimpleNamespace(speed_km_h=v * 3.6, velocity_km_h=np.array([v * math.cos(h), v * math.sin(h)]) * 3.6,
                              heading=np.array([math.cos(h), math.sin(h)]))
        front = SimpleNamespace(velocity_km_h=np.array([fv * math.cos(fh), fv * math.sin(fh)]) * 3.6, speed_km_h=fv * 3.6)
        p = policy(ego)
        p.target_speed = target
        a = IDMPolicy.acceleration(p, front if has_front else None, dist)
        gap = IDMPolicy.desired_gap(p, ego, front)
        acc.append(dict(v=v, h=h, fv=fv, fh=fh, target=target, has_front=has_front, dist=dist, acc=float(a), gap=float(gap)))
    out["acceleration"] = acc

    steer = []
    lanes = [StraightLane([0, 0], [80, 10], 3.5), CircularLane((5.0, -3.0), 40.0, 0.3, 1.5, True, 3.5),
             CircularLane((5.0, -3.0), 30.0, -2.0, 1.2, False, 3.5)]
    for lane in lanes:
        rec = lane_record(lane)
        seq = []
        ego = SimpleNamespace(position=None, heading_theta=0.0)
        p = policy(ego)
        for k in range(12):
            s_ = float(rng.uniform(2, lane.length - 3))
            lat = float(rng.uniform(-1.5, 1.5))
            pos = lane.position(s_, lat)
            ego.position = (float(pos[0]), float(pos[1]))
            ego.heading_theta = float(lane.heading_theta_at(s_) + rng.uniform(-0.3, 0.3))
            st = IDMPolicy.steering_control(p, lane)
            seq.append(dict(pos=list(ego.position), heading=ego.heading_theta, steering=float(st)))
        rec["sequence"] = seq
        steer.append(rec)
    out["steering"] = steer

    # front/back search on a 3-lane straight road followed by a 3-lane straight road
    def road(x0, x1):
        ls = []
        for i in range(3):
            l = StraightLane([x0, -3.5 * i], [x1, -3.5 * i], 3.5)
            l.index = ("a%d" % x0, "b%d" % x1, i)
            ls.append(l)
        return ls
    r1, r2 = road(0, 60), road(60, 140)
    fb = []
    for _ in range(60):
        ego_lane = r1[int(rng.randint(3))]
        es = float(rng.uniform(5, 58))
        epos = ego_lane.position(es, float(rng.uniform(-0.5, 0.5)))
        objs = []
        for j in range(int(rng.randint(1, 9))):
            lanes_all = r1 + r2
            ol = lanes_all[int(rng.randint(6))]
            os_ = float(rng.uniform(1, ol.length - 1))
            op = ol.position(os_, float(rng.uniform(-0.5, 0.5)))
            objs.append(SimpleNamespace(lane=ol, position=np.array([float(op[0]), float(op[1])]), slot=j))
        res = FrontBackObjects.get_find_front_back_objs(objs, ego_lane, (float(epos[0]), float(epos[1])), 30, r1)
        fb.append(dict(ego_lane=ego_lane.index[2], ego_pos=[float(epos[0]), float(epos[1])],
                       objs=[dict(road=0 if o.lane in r1 else 1, lane=o.lane.index[2], pos=[float(o.position[0]), float(o.position[1])]) for o in objs],
                       front=[(o.slot if o is not None else -1) for o in res.front_objs],
                       back=[(o.slot if o is not None else -1) for o in res.back_objs],
                       front_dist=[(float(d) if d is not None else None) for d in res.front_dist],
                       back_dist=[(float(d) if d is not None else None) for d in res.back_dist]))
    out["front_back"] = fb
    dump("idm.json", out)


def section_idm_policy():
    """IDMPolicy.act (policy/idm_policy.py:235-267) end to end on fake vehicles: move_to_next_road (:269-291),
    lane_change_policy (:330-402) over all its branches (lane-count drop on either side -> forced change / creep,
    overtaking left / right, timer, lane follow), the single-lane search when routing fails or lane change is off,
    and the bare-except fallback (:254-260) -- in particular with a traffic participant (an object WITHOUT `.lane`,
    traffic_participants/base_traffic_participant.py:12-32) among the surrounding objects.
    World: roads P (-60..0) -> A (0..60) -> B (60..140) along +x, 3 lanes each except the B variants with 2 lanes
    (aligned with A's lanes 0-1 or 1-2); Q is a road that nothing connects to.  The ego's current road is A."""
    from types import SimpleNamespace
    from metadrive.policy.idm_policy import IDMPolicy, FrontBackObjects
    from metadrive.component.vehicle.PID_controller import PIDController
    from metadrive.component.lane.straight_lane import StraightLane
    rng = np.random.RandomState(4242)
    W = 3.5

    def road(name0, name1, x0, x1, lane_ids):
        ls = []
        for k, i in enumerate(lane_ids):          # i = lateral slot (y = -W * i), k = index inside the road
            l = StraightLane([x0, -W * i], [x1, -W * i], W)
            l.index = (name0, name1, k)
            ls.append(l)
        return ls

    class Net:          # has_connection only reads .graph (road_network/base_road_network.py:106-113)
        def __init__(self, graph):
            self.graph = graph
    from metadrive.component.road_network.base_road_network import BaseRoadNetwork
    has_connection = BaseRoadNetwork.has_connection

    raised = []
    orig_lcp = IDMPolicy.lane_change_policy
    orig_find = FrontBackObjects.get_find_front_back_objs.__func__

    ret_line = []

    def lcp(self, objs):
        # which `return` of lane_change_policy was taken: the line number of the frame's return event
        def tracer(frame, event, arg):
            if frame.f_code is orig_lcp.__code__:
                def local(fr, ev, a):
                    if ev == "return":
                        ret_line.append(fr.f_lineno)
                    return local
                return local
            return None
        old_trace = sys.gettrace()
        sys.settrace(tracer)
        try:
            return orig_lcp(self, objs)
        except BaseException as e:
            raised.append(type(e).__name__)
            raise
        finally:
            sys.settrace(old_trace)

    def find(cls, *a, **k):
        try:
            return orig_find(cls, *a, **k)
        except BaseException as e:
            raised.append(type(e).__name__)
            raise

    IDMPolicy.lane_change_policy = lcp
    FrontBackObjects.get_find_front_back_objs = classmethod(find)
    cases = []
    try:
        for n in range(520):
            variant = int(rng.randint(4))      # 0: B has 3 lanes, 1: B = A's lanes 0-1, 2: B = A's lanes 1-2, 3: A is the last road
            P = road("n0", "n1", -60, 0, [0, 1, 2])
            A = road("n1", "n2", 0, 60, [0, 1, 2])
            B = road("n2", "n3", 60, 140, {0: [0, 1, 2], 1: [0, 1], 2: [1, 2], 3: [0, 1, 2]}[variant])
            Q = road("n4", "n5", 0, 60, [5, 6, 7])
            graph = {"n0": {"n1": P}, "n1": {"n2": A}, "n2": {"n3": B}, "n4": {"n5": Q}}
            net = SimpleNamespace(graph=graph)
            net.has_connection = lambda a, b, _n=net: has_connection(_n, a, b)
            all_lanes = dict(P=P, A=A, B=B, Q=Q)

            e_lane_road = "A" if rng.rand() < 0.85 else ("B" if rng.rand() < 0.5 else "P")
            e_lanes = all_lanes[e_lane_road]
            e_lane = e_lanes[int(rng.randint(len(e_lanes)))]
            on_lane = A[min(e_lane.index[2], 2)] if e_lane_road != "A" else e_lane
            es = float(rng.uniform(5, 55))
            elat = float(rng.uniform(-0.6, 0.6))
            epos = on_lane.position(es, elat)
            if e_lane_road == "B":
                epos = B[e_lane.index[2]].position(float(rng.uniform(1, 8)), elat)
            elif e_lane_road == "P":
                epos = P[e_lane.index[2]].position(float(rng.uniform(52, 59)), elat)
            eh = float(rng.uniform(-0.25, 0.25))
            ev = float(rng.choice([rng.uniform(0, 3), rng.uniform(7.5, 9.2), rng.uniform(10, 16)]))   # m/s; 30 km/h = 8.33
            kind = rng.rand()
            forced = variant in (1, 2) and rng.rand() < 0.6
            if forced:       # the routing target lane does not continue into B: forced lane change / creep branches
                e_lane_road, e_lane = "A", A[2 if variant == 1 else 0]
                epos = e_lane.position(es, elat)
                kind = 0.3
            if kind < 0.2:
                tgt = None
            elif kind < 0.55:
                tgt = e_lane
            elif kind < 0.75:
                tgt = A[int(rng.randint(3))]
            elif kind < 0.92:
                tgt = P[int(rng.randint(3))]
            else:
                tgt = Q[int(rng.randint(3))]
            timer0 = int(rng.randint(0, 90))
            enable_lc = bool(rng.rand() < 0.88)
            target_speed0 = float(rng.choice([30.0, 5.0]))

            nav = SimpleNamespace(current_ref_lanes=A, next_ref_lanes=None if variant == 3 else B,
                                  map=SimpleNamespace(road_network=net))
            ego = SimpleNamespace(lane=e_lane, navigation=nav, position=(float(epos[0]), float(epos[1])), heading_theta=eh,
                                  speed_km_h=ev * 3.6, velocity_km_h=np.array([math.cos(eh), math.sin(eh)]) * ev * 3.6,
                                  heading=np.array([math.cos(eh), math.sin(eh)]))
            objs, recs = [], []
            with_participant = rng.rand() < 0.12
            for j in range(int(rng.randint(0, 9))):
                rname = ["P", "A", "A", "A", "B", "B"][int(rng.randint(6))]
                lanes_r = all_lanes[rname]
                ol = lanes_r[int(rng.randint(len(lanes_r)))]
                x = float(np.clip(epos[0] + rng.uniform(-44, 44), ol.start[0] + 0.5, ol.end[0] - 0.5))
                op = ol.position(x - ol.start[0], float(rng.uniform(-0.4, 0.4)))
                is_cone = rng.rand() < 0.15
                ov = 0.0 if is_cone else float(rng.choice([rng.uniform(0, 3), rng.uniform(7.5, 9.2), rng.uniform(10, 16)]))
                o = SimpleNamespace(lane=ol, position=np.array([float(op[0]), float(op[1])]), speed_km_h=ov * 3.6,
                                    velocity_km_h=np.array([ov * 3.6, 0.0]), slot=j + 1)
                objs.append(o)
                recs.append(dict(kind="cone" if is_cone else "vehicle", road=rname, lane=ol.index[2],
                                 pos=[float(op[0]), float(op[1])], speed=ov))
            setup = rng.rand()
            if setup < 0.3 and e_lane_road == "A":
                # an object close by on a neighbouring lane (unsafe lane change -> creep) ...
                nb = A[int(np.clip(e_lane.index[2] + (1 if rng.rand() < 0.5 else -1), 0, 2))]
                x = float(np.clip(epos[0] + rng.uniform(-12, 6), 0.5, 59.5))
                ov = float(rng.uniform(0, 14))
                objs.append(SimpleNamespace(lane=nb, position=np.array([x, float(nb.start[1])]), speed_km_h=ov * 3.6,
                                            velocity_km_h=np.array([ov * 3.6, 0.0]), slot=len(objs) + 1))
                recs.append(dict(kind="vehicle", road="A", lane=nb.index[2], pos=[x, float(nb.start[1])], speed=ov))
            elif setup < 0.55 and e_lane_road == "A":
                # ... or an overtaking set-up: slow lead vehicle on the ego lane, timer past LANE_CHANGE_FREQ
                timer0 = int(rng.randint(51, 95))
                x = float(np.clip(epos[0] + rng.uniform(4, 25), 0.5, 59.5))
                ov = float(rng.uniform(0, 6))
                objs.append(SimpleNamespace(lane=e_lane, position=np.array([x, float(e_lane.start[1])]), speed_km_h=ov * 3.6,
                                            velocity_km_h=np.array([ov * 3.6, 0.0]), slot=len(objs) + 1))
                recs.append(dict(kind="vehicle", road="A", lane=e_lane.index[2], pos=[x, float(e_lane.start[1])], speed=ov))
            if with_participant:
                pp = [float(epos[0] + rng.uniform(-40, 40)), float(rng.uniform(-12, 5))]
                pos_at = int(rng.randint(0, len(objs) + 1))
                o = SimpleNamespace(position=np.array(pp), speed_km_h=3.0, velocity_km_h=np.array([0.0, 3.0]), slot=None)   # no .lane
                objs.insert(pos_at, o)
                recs.insert(pos_at, dict(kind="pedestrian" if rng.rand() < 0.6 else "cyclist", pos=pp, speed=0.0))
                for k, oo in enumerate(objs):
                    oo.slot = k + 1
            ego.lidar = SimpleNamespace(get_surrounding_objects=lambda v, _o=objs: list(_o))

            p = object.__new__(IDMPolicy)
            p.control_object = ego
            p.target_speed = target_speed0
            p.routing_target_lane = tgt
            p.available_routing_index_range = None
            p.overtake_timer = timer0
            p.enable_lane_change = enable_lc
            p.disable_idm_deceleration = False
            p.heading_pid = PIDController(1.7, 0.01, 3.5)
            p.lateral_pid = PIDController(0.3, .002, 0.05)
            p.action_info = {}
            p.np_random = SimpleNamespace(randint=lambda lo, hi: 7)
            del raised[:]
            del ret_line[:]
            action = IDMPolicy.act(p)

            def lane_name(l):
                if l is None:
                    return None
                for rn, ls in all_lanes.items():
                    for q in ls:
                        if q is l:
                            return [rn, q.index[2]]
                raise AssertionError("unknown lane")

            cases.append(dict(variant=variant, ego=dict(lane=lane_name(e_lane), pos=list(ego.position), heading=eh, speed=ev),
                              target0=lane_name(tgt), timer0=timer0, enable_lane_change=enable_lc, target_speed0=target_speed0,
                              objs=recs, action=[float(action[0]), float(action[1])], target1=lane_name(p.routing_target_lane),
                              timer1=int(p.overtake_timer), target_speed1=float(p.target_speed), fallback=bool(raised),
                              exc=(raised[0] if raised else None),
                              lcp_return_line=(ret_line[-1] if ret_line and not raised else None)))
    finally:
        IDMPolicy.lane_change_policy = orig_lcp
        FrontBackObjects.get_find_front_back_objs = classmethod(orig_find)
    n_fb = sum(c["fallback"] for c in cases)
    n_part = sum(any(o["kind"] in ("pedestrian", "cyclist") for o in c["objs"]) for c in cases)
    assert all(c["fallback"] and c["exc"] == "AttributeError" for c in cases
               if any(o["kind"] in ("pedestrian", "cyclist") for o in c["objs"])), "a lane-less object must raise"
    from collections import Counter
    print("lane_change_policy return lines:", sorted(Counter(c["lcp_return_line"] for c in cases).items(), key=lambda kv: str(kv[0])))
    print("idm_policy: %d cases, %d fallbacks, %d with a participant, creep %d, timer kept %d" % (
        len(cases), n_fb, n_part, sum(c["target_speed1"] == 5.0 for c in cases), sum(c["timer1"] == c["timer0"] for c in cases)))
    dump("idm_policy.json", dict(lane_width=W, rand_value=7, cases=cases))


def section_scenario():
    """Scenario mode (BASELINE configs[4]): the reference's own classes on synthetic tracks --
    InterpolatingLine / PointLane (segments, local_coordinates, position, heading_theta_at, lateral_direction, the outline
    polygon), TrajectoryNavigation.update_localization (22 dims incl. the unpacking quirk at trajectory_navigation.py:134,
    route completion), ScenarioEnv reward / cost / done, TrajectoryIDMPolicy.act (steering PID, speed control every fifth
    step, single-lane front search), get_max_valid_indicis and the static-car test.
    Supplied by the generator (placeholders in this container): BaseVehicle.convert_to_local_coordinates = the (forward,
    left) projection (as in agent_step.json); lane.point_on_lane (shapely) = an even-odd point-in-polygon test on the
    lane's own polygon."""
    import types
    from collections import deque
    from types import SimpleNamespace
    sys.path.insert(0, ROOT)
    from metadrive.component.lane.point_lane import PointLane
    from metadrive.component.navigation_module.trajectory_navigation import TrajectoryNavigation
    from metadrive.component.vehicle.base_vehicle import BaseVehicle
    from metadrive.component.vehicle.PID_controller import PIDController
    from metadrive.envs.scenario_env import ScenarioEnv, SCENARIO_ENV_CONFIG
    from metadrive.policy.idm_policy import TrajectoryIDMPolicy, FrontBackObjects
    from metadrive.scenario.parse_object_state import get_idm_route, get_max_valid_indicis
    from metadrive.manager.scenario_traffic_manager import ScenarioTrafficManager
    from metadrive.obs.state_obs import StateObservation
    from metadrive.component.map.base_map import BaseMap
    from gen_inputs import frozen_scenario as synthetic_scenario   # frozen inputs: never the product's own generator
    rng = np.random.RandomState(515)
    out = dict(supplied_by_generator=["convert_to_local_coordinates (forward, left)", "point_on_lane (even-odd on lane.polygon)"])

    def crossing(poly, p):
        inside = False
        n = len(poly)
        j = n - 1
        for i in range(n):
            xi, yi, xj, yj = poly[i][0], poly[i][1], poly[j][0], poly[j][1]
            if ((yi > p[1]) != (yj > p[1])) and (p[0] < (xj - xi) * (p[1] - yi) / (yj - yi) + xi):
                inside = not inside
            j = i
        return inside

    PointLane.point_on_lane = lambda self, point: crossing(self.polygon, point)

    # ---- polylines -------------------------------------------------------------------------------
    lines = []
    tracks_for_lines = []
    for seed in (1, 2):
        sc = synthetic_scenario(seed, T=120)
        for oid in ("0", "5", "12"):
            st = sc["tracks"][oid]["state"]
            v = st["valid"].astype(bool)
            tracks_for_lines.append(np.asarray(st["position"], np.float64)[v][:, :2])
    # a hand-made one with repeated points, a stop in the middle and a sharp corner
    tracks_for_lines.append(np.array([[0, 0], [0.3, 0], [0.6, 0], [0.6, 0], [0.6, 0], [1.4, 0.1], [2.9, 0.2], [4.0, 0.2], [4.0, 1.5],
                                      [4.0, 3.2], [3.0, 5.0], [3.0, 5.0], [2.0, 7.5]], np.float64))
    tracks_for_lines.append(np.array([[5.0, 5.0]] * 6, np.float64))          # never moves
    for pts in tracks_for_lines:
        lane = get_idm_route(pts)
        segs = [dict(start=[float(x) for x in sg["start_point"]], end=[float(x) for x in sg["end_point"]], length=float(sg["length"]),
                     heading=float(sg["heading"]), direction=[float(x) for x in sg["direction"]],
                     lateral_direction=[float(x) for x in sg["lateral_direction"]]) for sg in lane.segment_property]
        q = []
        for _ in range(30):
            s_ = float(rng.uniform(-3, lane.length + 3))
            lat = float(rng.uniform(-6, 6))
            base = lane.position(float(np.clip(s_, 0, lane.length)), 0)
            h = lane.heading_theta_at(float(np.clip(s_, 0, lane.length)))
            p = (float(base[0] + lat * math.sin(h) + rng.uniform(-0.3, 0.3)), float(base[1] - lat * math.cos(h) + rng.uniform(-0.3, 0.3)))
            lg, lt = lane.local_coordinates(p)
            pos = lane.position(s_, lat)
            ld = lane.lateral_direction(lg)
            q.append(dict(point=list(p), long=float(lg), lat=float(lt), s=s_, lateral=lat, position=[float(pos[0]), float(pos[1])],
                          heading_at_long=float(lane.heading_theta_at(lg)), lateral_direction=[float(ld[0]), float(ld[1])]))
        lines.append(dict(points=pts.tolist(), length=float(lane.length), width=float(lane.width), segments=segs, queries=q,
                          polygon=np.asarray(lane.polygon).tolist(), start=[float(x) for x in lane.start], end=[float(x) for x in lane.end]))
    out["polylines"] = lines

    # ---- TrajectoryNavigation + state observation + ScenarioEnv reward / cost / done ---------------------
    def make_vehicle(pos, heading, speed, last_heading, act, flags, lane, nav):
        hx, hy = math.cos(heading), math.sin(heading)
        veh = SimpleNamespace(position=np.array(pos, np.float64), heading=np.array([hx, hy]), heading_theta=heading,
                              last_heading_dir=np.array([math.cos(last_heading), math.sin(last_heading)]), speed=speed,
                              speed_km_h=speed * 3.6, max_speed_km_h=80.0, steering=act[0], MAX_STEERING=BaseVehicle.MAX_STEERING,
                              last_current_action=[(0.0, 0.0), (act[0], act[1])], current_action=act, lane=lane, navigation=nav,
                              engine=None, WIDTH=1.852, LENGTH=4.515,
                              config={"side_detector": {"num_lasers": 0, "distance": 50}, "lane_line_detector": {"num_lasers": 0, "distance": 20}},
                              **flags)
        veh.convert_to_local_coordinates = lambda vec, origin, hx=hx, hy=hy: np.array(
            [(vec[0] - (origin[0] if hasattr(origin, "__len__") else origin)) * hx + (vec[1] - (origin[1] if hasattr(origin, "__len__") else origin)) * hy,
             (vec[1] - (origin[1] if hasattr(origin, "__len__") else origin)) * hx - (vec[0] - (origin[0] if hasattr(origin, "__len__") else origin)) * hy])
        veh.heading_diff = types.MethodType(BaseVehicle.heading_diff, veh)
        return veh

    agent_cases = []
    for seed in (11, 12, 13):
        sc = synthetic_scenario(seed, T=120)
        pts = np.asarray(sc["tracks"]["0"]["state"]["position"], np.float64)[:, :2]
        lane = get_idm_route(pts)

        class Nav(TrajectoryNavigation):
            reference_trajectory = lane
            engine = SimpleNamespace(global_config=dict(max_lateral_dist=4.0))
            map = SimpleNamespace(MAX_LANE_NUM=BaseMap.MAX_LANE_NUM, MAX_LANE_WIDTH=BaseMap.MAX_LANE_WIDTH)

        nav = object.__new__(Nav)
        nav._navi_info = np.zeros((TrajectoryNavigation.get_navigation_info_dim(), ), dtype=np.float32)
        nav._show_navi_info = False
        nav._ckpt_vis_models = None
        nav._current_lane = lane
        nav.checkpoints = TrajectoryNavigation.discretize_reference_trajectory(nav)
        nav.last_current_long = deque([0.0, 0.0], maxlen=2)
        nav.last_current_lat = deque([0.0, 0.0], maxlen=2)
        nav.last_current_heading_theta_at_long = deque([0.0, 0.0], maxlen=2)
        nav._route_completion = 0
        samples = []
        s_prev = float(rng.uniform(0, 10))
        for k in range(60):
            kind = rng.rand()
            if kind < 0.08:
                s_now = float(rng.uniform(-6, 1))                     # before the start: negative completion
            elif kind < 0.2:
                s_now = float(rng.uniform(lane.length * 0.94, lane.length + 3))   # at the end
            else:
                s_now = float(np.clip(s_prev + rng.uniform(-0.5, 2.5), 0, lane.length))
            lat = float(rng.uniform(-5.5, 5.5) if rng.rand() < 0.35 else rng.uniform(-1.0, 1.0))
            base = lane.position(float(np.clip(s_now, 0, lane.length)), lat)
            if s_now < 0:
                base = lane.position(0, lat) + s_now * np.array([math.cos(lane.heading_theta_at(0)), math.sin(lane.heading_theta_at(0))])
            heading = float(lane.heading_theta_at(float(np.clip(s_now, 0, lane.length))) + rng.uniform(-0.6, 0.6))
            speed = float(rng.choice([rng.uniform(0, 0.5), rng.uniform(2, 20)]))
            last_heading = heading - float(rng.uniform(-0.08, 0.08))
            act = [float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))]
            flags = dict(crash_vehicle=bool(rng.rand() < 0.12), crash_object=bool(rng.rand() < 0.08), crash_building=bool(rng.rand() < 0.03),
                         crash_human=bool(rng.rand() < 0.05), crash_sidewalk=bool(rng.rand() < 0.06),
                         on_yellow_continuous_line=bool(rng.rand() < 0.06), on_white_continuous_line=bool(rng.rand() < 0.06))
            veh = make_vehicle([float(base[0]), float(base[1])], heading, speed, last_heading, act, flags, lane, nav)
            long_before = float(nav.last_current_long[1])
            TrajectoryNavigation.update_localization(nav, veh)
            veh.dist_to_left_side, veh.dist_to_right_side = BaseVehicle._dist_to_route_left_right(veh)
            obs_self = SimpleNamespace(config={"random_agent_model": False}, engine=None)
            state9 = [float(x) for x in StateObservation.vehicle_state(obs_self, veh)]
            cfgv = int(rng.randint(4))
            cfg = dict(SCENARIO_ENV_CONFIG)
            steps, horizon = int(rng.randint(0, 200)), int(rng.choice([0, 100]))
            cfg.update(horizon=(horizon or None), truncate_as_terminate=bool(rng.rand() < 0.3), relax_out_of_road_done=bool(cfgv != 1),
                       out_of_route_done=bool(cfgv == 1 and rng.rand() < 0.5), crash_vehicle_done=bool(cfgv == 2), crash_object_done=bool(cfgv == 2),
                       crash_human_done=bool(cfgv == 2), no_negative_reward=bool(cfgv != 3),
                       allowed_more_steps=(int(rng.choice([0, 20])) or None))
            eng = SimpleNamespace(curriculum_manager=MagicMock(), data_manager=SimpleNamespace(current_scenario_length=160, current_scenario_id="x",
                                  current_scenario_difficulty=0, data_coverage=0), current_level=0, current_seed=seed,
                                  map_manager=SimpleNamespace(num_stored_maps=0))
            env = SimpleNamespace(agents={"a": veh}, config=cfg, episode_lengths={"a": steps + 1}, current_seed=seed, logger=MagicMock(), engine=eng)
            env._is_out_of_road = types.MethodType(ScenarioEnv._is_out_of_road, env)
            env._is_arrive_destination = ScenarioEnv._is_arrive_destination
            reward, rinfo = ScenarioEnv.reward_function(env, "a")
            done, dinfo = ScenarioEnv.done_function(env, "a")
            cost, _ = ScenarioEnv.cost_function(env, "a")
            samples.append(dict(pos=[float(base[0]), float(base[1])], heading=heading, speed=speed, last_heading=last_heading, action=act,
                                flags=flags, steps=steps, horizon=horizon, long_before=long_before, long=float(nav.last_current_long[1]),
                                lat=float(nav.last_current_lat[1]), heading_at=float(nav.last_current_heading_theta_at_long[1]),
                                navi=[float(x) for x in nav._navi_info], route_completion=float(nav.route_completion), state9=state9,
                                config={k_: cfg[k_] for k_ in ("truncate_as_terminate", "relax_out_of_road_done", "out_of_route_done",
                                                               "crash_vehicle_done", "crash_object_done", "crash_human_done",
                                                               "no_negative_reward", "allowed_more_steps")},
                                reward=float(reward), step_reward=float(rinfo["step_reward"]), cost=float(cost), done=bool(done),
                                done_info={k_: bool(v) for k_, v in dinfo.items() if k_ != "env_seed"}))
            s_prev = max(s_now, 0.0)
        agent_cases.append(dict(points=pts.tolist(), checkpoints=np.asarray(nav.checkpoints).tolist(), scenario_length=160, samples=samples))
    out["agent"] = agent_cases

    # ---- TrajectoryIDMPolicy.act ----------------------------------------------------------------------
    idm_cases = []
    for seed in (21, 22):
        sc = synthetic_scenario(seed, T=120)
        for oid in ("1", "2", "3"):
            st = sc["tracks"][oid]["state"]
            v = st["valid"].astype(bool)
            pts = np.asarray(st["position"], np.float64)[v][:, :2]
            route = get_idm_route(pts)
            p = object.__new__(TrajectoryIDMPolicy)
            p.traj_to_follow = route
            p.routing_target_lane = route
            p.target_speed = TrajectoryIDMPolicy.NORMAL_SPEED
            p.destination = np.asarray(route.end)
            p.enable_lane_change = False
            p.disable_idm_deceleration = False
            p.heading_pid = PIDController(1.2, 0.1, 3.5)
            p.lateral_pid = PIDController(0.3, .0, 0.0)
            p.last_action = [0, 0]
            p.action_info = {}
            p.policy_index = int(rng.randint(5))
            seq = []
            s_now = float(rng.uniform(2, 20))
            for step in range(1, 26):
                s_now = float(min(s_now + rng.uniform(0.2, 1.5), route.length - 0.5))
                lat = float(rng.uniform(-0.8, 0.8))
                pos = route.position(s_now, lat)
                heading = float(route.heading_theta_at(s_now) + rng.uniform(-0.3, 0.3))
                speed = float(rng.uniform(0, 14))
                hx, hy = math.cos(heading), math.sin(heading)
                objs, recs = [], []
                for j in range(int(rng.randint(0, 5))):
                    ds = float(rng.uniform(-15, 26))
                    olat = float(rng.choice([0.0, 0.0, 1.6, 3.5, -3.5]) + rng.uniform(-0.3, 0.3))
                    op = route.position(float(np.clip(s_now + ds, 0, route.length)), olat)
                    oh = float(route.heading_theta_at(float(np.clip(s_now + ds, 0, route.length))) + rng.uniform(-0.2, 0.2))
                    ov = float(rng.uniform(0, 12))
                    L, W = 4.6, 1.9
                    c_, s__ = math.cos(oh), math.sin(oh)
                    bb = [np.array([op[0] + c_ * L / 2 - s__ * W / 2 * sg2, op[1] + s__ * L / 2 * 1 + c_ * W / 2 * sg2]) if False else None for sg2 in (1, )]
                    corners = [np.array([op[0] + sx * c_ * L / 2 - sy * s__ * W / 2, op[1] + sx * s__ * L / 2 + sy * c_ * W / 2])
                               for sx, sy in ((1, 1), (1, -1), (-1, -1), (-1, 1))]
                    objs.append(SimpleNamespace(position=np.array([float(op[0]), float(op[1])]), bounding_box=corners,
                                                velocity_km_h=np.array([ov * c_, ov * s__]) * 3.6, speed_km_h=ov * 3.6, slot=j + 1))
                    recs.append(dict(pos=[float(op[0]), float(op[1])], heading=oh, speed=ov, length=L, width=W))
                ego = SimpleNamespace(position=np.array([float(pos[0]), float(pos[1])]), heading_theta=heading, speed_km_h=speed * 3.6,
                                      velocity_km_h=np.array([hx, hy]) * speed * 3.6, heading=np.array([hx, hy]),
                                      lidar=SimpleNamespace(get_surrounding_objects=lambda v_, _o=objs: list(_o)))
                p.control_object = ego
                do_speed = (step % ScenarioTrafficManager.IDM_ACT_BATCH_SIZE) == p.policy_index
                arrived = bool(p.arrive_destination)
                action = TrajectoryIDMPolicy.act(p, do_speed)
                seq.append(dict(step=step, pos=[float(pos[0]), float(pos[1])], heading=heading, speed=speed, objs=recs,
                                do_speed_control=bool(do_speed), arrived=arrived, action=[float(action[0]), float(action[1])]))
            idm_cases.append(dict(points=pts.tolist(), policy_index=p.policy_index, polygon=np.asarray(route.polygon).tolist(), sequence=seq))
    out["traj_idm"] = idm_cases

    # ---- track bookkeeping --------------------------------------------------------------------------
    book = []
    for seed in (31, 32):
        sc = synthetic_scenario(seed, T=120)
        for oid, tr in sc["tracks"].items():
            st = tr["state"]
            valid = st["valid"].astype(bool)
            if not valid.any():
                continue
            t0 = int(np.nonzero(valid)[0][0])
            a, b = get_max_valid_indicis(tr, t0)
            vp = st["position"][np.where(st["valid"])]
            moving = bool(np.max(np.std(vp, axis=0)[:2]) > ScenarioTrafficManager.STATIC_THRESHOLD)
            d = st["position"][a][..., :2] - st["position"][b - 1][..., :2]
            book.append(dict(seed=seed, oid=oid, type=tr["type"], t0=t0, run=[int(a), int(b)], moving=moving,
                             length_ok=bool(math.hypot(float(d[0]), float(d[1])) > ScenarioTrafficManager.IDM_CREATE_MIN_LENGTH),
                             noise=bool(np.sum(st["valid"]) < ScenarioTrafficManager.MIN_VALID_FRAME_LEN)))
    out["bookkeeping"] = book
    dump("scenario.json", out)


def section_scenario_spawn():
    """ScenarioTrafficManager.spawn_vehicle (manager/scenario_traffic_manager.py:171-236) -- the reference's own method on a manager
    whose engine is a recording stand-in (spawn_object / add_policy need Bullet): for posed egos and single tracks at an episode
    step k, whether a vehicle is created, with which policy (replay / TrajectoryIDMPolicy), its policy index, and the route the
    reactive policy gets -- PointLane(positions[k : end of the valid run]) (get_max_valid_indicis, get_idm_route): start, end,
    length, number of pieces.
    Supplied by the generator: BaseVehicle.convert_to_local_coordinates = the (forward, left) projection (as in agent_step.json)."""
    from types import SimpleNamespace
    sys.path.insert(0, ROOT)
    from metadrive.manager.scenario_traffic_manager import ScenarioTrafficManager
    from metadrive.policy.idm_policy import TrajectoryIDMPolicy
    from metadrive.policy.replay_policy import ReplayTrafficParticipantPolicy
    from gen_inputs import _track_dict
    rng = np.random.RandomState(99)
    T = 60

    class Probe(ScenarioTrafficManager):
        """spawn_vehicle with everything Bullet-side recorded instead of executed"""
        def __init__(self, cfg, ego, step, idm_count):
            self._cfg, self._ego, self._step = cfg, ego, step
            self._static_car_id, self._moving_car_id = set(), set()
            self._scenario_id_to_obj_id, self._obj_id_to_scenario_id = {}, {}
            self.idm_policy_count = idm_count
            self.even_sample_v, self.need_default_vehicle = True, False
            self.is_ego_vehicle_replay = False
            self._filter_overlapping_car = cfg["filter_overlapping_car"]
            self._traffic_v_config = {}
            self.np_random = np.random.RandomState(0)
            self.spawned, self.policy = None, None

        engine = property(lambda self: SimpleNamespace(global_config=self._cfg))
        episode_step = property(lambda self: self._step)
        ego_vehicle = property(lambda self: self._ego)

        def spawn_object(self, cls, **kw):
            self.spawned = dict(cls=cls.__name__, position=[float(x) for x in kw["position"]], heading=float(kw["heading"]))
            return SimpleNamespace(name="obj")

        def add_policy(self, name, cls, *args):
            self.policy = dict(cls=cls.__name__)
            if cls is TrajectoryIDMPolicy:
                lane, index = args[2], args[3]
                self.policy.update(policy_index=int(index), route_start=[float(x) for x in lane.start], route_end=[float(x) for x in lane.end],
                                   route_length=float(lane.length), route_pieces=len(lane.segment_property))
            return SimpleNamespace(act=lambda *a, **k: None)

        def generate_seed(self):
            return 0

    cases = []
    for ci in range(400):
        ego_h = float(rng.uniform(-math.pi, math.pi))
        ego_p = np.array([rng.uniform(-20, 20), rng.uniform(-20, 20)])
        ego = SimpleNamespace(position=ego_p, heading_theta=ego_h)
        c_, s_ = math.cos(ego_h), math.sin(ego_h)
        ego.convert_to_local_coordinates = lambda v, o, c_=c_, s_=s_: ((v[0] - o[0]) * c_ + (v[1] - o[1]) * s_, (v[1] - o[1]) * c_ - (v[0] - o[0]) * s_)
        k = int(rng.choice([0, 0, 7, 13, 25]))
        # the track at frame k, relative to the ego: near the thresholds of every filter in a good share of the cases
        if ci % 2 == 0:     # a candidate for a reactive policy: behind the ego, beside it, heading its way, moving
            fwd = float(rng.choice([rng.uniform(-30, -1.2), rng.uniform(-9, -7), rng.uniform(-1.5, -0.5)]))
            side = float(rng.choice([rng.uniform(-14, 14), rng.uniform(-2.5, 2.5), rng.uniform(14, 16), rng.uniform(-16, -14)]))
            rel_h = float(rng.choice([rng.uniform(-1.4, 1.4), math.pi / 2 + rng.uniform(-0.05, 0.05)]))
            speed_mps = float(rng.choice([rng.uniform(2, 12), rng.uniform(2, 12), rng.uniform(0.3, 2.5)]))
        else:
            fwd = float(rng.choice([rng.uniform(-40, 40), rng.uniform(-9, -7), rng.uniform(7, 9), rng.uniform(-1.5, -0.5)]))
            side = float(rng.choice([rng.uniform(-25, 25), rng.uniform(-2.5, 2.5), rng.uniform(14, 16), rng.uniform(-16, -14)]))
            rel_h = float(rng.choice([rng.uniform(-math.pi, math.pi), rng.uniform(-0.3, 0.3), math.pi / 2 + rng.uniform(-0.05, 0.05)]))
            speed_mps = float(rng.choice([0.0, 0.2, rng.uniform(0.5, 3.0), rng.uniform(2, 12)]))
        h = ego_h + rel_h
        t = np.arange(T)
        xk, yk = float(ego_p[0] + fwd * c_ - side * s_), float(ego_p[1] + fwd * s_ + side * c_)
        px = xk + (t - k) * 0.1 * speed_mps * math.cos(h)       # (the test rebuilds the positions with these two lines)
        py = yk + (t - k) * 0.1 * speed_mps * math.sin(h)
        valid = np.ones(T, bool)
        pattern = int(rng.randint(0, 5))
        if pattern == 1:
            valid[:k] = False                                     # appears at k
        elif pattern == 2 and k > 3:
            valid[k - 3:k] = False                                # a gap right before k: second run
        elif pattern == 3:
            valid[k + int(rng.randint(2, 20)):] = False           # ends soon after k
        elif pattern == 4:
            valid[k] = bool(rng.rand() < 0.5)                     # maybe not valid at k at all
        tr = _track_dict("9", "VEHICLE", T, valid, px, py, np.full(T, h), np.full(T, speed_mps), float(rng.choice([3.6, 4.6, 6.5])), 1.9, 1.6)
        cfg = dict(no_static_vehicles=bool(rng.rand() < 0.25), reactive_traffic=bool(rng.rand() < 0.8), force_reuse_object_name=False,
                   top_down_show_real_size=False, filter_overlapping_car=bool(rng.rand() < 0.8))
        idm_count = int(rng.randint(0, 9))
        m = Probe(cfg, ego, k, idm_count)
        m.spawn_vehicle("9", tr)
        st = tr["state"]
        cases.append(dict(ego_position=[float(x) for x in ego_p], ego_heading=ego_h, step=k, idm_count=idm_count, config=cfg,
                          track=dict(xk=xk, yk=yk, heading=float(h), speed=speed_mps, valid="".join("1" if v else "0" for v in valid),
                                     length=float(st["length"].max())),
                          spawned=m.spawned, policy=m.policy, idm_count_after=int(m.idm_policy_count),
                          moving=("9" in m._moving_car_id)))
    kinds = [(c["spawned"] is not None, (c["policy"] or {}).get("cls")) for c in cases]
    print("scenario_spawn: %d cases, not spawned %d, replay %d, reactive %d (of them at a later frame than the run's start: %d)" % (
        len(cases), sum(1 for a, b in kinds if not a), sum(1 for a, b in kinds if b == "ReplayTrafficParticipantPolicy"),
        sum(1 for a, b in kinds if b == "TrajectoryIDMPolicy"),
        sum(1 for c in cases if (c["policy"] or {}).get("cls") == "TrajectoryIDMPolicy" and c["step"] > 0 and c["track"]["valid"][c["step"] - 1] == "1")))
    dump("scenario_spawn.json", dict(supplied_by_generator=["convert_to_local_coordinates (forward, left)"], frames=T, cases=cases))


def _cls_const(cls, name):
    """Class constant of a vehicle class; LENGTH / WIDTH are properties returning a literal (vehicle_type.py:155-165)."""
    v = getattr(cls, name)
    return v.fget(None) if isinstance(v, property) else v


def section_traffic_spawn():
    """Per-seed traffic placement: the reference's own PGTrafficManager.reset
    (manager/traffic_manager.py:51-72) -> _create_vehicles_once (:230-277) / _create_respawn_vehicles (:213-228) /
    _propose_vehicle_configs (:201-211) / _get_available_respawn_lanes (:279-296) on BIG maps of the default block
    distribution, with the manager's own seeded stream.  spawn_object is a recorder that consumes the ENGINE stream
    exactly as BaseEngine.spawn_object does (engine/base_engine.py:133-134: random_seed = generate_seed()) -- it would
    create a Bullet vehicle otherwise -- and add_policy records the policy seed the manager draws.  The vehicle's own
    parameter draw is the reference's BaseRunnable.__init__ -> sample_parameters (base_class/base_runnable.py:14-27,
    81-91) on the vehicle class's PARAMETER_SPACE with that seed.  The engine stream is advanced by the draws that
    precede the traffic manager in the reset chain (one: the agent's spawn_object, manager/agent_manager.py:43);
    the count is part of the fixture."""
    from types import SimpleNamespace
    from metadrive.base_class.base_runnable import BaseRunnable
    from metadrive.base_class.randomizable import Randomizable
    from metadrive.component.algorithm.blocks_prob_dist import PGBlockDistConfig
    from metadrive.manager.traffic_manager import PGTrafficManager
    from metadrive.utils.random_utils import get_np_random
    type_key = {"SVehicle": "s", "MVehicle": "m", "LVehicle": "l", "XLVehicle": "xl", "DefaultVehicle": "default"}
    ENGINE_DRAWS_BEFORE = 1
    specs = []
    for mode in ("trigger", "respawn", "hybrid"):
        specs += [dict(seed=s, mode=mode, density=0.1, inverse=False, lane_num=3, lane_width=3.5, blocks=3) for s in range(0, 32)]
    specs += [dict(seed=100 + s, mode="trigger", density=0.3, inverse=False, lane_num=3, lane_width=3.5, blocks=3) for s in range(8)]
    specs += [dict(seed=200 + s, mode="trigger", density=0.15, inverse=True, lane_num=3, lane_width=3.5, blocks=3) for s in range(8)]
    specs += [dict(seed=300 + s, mode=m, density=0.2, inverse=False, lane_num=2, lane_width=3.0, blocks=5)
              for s in range(4) for m in ("trigger", "respawn")]
    cases = []
    maps = {}       # one BIG run per (seed, shape): the traffic manager only reads the map
    for sp in specs:
        seed = sp["seed"]
        mkey = (seed, sp["lane_num"], sp["lane_width"], sp["blocks"])
        if mkey not in maps:
            maps[mkey] = build_reference_map(seed, sp["lane_num"], sp["lane_width"], 50, "block_num", sp["blocks"], PGBlockDistConfig)
            for f, td in maps[mkey][1].graph.items():
                for t, lanes in td.items():
                    for i, l in enumerate(lanes):
                        l.index = (f, t, i)
        big, net = maps[mkey]
        engine_stream = Randomizable(seed)                  # BaseEngine.seed(s) (engine/base_engine.py:546-553)
        for _ in range(ENGINE_DRAWS_BEFORE):
            engine_stream.generate_seed()
        spawned = []

        class Rec(PGTrafficManager):
            def __init__(self):                             # no BaseManager / engine machinery
                Randomizable.__init__(self, seed)           # BaseManager.seed(s) at reset
                self._traffic_vehicles = []
                self.block_triggered_vehicles = []
                self.mode = sp["mode"]
                self.random_traffic = False
                self.density = sp["density"]
                self.respawn_lanes = None

            def spawn_object(self, cls, vehicle_config=None, **kw):
                vseed = int(engine_stream.generate_seed())

                class Params(BaseRunnable):
                    PARAMETER_SPACE = cls.PARAMETER_SPACE
                prm = Params(random_seed=vseed)
                lane = net.get_lane(vehicle_config["spawn_lane_index"])
                long = float(vehicle_config["spawn_longitude"])
                pos = lane.position(long, 0.0)
                rec = dict(cls=type_key[cls.__name__], vehicle_seed=vseed, lane=list(vehicle_config["spawn_lane_index"]),
                           longitude=long, position=[float(pos[0]), float(pos[1])], heading=float(lane.heading_theta_at(long)),
                           params={k: float(v) for k, v in dict(prm.config).items()},
                           length=float(_cls_const(cls, "LENGTH")), width=float(_cls_const(cls, "WIDTH")),
                           mass=float(_cls_const(cls, "MASS")), front_wheelbase=float(_cls_const(cls, "FRONT_WHEELBASE")),
                           rear_wheelbase=float(_cls_const(cls, "REAR_WHEELBASE")))
                spawned.append(rec)
                return SimpleNamespace(id="v%d" % (len(spawned) - 1), name="v%d" % (len(spawned) - 1))

            def add_policy(self, object_id, policy_class, obj, policy_seed):
                spawned[int(object_id[1:])]["policy_seed"] = int(policy_seed)
                spawned[int(object_id[1:])]["policy"] = policy_class.__name__

        map_ = SimpleNamespace(blocks=big.blocks, road_network=net)
        fake_engine = SimpleNamespace(
            global_config=dict(traffic_mode=sp["mode"], random_traffic=False, traffic_density=sp["density"],
                               need_inverse_traffic=sp["inverse"], traffic_vehicle_config=dict(enable_reverse=False)),
            map_manager=SimpleNamespace(current_map=map_))
        Rec.engine = property(lambda self: fake_engine)
        mgr = Rec()
        PGTrafficManager.reset(mgr)
        blocks = [dict(trigger_road=[bv.trigger_road.start_node, bv.trigger_road.end_node], vehicles=[int(n[1:]) for n in bv.vehicles])
                  for bv in mgr.block_triggered_vehicles]
        cases.append(dict(spec=sp, map_blocks=[b.ID for b in big.blocks], vehicles=spawned, block_triggered=blocks,
                          driving_from_reset=[int(v.id[1:]) for v in mgr._traffic_vehicles],
                          respawn_lanes=[list(l.index) for l in (mgr.respawn_lanes or [])],
                          engine_stream_next=float(engine_stream.np_random.rand()), traffic_stream_next=float(mgr.np_random.rand())))
    dump("traffic_spawn.json", dict(engine_draws_before=ENGINE_DRAWS_BEFORE, cases=cases))


def section_fork_blocks():
    """InFork 'f' / OutFork 'F' (component/pgblock/fork.py): what happens when a block sequence names one.  Both raise as the first
    statement of _try_plug_into_previous_block (:27, :172); the fixture holds the exception the reference's own BIG run ends with."""
    from metadrive.component.algorithm.blocks_prob_dist import PGBlockDistConfig
    cases = []
    for seq in ("F", "f", "SF", "Cf"):
        try:
            build_reference_map(600, 3, 3.5, 50, "block_sequence", seq, PGBlockDistConfig)
            cases.append(dict(sequence=seq, raised=None))
        except Exception as ex:       # noqa: BLE001 -- the class and text are what is recorded
            cases.append(dict(sequence=seq, raised=type(ex).__name__, message=str(ex)))
    dump("fork_blocks.json", dict(cases=cases))


def section_pg_maps_v2():
    """PG topology with the reference's DEFAULT block distribution (BLOCK_TYPE_DISTRIBUTION_V2: curves,
    straights, in/out ramps, X and T intersections, roundabouts) -- what `MetaDriveEnv(map=3)` builds."""
    from metadrive.component.algorithm.blocks_prob_dist import PGBlockDistConfig
    cases = []
    specs = [(seed, 3, 3.5, 50, "block_num", 3) for seed in range(0, 24)]
    specs += [(200 + seed, 2, 3.0, 50, "block_num", 5) for seed in range(0, 4)]
    specs += [(300, 3, 3.5, 50, "block_sequence", "XTO"), (301, 3, 3.5, 50, "block_sequence", "rRX"),
              (302, 2, 3.5, 50, "block_sequence", "TXT")]
    _pg_maps_cases("pg_maps_v2.json", specs)


def section_pg_maps_v3():
    """Block sequences with the lane-count changing blocks Merge 'y' / Split 'Y' (zero probability in the default
    distribution, reachable through `map="..."`)."""
    specs = [(400, 3, 3.5, 50, "block_sequence", "yY"), (401, 3, 3.5, 50, "block_sequence", "SyY"),
             (402, 2, 3.5, 50, "block_sequence", "YyC"), (403, 3, 3.0, 50, "block_sequence", "yYX"),
             (404, 2, 3.5, 50, "block_sequence", "YS"), (405, 3, 3.5, 50, "block_sequence", "CyS")]
    _pg_maps_cases("pg_maps_v3.json", specs)


def section_pg_maps_v4():
    """Block sequences with the two-way single-lane block Bidirection 'B' (zero probability in the default distribution):
    what follows it plugs into its one-lane-per-direction socket."""
    specs = [(410, 3, 3.5, 50, "block_sequence", "SBS"), (411, 2, 3.5, 50, "block_sequence", "BC"),
             (412, 3, 3.5, 50, "block_sequence", "yBY"), (413, 3, 3.0, 50, "block_sequence", "BSC")]
    _pg_maps_cases("pg_maps_v4.json", specs)


def section_pg_maps_v5():
    """Block sequences with the ParkingLot block 'P' (zero probability in the default distribution; needs one lane per
    direction before it)."""
    specs = [(420, 1, 3.5, 50, "block_sequence", "P"), (421, 1, 3.5, 50, "block_sequence", "SP"),
             (422, 1, 3.0, 50, "block_sequence", "PS"), (423, 1, 3.5, 50, "block_sequence", "CPC")]
    _pg_maps_cases("pg_maps_v5.json", specs)


def section_pg_maps_v6():
    """Block sequences with the TollGate block '$' (zero probability in the default distribution): topology, line types, and
    the toll booths the block spawns -- lane, position, heading of every TollGateBuilding, in spawn order (the engine's
    spawn_object is recorded, not executed: it needs Bullet)."""
    specs = [(520, 3, 3.5, 50, "block_sequence", "$"), (521, 3, 3.5, 50, "block_sequence", "S$S"),
             (522, 2, 3.0, 50, "block_sequence", "$C"), (523, 4, 3.5, 50, "block_sequence", "C$"),
             (524, 3, 3.5, 50, "block_sequence", "$$")]
    _pg_maps_cases("pg_maps_v6.json", specs, record_buildings=True)


def _pg_maps_cases(fname, specs, record_buildings=False):
    from metadrive.component.algorithm.blocks_prob_dist import PGBlockDistConfig
    cases = []
    for seed, lane_num, lane_width, exit_length, method, parameter in specs:
        spawned = []
        if record_buildings:
            import metadrive.component.pgblock.tollgate as tg

            def spawn_object(cls, lane=None, position=None, heading_theta=None, **kw):
                spawned.append((cls.__name__, lane, [float(position[0]), float(position[1])], float(heading_theta)))
                return MagicMock()
            tg.get_engine = lambda: MagicMock(spawn_object=spawn_object)
        big, net = build_reference_map(seed, lane_num, lane_width, exit_length, method, parameter, PGBlockDistConfig)
        for f, td in net.graph.items():
            for t, lanes in td.items():
                for i, l in enumerate(lanes):
                    l.index = (f, t, i)
        roads = []
        for f, td in net.graph.items():
            for t, lanes in td.items():
                roads.append(dict(start=f, end=t, lanes=[lane_record(l) for l in lanes]))
        blocks = [dict(name=b.name, config={k: float(v) for k, v in dict(b.get_config()).items()},
                       trials=int(b.number_of_sample_trial),
                       sockets=[[s.positive_road.start_node, s.positive_road.end_node] for s in b.get_socket_list()],
                       respawn_roads=[[r.start_node, r.end_node] for r in b.get_respawn_roads()])
                  for b in big.blocks]
        spawn = []
        for b in big.blocks[1:]:
            spawn.append([[list(l.index) for l in lanes] for lanes in b.get_intermediate_spawn_lanes()])
        case = dict(seed=seed, lane_num=lane_num, lane_width=lane_width, exit_length=exit_length,
                    method=method, parameter=parameter, blocks=blocks, roads=roads, spawn_lanes=spawn)
        if record_buildings:
            from metadrive.component.buildings.tollgate_building import TollGateBuilding
            case["buildings"] = [dict(cls=c, lane=list(l.index), position=p, heading=h, width=float(l.width),
                                      length=float(TollGateBuilding.BUILDING_LENGTH)) for c, l, p, h in spawned]
        cases.append(case)
    dump(fname, dict(cases=cases))


def scenario_export_case():
    """The recorded episode both the generator and tests/test_scenario_export.py export (oracle-driven, CPU only)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import oracle_binding as ob
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.scenario_export import tracks_to_scenarios
    E, T = 3, 80
    host = HostScene(make_config(dict(num_envs=E, num_scenarios=E, start_seed=20, traffic_density=0.2, accident_prob=1.0,
                                      traffic_mode="respawn", horizon=1000, auto_reset=False)))
    o = ob.OracleWorld(host)
    o.reset()
    acts = [np.tile(np.array([0.03 * math.sin(0.07 * t), 0.6], np.float32), (E, 1, 1)) for t in range(T)]
    return tracks_to_scenarios(ob.record_episode(o, acts), host)


def section_scenario_export():
    """Episodes exported by metadrive_ped_amd.scenario_export, judged by the REFERENCE's own
    ScenarioDescription.sanity_check(valid_check=True) and summarised by its update_summaries
    (scenario/scenario_description.py:199-257, :417-437).  The fixture holds the reference's summaries of our export."""
    from metadrive.scenario.scenario_description import ScenarioDescription as SD
    out = []
    for sc in scenario_export_case():
        sc["metadata"].pop("object_summary")
        sc["metadata"].pop("number_summary")
        SD.sanity_check(sc, check_self_type=True, valid_check=True)
        sd = SD(sc)
        SD.update_summaries(sd)
        ns = dict(sd["metadata"]["number_summary"])
        ns.pop("map_height_diff")          # -inf for a flat map without road-line features; not JSON
        for k in ("object_types", "num_traffic_light_types"):
            ns[k] = sorted(ns[k])
        ns["num_moving_objects_each_type"] = dict(ns["num_moving_objects_each_type"])
        out.append(dict(id=sc["id"], length=sc["length"], sdc_moving_dist=float(SD.sdc_moving_dist(sd)),
                        object_summary=sd["metadata"]["object_summary"], number_summary=ns))
    dump("scenario_export.json", dict(accepted_by_reference_sanity_check=True, scenarios=out))


SECTIONS = OrderedDict(pg_maps=section_pg_maps, lanes=section_lanes, utils=section_utils, agent_step=section_agent_step, objects=section_objects, roundabout=section_roundabout, idm=section_idm, idm_policy=section_idm_policy, scenario=section_scenario, pg_maps_v2=section_pg_maps_v2, pg_maps_v6=section_pg_maps_v6, ma_intersection=section_ma_intersection, ma_bottleneck=section_ma_bottleneck, pg_maps_v3=section_pg_maps_v3,
                       scenario_export=section_scenario_export, ma_bidirection=section_ma_bidirection, ma_tollgate=section_ma_tollgate, ma_parking_lot=section_ma_parking_lot, scenario_lines=section_scenario_lines, others=section_others, pg_maps_v4=section_pg_maps_v4, pg_maps_v5=section_pg_maps_v5, scenario_spawn=section_scenario_spawn,
                       traffic_spawn=section_traffic_spawn, fork_blocks=section_fork_blocks, ma_tinyinter=section_ma_tinyinter, ma_racing=section_ma_racing, ma_racing_rules=section_ma_racing_rules)

if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    names = sys.argv[1:] or list(SECTIONS)
    for n in names:
        SECTIONS[n]()

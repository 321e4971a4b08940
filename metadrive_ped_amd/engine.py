"""BatchedEngine: the host side of the batched step().  Owns every table and state array as
PyTorch-ROCm tensors (device memory + streams are torch's job; arithmetic is the HIP library's) and
drives the C-ABI.  Plays the role of BaseEngine + the manager chain for E lock-stepped worlds
(metadrive/engine/base_engine.py:306-478): reset() builds the scenes on the host with numpy
RandomState streams, step() is ONE md_step launch.

Sharding: an engine owns envs [env_seed_offset, env_seed_offset + num_envs) of the global batch;
env g uses scenario seed start_seed + g % num_scenarios, so results do not depend on how many
GPUs the batch is split over.
"""
import copy
import ctypes as C
import os
from collections import OrderedDict

import numpy as np

from metadrive_ped_amd import abi, hostpool
from metadrive_ped_amd.mapgen.pg import PGMap
from metadrive_ped_amd.mapgen.tables import MapTables, WorldTables, beam_table
from metadrive_ped_amd.scene import EnvScene

STATE_ARRAY_SPECS = None  # filled below


# One wave per env beats the 4-wave workgroup when a LARGE batch shares few distinct maps: the lane / grid tables stay in L2, so
# the wave's serial chain of map reads is short, and all envs are resident at once.  Measured on the MI355X
# (tools/locality_probe.py, profiles/r03_step_kernel_by_maps.txt): 4096 envs on 1 / 8 / 16 / 32 maps 70-72 us against 83-86 us;
# even at 64 maps, behind at 512 (82.6 vs 79.2) and with a map per env (94 vs 87); at 1024 envs and below the workgroup kernel's
# shorter chain wins whatever the maps (38 vs 60 us).  The reference's default is num_scenarios = 1.
WAVE_KERNEL_MAX_MAPS = 32
WAVE_KERNEL_MIN_ENVS = 3072


def pick_step_kernel(cfg, n_maps):
    """config["step_kernel"] -> "wg" | "wave".  "auto": by the batch size and the number of distinct maps; MD_STEP_KERNEL in the
    environment overrides "auto" only (A/B runs) -- read HERE, once per engine, never inside the library."""
    want = cfg.get("step_kernel", "auto")
    if cfg.get("is_multi_agent") or cfg.get("scenario_mode"):
        return "wg"
    if want == "auto":
        env = os.environ.get("MD_STEP_KERNEL", "")
        if env in ("wg", "wave"):
            return env
        return "wave" if (n_maps <= WAVE_KERNEL_MAX_MAPS and int(cfg["num_envs"]) >= WAVE_KERNEL_MIN_ENVS) else "wg"
    return want


def _build_marl(cfg, scene_cfg, uniq):
    """Multi-agent maps (roundabout, intersection): one shared map, one scene per env seed."""
    from metadrive_ped_amd.mapgen.pg import (MABidirectionMap, MABottleneckMap, MAIntersectionMap, MAParkingLotMap, MARoundaboutMap,
                                             MATollGateMap)
    from metadrive_ped_amd.marl import FIXED_DESTINATION, SPAWN_ROADS, RoundaboutScene
    from metadrive_ped_amd.mapgen.tables import spawn_tables
    mc = cfg["map_config"]
    kind = cfg["marl_map"]
    if kind not in SPAWN_ROADS:
        raise NotImplementedError("multi-agent map {!r} is not built (built: {})".format(kind, sorted(SPAWN_ROADS)))
    if kind in ("bottleneck", "bidirection"):
        pg = (MABottleneckMap if kind == "bottleneck" else MABidirectionMap)(lane_num=mc["lane_num"], lane_width=mc["lane_width"], exit_length=mc["exit_length"],
                             neck_lane_num=mc["neck_lane_num"], neck_length=mc["neck_length"])
    elif kind == "parking_lot":
        pg = MAParkingLotMap(lane_num=mc["lane_num"], lane_width=mc["lane_width"], exit_length=mc["exit_length"],
                             parking_space_num=cfg["parking_space_num"])
    elif kind == "racing":
        from metadrive_ped_amd.mapgen.pg import RacingMap
        pg = RacingMap(lane_num=mc["lane_num"], lane_width=mc["lane_width"], exit_length=mc["exit_length"])
    elif kind == "tollgate":
        pg = MATollGateMap(lane_num=mc["lane_num"], lane_width=mc["lane_width"], exit_length=mc["exit_length"],
                           toll_lane_num=mc["toll_lane_num"], toll_length=mc["toll_length"])
    else:
        cls = dict(roundabout=MARoundaboutMap, intersection=MAIntersectionMap)[kind]
        kw = dict(radius=mc.get("radius")) if kind == "intersection" else {}
        pg = cls(lane_num=mc["lane_num"], lane_width=mc["lane_width"], exit_length=mc["exit_length"], **kw)
    mt = MapTables(pg)
    sc_cfg = dict(scene_cfg, exit_length=mc["exit_length"])
    fixed = FIXED_DESTINATION[kind]
    roads = _user_spawn_roads(cfg, mt) or SPAWN_ROADS[kind]
    parking, dests = None, None
    if kind == "parking_lot":
        from metadrive_ped_amd.marl import PARKING_IN_ROADS, parking_lot_roads
        if cfg.get("spawn_roads"):
            raise NotImplementedError("spawn_roads: the parking-lot env fills them itself (marl_parking_lot.py:194-203)")
        roads, dests = parking_lot_roads(cfg["parking_space_num"])
        assert [tuple(r) for r in pg.parking_space] == [(d[:-2] + "1_", d) for d in dests[:cfg["parking_space_num"]]]
        parking = (len(PARKING_IN_ROADS), cfg["parking_space_num"], dests)
    # MAIntersectionSpawnManager(disable_u_turn = lane_num < 2) (marl_intersection.py:73-85, :104): on the one-lane intersection a
    # vehicle is never sent back out of the arm it came in by
    no_u_turn = kind == "intersection" and mc["lane_num"] < 2
    sc_cfg["exclude_own_road"] = no_u_turn
    scenes = {s: RoundaboutScene(s, mt, sc_cfg, roads, fixed, parking) for s in uniq}
    return mt, scenes, spawn_tables(mt, roads, mc["lane_num"], fixed, dests, exclude_own_road=no_u_turn)


def _user_spawn_roads(cfg, mt):
    """config["spawn_roads"] (multi_agent_metadrive.py:27,84-92): the user's own list of (start node, end node) roads."""
    if not cfg.get("spawn_roads"):
        return None
    roads = [tuple(r) for r in cfg["spawn_roads"]]
    for r in roads:
        if len(r) != 2 or r not in mt.road_id:
            raise ValueError("spawn_roads: {!r} is not a road of this map".format(r))
    return roads


def _build_one_marl_pg(job):
    """MultiAgentMetaDrive on procedurally generated maps (envs/marl_envs/multi_agent_metadrive.py:12-61): one PG map per
    scenario seed like the single-agent env, agents spawn on the first block's exit road, destination = the far end."""
    from metadrive_ped_amd.mapgen.tables import spawn_tables
    from metadrive_ped_amd.marl import PG_SPAWN_ROADS, RoundaboutScene
    s, mc, dist, scene_cfg = job
    mt = copy.copy(_map_tables_for(s, mc, dist))   # the cached tables stay as generated
    roads = [tuple(r) for r in scene_cfg["spawn_roads"]] if scene_cfg.get("spawn_roads") else PG_SPAWN_ROADS
    mt.respawn = spawn_tables(mt, roads, mc["lane_num"], fixed_destination=True)   # stacked per map by WorldTables
    sc_cfg = dict(scene_cfg, exit_length=mc["exit_length"])
    return mt, RoundaboutScene(s, mt, sc_cfg, roads, True)


# Maps built by THIS process, by what they were generated from: a rebuild that only changes the scenes on them (random_traffic: new
# traffic at every env.reset(); another traffic density) does not generate them again.  The build workers are persistent and
# build_all keeps a job on the same worker (sticky), so their caches hit as well.
_MAP_CACHE = OrderedDict()
_MAP_CACHE_MAX = 512
MAPS_GENERATED = [0]          # what the cache did not have (tests read it)


def _map_tables_for(s, mc, dist):
    import pickle
    key = pickle.dumps((s, sorted(mc.items(), key=lambda kv: str(kv[0])), dist), protocol=4)
    mt = _MAP_CACHE.get(key)
    if mt is not None:
        _MAP_CACHE.move_to_end(key)
        return mt
    pg = PGMap(s, lane_num=mc["lane_num"], lane_width=mc["lane_width"], exit_length=mc["exit_length"],
               generate_type=mc["type"], generate_config=mc["config"], block_dist=dist)
    mt = MapTables(pg)
    MAPS_GENERATED[0] += 1
    _MAP_CACHE[key] = mt
    while len(_MAP_CACHE) > _MAP_CACHE_MAX:
        _MAP_CACHE.popitem(last=False)
    return mt


def _build_one(job):
    """One scenario seed -> (MapTables, EnvScene).  Module-level so that a fork pool can run it."""
    s, mc, dist, scene_cfg = job
    if scene_cfg.get("random_lane_width") or scene_cfg.get("random_lane_num"):
        # PGMapManager.add_random_to_map (manager/pg_map_manager.py:68-74): the map manager's stream, re-seeded with
        # the scenario index at every reset; width first, then the lane count
        from metadrive_ped_amd.rng import get_np_random
        rng = get_np_random(s)
        mc = dict(mc)
        if scene_cfg.get("random_lane_width"):
            mc["lane_width"] = float(rng.rand() * (4.5 - 3.0) + 3.0)     # MAX_LANE_WIDTH / MIN_LANE_WIDTH (base_map.py:38-39)
        if scene_cfg.get("random_lane_num"):
            mc["lane_num"] = int(rng.randint(2, 3 + 1))                   # MIN_LANE_NUM .. MAX_LANE_NUM (base_map.py:40-41)
    mt = _map_tables_for(s, mc, dist)
    if scene_cfg["traffic_mode"] in ("respawn", "hybrid") and abs(scene_cfg["traffic_density"]) >= 1e-2:
        from metadrive_ped_amd.mapgen.tables import respawn_tables
        mt = copy.copy(mt)                      # the cached tables stay as generated
        mt.respawn = respawn_tables(mt, s)
    return mt, EnvScene(s, mt, scene_cfg)


class HostScene:
    """Host (numpy) copy of everything: world tables + reset snapshot.  Also what the tests hand to
    the CPU oracle."""
    def __init__(self, cfg):
        self.cfg = cfg
        E = cfg["num_envs"]
        cap = cfg["mover_capacity"] or abi.MD_MAX_CAP  # 0 = auto: build with the maximum, trim below
        A = cfg["num_agents"]
        self.E, self.cap, self.A = E, cap, A
        self.n_beams = int(cfg["vehicle_config"]["lidar"]["num_lasers"]) if cfg["vehicle_config"]["lidar"]["distance"] > 0 else 0
        vc = cfg["vehicle_config"]
        self.n_side = int(vc["side_detector"]["num_lasers"]) if vc["side_detector"]["distance"] > 0 else 0
        self.n_ll = int(vc["lane_line_detector"]["num_lasers"]) if vc["lane_line_detector"]["distance"] > 0 else 0
        self.obs_base = 2 if cfg["random_agent_model"] else 0            # [length, width] lead the state dims
        self.tollgate = bool(cfg["is_multi_agent"]) and cfg["marl_map"] == "tollgate"
        # 19 with everything off; the tollgate env's state observation has no navigation dims (marl_tollgate.py:62-74)
        self.state_dim = self.obs_base + (self.n_side or 2) + 6 + (self.n_ll or 1) + (0 if self.tollgate else 10)
        # "others" block only exists with the lidar on (obs/state_obs.py:172-183)
        self.num_others = int(vc["lidar"]["num_others"]) if self.n_beams > 0 else 0
        self.add_others_navi = bool(vc["lidar"]["add_others_navi"]) and self.num_others > 0
        self.others_dim = self.num_others * (8 if self.add_others_navi else 4)
        self.obs_dim = self.state_dim + self.others_dim + self.n_beams + (2 if self.tollgate else 0)   # + the two toll dims
        mc = cfg["map_config"]
        seeds = [cfg["start_seed"] + ((cfg["env_seed_offset"] + e) % cfg["num_scenarios"]) for e in range(E)]
        self.seeds = seeds
        uniq = sorted(set(seeds))
        map_of_seed = {}
        tables, scenes = [], {}
        scene_cfg = dict(cap=cap, agents_per_env=A, physics_world_step_size=cfg["physics_world_step_size"],
                         random_spawn_lane_index=cfg["random_spawn_lane_index"],
                         spawn_lane_index=cfg["agent_configs"]["default_agent"]["spawn_lane_index"],
                         agent_vehicle_model=cfg["vehicle_config"]["vehicle_model"],
                         spawn_longitude=cfg["vehicle_config"]["spawn_longitude"],
                         spawn_lateral=cfg["vehicle_config"]["spawn_lateral"],
                         agent_size_mass={k: cfg["vehicle_config"][k] for k in ("width", "length", "height", "mass")},
                         spawn_velocity=cfg["vehicle_config"]["spawn_velocity"],
                         spawn_velocity_car_frame=cfg["vehicle_config"]["spawn_velocity_car_frame"],
                         traffic_density=cfg["traffic_density"], traffic_mode=cfg["traffic_mode"],
                         accident_prob=cfg["accident_prob"], static_traffic_object=cfg["static_traffic_object"],
                         need_inverse_traffic=cfg["need_inverse_traffic"], random_lane_width=cfg["random_lane_width"],
                         random_lane_num=cfg["random_lane_num"], random_agent_model=cfg["random_agent_model"],
                         random_dynamics=cfg["random_dynamics"], initial_agents=cfg["initial_agents"],
                         agent_policy=cfg["agent_policy"], spawn_roads=cfg["spawn_roads"],
                         random_traffic=cfg["random_traffic"], traffic_epoch=cfg.get("traffic_epoch", 0),
                         destination=cfg["vehicle_config"]["destination"])
        self.spawn = None
        if not cfg["is_multi_agent"]:
            scene_cfg["cap"] = abi.MD_MAX_CAP
        jobs = [(s, dict(mc), cfg["block_dist_config"], scene_cfg) for s in uniq]
        # reset-time host work goes to the persistent build workers (metadrive_ped_amd/hostpool.py): they are started before
        # this process touches the GPU and kept; a process that already has a GPU context and no workers builds serially
        # (it must not fork).  build_workers = 1 forces the serial path.
        shared_map = cfg["is_multi_agent"] and cfg["marl_map"] != "pg"
        build_fn = _build_one
        if cfg["is_multi_agent"] and not cfg["mover_capacity"]:
            scene_cfg["cap"] = cap = A
            if self.tollgate:     # + the toll booths (one on every odd lane of both directions), slots in multiples of 8
                cap = min(abi.MD_MAX_CAP, (A + 2 * (mc["toll_lane_num"] // 2) + 7) // 8 * 8)
                scene_cfg["cap"] = cap
            self.cap = cap
            jobs = [(s, dict(mc), cfg["block_dist_config"], scene_cfg) for s in uniq]
        if cfg["is_multi_agent"] and not shared_map:
            build_fn = _build_one_marl_pg
            self.spawn = dict(n_dest=1)          # per-map spawn tables travel with the map tables
        if shared_map:
            mt, marl_scenes, self.spawn = _build_marl(cfg, scene_cfg, uniq)
            built = [(mt, marl_scenes[s]) for s in uniq]
        else:
            built = hostpool.build_all(build_fn, jobs, workers=int(cfg.get("build_workers", 0)), cache=bool(cfg.get("build_cache", False)), sticky=True)
        for s, (mt, sc) in zip(uniq, built):
            if shared_map:
                map_of_seed[s] = 0
                if not tables:
                    tables.append(mt)
            else:
                map_of_seed[s] = len(tables)
                tables.append(mt)
            scenes[s] = sc
        if not cfg["is_multi_agent"]:
            # single-agent scenes are always generated with the maximum slot count and cut to size here (vehicles keep their
            # low slots, props the top ones: the same arrays as a build at that size), so that one built scene serves every
            # capacity -- the build memo keys on the job
            need = max(A + sc.n_traffic + sc.n_props for sc in scenes.values())
            cap = cfg["mover_capacity"] or min(abi.MD_MAX_CAP, max(8, (need + 7) // 8 * 8))
            if need > cap:
                raise ValueError("more than cap={} movers in an env ({}); raise `mover_capacity`".format(cap, need))
            for sc in scenes.values():
                sc.trim(cap)
            self.cap = cap
        self.map_tables = tables
        self.scenes = scenes
        env_map = [map_of_seed[s] for s in seeds]
        self.world = WorldTables(tables, env_map, beam_table(self.n_beams))
        if self.spawn is not None and shared_map:
            a = self.world.arrays
            a["spawn_off"] = np.asarray([0, len(self.spawn["spawn_lane"])], np.int32)
            for k in ("spawn_place", "spawn_lane", "spawn_route", "spawn_route_meta"):
                a[k] = np.ascontiguousarray(self.spawn[k])
        N = E * cap

        def stack(field):
            return np.concatenate([getattr(scenes[s], field) for s in seeds], axis=0)

        st = {}
        st["shape0"] = stack("shape")
        st["dyn0"] = stack("dyn")
        st["nav0"] = stack("nav")
        st["pid0"] = stack("pid")
        st["param"] = stack("param")
        st["route_nodes"] = stack("route_nodes")
        st["route_roads"] = stack("route_roads")
        st["final_lane"] = stack("final_lane")
        st["idm_rand"] = stack("idm_rand")
        # MdNav.road0 / road1: the road ids under the two route cursors, kept beside them (ABI v7) so that the per-step
        # logic never indexes the route arrays
        rows = np.arange(N)
        rr = st["route_roads"].reshape(N, abi.MD_ROUTE_LEN)
        st["nav0"]["road0"] = rr[rows, np.clip(st["nav0"]["ck0"], 0, abi.MD_ROUTE_LEN - 1)]
        st["nav0"]["road1"] = rr[rows, np.clip(st["nav0"]["ck1"], 0, abi.MD_ROUTE_LEN - 1)]
        st["shape"] = st["shape0"].copy()
        st["dyn"] = st["dyn0"].copy()
        st["nav"] = st["nav0"].copy()
        st["pid"] = st["pid0"].copy()
        st["action"] = np.zeros((N, 2), np.float32)
        st["flags"] = np.zeros(N, np.uint32)
        st["obs"] = np.zeros((E * A, self.obs_dim), np.float32)
        st["reward"] = np.zeros(E * A, np.float32)
        st["cost"] = np.zeros(E * A, np.float32)
        st["step_info"] = np.zeros((E * A, 8), np.float32)
        st["done_out"] = np.zeros((E * A, 4), np.uint8)          # (terminated, truncated, flag word lo / hi) straight from the kernel
        st["need_reset"] = np.ones(E, np.int32)
        self.traffic_respawns = "spawn_off" in self.world.arrays and not cfg["is_multi_agent"]
        if cfg["is_multi_agent"] or self.traffic_respawns:
            # respawns (agents in MARL, traffic in the respawn / hybrid modes) rewrite routes and draw random numbers
            st["route_nodes0"] = st["route_nodes"].copy()
            st["route_roads0"] = st["route_roads"].copy()
            st["final_lane0"] = st["final_lane"].copy()
            # xorshift32 needs a non-zero state; derive it from the env's scenario seed
            st["rng"] = np.asarray([((s * 2654435761) ^ 0x9E3779B9) & 0xFFFFFFFF or 1 for s in seeds], np.uint32)
        if cfg["is_multi_agent"] and cfg["random_agent_model"]:
            from metadrive_ped_amd.marl import vehicle_class_table
            st["param0"] = st["param"].copy()
            self.world.arrays["vclass"] = vehicle_class_table(cfg["physics_world_step_size"])
        if cfg["is_multi_agent"]:
            st["env_steps"] = np.zeros(E, np.int32)
            st["agent_id"] = np.tile(np.arange(cap, dtype=np.int32), E)
            st["next_agent_id"] = np.full(E, cfg["initial_agents"] or A, np.int32)   # names agent0 .. agent{n-1} are taken
        if self.num_others > 0:
            st["detected"] = np.zeros((E * A, 2), np.uint64)
        if cfg["is_multi_agent"] and cfg["marl_map"] == "racing":
            st["idle_ring"] = np.zeros((E * A, abi.MD_IDLE_WINDOW), np.float32)     # movement_between_steps of every agent
        self.state = st
        self.md_config = make_md_config(cfg, E, A, cap, self.n_beams)
        self.md_config.n_side, self.md_config.n_lane_line = self.n_side, self.n_ll
        self.md_config.num_others, self.md_config.add_others_navi = self.num_others, int(self.add_others_navi)
        self.md_config.random_agent_model = int(bool(cfg["random_agent_model"]))
        self.md_config.agent_idm = int(cfg["agent_policy"] == "IDMPolicy")
        self.md_config.enable_reverse = int(bool(cfg["vehicle_config"]["enable_reverse"]))
        self.step_kernel = pick_step_kernel(cfg, len(tables))
        self.md_config.step_kernel = {"wg": 0, "wave": 1}[self.step_kernel]
        self.md_config.obs_dim = self.obs_dim
        # detector beam fans start 90 deg off the heading (SideDetector.__init__, distance_detector.py:197)
        self.side_beams = beam_table(self.n_side, np.pi / 2) if self.n_side else None
        self.ll_beams = beam_table(self.n_ll, np.pi / 2) if self.n_ll else None

    def clone_state(self):
        return {k: v.copy() for k, v in self.state.items()}


def make_md_config(cfg, E, A, cap, n_beams):
    k = abi.MdConfig()
    k.struct_size = C.sizeof(abi.MdConfig)
    k.n_envs, k.agents_per_env, k.cap = E, A, cap
    k.n_beams, k.obs_dim = n_beams, 19 + n_beams
    k.substeps = int(cfg["decision_repeat"])
    k.horizon = int(cfg["horizon"]) if cfg["horizon"] else 0
    k.dt = float(cfg["physics_world_step_size"])
    k.lidar_range = float(cfg["vehicle_config"]["lidar"]["distance"])
    for name in ("success_reward", "out_of_road_penalty", "crash_vehicle_penalty", "crash_object_penalty",
                 "driving_reward", "speed_reward", "crash_vehicle_cost", "crash_object_cost", "out_of_road_cost"):
        setattr(k, name, float(cfg[name]))
    for name in ("use_lateral_reward", "out_of_route_done", "on_continuous_line_done", "crash_vehicle_done",
                 "crash_object_done", "crash_human_done", "truncate_as_terminate", "enable_idm_lane_change",
                 "auto_reset"):
        setattr(k, name, int(bool(cfg[name])))
    # density ~ 0: PGTrafficManager.reset returns before any mode-specific set-up (traffic_manager.py:62-63)
    k.traffic_mode = {"trigger": 0, "respawn": 1, "hybrid": 2, "replay": 3}[cfg["traffic_mode"]] \
        if abs(cfg["traffic_density"]) >= 1e-2 else 0
    k.max_lane_width = 4.5      # BaseMap.MAX_LANE_WIDTH (component/map/base_map.py:38)
    k.total_width = (3 + 1) * 4.5  # (MAX_LANE_NUM + 1) * MAX_LANE_WIDTH (obs/state_obs.py:92)
    k.curve_radius_max = 60.0   # BlockParameterSpace.CURVE radius max
    k.curve_angle_max = 135.0
    k.is_multi_agent = int(bool(cfg["is_multi_agent"]))
    k.delay_done = int(cfg["delay_done"])
    k.allow_respawn = int(bool(cfg["allow_respawn"]))
    k.crash_done = int(bool(cfg["crash_done"]))
    k.out_of_road_done = int(bool(cfg["out_of_road_done"]))
    if cfg["is_multi_agent"] and cfg["marl_map"] == "parking_lot":
        k.ma_kind = abi.MA_PARKING_LOT
        k.n_parking = int(cfg["parking_space_num"])
    if cfg["is_multi_agent"] and cfg["marl_map"] == "racing":
        k.ma_kind = abi.MA_RACING
        k.crash_sidewalk_penalty, k.idle_penalty = float(cfg["crash_sidewalk_penalty"]), float(cfg["idle_penalty"])
        k.idle_done, k.crash_sidewalk_done = int(bool(cfg["idle_done"])), int(bool(cfg["crash_sidewalk_done"]))
    if cfg["is_multi_agent"] and cfg["marl_map"] == "tollgate":
        k.ma_kind = abi.MA_TOLLGATE
        k.min_pass_steps = int(cfg["vehicle_config"]["min_pass_steps"])
        k.overspeed_penalty = float(cfg["overspeed_penalty"])
        k.on_continuous_line_done = int(bool(cfg["cross_yellow_line_done"]))   # _is_out_of_road (marl_tollgate.py:241-247)
    return k


def make_structs(world_arrays, state_arrays, md_config, n_maps, n_envs, ptr_of):
    w = abi.MdWorld()
    w.n_maps, w.n_envs = n_maps, n_envs
    abi.fill_struct(w, abi.WORLD_FIELDS, world_arrays, ptr_of)
    lane_off = np.asarray(world_arrays["lane_off_host"])
    road_off = np.asarray(world_arrays["road_off_host"])
    w.max_lanes = int(np.diff(lane_off).max())
    w.max_roads = int(np.diff(road_off).max())
    w.n_dest = int(world_arrays.get("n_dest_host", 0))
    w.n_vclass = int(world_arrays.get("n_vclass_host", 0))
    s = abi.MdState()
    abi.fill_struct(s, abi.STATE_FIELDS, state_arrays, ptr_of)
    return w, s, md_config


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL_CTX = _NullCtx()


class BatchedEngine:
    def __init__(self, cfg, host=None):
        import torch
        from metadrive_ped_amd import _lib
        self.torch = torch
        self.lib = _lib.load()
        self._check = _lib.check
        self.cfg = cfg
        self.device = torch.device(cfg["device"])
        self._dev_index = self.device.index if self.device.index is not None else 0
        if self.device.type != "cuda":
            raise _lib.MdStepError("BatchedEngine needs a ROCm device (config['device']={!r}); there is no CPU "
                                   "fallback".format(cfg["device"]))
        self.host = host
        self._noise_gen = None
        self._rec = None
        self._tracks = None
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
            self._dev_index = self.device.index
        self.build()

    # -- upload helpers ---------------------------------------------------------------------------
    def _to_dev(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1))
        return t.to(self.device)

    DRAW_ARRAYS = ("shape0", "dyn0", "nav0", "pid0", "param", "route_nodes", "route_roads", "final_lane", "idm_rand")
    DRAW_EPOCH_STRIDE = 4099          # traffic_epoch of draw k = the env's epoch + k * stride

    def n_traffic_draws(self):
        """random_traffic with envs that reset themselves: how many traffic draws are staged on the device (md_swap_draw)."""
        c = self.cfg
        if c.get("scenario_mode") or c["is_multi_agent"] or not (c["random_traffic"] and c["auto_reset"]):
            return 1
        return max(1, int(c.get("traffic_draws", 1)))

    def _draw_cfg(self, k, cap=None):
        c = dict(self.host.cfg if self.host is not None else self.cfg)
        c["traffic_epoch"] = int(c.get("traffic_epoch", 0)) + k * self.DRAW_EPOCH_STRIDE
        if cap is not None:
            c["mover_capacity"] = cap
        return c

    def build(self):
        """(Re)generate maps + scenes on the host and upload.  BaseEnv.reset's map/agent/traffic managers."""
        torch = self.torch
        K = self.n_traffic_draws()
        draws = None
        if self.host is None:
            self.host = HostScene(self.cfg)
            if K > 1 and not self.cfg["mover_capacity"]:
                # the draws share one capacity: the largest any of them needs (the maps are cached per process, the scenes are cheap)
                draws = [HostScene(self._draw_cfg(k)) for k in range(1, K)]
                need = max([self.host.cap] + [d.cap for d in draws])
                if self.host.cap != need:
                    self.host = HostScene(dict(self.cfg, mover_capacity=need))
                draws = [d if d.cap == need else HostScene(self._draw_cfg(k + 1, need)) for k, d in enumerate(draws)]
        if K > 1 and draws is None:      # a host handed in, or a fixed capacity: every draw must fit it (ValueError names the capacity)
            draws = [HostScene(self._draw_cfg(k, self.host.cap)) for k in range(1, K)]
        self.draw_hosts_ = [self.host] + (draws or [])
        h = self.host
        self.E, self.A, self.cap = h.E, h.A, h.cap
        self.n_beams, self.obs_dim = h.n_beams, h.obs_dim
        self.world_dev = {k: self._to_dev(v) for k, v in h.world.arrays.items()}
        self.state_dev = {k: self._to_dev(v) for k, v in h.state.items()}
        self._pack_step_outputs()
        ptr = lambda t: t.data_ptr()
        wd = dict(self.world_dev)
        wd["lane_off_host"], wd["road_off_host"] = h.world.arrays["lane_off"], h.world.arrays["road_off"]
        wd["n_dest_host"] = h.spawn["n_dest"] if h.spawn is not None else (1 if h.traffic_respawns else 0)
        wd["n_vclass_host"] = len(h.world.arrays["vclass"]) if "vclass" in h.world.arrays else 0
        self._side_beams = self._to_dev(h.side_beams) if h.side_beams is not None else None
        self._ll_beams = self._to_dev(h.ll_beams) if h.ll_beams is not None else None
        # scenario mode: md_step runs the side / lane-line detectors itself (MdWorld.side_beam_cs / ll_beam_cs) on waves that
        # idle while the agent is observed; the other modes call md_line_detector after md_step
        self._fused_detectors = bool(self.cfg.get("scenario_mode")) and (h.n_side > 0 or h.n_ll > 0)
        if self._fused_detectors:
            vc = self.cfg["vehicle_config"]
            if self._side_beams is not None:
                wd["side_beam_cs"] = self._side_beams
            if self._ll_beams is not None:
                wd["ll_beam_cs"] = self._ll_beams
            h.md_config.side_range = float(vc["side_detector"]["distance"])
            h.md_config.ll_range = float(vc["lane_line_detector"]["distance"])
            h.md_config.side_mask, h.md_config.ll_mask = self.SIDE_MASK, self.LANE_LINE_MASK
        self.w, self.s, self.k = make_structs(wd, self.state_dev, h.md_config, h.world.n_maps, h.E, ptr)
        # random_traffic: the staged draws (md_swap_draw after every step hands an env that finished its episode the next one)
        self._staged = None
        if K > 1:
            hosts = self.draw_hosts_
            names = [k for k in self.DRAW_ARRAYS if k in h.state]
            self._staged_dev = {k: self._to_dev(np.stack([np.ascontiguousarray(x.state[k]).view(np.uint8).reshape(-1) for x in hosts])) for k in names}
            self._staged = abi.MdState()
            abi.fill_struct(self._staged, abi.STATE_FIELDS, self._staged_dev, ptr)
            self.draw_idx = torch.zeros(self.E, dtype=torch.int32, device=self.device)
        sd = self.state_dev
        # typed views for the env API
        self.obs = sd["obs"].view(torch.float32).view(self.E, self.A, self.obs_dim)
        self.reward = sd["reward"].view(torch.float32).view(self.E, self.A)
        self.cost = sd["cost"].view(torch.float32).view(self.E, self.A)
        self.flags = sd["flags"].view(torch.int32).view(self.E, self.cap)
        self.action = sd["action"].view(torch.float32).view(self.E, self.cap, 2)
        self.step_info = sd["step_info"].view(torch.float32).view(self.E, self.A, 8)
        self.done_tt = sd["done_out"].view(torch.bool).view(self.E, self.A, 4)[:, :, 0:2] if "done_out" in sd else None
        self.step_flags = sd["done_out"].view(torch.int16).view(self.E, self.A, 2)[:, :, 1] if "done_out" in sd else None   # MD_FL_* of the step
        self.need_reset = sd["need_reset"].view(torch.int32)
        self.shape_f = sd["shape"].view(torch.float32).view(self.E, self.cap, 8)
        self.dyn_f = sd["dyn"].view(torch.float32).view(self.E, self.cap, 8)
        self.nav_i = sd["nav"].view(torch.int32).view(self.E, self.cap, 16)
        self.agent_id = sd["agent_id"].view(torch.int32).view(self.E, self.cap) if "agent_id" in sd else None
        if getattr(h, "tracks", None) is not None:     # scenario mode: the scenes' recorded frames come with the host scene
            tr = h.tracks
            self.set_tracks(dict(shape=torch.from_numpy(np.ascontiguousarray(tr["shape"]).view(np.uint8).reshape(tr["shape"].shape[0], -1)),
                                 dyn=torch.from_numpy(np.ascontiguousarray(tr["dyn"])), seeds=tr["seeds"], cap=tr["cap"]))

    def _pack_step_outputs(self):
        """obs | reward | done_out (terminated, truncated, step flags) of this rank in ONE allocation, in that order, each part
        16-byte aligned: the kernel writes the three arrays where it always did, and the whole step output of the shard is one
        contiguous slab -- the single collective of SURVEY 8(e) (sharding.gather_step_slab) moves it as it lies."""
        torch = self.torch
        sd = self.state_dev
        names = [k for k in ("obs", "reward", "done_out") if k in sd]
        layout, at = {}, 0
        for k in names:
            layout[k] = (at, sd[k].numel())
            at = (at + sd[k].numel() + 15) // 16 * 16
        slab = torch.zeros(at, dtype=torch.uint8, device=self.device)
        for k in names:
            o, n = layout[k]
            view = slab[o:o + n]
            view.copy_(sd[k])
            sd[k] = view
        self.out_slab, self.out_layout = slab, layout

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _on_device(self):
        """Context in which HIP's current device is the engine's (config["device"]): a no-op object when it already is
        (the usual one-process-per-GPU case), torch.cuda.device(...) otherwise -- a launch must not land on another
        device's stream because the caller's current device differs."""
        cuda = self.torch.cuda
        if cuda.current_device() == self._dev_index:
            return _NULL_CTX
        return cuda.device(self._dev_index)

    def reset(self):
        """All envs back to their reset snapshot; returns after the reset observation is computed.
        (BaseEnv.reset -> engine.reset -> _get_reset_return, envs/base_env.py:502-584)"""
        self.need_reset.fill_(1)
        self.step_raw()

    SIDE_MASK = (1 << abi.Q_LINE_WHITE_CONT) | (1 << abi.Q_LINE_YELLOW_CONT)      # CollisionGroup.ContinuousLaneLine
    LANE_LINE_MASK = SIDE_MASK | (1 << abi.Q_LINE_BROKEN)                         # ... | BrokenLaneLine

    def step_raw(self):
        with self._on_device():
            self._step_raw()

    def _step_raw(self):
        self._check(self.lib.md_step(C.byref(self.w), C.byref(self.s), C.byref(self.k), self._stream()), "md_step")
        if self._staged is not None:     # random_traffic: the envs whose episode just ended get other traffic for the next one
            self._check(self.lib.md_swap_draw(C.byref(self.s), C.byref(self._staged), C.byref(self.k), len(self.draw_hosts_),
                                              C.c_void_p(self.draw_idx.data_ptr()), self._stream()), "md_swap_draw")
        h = self.host
        vc = self.cfg["vehicle_config"]
        if self._fused_detectors:
            pass
        elif h.n_side and h.n_ll and h.n_side + h.n_ll <= 255:
            # both detector clouds in ONE launch and one pass over the line pieces (md_line_detectors)
            self._check(self.lib.md_line_detectors(
                C.byref(self.w), C.byref(self.s), C.byref(self.k),
                C.c_void_p(self._side_beams.data_ptr()), h.n_side, C.c_float(float(vc["side_detector"]["distance"])), C.c_uint32(self.SIDE_MASK), h.obs_base,
                C.c_void_p(self._ll_beams.data_ptr()), h.n_ll, C.c_float(float(vc["lane_line_detector"]["distance"])), C.c_uint32(self.LANE_LINE_MASK),
                h.obs_base + (h.n_side or 2) + 6, C.c_void_p(self.state_dev["obs"].data_ptr()), h.obs_dim, self._stream()), "md_line_detectors")
        elif h.n_side:   # SideDetector cloud replaces obs[0:2] (obs/state_obs.py:77-86)
            self.line_detector(self._side_beams, h.n_side, float(self.cfg["vehicle_config"]["side_detector"]["distance"]),
                               self.SIDE_MASK, self.state_dev["obs"], h.obs_dim, h.obs_base)
        if h.n_ll and not self._fused_detectors and not (h.n_side and h.n_side + h.n_ll <= 255):     # LaneLineDetector cloud replaces the lateral dim (obs/state_obs.py:129-140)
            self.line_detector(self._ll_beams, h.n_ll, float(self.cfg["vehicle_config"]["lane_line_detector"]["distance"]),
                               self.LANE_LINE_MASK, self.state_dev["obs"], h.obs_dim, h.obs_base + (h.n_side or 2) + 6)
        self._lidar_noise()
        if self._rec is not None:
            self._record_frame()

    # -- record / replay of the traffic (the role of RecordManager / ReplayManager for the movers of the batch,
    #    manager/record_manager.py:35-135, manager/replay_manager.py:21-195, policy/replay_policy.py:43-67) ---------
    def start_recording(self, max_steps):
        """Call right after reset(): frame 0 is the reset state, frame k the state after the k-th step.  Frames are
        device tensors (32 + 8 bytes per slot and step); recording stops by itself when the buffer is full."""
        torch = self.torch
        n = self.E * self.cap
        self._rec = dict(shape=torch.empty((max_steps + 1, n * 32), dtype=torch.uint8, device=self.device),
                         dyn=torch.empty((max_steps + 1, n, 2), dtype=torch.float32, device=self.device), n=0)
        self._record_frame()

    def _record_frame(self):
        r = self._rec
        if r["n"] >= r["shape"].shape[0]:
            return
        r["shape"][r["n"]].copy_(self.state_dev["shape"])
        r["dyn"][r["n"], :, 0].copy_(self.dyn_f.reshape(-1, 8)[:, 0])      # heading
        r["dyn"][r["n"], :, 1].copy_(self.dyn_f.reshape(-1, 8)[:, 1])      # speed
        r["n"] += 1

    def stop_recording(self):
        """-> dict(shape=[T, E*cap*32] uint8, dyn=[T, E*cap, 2] float32, seeds=...) of the T recorded frames."""
        r, self._rec = self._rec, None
        return dict(shape=r["shape"][:r["n"]].contiguous(), dyn=r["dyn"][:r["n"]].contiguous(),
                    seeds=list(self.host.seeds), cap=self.cap)

    def set_tracks(self, tracks):
        """traffic_mode 'replay': every non-agent slot follows these recorded frames (episode step k -> frame k; the
        last frame is held afterwards).  The tracks must come from the same scenario assignment and capacity."""
        if list(tracks["seeds"]) != list(self.host.seeds) or tracks["cap"] != self.cap:
            raise ValueError("tracks were recorded with another scenario assignment or mover capacity")
        self._tracks = dict(shape=tracks["shape"].to(self.device).contiguous(), dyn=tracks["dyn"].to(self.device).contiguous())
        self.s.track_shape = self._tracks["shape"].data_ptr()
        self.s.track_dyn = self._tracks["dyn"].data_ptr()
        self.k.track_len = int(self._tracks["shape"].shape[0])

    def _lidar_noise(self):
        """LidarStateObservation._add_noise_to_cloud_points (obs/state_obs.py:234-244) and the same call on the side /
        lane-line detector clouds (state_obs.py:82-85,134-137): gaussian noise (clipped to [0,1]) then dropout to 0,
        each cloud with its own detector's `gaussian_noise` / `dropout_prob`.  The reference draws from the global,
        unseeded numpy stream, so no stream can be 'the' stream; this one is a device generator seeded with
        start_seed + env_seed_offset (reproducible, shard-dependent)."""
        vc = self.cfg["vehicle_config"]
        h = self.host
        clouds = [(vc["lidar"], self.obs_dim - self.n_beams, self.n_beams),
                  (vc["side_detector"], h.obs_base, h.n_side),
                  (vc["lane_line_detector"], h.obs_base + (h.n_side or 2) + 6, h.n_ll)]
        torch = self.torch
        for dc, off, n in clouds:
            g, p = float(dc["gaussian_noise"]), float(dc["dropout_prob"])
            if (g <= 0.0 and p <= 0.0) or n <= 0:
                continue
            if self._noise_gen is None:
                self._noise_gen = torch.Generator(device=self.device)
                self._noise_gen.manual_seed(int(self.cfg["start_seed"]) + int(self.cfg["env_seed_offset"]))
            cloud = self.obs[..., off:off + n]
            if g > 0.0:
                noise = torch.empty_like(cloud).normal_(0.0, g, generator=self._noise_gen)
                cloud.copy_((cloud + noise).clamp_(0.0, 1.0))
            if p > 0.0:
                assert p <= 1.0
                drop = torch.empty_like(cloud).uniform_(0.0, 1.0, generator=self._noise_gen) < p
                cloud.masked_fill_(drop, 0.0)

    def line_detector(self, beams, n, dist, mask, out, stride, offset):
        self._check(self.lib.md_line_detector(C.byref(self.w), C.byref(self.s), C.byref(self.k), C.c_void_p(beams.data_ptr()),
                                              n, C.c_float(dist), C.c_uint32(mask), C.c_void_p(out.data_ptr()), stride, offset,
                                              self._stream()), "md_line_detector")

    def step(self, actions):
        """actions: tensor [E, A, 2] (or [E, 2] when A == 1), float32, on the engine's device.  With agent_policy =
        IDMPolicy the agents drive themselves: `actions` is ignored (None is fine), as the reference's IDMPolicy ignores
        what env.step() is given."""
        if self.k.agent_idm:
            self.s.agent_action = None
            self.step_raw()
            return
        a = actions
        if a.dim() == 2:
            a = a.unsqueeze(1)
        if tuple(a.shape) != (self.E, self.A, 2):
            raise ValueError("actions must have shape [{}, {}, 2], got {}".format(self.E, self.A, tuple(a.shape)))
        if a.dtype != self.torch.float32 or a.device != self.device or not a.is_contiguous():
            a = a.to(self.device, self.torch.float32).contiguous()
        # zero-copy: the kernel reads the agents' actions from the caller's tensor (MdState.agent_action) and
        # writes the sanitised values into the per-slot `action` array itself.  The struct is passed by value at
        # launch, so the pointer is cleared again right away; the tensor is kept alive until the next step.
        self._held_actions = a
        self.s.agent_action = a.data_ptr()
        try:
            self.step_raw()
        finally:
            self.s.agent_action = None

    def call(self, name):
        """Single-phase entry points (parity tests): md_integrate, md_localize, ..."""
        fn = getattr(self.lib, name)
        with self._on_device():
            self._check(fn(C.byref(self.w), C.byref(self.s), C.byref(self.k), self._stream()), name)

    def lidar(self, out, stride, offset):
        with self._on_device():
            self._lidar(out, stride, offset)

    def _lidar(self, out, stride, offset):
        self._check(self.lib.md_lidar(C.byref(self.w), C.byref(self.s), C.byref(self.k), C.c_void_p(out.data_ptr()),
                                      stride, offset, self._stream()), "md_lidar")

    # -- user-spawned traffic participants (engine.spawn_object(Pedestrian, ...) + set_velocity of the reference,
    #    tests/test_functionality/test_pedestrian.py:38-55); see participants.py ---------------------------------
    def _edit_participants(self, fn):
        names = ("shape", "shape0", "dyn", "flags")
        st = {k: self.state_dev[k].cpu().numpy().view(self.host.state[k].dtype).reshape(self.host.state[k].shape).copy()
              for k in names}
        out = fn(st)
        self.upload_state({k: st[k] for k in ("shape", "dyn", "flags")})
        return out

    def spawn_object(self, kind, position, heading_theta=0.0, envs=None):
        """-> slot handle.  The participant lives until its env resets (auto-reset included)."""
        from metadrive_ped_amd import participants as P
        return self._edit_participants(lambda st: P.spawn(st, self.E, self.cap, self.A, kind, position, heading_theta, envs))

    def set_velocity(self, slot, direction, value=None, in_local_frame=False, envs=None):
        from metadrive_ped_amd import participants as P
        self._edit_participants(lambda st: P.set_velocity(st, self.E, self.cap, slot, direction, value, in_local_frame, envs))

    def clear_objects(self, slots, envs=None):
        from metadrive_ped_amd import participants as P
        self._edit_participants(lambda st: [P.clear(st, self.E, self.cap, s, envs) for s in slots])

    def object_positions(self, slot):
        """[E, 2] tensor view of the slot's centre in every env."""
        return self.shape_f[:, slot, 0:2]

    def download_state(self):
        """Device state -> dict of numpy arrays with the host dtypes (tests / checkpoints)."""
        out = {}
        for k, v in self.host.state.items():
            out[k] = self.state_dev[k].cpu().numpy().view(v.dtype).reshape(v.shape).copy()
        return out

    def upload_state(self, arrays):
        for k, v in arrays.items():
            self.state_dev[k].copy_(self.torch.from_numpy(np.ascontiguousarray(v).view(np.uint8).reshape(-1)))

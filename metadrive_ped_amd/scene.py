"""Reset-time scene construction for ONE environment (host, numpy): which map, where the agent
spawns, its route, which traffic vehicles exist and when they wake up.

Restates the reset chain of the reference in manager PRIORITY order (engine/base_engine.py:306-400):

  engine.seed(s)                  every manager's np_random = get_np_random(s)   (base_engine.py:546-553)
  PGMapManager.reset              PGMap via BIG with random_seed = s            (manager/pg_map_manager.py:52-66)
  VehicleAgentManager.reset       spawn lane index = np_random.randint(lane_num); vehicle seed from the
                                  ENGINE stream; policy seed from the manager stream
                                  (manager/agent_manager.py:37-52,88-113)
  PGTrafficManager.reset          _create_vehicles_once per block: shuffle slots, pick type, spawn, policy seed
                                  (manager/traffic_manager.py:51-72,230-277)

Vehicle parameters come from the vehicle's own stream (first sample_parameters draw after seeding:
component/vehicle/base_vehicle.py:296-303), including the BoxSpace(max,min) quirk.

Pinned by fixtures: map topology per seed (tests/golden/pg_maps*.json); the traffic placement per seed -- which lane,
longitude, vehicle class, vehicle seed and sampled parameters, policy seed, trigger block, and where the engine and
traffic streams stand afterwards -- by tests/golden/traffic_spawn.json: the reference's own PGTrafficManager.reset in
the trigger / respawn / hybrid modes with spawn_object replaced by a recorder that draws the engine seed exactly as
BaseEngine.spawn_object does (oracle/gen/gen_golden.py: section_traffic_spawn; tests/test_oracle_golden.py).
"""
import math

import numpy as np

from metadrive_ped_amd import abi
from metadrive_ped_amd.mapgen.lanes import wrap_to_pi
from metadrive_ped_amd.mapgen.pg import FirstBlock
from metadrive_ped_amd.mapgen.tables import destination_for, respawn_lanes
from metadrive_ped_amd.pg_space import VEHICLE_TYPES, sample_parameters
from metadrive_ped_amd.rng import Randomizable, get_np_random

VEHICLE_GAP = 10  # PGTrafficManager.VEHICLE_GAP
GRAVITY = 9.8
TRAFFIC_TYPE_KEYS = ["s", "m", "l", "xl", "default"]  # vehicle_type.py:269-275 (dict order)
TRAFFIC_TYPE_P = [0.2, 0.3, 0.3, 0.2, 0.0]            # traffic_manager.py:298-301


def _peek(rng):
    """The next rand() of a RandomState without advancing it."""
    r = np.random.RandomState()
    r.set_state(rng.get_state())
    return float(r.rand())


def vehicle_param_record(vtype, vehicle_seed, substep_dt, overrides=None):
    """MdParam + (length, width) of a vehicle of class `vtype` seeded with `vehicle_seed`.  `overrides` are the
    agent-config values laid over the sampled parameters (BaseVehicle.__init__: sample_parameters through
    BaseObject.__init__, then update_config(vehicle_config), base_vehicle.py:136-138): max_engine_force,
    max_brake_force, wheel_friction, max_steering, mass of the varying-dynamics vehicle (vehicle_type.py:168-187)."""
    spec = VEHICLE_TYPES[vtype]
    rng = get_np_random(vehicle_seed)
    cfg = sample_parameters(rng, spec["space"])
    mass = spec["mass"]
    if overrides:
        cfg.update({k: v for k, v in overrides.items() if k != "mass"})
        mass = overrides.get("mass", mass)
        cfg["mass"] = mass
    rec = np.zeros((), dtype=abi.PARAM_DT)
    rec["max_steer"] = math.radians(cfg["max_steering"])
    rec["accel_gain"] = 4.0 * cfg["max_engine_force"] / mass
    rec["brake_gain"] = 4.0 * cfg["max_brake_force"] / (mass * substep_dt)
    rec["roll_decel"] = 4.0 * 2.0 / (mass * substep_dt)
    rec["max_speed_kmh"] = cfg["max_speed_km_h"]
    rec["lf"], rec["lr"] = spec["lf"], spec["lr"]
    rec["fric_decel"] = cfg["wheel_friction"] * GRAVITY
    return rec, spec["length"], spec["width"], cfg


class EnvScene:
    """All per-env arrays (cap slots) for one seed."""
    def __init__(self, seed, map_tables, cfg):
        cap = cfg["cap"]
        A = cfg["agents_per_env"]
        self.seed = seed
        self.tables = map_tables
        pg_map = map_tables.pg_map
        self.shape = np.zeros(cap, dtype=abi.SHAPE_DT)
        self.shape["aux"] = -1
        self.dyn = np.zeros(cap, dtype=abi.DYN_DT)
        self.param = np.zeros(cap, dtype=abi.PARAM_DT)
        self.param["max_speed_kmh"] = 80.0
        self.param["lf"], self.param["lr"] = 1.0, 1.0
        self.nav = np.zeros(cap, dtype=abi.NAV_DT)
        self.nav["lane"] = -1
        self.nav["target_lane"] = -1
        self.pid = np.zeros(cap, dtype=abi.PID_DT)
        self.pid["target_speed"] = 30.0
        self.route_nodes = np.full((cap, abi.MD_ROUTE_LEN), -1, dtype=np.int32)
        self.route_roads = np.full((cap, abi.MD_ROUTE_LEN), -1, dtype=np.int32)
        self.final_lane = np.zeros(cap, dtype=np.int32)
        self.idm_rand = np.zeros((cap, abi.MD_IDM_RAND), dtype=np.int32)
        self.n_traffic = 0
        self.vehicle_cfgs = [None] * cap

        engine = Randomizable(seed)          # BaseEngine is a Randomizable seeded with the scenario index
        object_mgr = Randomizable(seed)
        agent_mgr = Randomizable(seed)
        traffic_mgr = Randomizable(seed)
        if cfg.get("random_traffic"):
            # random_traffic=True: PGTrafficManager.seed() skips the re-seeding at reset (manager/traffic_manager.py:335-337),
            # so the traffic stream runs on from wherever it is and every episode gets other traffic.  The reference's
            # stream starts unseeded; here it is keyed by (scenario seed, how many resets the env has seen): the traffic
            # differs from reset to reset and the run as a whole stays reproducible.
            traffic_mgr = Randomizable((seed * 1000003 + 7919 * (int(cfg.get("traffic_epoch", 0)) + 1)) % (2 ** 31))
        dt = cfg["physics_world_step_size"]
        self._dt = dt
        self._next_prop_slot = cap - 1       # props fill the slot array from the top down
        self.accident_lanes = []

        # ---- toll booths: spawned by their block while the map is built (PGMapManager, PRIORITY 0) ----
        for _ in range(getattr(pg_map.net, "building_spawns", 0)):
            engine.generate_seed()
        for block in pg_map.blocks:
            for lane, pos, heading in getattr(block, "buildings", ()):
                self._place_building(lane, pos, heading, block.BUILDING_LENGTH)

        # ---- static props (TrafficObjectManager, PRIORITY 9: before agents and traffic) ----
        if abs(cfg.get("accident_prob", 0.0)) >= 1e-2:
            self._object_scenes(cfg, engine, object_mgr, traffic_mgr)

        # ---- agents (single agent: slot 0) ----
        assert A == 1, "multi-agent scenes are built by the MARL scene builder"
        # VaryingDynamicsAgentManager.reset (envs/varying_dynamics_env.py:29-49): one uniform draw per randomised
        # parameter, in the config's key order, from the agent manager's stream BEFORE the agents are created
        dynamics = None
        if cfg.get("random_dynamics"):
            dynamics = {}
            for name, rng_ in cfg["random_dynamics"].items():
                if rng_ is None:
                    continue
                if not isinstance(rng_, (tuple, list)) or len(rng_) != 2 or rng_[1] < rng_[0]:
                    raise ValueError("Unknown parameter range: {}".format(rng_))
                dynamics[name] = rng_[0] if rng_[1] == rng_[0] else float(agent_mgr.np_random.uniform(rng_[0], rng_[1]))
        lane_num = pg_map.lane_num
        if cfg["random_spawn_lane_index"]:
            spawn_idx = int(agent_mgr.np_random.randint(lane_num))
        else:
            spawn_idx = cfg["spawn_lane_index"][2]
        spawn_lane_index = (FirstBlock.NODE_1, FirstBlock.NODE_2, spawn_idx)
        agent_model = cfg["agent_vehicle_model"]
        if cfg.get("random_agent_model", False):
            # VehicleAgentManager._create_agents -> random_vehicle_type(self.np_random), uniform over the five
            # classes (manager/agent_manager.py:41, component/vehicle/vehicle_type.py:269-281)
            agent_model = str(agent_mgr.np_random.choice(["s", "m", "l", "xl", "default"], p=[0.2] * 5))
        vehicle_seed = engine.generate_seed()
        policy_seed = agent_mgr.generate_seed()  # policy seed (EnvInputPolicy does not use it)
        size = cfg.get("agent_size_mass") or {}
        if agent_model == "varying_dynamics" and size.get("mass") is not None:
            # VaryingDynamicsVehicle.MASS / WIDTH / LENGTH come from its config when given (vehicle_type.py:168-187)
            dynamics = dict(dynamics or {})
            dynamics.setdefault("mass", size["mass"])
        self._place_vehicle(0, agent_model, vehicle_seed, spawn_lane_index, cfg["spawn_longitude"],
                            cfg["spawn_lateral"], dt, abi.F_ALIVE | abi.F_AGENT, overrides=dynamics,
                            destination=cfg.get("destination"))
        if agent_model == "varying_dynamics":
            if size.get("length") is not None:
                self.shape[0]["hl"] = float(size["length"]) / 2
            if size.get("width") is not None:
                self.shape[0]["hw"] = float(size["width"]) / 2
        if cfg.get("spawn_velocity") is not None:
            # BaseVehicle.reset -> set_velocity(spawn_velocity, in_local_frame=spawn_velocity_car_frame); the vehicle model
            # here carries a signed speed along the heading, so the component along the heading is what starts the episode
            vx, vy = float(cfg["spawn_velocity"][0]), float(cfg["spawn_velocity"][1])
            if cfg.get("spawn_velocity_car_frame"):
                self.dyn[0]["speed"] = vx
            else:
                self.dyn[0]["speed"] = vx * float(self.shape[0]["c"]) + vy * float(self.shape[0]["s"])
        if cfg.get("agent_policy") == "IDMPolicy":
            # IDMPolicy.__init__ (policy/idm_policy.py:225-233): overtake_timer = randint(0, LANE_CHANGE_FREQ) from the
            # policy's own stream
            self.nav[0]["timer"] = int(get_np_random(policy_seed).randint(0, 50))

        # ---- traffic (trigger mode) ----
        density = cfg["traffic_density"]
        slot = A
        if abs(density) >= 1e-2:
            if cfg["traffic_mode"] not in ("trigger", "respawn", "hybrid", "replay"):
                raise ValueError("No such mode named {}".format(cfg["traffic_mode"]))  # traffic_manager.py:71
            if cfg["traffic_mode"] == "respawn":
                # _create_respawn_vehicles (traffic_manager.py:213-228): every vehicle drives from step 0
                for lane in respawn_lanes(pg_map):
                    total_num = int(lane.length / VEHICLE_GAP)
                    longs = [i * VEHICLE_GAP for i in range(total_num)]
                    traffic_mgr.np_random.shuffle(longs)
                    for long in longs[:int(np.ceil(density * len(longs)))]:
                        vtype = str(traffic_mgr.np_random.choice(TRAFFIC_TYPE_KEYS, p=TRAFFIC_TYPE_P))
                        vseed = engine.generate_seed()
                        policy_seed = traffic_mgr.generate_seed()
                        if slot > self._next_prop_slot:
                            raise ValueError("env seed {}: more than cap={} movers; raise `mover_capacity`".format(seed, cap))
                        self._place_vehicle(slot, vtype, vseed, tuple(lane.index), float(long), 0.0, dt, abi.F_ALIVE)
                        self.nav[slot]["trigger_road"] = -1
                        prng = get_np_random(policy_seed)
                        self.nav[slot]["timer"] = int(prng.randint(0, 50))
                        self.idm_rand[slot] = [int(prng.randint(0, 25)) for _ in range(abi.MD_IDM_RAND)]
                        slot += 1
            for bi, block in enumerate(pg_map.blocks[1:] if cfg["traffic_mode"] != "respawn" else [], start=1):
                trigger_lanes = block.intermediate_spawn_lanes()
                if cfg.get("need_inverse_traffic", False) and block.ID in ("S", "C", "r", "R"):
                    # oncoming traffic on the simple blocks (traffic_manager.py:242-245)
                    neg_lanes = block.negative_roads_lanes()
                    traffic_mgr.np_random.shuffle(neg_lanes)
                    trigger_lanes = trigger_lanes + neg_lanes
                potential = []
                for lanes in trigger_lanes:
                    for l in lanes:
                        if l in self.accident_lanes:   # traffic_manager.py:247-248
                            continue
                        total_num = int(l.length / VEHICLE_GAP)
                        potential += [(l.index, i * VEHICLE_GAP) for i in range(total_num)]
                total_length = sum(l.length for lanes in trigger_lanes for l in lanes)
                total_spawn_points = int(math.floor(total_length / VEHICLE_GAP))
                total_vehicles = int(math.floor(total_spawn_points * density))
                traffic_mgr.np_random.shuffle(potential)
                selected = potential[:min(total_vehicles, len(potential))]
                trigger_road = block.pre_socket.positive
                for lane_index, long in selected:
                    vtype = str(traffic_mgr.np_random.choice(TRAFFIC_TYPE_KEYS, p=TRAFFIC_TYPE_P))
                    vseed = engine.generate_seed()
                    policy_seed = traffic_mgr.generate_seed()
                    if slot > self._next_prop_slot:
                        raise ValueError("env seed {}: more than cap={} movers; raise `mover_capacity`".format(seed, cap))
                    self._place_vehicle(slot, vtype, vseed, lane_index, float(long), 0.0, dt,
                                        abi.F_ALIVE | abi.F_PENDING)
                    self.nav[slot]["trigger_road"] = map_tables.road_id[trigger_road]
                    self.nav[slot]["trigger_order"] = bi
                    prng = get_np_random(policy_seed)
                    self.nav[slot]["timer"] = int(prng.randint(0, 50))  # IDMPolicy.__init__ (idm_policy.py:229)
                    self.idm_rand[slot] = [int(prng.randint(0, 25)) for _ in range(abi.MD_IDM_RAND)]
                    slot += 1
        self.n_traffic = slot - A
        # where the two streams stand after the reset chain (copies: nothing is consumed) -- what the golden fixture of the
        # reference's PGTrafficManager.reset ends with (tests/golden/traffic_spawn.json)
        self.stream_probe = dict(engine=_peek(engine.np_random), traffic=_peek(traffic_mgr.np_random))

    @property
    def n_props(self):
        return len(self.shape) - 1 - self._next_prop_slot

    def trim(self, cap):
        """Shrink the slot arrays to `cap`: vehicles keep their low slots, props keep filling from the top."""
        old = len(self.shape)
        used_low = 1 + self.n_traffic
        n_props = self.n_props
        assert cap >= used_low + n_props
        for k in ("shape", "dyn", "param", "nav", "pid", "route_nodes", "route_roads", "final_lane", "idm_rand"):
            arr = getattr(self, k)
            new = arr[:cap].copy()
            if n_props:
                new[cap - n_props:] = arr[old - n_props:]
            setattr(self, k, new)
        cfgs = self.vehicle_cfgs[:cap]
        if n_props:
            cfgs[cap - n_props:] = self.vehicle_cfgs[old - n_props:]
        self.vehicle_cfgs = cfgs
        self._next_prop_slot = cap - 1 - n_props

    # -- TrafficObjectManager.reset (manager/object_manager.py:40-151) ---------------------------
    ALERT_DIST, ACCIDENT_AREA_LEN, CONE_LONGITUDE, CONE_LATERAL, PROHIBIT_SCENE_PROB = 10, 10, 2, 1, 0.67

    def _object_scenes(self, cfg, engine, object_mgr, traffic_mgr):
        if not cfg.get("static_traffic_object", True):
            raise NotImplementedError("static_traffic_object=False (props that react to collisions) is not built")
        t = self.tables
        pg_map = t.pg_map
        rng = object_mgr.np_random
        prob = cfg["accident_prob"]
        for block in pg_map.blocks:
            if block.ID not in ("S", "C", "r", "R"):
                continue
            if rng.rand() > prob:
                continue
            node0 = "{}{}0_0_".format(block.index, block.ID)
            node1 = "{}{}0_1_".format(block.index, block.ID)
            road_1 = (block.pre_socket.positive[1], node0)
            road_2 = (node0, node1) if block.ID != "S" else None
            is_ramp = block.ID in ("r", "R")
            if rng.rand() > self.PROHIBIT_SCENE_PROB:
                if block.ID != "C":
                    accident_road = [road_1, road_2][int(rng.choice(2))]
                else:
                    accident_road = road_2
                accident_road = road_1 if accident_road is None else accident_road
                on_left = bool(rng.rand() > 0.5 or (accident_road is road_2 and is_ramp))
                lanes = pg_map.net.lanes(*accident_road)
                lane = lanes[0 if on_left else -1]
                longitude = lane.length - self.ACCIDENT_AREA_LEN - 5
                self.accident_lanes.append(lane)
                self._prohibit_scene(engine, lane, longitude, pg_map.lane_width, on_left)
            else:
                accident_road = [road_1, road_2][int(rng.choice(2))]
                accident_road = road_1 if accident_road is None else accident_road
                on_left = bool(rng.rand() > 0.5 or (accident_road is road_2 and is_ramp))
                lanes = pg_map.net.lanes(*accident_road)
                if len(lanes) - 1 == 0:
                    idx = -1
                else:
                    idx = int(rng.randint(0, len(lanes) - 1)) if on_left else -1
                lane = lanes[idx]
                self.accident_lanes.append(lane)
                longitude = rng.rand() * lane.length / 2 + lane.length / 2
                if rng.rand() > 0.5:
                    # break_down_scene: vehicle type comes from the TRAFFIC manager's stream (object_manager.py:96-98)
                    vtype = str(traffic_mgr.np_random.choice(TRAFFIC_TYPE_KEYS, p=TRAFFIC_TYPE_P))
                    vseed = engine.generate_seed()
                    slot = self._take_prop_slot()
                    self._place_vehicle(slot, vtype, vseed, lane.index, float(longitude), 0.0, self._dt,
                                        abi.F_ALIVE | abi.F_STATIC)
                    self.shape[slot]["aux"] = t.lane_id[tuple(lane.index)]
                    engine.generate_seed()
                    self._place_prop(abi.KIND_WARNING, lane, longitude - self.ALERT_DIST, 0.0)
                else:
                    engine.generate_seed()
                    self._place_prop(abi.KIND_BARRIER, lane, longitude, 0.0)

    def _prohibit_scene(self, engine, lane, longitude_position, lateral_len, on_left):
        lat_num = int(lateral_len / self.CONE_LATERAL)
        longitude_num = int(self.ACCIDENT_AREA_LEN / self.CONE_LONGITUDE)
        lat_1 = [lat * self.CONE_LATERAL for lat in range(lat_num)]
        lat_2 = [lat_num * self.CONE_LATERAL] * (longitude_num + 1)
        lat_3 = [(lat_num - lat - 1) * self.CONE_LATERAL for lat in range(int(lat_num))]
        total_long_num = lat_num * 2 + longitude_num + 1
        pos = [(lg * self.CONE_LONGITUDE, lat - lane.width / 2)
               for lg, lat in zip(range(-int(total_long_num / 2), int(total_long_num / 2)), lat_1 + lat_2 + lat_3)]
        left = 1 if on_left else -1
        for p in pos:
            engine.generate_seed()
            self._place_prop(abi.KIND_CONE, lane, p[0] + longitude_position, left * p[1])

    def _take_prop_slot(self):
        slot = self._next_prop_slot
        if slot < 1:
            raise ValueError("env seed {}: no free slot for props; raise `mover_capacity`".format(self.seed))
        self._next_prop_slot -= 1
        return slot

    def _place_prop(self, kind, lane, longitude, lateral):
        """Cone r=0.2, warning r=0.5 (cylinders), barrier box 0.3 (along lane) x 2.0
        (component/static_object/traffic_object.py:43-177)."""
        slot = self._take_prop_slot()
        pos = lane.position(longitude, lateral)
        heading = wrap_to_pi(lane.heading_theta_at(longitude))
        sh = self.shape[slot]
        sh["cx"], sh["cy"] = pos
        sh["c"], sh["s"] = math.cos(heading), math.sin(heading)
        if kind == abi.KIND_CONE:
            sh["hl"] = sh["hw"] = 0.2
        elif kind == abi.KIND_WARNING:
            sh["hl"] = sh["hw"] = 0.5
        else:
            sh["hl"], sh["hw"] = 0.15, 1.0
        sh["flags"] = kind | abi.F_ALIVE | abi.F_STATIC
        sh["aux"] = self.tables.lane_id[tuple(lane.index)]
        self.dyn[slot]["heading"] = heading
        return slot

    def _place_building(self, lane, pos, heading, length):
        """TollGateBuilding (component/buildings/tollgate_building.py:7-27): a solid box `length` x the lane's width; a
        BaseStaticObject with `.lane`, so the IDM of the traffic on that lane queues up behind it; crashing into it raises
        crash_building (base_vehicle.py:737-738), which always ends the episode (envs/metadrive_env.py:170-175)."""
        slot = self._take_prop_slot()
        sh = self.shape[slot]
        heading = wrap_to_pi(heading)
        sh["cx"], sh["cy"] = pos
        sh["c"], sh["s"] = math.cos(heading), math.sin(heading)
        sh["hl"], sh["hw"] = length / 2.0, lane.width / 2.0
        sh["flags"] = abi.KIND_BUILDING | abi.F_ALIVE | abi.F_STATIC
        sh["aux"] = self.tables.lane_id[tuple(lane.index)]
        self.dyn[slot]["heading"] = heading
        return slot

    def _place_vehicle(self, slot, vtype, vehicle_seed, lane_index, longitude, lateral, dt, flags, overrides=None,
                       destination=None):
        t = self.tables
        pg_map = t.pg_map
        lane = pg_map.net.lanes(lane_index[0], lane_index[1])[lane_index[2]]
        prm, length, width, cfg = vehicle_param_record(vtype, vehicle_seed, dt, overrides)
        self.vehicle_cfgs[slot] = dict(type=vtype, seed=vehicle_seed, **cfg)
        pos = lane.position(longitude, lateral)
        heading = lane.heading_theta_at(longitude)
        heading = wrap_to_pi(heading)  # BaseVehicle.heading_theta (base_vehicle.py:990-992)
        sh = self.shape[slot]
        sh["cx"], sh["cy"] = pos
        sh["c"], sh["s"] = math.cos(heading), math.sin(heading)
        sh["hl"], sh["hw"] = length / 2, width / 2
        sh["flags"] = abi.KIND_VEHICLE | flags
        sh["aux"] = -1
        d = self.dyn[slot]
        d["heading"] = heading
        d["last_x"], d["last_y"] = pos
        d["last_c"], d["last_s"] = sh["c"], sh["s"]
        self.param[slot] = prm
        # navigation.reset + set_route (node_network_navigation.py:43-128)
        lane_id = t.lane_id[tuple(lane_index)]
        start_node = lane_index[0]
        dest = destination_for(pg_map, self.seed, lane_index)
        if destination is not None:      # vehicle_config.destination: the agent's own end node (node_network_navigation.py:54-56)
            dest = destination
        ckpts = pg_map.bfs_route(start_node, dest)
        if destination is not None and (not ckpts or ckpts[-1] != destination):
            raise ValueError("vehicle_config.destination {!r}: no route from {!r} on the map of seed {}".format(
                destination, start_node, self.seed))
        ck0, ck1 = 0, 1
        if len(ckpts) <= 2:
            ckpts = [lane_index[0], lane_index[1]]
            ck0, ck1 = 0, 0
        if len(ckpts) > abi.MD_ROUTE_LEN:
            raise ValueError("route with {} checkpoints exceeds MD_ROUTE_LEN".format(len(ckpts)))
        for j, name in enumerate(ckpts):
            self.route_nodes[slot, j] = t.node_index[name]
        for j in range(len(ckpts) - 1):
            self.route_roads[slot, j] = t.road_id[(ckpts[j], ckpts[j + 1])]
        final_road = (ckpts[-2], ckpts[-1])
        fr = t.roads[t.road_id[final_road]]
        self.final_lane[slot] = fr["first_lane"] + fr["n_lanes"] - 1
        nv = self.nav[slot]
        nv["lane"] = lane_id
        nv["ck0"], nv["ck1"] = ck0, ck1
        nv["route_len"] = len(ckpts)
        nv["target_lane"] = -1
        self.pid[slot]["target_speed"] = 30.0

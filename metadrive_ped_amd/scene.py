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

Pinned by fixtures: map topology per seed (tests/golden/pg_maps.json).  The order in which the
engine / manager streams are consumed by spawn_object and add_policy is restated from the code
above; the reference cannot spawn vehicles without Bullet, so the traffic placement stream is
parity-unpinned (documented in DESIGN.md).
"""
import math

import numpy as np

from metadrive_ped_amd import abi
from metadrive_ped_amd.mapgen.lanes import wrap_to_pi
from metadrive_ped_amd.mapgen.pg import FirstBlock
from metadrive_ped_amd.pg_space import VEHICLE_TYPES, sample_parameters
from metadrive_ped_amd.rng import Randomizable, get_np_random

VEHICLE_GAP = 10  # PGTrafficManager.VEHICLE_GAP
GRAVITY = 9.8
TRAFFIC_TYPE_KEYS = ["s", "m", "l", "xl", "default"]  # vehicle_type.py:269-275 (dict order)
TRAFFIC_TYPE_P = [0.2, 0.3, 0.3, 0.2, 0.0]            # traffic_manager.py:298-301


def vehicle_param_record(vtype, vehicle_seed, substep_dt):
    """MdParam + (length, width) of a vehicle of class `vtype` seeded with `vehicle_seed`."""
    spec = VEHICLE_TYPES[vtype]
    rng = get_np_random(vehicle_seed)
    cfg = sample_parameters(rng, spec["space"])
    rec = np.zeros((), dtype=abi.PARAM_DT)
    rec["max_steer"] = math.radians(cfg["max_steering"])
    rec["accel_gain"] = 4.0 * cfg["max_engine_force"] / spec["mass"]
    rec["brake_gain"] = 4.0 * cfg["max_brake_force"] / (spec["mass"] * substep_dt)
    rec["roll_decel"] = 4.0 * 2.0 / (spec["mass"] * substep_dt)
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
        agent_mgr = Randomizable(seed)
        traffic_mgr = Randomizable(seed)
        dt = cfg["physics_world_step_size"]

        # ---- agents (single agent: slot 0) ----
        assert A == 1, "multi-agent scenes are built by the MARL scene builder"
        lane_num = pg_map.lane_num
        if cfg["random_spawn_lane_index"]:
            spawn_idx = int(agent_mgr.np_random.randint(lane_num))
        else:
            spawn_idx = cfg["spawn_lane_index"][2]
        spawn_lane_index = (FirstBlock.NODE_1, FirstBlock.NODE_2, spawn_idx)
        vehicle_seed = engine.generate_seed()
        agent_mgr.generate_seed()  # policy seed (EnvInputPolicy does not use it)
        self._place_vehicle(0, cfg["agent_vehicle_model"], vehicle_seed, spawn_lane_index, cfg["spawn_longitude"],
                            cfg["spawn_lateral"], dt, abi.F_ALIVE | abi.F_AGENT)

        # ---- traffic (trigger mode) ----
        density = cfg["traffic_density"]
        slot = A
        if abs(density) >= 1e-2:
            if cfg["traffic_mode"] != "trigger":
                raise NotImplementedError("traffic_mode '{}' is not built yet (only 'trigger')".format(cfg["traffic_mode"]))
            for bi, block in enumerate(pg_map.blocks[1:], start=1):
                trigger_lanes = block.intermediate_spawn_lanes()
                potential = []
                for lanes in trigger_lanes:
                    for l in lanes:
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
                    if slot >= cap:
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

    def trim(self, cap):
        """Shrink the slot arrays to `cap` (>= used slots)."""
        assert cap >= self.n_traffic + 1
        for k in ("shape", "dyn", "param", "nav", "pid", "route_nodes", "route_roads", "final_lane", "idm_rand"):
            setattr(self, k, getattr(self, k)[:cap].copy())
        self.vehicle_cfgs = self.vehicle_cfgs[:cap]

    def _place_vehicle(self, slot, vtype, vehicle_seed, lane_index, longitude, lateral, dt, flags):
        t = self.tables
        pg_map = t.pg_map
        lane = pg_map.net.lanes(lane_index[0], lane_index[1])[lane_index[2]]
        prm, length, width, cfg = vehicle_param_record(vtype, vehicle_seed, dt)
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
        negative = lane_index[1].find("-") != -1
        block = pg_map.blocks[0] if negative else pg_map.blocks[-1]
        sockets = list(block.sockets.values())
        socket = sockets[0] if len(sockets) == 1 else sockets[int(get_np_random(self.seed).choice(len(sockets)))]
        dest = socket.negative[1] if negative else socket.positive[1]
        ckpts = pg_map.bfs_route(start_node, dest)
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

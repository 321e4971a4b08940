"""Multi-agent maps: scene construction (host) for MultiAgentRoundaboutEnv / MultiAgentIntersectionEnv.

Restates the reset-time part of envs/marl_envs/marl_inout_roundabout.py + manager/spawn_manager.py:
one fixed map (FirstPGBlock 60 m, 2 lanes + Roundabout exit 10 / inner 30 / angle 70), 4 spawn roads x 2
lanes x 6 longitudinal slots = 48 spawn points, `num_agents` (40) of them drawn without replacement, a
+-1 m / +-0.25 m jitter inside the slot (spawn_manager.py:211-218 -- note RESPAWN_REGION_LONGITUDE -
MAX_VEHICLE_LENGTH = -2, so the reference draws uniform(1, -1)), a random destination among the four
arms, `static_default` vehicles (fixed engine 800 / brake 150).  The reference seeds none of these
draws (force_seed_spawn_manager=False); here they come from RandomState(env seed) so that a batch is
reproducible.  The per-step lifecycle (dying queue, respawn) runs on the device: md_lifecycle.
"""
import math

import numpy as np

from metadrive_ped_amd import abi
from metadrive_ped_amd.mapgen.lanes import wrap_to_pi
from metadrive_ped_amd.mapgen.pg import negate_road
from metadrive_ped_amd.mapgen.tables import route_arrays
from metadrive_ped_amd.rng import get_np_random
from metadrive_ped_amd.scene import vehicle_param_record

ROUNDABOUT_SPAWN_ROADS = [(">>", ">>>"), negate_road("1O0_2_", "1O0_3_"), negate_road("1O1_2_", "1O1_3_"),
                          negate_road("1O2_2_", "1O2_3_")]
# MAIntersectionConfig.spawn_roads (envs/marl_envs/marl_intersection.py:13-18)
INTERSECTION_SPAWN_ROADS = [(">>", ">>>"), negate_road("1X0_0_", "1X0_1_"), negate_road("1X1_0_", "1X1_1_"),
                            negate_road("1X2_0_", "1X2_1_")]
# MABottleneckConfig.spawn_roads (envs/marl_envs/marl_bottleneck.py:11)
BOTTLENECK_SPAWN_ROADS = [(">>", ">>>"), negate_road("2Y0_0_", "2Y0_1_")]
# MultiAgentMetaDrive on PG maps: MULTI_AGENT_METADRIVE_DEFAULT_CONFIG.spawn_roads (multi_agent_metadrive.py:27)
PG_SPAWN_ROADS = [(">>", ">>>")]
# MABidirectionConfig.spawn_roads (envs/marl_envs/marl_bidirection.py:12): the Split is block 3 there
BIDIRECTION_SPAWN_ROADS = [(">>", ">>>"), negate_road("3Y0_0_", "3Y0_1_")]
# MATollConfig.spawn_roads (envs/marl_envs/marl_tollgate.py:16): the Merge is block 3 there
TOLLGATE_SPAWN_ROADS = [(">>", ">>>"), negate_road("3y0_0_", "3y0_1_")]
# MAParkingLotConfig.in_spawn_roads (envs/marl_envs/marl_parking_lot.py:22-27): the three ways in; the spaces themselves
# (out direction, ParkingLot.node(1, i, 5) -> node(1, i, 6), :188-192) follow
PARKING_IN_ROADS = [(">>", ">>>"), negate_road("2T0_0_", "2T0_1_"), negate_road("2T2_0_", "2T2_1_")]


def parking_lot_roads(parking_space_num):
    """(spawn roads, destination nodes): entrances then spaces; destinations = the spaces (in direction: the lot's
    dest_roads, node(1, i, 1) -> node(1, i, 2)) then the entrances driven the other way (update_destination_for, :80-88)."""
    out_roads = [("1P{}_5_".format(i), "1P{}_6_".format(i)) for i in range(1, parking_space_num + 1)]
    dests = ["1P{}_2_".format(i) for i in range(1, parking_space_num + 1)] + [negate_road(*r)[1] for r in PARKING_IN_ROADS]
    return PARKING_IN_ROADS + out_roads, dests


SPAWN_ROADS = dict(roundabout=ROUNDABOUT_SPAWN_ROADS, intersection=INTERSECTION_SPAWN_ROADS, bottleneck=BOTTLENECK_SPAWN_ROADS,
                   bidirection=BIDIRECTION_SPAWN_ROADS, tollgate=TOLLGATE_SPAWN_ROADS, parking_lot=parking_lot_roads(8)[0],
                   racing=PG_SPAWN_ROADS)      # MultiAgentRacingEnv keeps MultiAgentMetaDrive's spawn road (the first block's exit)
# roundabout / intersection: the spawn manager draws a destination among the arms (update_destination_for overrides);
# bottleneck: the base SpawnManager leaves it to NodeNetworkNavigation.reset (the far end of the map)
FIXED_DESTINATION = dict(roundabout=False, intersection=False, bottleneck=True, bidirection=True, tollgate=True, parking_lot=False,
                         racing=True)
MAX_VEHICLE_LENGTH, MAX_VEHICLE_WIDTH = 10.0, 2.5   # BaseVehicle.MAX_LENGTH / MAX_WIDTH
REGION_LONG, REGION_LAT = 8.0, 3.0


def vehicle_class_table(dt):
    """[5][12] float32: MdParam (8 floats), half length, half width, 0, 0 of the classes random_vehicle_type draws from
    (component/vehicle/vehicle_type.py:269-281), in its order."""
    rows = []
    for name in ("s", "m", "l", "xl", "default"):
        prm, length, width, _ = vehicle_param_record(name, 0, dt)
        rows.append(list(np.frombuffer(prm.tobytes(), np.float32)) + [length / 2, width / 2, 0.0, 0.0])
    return np.asarray(rows, np.float32)


class RoundaboutScene:
    """Per-env arrays (cap == num_agents slots, all agents) for one env seed.  `spawn_roads` selects the map
    family (the roundabout's by default); everything else is SpawnManager's and shared."""
    def __init__(self, seed, mt, cfg, spawn_roads=None, fixed_destination=False, parking=None):
        """`parking`: (number of entrances, number of parking spaces, destination nodes) of the parking-lot env"""
        ROUNDABOUT_SPAWN_ROADS = spawn_roads if spawn_roads is not None else globals()["ROUNDABOUT_SPAWN_ROADS"]
        A = cfg["agents_per_env"]
        cap = cfg["cap"]
        assert cap >= A
        self.seed, self.tables = seed, mt
        self.n_traffic, self.n_props = 0, 0
        self.shape = np.zeros(cap, dtype=abi.SHAPE_DT)
        self.shape["aux"] = -1
        self.dyn = np.zeros(cap, dtype=abi.DYN_DT)
        self.param = np.zeros(cap, dtype=abi.PARAM_DT)
        self.nav = np.zeros(cap, dtype=abi.NAV_DT)
        self.nav["lane"] = -1
        self.nav["target_lane"] = -1
        self.pid = np.zeros(cap, dtype=abi.PID_DT)
        self.pid["target_speed"] = 30.0
        self.route_nodes = np.full((cap, abi.MD_ROUTE_LEN), -1, dtype=np.int32)
        self.route_roads = np.full((cap, abi.MD_ROUTE_LEN), -1, dtype=np.int32)
        self.final_lane = np.zeros(cap, dtype=np.int32)
        self.idm_rand = np.zeros((cap, abi.MD_IDM_RAND), dtype=np.int32)
        rng = get_np_random(seed)
        pg = mt.pg_map
        lane_num = pg.lane_num
        exit_length = cfg["exit_length"] - 10   # minus FirstPGBlock.ENTRANCE_LENGTH
        num_slots = int(math.floor(exit_length / REGION_LONG))
        spots = [(road, li, j) for road in ROUNDABOUT_SPAWN_ROADS for li in range(lane_num) for j in range(num_slots)]
        if A > len(spots) and not cfg.get("initial_agents", 0):
            raise ValueError("Too many agents! We only accept {} agents, but you have {} agents!".format(len(spots), A))
        if cfg.get("initial_agents", 0):
            # num_agents = -1: every spawn point in order, no draw (spawn_manager.py:77-78); slots beyond stay free
            if cfg["initial_agents"] != len(spots):
                raise ValueError("initial_agents {} != spawn points {}".format(cfg["initial_agents"], len(spots)))
            chosen = np.arange(len(spots))
        else:
            chosen = rng.choice(len(spots), A, replace=False)
            if A == 1:
                chosen = np.array([0])   # a lone agent takes the FIRST spawn point, whatever was drawn (spawn_manager.py:85-91)
        dests = [negate_road(*r)[1] for r in ROUNDABOUT_SPAWN_ROADS]
        free_spaces = list(range(parking[1])) if parking else []     # ParkingLotSpawnManager.parking_space_available
        prm, length, width, vcfg = vehicle_param_record(cfg["agent_vehicle_model"], 0, cfg["physics_world_step_size"])
        classes = vehicle_class_table(cfg["physics_world_step_size"]) if cfg.get("random_agent_model") else None
        self.param[:] = prm
        for a, k in enumerate(chosen):
            road, li, j = spots[int(k)]
            lane = pg.net.lanes(*road)[li]
            long = REGION_LONG / 2 + j * REGION_LONG + rng.uniform(-(REGION_LONG - MAX_VEHICLE_LENGTH) / 2,
                                                                   (REGION_LONG - MAX_VEHICLE_LENGTH) / 2)
            lat = rng.uniform(-(REGION_LAT - MAX_VEHICLE_WIDTH) / 2, (REGION_LAT - MAX_VEHICLE_WIDTH) / 2)
            space = 0
            if parking:
                # ParkingLotSpawnManager.update_destination_for (marl_parking_lot.py:80-88): from an entrance to a parking
                # space nobody else is heading for, from a space out through one of the entrances
                n_in, n_space, park_dests = parking
                if ROUNDABOUT_SPAWN_ROADS.index(road) < n_in:
                    d = free_spaces.pop(int(rng.randint(len(free_spaces))))
                    space = d + 1
                else:
                    d = n_space + int(rng.randint(n_in))
                dest = park_dests[d]
            elif fixed_destination:
                from metadrive_ped_amd.mapgen.tables import destination_for
                dest = destination_for(pg, seed, (road[0], road[1], li))
            elif cfg.get("exclude_own_road"):     # MAIntersectionSpawnManager with disable_u_turn (marl_intersection.py:78-85)
                others = [d_ for r_, d_ in zip(ROUNDABOUT_SPAWN_ROADS, dests) if tuple(r_) != tuple(road)]
                dest = others[int(rng.randint(len(others)))]
            else:
                dest = dests[int(rng.randint(len(dests)))]
            pos = lane.position(long, lat)
            h = wrap_to_pi(lane.heading_theta_at(long))
            sh = self.shape[a]
            sh["cx"], sh["cy"], sh["c"], sh["s"] = pos[0], pos[1], math.cos(h), math.sin(h)
            sh["hl"], sh["hw"] = length / 2, width / 2
            sh["flags"] = abi.KIND_VEHICLE | abi.F_ALIVE | abi.F_AGENT
            d = self.dyn[a]
            d["heading"], d["last_x"], d["last_y"], d["last_c"], d["last_s"] = h, pos[0], pos[1], sh["c"], sh["s"]
            self.param[a] = prm
            if classes is not None:     # random_agent_model: every agent its own class (agent_manager.py:41)
                v = classes[int(rng.randint(len(classes)))]
                self.param[a] = v[:8].view(abi.PARAM_DT)[0]
                sh["hl"], sh["hw"] = v[8], v[9]
            nodes, roads, n, fin = route_arrays(mt, (road[0], road[1], li), dest)
            self.route_nodes[a], self.route_roads[a], self.final_lane[a] = nodes, roads, fin
            nv = self.nav[a]
            nv["lane"] = mt.lane_id[(road[0], road[1], li)]
            nv["ck0"], nv["ck1"] = (0, 1) if n > 2 else (0, 0)
            nv["route_len"] = n
            nv["toll_entry"] = space     # parking-lot env: the space this agent holds (+1), see md_lifecycle_env
            if cfg.get("agent_policy") == "IDMPolicy":
                # IDMPolicy.__init__ (policy/idm_policy.py:225-233): overtake_timer = randint(0, LANE_CHANGE_FREQ); the pre-drawn
                # randint(0, 25) values its move_to_next_road consumes (:285)
                nv["timer"] = int(rng.randint(0, 50))
                self.idm_rand[a] = [int(rng.randint(0, 25)) for _ in range(abi.MD_IDM_RAND)]
        # free slots keep the vehicle's dimensions / parameters so that a respawn only rewrites the pose
        for a in range(len(chosen), A):
            self.shape["hl"][a], self.shape["hw"][a] = length / 2, width / 2
        for a in range(len(chosen), A):
            self.shape[a]["flags"] = abi.KIND_VEHICLE      # a free agent slot: what the lifecycle hands to the next spawn
        # static bodies of the map (toll booths): slots from the top, like the props of the single-agent scenes
        top = cap
        for blk in pg.blocks:
            for lane, pos, heading in getattr(blk, "buildings", ()):
                top -= 1
                if top < A:
                    raise ValueError("mover_capacity {} leaves no slot for the map's buildings".format(cap))
                sh = self.shape[top]
                h = wrap_to_pi(heading)
                sh["cx"], sh["cy"], sh["c"], sh["s"] = pos[0], pos[1], math.cos(h), math.sin(h)
                sh["hl"], sh["hw"] = blk.BUILDING_LENGTH / 2.0, lane.width / 2.0
                sh["flags"] = abi.KIND_BUILDING | abi.F_ALIVE | abi.F_STATIC
                sh["aux"] = mt.lane_id[tuple(lane.index)]
                self.dyn[top]["heading"] = h
                self.n_props += 1

    def trim(self, cap):
        pass

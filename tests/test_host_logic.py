"""Host-side logic: config contract, spaces, scene construction, static grid."""
import math
import numpy as np
import pytest

from helpers import make_cfg
from metadrive_ped_amd import abi


def test_config_contract(cs_dist):
    from metadrive_ped_amd.config import make_config
    cfg = make_config({})
    assert cfg["map"] == 3 and cfg["traffic_density"] == 0.1 and cfg["decision_repeat"] == 5
    assert cfg["vehicle_config"]["lidar"]["num_lasers"] == 240 and cfg["success_reward"] == 10.0
    with pytest.raises(KeyError):       # utils/config.py:23-38 unknown key
        make_config(dict(not_a_key=1))
    with pytest.raises(KeyError):
        make_config(dict(vehicle_config=dict(lidar=dict(nope=1))))
    with pytest.raises(TypeError):      # type check against the default
        make_config(dict(traffic_density="dense"))
    with pytest.raises(NotImplementedError):
        make_config(dict(use_render=True))
    assert make_config(dict(map="SCS"))["map_config"]["type"] == "block_sequence"
    assert make_config(dict(map=5))["map_config"]["config"] == 5


def test_unbuilt_block_type_fails_loudly():
    """Every block type of the default distribution, Merge / Split, Bidirection, ParkingLot and TollGate are built; the
    forks raise the reference's own ValueError (they are broken there: pgblock/fork.py:27) instead of silently changing the map.
    A parking lot after a three-lane road fails like the reference's assertion."""
    from collections import OrderedDict
    from metadrive_ped_amd.mapgen.pg import BLOCK_TYPE_DISTRIBUTION_V2, BlockDist, PGMap
    for seed in range(10):
        PGMap(seed)
    d = OrderedDict((k, 0.0) for k in BLOCK_TYPE_DISTRIBUTION_V2)
    d["TollGate"] = 1.0
    assert [b.ID for b in PGMap(0, block_dist=BlockDist(d)).blocks] == ["I", "$", "$", "$"]
    d["TollGate"], d["InFork"] = 0.0, 1.0
    with pytest.raises(ValueError, match="Bug exists in this block"):     # the reference's own error (pgblock/fork.py:27, :172)
        PGMap(0, block_dist=BlockDist(d))
    with pytest.raises(AssertionError, match="must be 1 in each direction"):
        PGMap(0, generate_type="block_sequence", generate_config="P")            # default lane_num = 3
    assert [b.ID for b in PGMap(0, lane_num=1, generate_type="block_sequence", generate_config="P").blocks] == ["I", "P"]


def test_spaces_and_env_surface():
    from metadrive_ped_amd.envs import BatchedMetaDriveEnv
    env = BatchedMetaDriveEnv(dict(num_envs=3, map="S"))
    assert env.observation_space.shape == (259, ) and env.observation_space.dtype == np.float32
    assert env.action_space.shape == (2, ) and env.action_space.contains(env.action_space.sample())
    env2 = BatchedMetaDriveEnv(dict(num_envs=1, map="S", vehicle_config=dict(lidar=dict(num_lasers=0))))
    assert env2.observation_space.shape == (19, )   # config 1 of BASELINE: lidar off -> 19 dims
    with pytest.raises(RuntimeError):
        env.step(np.zeros((3, 2), np.float32))      # step before reset


def test_discrete_action_tables():
    """The reference's known answers for the action grids (tests/test_functionality/test_discrete_action.py:
    48-56,78-87): Discrete(15) with steering_dim 3 / throttle_dim 5, and MultiDiscrete([3, 5])."""
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.envs.metadrive_env import discrete_to_continuous, make_action_space
    from metadrive_ped_amd.envs.spaces import Discrete, MultiDiscrete
    cfg = make_config(dict(discrete_action=True, use_multi_discrete=False, discrete_steering_dim=3,
                           discrete_throttle_dim=5, action_check=True))
    sp = make_action_space(cfg)
    assert isinstance(sp, Discrete) and sp.n == 15 and sp.contains(sp.sample())
    got = discrete_to_continuous(torch, cfg, torch.tensor([0, 1, 2, 7, 14]), (5, ), "cpu")
    assert got.tolist() == [[-1, -1], [0, -1], [1, -1], [0, 0], [1, 1]]
    with pytest.raises(AssertionError):
        discrete_to_continuous(torch, cfg, torch.tensor([15]), (1, ), "cpu")
    with pytest.raises(TypeError):
        discrete_to_continuous(torch, cfg, torch.tensor([0.5]), (1, ), "cpu")
    cfg = make_config(dict(discrete_action=True, use_multi_discrete=True, discrete_steering_dim=3,
                           discrete_throttle_dim=5, action_check=True))
    sp = make_action_space(cfg)
    assert isinstance(sp, MultiDiscrete) and sp.shape == (2, ) and all(sp.nvec == (3, 5)) and sp.contains(sp.sample())
    got = discrete_to_continuous(torch, cfg, torch.tensor([[0, 0], [1, 0], [2, 0], [1, 2], [2, 4]]), (5, ), "cpu")
    assert got.tolist() == [[-1, -1], [0, -1], [1, -1], [0, 0], [1, 1]]
    # default grid 5 x 5 (base_env.py:74-77)
    cfg = make_config(dict(discrete_action=True))
    assert make_action_space(cfg).n == 25
    assert discrete_to_continuous(torch, cfg, torch.tensor([12]), (1, ), "cpu").tolist() == [[0.0, 0.0]]


def test_scene_routes_and_traffic(cs_dist):
    from metadrive_ped_amd.engine import HostScene
    h = HostScene(make_cfg(cs_dist, num_envs=6, num_scenarios=6))
    cap = h.cap
    shape = h.state["shape0"].reshape(6, cap)
    nav = h.state["nav0"].reshape(6, cap)
    for e in range(6):
        t = h.map_tables[h.world.arrays["env_map"][e]]
        assert shape["flags"][e, 0] == (abi.KIND_VEHICLE | abi.F_ALIVE | abi.F_AGENT)
        # agent spawns 5 m into its lane with zero lateral offset (envs/base_env.py:139-140)
        lane = t.lane_objs[nav["lane"][e, 0]]
        s, lat = lane.local_coordinates((shape["cx"][e, 0], shape["cy"][e, 0]))
        assert abs(s - 5.0) < 1e-4 and abs(lat) < 1e-4
        rn = h.state["route_nodes"].reshape(6, cap, -1)[e, 0]
        k = nav["route_len"][e, 0]
        assert t.node_names[rn[0]] == ">" and k >= 3 and (rn[k:] == -1).all()
        n_tr = ((shape["flags"][e] & abi.F_PENDING) != 0).sum()
        assert n_tr == h.scenes[h.seeds[e]].n_traffic
        for j in range(1, cap):
            if shape["flags"][e, j] & abi.F_ALIVE:
                # traffic sits on a 10 m slot of a positive lane of the block that triggers it
                lane = t.lane_objs[nav["lane"][e, j]]
                s, lat = lane.local_coordinates((shape["cx"][e, j], shape["cy"][e, j]))
                assert abs(s / 10 - round(s / 10)) < 1e-4 and abs(lat) < 1e-4
                assert nav["trigger_order"][e, j] >= 1 and 0 <= nav["timer"][e, j] < 50


def test_static_grid_is_conservative(cs_dist):
    """Every lane hull / quad AABB is listed in every cell it touches (what makes grid culling exact)."""
    from metadrive_ped_amd.engine import HostScene
    h = HostScene(make_cfg(cs_dist, num_envs=2, num_scenarios=2))
    for t in h.map_tables:
        g = t.grid[0]
        nx, ny, x0, y0, inv = int(g["nx"]), int(g["ny"]), float(g["x0"]), float(g["y0"]), float(g["inv_cell"])

        def cells_of(bx0, by0, bx1, by1):
            gx0, gx1 = int(np.floor((bx0 - x0) * inv)), int(np.floor((bx1 - x0) * inv))
            gy0, gy1 = int(np.floor((by0 - y0) * inv)), int(np.floor((by1 - y0) * inv))
            return [(gx, gy) for gy in range(gy0, gy1 + 1) for gx in range(gx0, gx1 + 1)]

        def listed(item, gx, gy):
            c = gy * nx + gx
            return item in t.cell_items[t.cell_start[c]:t.cell_start[c + 1]]

        for k, r in enumerate(t.lanes):
            for gx, gy in cells_of(r["x0"], r["y0"], r["x1"], r["y1"]):
                assert 0 <= gx < nx and 0 <= gy < ny and listed(k, gx, gy)
        for k in range(0, len(t.quads), 7):
            q = t.quads[k].reshape(4, 2)
            for gx, gy in cells_of(q[:, 0].min(), q[:, 1].min(), q[:, 0].max(), q[:, 1].max()):
                assert listed(~k, gx, gy)
        # lane items of a cell are ascending (tie-break of the localisation scan)
        for c in range(nx * ny):
            it = t.cell_items[t.cell_start[c]:t.cell_start[c + 1]]
            ln = it[it >= 0]
            assert (np.diff(ln) > 0).all()


def test_line_quads_follow_reference_geometry(cs_dist):
    from metadrive_ped_amd.engine import HostScene
    h = HostScene(make_cfg(cs_dist, num_envs=1, num_scenarios=1, start_seed=1))
    t = h.map_tables[0]
    kinds = set(t.quad_kind.tolist())
    assert {abi.Q_LINE_WHITE_CONT, abi.Q_LINE_YELLOW_CONT, abi.Q_LINE_BROKEN, abi.Q_SIDEWALK} <= kinds
    q = t.quads[t.quad_kind == abi.Q_LINE_BROKEN].reshape(-1, 4, 2)
    w = np.linalg.norm(q[:, 1] - q[:, 2], axis=1)
    ln = np.linalg.norm(q[:, 0] - q[:, 1], axis=1)
    np.testing.assert_allclose(np.minimum(w, ln), 0.075, atol=1e-5)   # 2 * LANE_LINE_WIDTH/4
    # STRIPE_LENGTH stripes; the LAST stripe of a line runs to length - 1.5 (pg_block.py:268-269) so it is longer
    assert np.median(np.maximum(w, ln)) == pytest.approx(1.5, abs=1e-4) and np.all(np.maximum(w, ln) < 4.5 + 1e-4)


def test_random_lane_width_num_and_inverse_traffic():
    """random_lane_width / random_lane_num (PGMapManager.add_random_to_map, manager/pg_map_manager.py:68-74: the
    manager's stream re-seeded with the scenario index: rand() for the width, then randint for the count) and
    need_inverse_traffic (traffic_manager.py:242-245: oncoming traffic on S / C / r / R blocks)."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.rng import get_np_random
    E = 12
    base = dict(num_envs=E, num_scenarios=E, traffic_density=0.3)
    host = HostScene(make_config(dict(base, random_lane_width=True, random_lane_num=True)))
    for s in host.seeds:
        pg = host.scenes[s].tables.pg_map
        rng = get_np_random(s)
        assert pg.lane_width == pytest.approx(rng.rand() * 1.5 + 3.0) and 3.0 <= pg.lane_width <= 4.5
        assert pg.lane_num == int(rng.randint(2, 4)) and pg.lane_num in (2, 3)
    assert len({host.scenes[s].tables.pg_map.lane_num for s in host.seeds}) == 2
    only_w = HostScene(make_config(dict(base, random_lane_width=True)))
    assert all(only_w.scenes[s].tables.pg_map.lane_num == 3 for s in only_w.seeds)
    # inverse traffic: some traffic vehicles start on negative roads of the simple blocks
    inv = HostScene(make_config(dict(base, need_inverse_traffic=True)))
    plain = HostScene(make_config(dict(base)))
    def oncoming_on_simple_blocks(host):
        n = 0
        for s in host.seeds:
            sc = host.scenes[s]
            mt = sc.tables
            for j in range(1, 1 + sc.n_traffic):
                lane_id = sc.nav["lane"][j]
                if not mt.roads[mt.lanes[lane_id]["road"]]["negative"]:
                    continue
                blk = [b for b in mt.pg_map.blocks if any(l is mt.lane_objs[lane_id] for _, _, ls in b.net.roads() for l in ls)]
                n += bool(blk and blk[0].ID in ("S", "C", "r", "R"))
        return n

    assert oncoming_on_simple_blocks(inv) > 0 and oncoming_on_simple_blocks(plain) == 0
    assert sum(inv.scenes[s].n_traffic for s in inv.seeds) > sum(plain.scenes[s].n_traffic for s in plain.seeds)


def test_varying_dynamics_env_draws_and_behaviour():
    """VaryingDynamicsEnv (envs/varying_dynamics_env.py:14-60): per scenario seed the agent manager's stream gives
    one uniform per randomised parameter in config order BEFORE the agents are created; the vehicle then takes
    them over its sampled parameters.  Same seed -> same dynamics; a stronger engine accelerates harder (oracle)."""
    import oracle_binding as ob
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.envs import BatchedVaryingDynamicsEnv
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.rng import get_np_random
    cfg = BatchedVaryingDynamicsEnv.default_config()
    assert cfg["vehicle_config"]["vehicle_model"] == "varying_dynamics"
    ranges = cfg["random_dynamics"]
    assert list(ranges) == ["max_engine_force", "max_brake_force", "wheel_friction", "max_steering", "mass"]
    E = 8
    cfg.update(num_envs=E, num_scenarios=E, start_seed=40, traffic_density=0.0, map="SS")
    cfg["random_dynamics"]["wheel_friction"] = None          # not randomised: keeps the class value, draws nothing
    cfg["random_dynamics"]["max_steering"] = (35, 35)        # degenerate range: the value, no draw
    h = HostScene(cfg)
    for e, s in enumerate(h.seeds):
        rng = get_np_random(s)
        want = {k: float(rng.uniform(*ranges[k])) for k in ("max_engine_force", "max_brake_force", "mass")}
        got = h.scenes[s].vehicle_cfgs[0]
        for k, v in want.items():
            assert got[k] == v and ranges[k][0] <= v <= ranges[k][1]
        assert got["max_steering"] == 35 and abs(got["wheel_friction"] - 0.9) < 1e-6
        p = h.state["param0"][e * h.cap] if "param0" in h.state else h.scenes[s].param[0]
        assert abs(float(p["accel_gain"]) - 4.0 * want["max_engine_force"] / want["mass"]) < 1e-4 * float(p["accel_gain"])
        assert abs(float(p["max_steer"]) - math.radians(35)) < 1e-6
    h2 = HostScene(cfg)
    assert all(h2.scenes[s].vehicle_cfgs[0] == h.scenes[s].vehicle_cfgs[0] for s in h.seeds)
    # full throttle for 2 s: speed ordering follows accel_gain (below the speed limit)
    o = ob.OracleWorld(h)
    o.reset()
    for _ in range(20):
        o.step(np.tile(np.array([0.0, 1.0], np.float32), (E, 1, 1)))
    speed = o.state["dyn"]["speed"].reshape(E, -1)[:, 0]
    gain = np.array([float(h.scenes[s].param[0]["accel_gain"]) for s in h.seeds])
    slow = speed < 20.0
    assert slow.sum() >= 3 and (np.argsort(speed[slow]) == np.argsort(gain[slow])).all()
    with pytest.raises(KeyError):
        HostScene(make_config(dict(random_dynamics=dict(wheel_radius=(1, 2)))))
    with pytest.raises(NotImplementedError):
        from metadrive_ped_amd.envs import BatchedMultiAgentRoundaboutEnv
        BatchedMultiAgentRoundaboutEnv(dict(random_dynamics=dict(mass=(300, 3000))))


def test_sub_batches_hold_the_same_environments_as_the_whole_batch():
    """envs/pipeline.py: sub-batch k of S is the env range [k E/S, (k+1) E/S) of the whole batch -- same seeds, same
    maps, same reset snapshot -- so double-buffered stepping changes the schedule, not the results."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.envs import BatchedMetaDriveEnv
    from metadrive_ped_amd.envs.pipeline import SubBatchedEnvs
    E, S = 12, 3
    user = dict(num_envs=E, num_scenarios=E, start_seed=7, env_seed_offset=24, traffic_density=0.2, mover_capacity=48)
    whole = HostScene(make_config(user))
    sub = SubBatchedEnvs(BatchedMetaDriveEnv, user, sub_batches=S)
    hosts = sub.build_host()
    assert [h.E for h in hosts] == [E // S] * S
    assert sum((h.seeds for h in hosts), []) == whole.seeds
    n = E // S * whole.cap
    for k, h in enumerate(hosts):
        for name in ("shape0", "dyn0", "nav0", "pid0", "param"):
            assert h.state[name].tobytes() == whole.state[name][k * n:(k + 1) * n].tobytes(), name
    with pytest.raises(ValueError):
        SubBatchedEnvs(BatchedMetaDriveEnv, user, sub_batches=5)


def test_reference_test_configs_are_accepted_verbatim():
    """Configs copied from the reference's own tests (test_pedestrian.py:8-36, test_traffic_light.py:8-22,
    test_marl_infinite_agents.py:5-15): rendering / debug / camera keys are inert here and accepted, behavioural keys of
    subsystems that are not built are rejected loudly, unknown keys raise KeyError like utils/config.py."""
    from metadrive_ped_amd.config import make_config
    ped = {"num_scenarios": 1, "traffic_density": 0., "traffic_mode": "hybrid", "start_seed": 22, "debug": False,
           "manual_control": False, "use_render": False, "decision_repeat": 5, "need_inverse_traffic": False, "norm_pixel": True,
           "map": "X", "random_traffic": False, "random_lane_width": True, "driving_reward": 1.0, "force_destroy": False,
           "window_size": (2400, 1600), "vehicle_config": {"enable_reverse": False}}
    c = make_config(ped)
    assert c["window_size"] == (2400, 1600) and c["traffic_mode"] == "hybrid" and c["random_lane_width"] is True
    light = {"num_scenarios": 1, "traffic_density": 0., "traffic_mode": "hybrid", "manual_control": False, "use_render": False,
             "debug": False, "debug_static_world": False, "map": "X", "window_size": (1200, 800),
             "vehicle_config": {"enable_reverse": True, "show_dest_mark": True}}
    c = make_config(light)
    assert c["vehicle_config"]["enable_reverse"] is True and c["vehicle_config"]["show_dest_mark"] is True
    for bad in (dict(manual_control=True), dict(record_episode=True), dict(use_render=True), dict(image_observation=True),
                dict(vehicle_config=dict(spawn_position_heading=((0, 0), 0))), dict(vehicle_config=dict(light=True)), dict(replay_episode="x.pkl")):
        with pytest.raises(NotImplementedError):
            make_config(bad)
    with pytest.raises(KeyError):
        make_config(dict(vehicle_config=dict(warp_drive=True)))


def test_spawn_velocity_starts_the_episode_rolling():
    """vehicle_config.spawn_velocity (base_vehicle.py:371-372): the agent leaves the reset with that velocity (its component
    along the heading; world axes, or the car's own with spawn_velocity_car_frame) -- also after every auto-reset."""
    import oracle_binding as ob
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    E = 3
    h = HostScene(make_config(dict(num_envs=E, num_scenarios=E, traffic_density=0.0, horizon=20,
                                   vehicle_config=dict(spawn_velocity=[6.0, 0.0]))))
    o = ob.OracleWorld(h)
    o.reset()
    x0 = o.state["shape"]["cx"].reshape(E, -1)[:, 0].copy()
    assert np.allclose(o.state["dyn"]["speed"].reshape(E, -1)[:, 0], 6.0)
    o.step(np.zeros((E, 1, 2), np.float32))
    assert (o.state["shape"]["cx"].reshape(E, -1)[:, 0] - x0 > 0.5).all()          # 0.1 s at ~6 m/s
    for _ in range(21):                                                             # horizon 20 -> truncation -> auto-reset
        o.step(np.zeros((E, 1, 2), np.float32))
    assert (o.state["nav"]["steps"].reshape(E, -1)[:, 0] <= 2).all()
    assert (o.state["dyn"]["speed"].reshape(E, -1)[:, 0] > 5.0).all()


def test_varying_dynamics_vehicle_takes_size_and_mass_from_its_config():
    """VaryingDynamicsVehicle.WIDTH / LENGTH / MASS (vehicle_type.py:168-187; tests/test_component/
    test_varying_dynamics_vehicle.py): given in the vehicle config they replace the default car's; any other vehicle class
    ignores them."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    for width in (1.852, 2, 3, 4):
        for length in (4.515, 6, 9, 13):
            h = HostScene(make_config(dict(num_envs=1, traffic_density=0.0, vehicle_config=dict(
                vehicle_model="varying_dynamics", width=width, length=length, mass=2200))))
            sh = h.state["shape0"][0]
            assert abs(2 * sh["hw"] - width) < 1e-6 and abs(2 * sh["hl"] - length) < 1e-6
            p = h.state["param"][0]
            cfgv = h.scenes[h.seeds[0]].vehicle_cfgs[0]
            assert abs(p["accel_gain"] - 4.0 * cfgv["max_engine_force"] / 2200) < 1e-5
    h = HostScene(make_config(dict(num_envs=1, traffic_density=0.0, vehicle_config=dict(width=3, length=9, mass=2200))))
    assert abs(2 * h.state["shape0"][0]["hl"] - 4.515) < 1e-6 and abs(2 * h.state["shape0"][0]["hw"] - 1.852) < 1e-6


def test_lazy_info_behaves_like_a_dict_and_computes_on_first_read():
    from metadrive_ped_amd.envs.spaces import LazyInfo
    calls = []
    info = LazyInfo(dict(velocity=1.5), dict(crash=lambda: calls.append("crash") or True, out_of_road=lambda: calls.append("oor") or False))
    assert "velocity" in info and "crash" in info and "nope" not in info and len(info) == 3
    assert calls == []                                           # nothing derived yet
    assert info["crash"] is True and info["crash"] is True and calls == ["crash"]     # computed once, then cached
    assert sorted(info.keys()) == ["crash", "out_of_road", "velocity"]
    assert dict(info.items()) == dict(velocity=1.5, crash=True, out_of_road=False) and calls == ["crash", "oor"]
    assert info.get("nope", 7) == 7 and info.get("velocity") == 1.5
    with pytest.raises(KeyError):
        info["nope"]
    info["extra"] = 3                                            # still an ordinary dict for writes
    assert info["extra"] == 3 and len(info) == 4


def test_vehicle_config_destination_is_honoured():
    """vehicle_config.destination (node_network_navigation.py:54-56): the agent's route ends at the node the user names
    instead of the far socket of the last block; an unreachable node is an error."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    base = dict(num_envs=1, start_seed=3, map="SSS", traffic_density=0.0)
    h0 = HostScene(make_config(dict(base)))
    mt = h0.map_tables[0]
    full = [mt.node_names[i] for i in h0.state["route_nodes"].reshape(1, -1, abi.MD_ROUTE_LEN)[0, 0] if i >= 0]
    assert len(full) >= 4
    mid = full[-2]
    h1 = HostScene(make_config(dict(base, vehicle_config=dict(destination=mid))))
    short = [h1.map_tables[0].node_names[i] for i in h1.state["route_nodes"].reshape(1, -1, abi.MD_ROUTE_LEN)[0, 0] if i >= 0]
    assert short == full[:-1] and int(h1.state["nav0"]["route_len"][0]) == len(full) - 1
    with pytest.raises(ValueError):
        HostScene(make_config(dict(base, vehicle_config=dict(destination="no_such_node"))))


def test_examples_and_tools_compile():
    """The runnable scripts need an MI355X; here they at least have to be valid Python."""
    import glob
    import os
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "examples", "*.py")) + glob.glob(os.path.join(root, "tools", "*.py")) + \
        [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    assert len(files) >= 12
    for f in files:
        with open(f) as fh:
            ast.parse(fh.read(), filename=f)


def test_random_traffic_redraws_the_traffic_per_reset(cs_dist):
    """random_traffic=True (manager/traffic_manager.py:335-337: the traffic manager is not re-seeded at reset): another
    traffic layout for every reset epoch, the map and the agent unchanged; off: identical scenes."""
    from helpers import make_cfg
    from metadrive_ped_amd.engine import HostScene
    a = HostScene(make_cfg(cs_dist, num_envs=6, num_scenarios=6, traffic_density=0.3, random_traffic=True, traffic_epoch=0))
    b = HostScene(make_cfg(cs_dist, num_envs=6, num_scenarios=6, traffic_density=0.3, random_traffic=True, traffic_epoch=1))
    c = HostScene(make_cfg(cs_dist, num_envs=6, num_scenarios=6, traffic_density=0.3))
    d = HostScene(make_cfg(cs_dist, num_envs=6, num_scenarios=6, traffic_density=0.3))
    assert c.state["shape0"].tobytes() == d.state["shape0"].tobytes()
    sa, sb = a.state["shape0"].reshape(6, -1), b.state["shape0"].reshape(6, -1)
    assert sa[:, 0].tobytes() == sb[:, 0].tobytes()                       # the agents
    assert sa[:, 1:].tobytes() != sb[:, 1:].tobytes()                     # the traffic
    assert a.world.arrays["lanes"].tobytes() == b.world.arrays["lanes"].tobytes()


def test_step_kernel_auto_selection(monkeypatch):
    """config["step_kernel"] = "auto": one wave per env for a large batch on few distinct maps (the reference's default
    num_scenarios = 1), the 4-wave workgroup otherwise; explicit choices are kept; the environment variable only steers "auto"."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import WAVE_KERNEL_MAX_MAPS, WAVE_KERNEL_MIN_ENVS, pick_step_kernel
    monkeypatch.delenv("MD_STEP_KERNEL", raising=False)
    big = make_config(dict(num_envs=4096, num_scenarios=1))
    assert big["step_kernel"] == "auto"
    assert pick_step_kernel(big, 1) == "wave" and pick_step_kernel(big, WAVE_KERNEL_MAX_MAPS) == "wave"
    assert pick_step_kernel(big, WAVE_KERNEL_MAX_MAPS + 1) == "wg" and pick_step_kernel(big, 4096) == "wg"
    small = make_config(dict(num_envs=WAVE_KERNEL_MIN_ENVS - 1, num_scenarios=1))
    assert pick_step_kernel(small, 1) == "wg"
    assert pick_step_kernel(make_config(dict(num_envs=4096, step_kernel="wg")), 1) == "wg"
    assert pick_step_kernel(make_config(dict(num_envs=8, step_kernel="wave")), 8) == "wave"
    monkeypatch.setenv("MD_STEP_KERNEL", "wave")
    assert pick_step_kernel(small, 4096) == "wave"
    assert pick_step_kernel(make_config(dict(num_envs=4096, step_kernel="wg")), 1) == "wg"     # an explicit choice wins
    with pytest.raises(ValueError):
        make_config(dict(step_kernel="pm"))


def test_random_traffic_draws_share_the_maps_and_the_capacity():
    """random_traffic with auto_reset: the staged draws (engine.BatchedEngine.draw_hosts_) are built on the same maps and agents with
    other traffic; traffic_draws=1 keeps the one-draw-per-reset() behaviour and warns."""
    import warnings
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine, HostScene
    from metadrive_ped_amd.envs.metadrive_env import BatchedMetaDriveEnv
    cfg = make_config(dict(num_envs=5, num_scenarios=5, map=2, traffic_density=0.3, start_seed=3, random_traffic=True, build_workers=1))

    class HostOnly(BatchedEngine):          # the host part of build() without a device
        def __init__(self, cfg):
            self.cfg, self.host = cfg, None
    eng = HostOnly(cfg)
    assert eng.n_traffic_draws() == 4
    h0 = HostScene(cfg)
    h1 = HostScene(dict(cfg, traffic_epoch=BatchedEngine.DRAW_EPOCH_STRIDE))
    for k in h0.world.arrays:
        assert np.array_equal(np.ascontiguousarray(h0.world.arrays[k]).view(np.uint8), np.ascontiguousarray(h1.world.arrays[k]).view(np.uint8)), k
    a0, a1 = h0.state["shape0"].reshape(5, -1), h1.state["shape0"].reshape(5, -1)
    assert np.array_equal(np.ascontiguousarray(a0[:, 0]).view(np.uint8), np.ascontiguousarray(a1[:, 0]).view(np.uint8))   # the agents' spawn is not traffic
    assert not np.array_equal(np.ascontiguousarray(a0).view(np.uint8), np.ascontiguousarray(a1).view(np.uint8))
    assert HostOnly(make_config(dict(cfg, auto_reset=False))).n_traffic_draws() == 1
    assert HostOnly(make_config(dict(cfg, random_traffic=False))).n_traffic_draws() == 1
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        BatchedMetaDriveEnv(dict(num_envs=2, random_traffic=True))                     # draws staged: nothing to warn about
    with pytest.warns(UserWarning, match="traffic_draws=1"):
        BatchedMetaDriveEnv(dict(num_envs=2, random_traffic=True, traffic_draws=1))

"""GPU parity: the HIP path (through the C-ABI, libmdstep.so) against the CPU oracle on the same
seeded scenes -- bit-exact on every state array (poses, flags, obs, reward, done)."""
import numpy as np
import pytest

from helpers import assert_state_equal, make_cfg, scripted_actions

pytestmark = pytest.mark.gpu


def _engine_and_oracle(cs_dist, **kw):
    import torch
    from metadrive_ped_amd.engine import BatchedEngine
    import oracle_binding as ob
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    cfg = make_cfg(cs_dist, **kw)
    eng = BatchedEngine(cfg)
    orc = ob.OracleWorld(eng.host)
    return eng, orc


def test_reset_parity(cs_dist):
    eng, orc = _engine_and_oracle(cs_dist, num_envs=32, num_scenarios=32)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="after reset")
    obs = eng.obs.cpu().numpy()
    assert obs.shape == (32, 1, 259)
    assert np.all(obs >= 0) and np.all(obs <= 1)


@pytest.mark.parametrize("auto_reset", [False, True])
def test_step_parity_rollout(cs_dist, auto_reset):
    import torch
    E = 48
    eng, orc = _engine_and_oracle(cs_dist, num_envs=E, num_scenarios=E, auto_reset=auto_reset, horizon=150)
    eng.reset()
    orc.reset()
    for t in range(220):
        a = scripted_actions(E, 1, t)
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 10 == 0 or t > 200:
            assert_state_equal(eng.download_state(), orc.state, where="step %d" % t)
    st = eng.download_state()
    assert_state_equal(st, orc.state, where="final")
    # the rollout must have exercised the interesting paths
    fl = st["flags"].reshape(E, -1)[:, 0]
    assert (fl != 0).any()


def test_single_phase_entry_points(cs_dist):
    """Every md_* phase entry point on its own against ref_* (SURVEY 8b list)."""
    import torch
    import ctypes as C
    E = 24
    eng, orc = _engine_and_oracle(cs_dist, num_envs=E, num_scenarios=E, auto_reset=False)
    eng.reset()
    orc.reset()
    for t in range(60):
        a = scripted_actions(E, 1, t, seed=3)
        eng.action[:, :1, :] = torch.from_numpy(a).to(eng.device)
        orc.state["action"].reshape(E, -1, 2)[:, :1, :] = a
        for md, ref in (("md_idm", "ref_idm"), ("md_integrate", "ref_integrate"), ("md_localize", "ref_localize"),
                        ("md_contacts", "ref_contacts"), ("md_traffic_after_step", "ref_traffic_after_step"),
                        ("md_observe", "ref_observe")):
            eng.call(md)
            orc.call(ref)
            if t % 15 == 0:
                assert_state_equal(eng.download_state(), orc.state, where="t=%d after %s" % (t, md))
        out = torch.zeros(E, 240, device=eng.device)
        eng.lidar(out, 240, 0)
        ref = orc.lidar()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32)), "lidar t=%d" % t
    assert_state_equal(eng.download_state(), orc.state, where="final")


def test_safe_env_rollout_parity(cs_dist):
    """SafeMetaDrive config (props on the road, crashes do not terminate): cones / tripods / barriers /
    broken-down vehicles are seen by the lidar, hit by the contact test and scanned by the IDM."""
    import torch
    from metadrive_ped_amd import abi
    E = 40
    eng, orc = _engine_and_oracle(cs_dist, num_envs=E, num_scenarios=E, accident_prob=0.8, traffic_density=0.05,
                                  crash_vehicle_done=False, crash_object_done=False, horizon=200)
    kinds = eng.host.state["shape0"]["flags"] & abi.KIND_MASK
    assert (kinds == abi.KIND_CONE).sum() > 50 and (kinds == abi.KIND_BARRIER).sum() > 3
    eng.reset()
    orc.reset()
    hit_obj = False
    for t in range(260):
        a = scripted_actions(E, 1, t, seed=11)
        a[:, :, 0] *= 0.3
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 20 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="safe step %d" % t)
        hit_obj |= bool(((orc.state["flags"].reshape(E, -1)[:, 0] & abi.FL_CRASH_OBJECT) != 0).any())
    assert_state_equal(eng.download_state(), orc.state, where="safe final")
    assert hit_obj
    assert (orc.state["step_info"][:, 5] > 0).any()          # total_cost accumulates


def test_side_and_lane_line_detectors_in_obs(cs_dist):
    """side_detector / lane_line_detector on (reference test configs: tests/test_env/test_metadrive_env.py:26-33):
    the two clouds take the place of the border-distance / lateral dims; md_line_detector vs ref_line_detector."""
    import torch
    E = 16
    eng, orc = _engine_and_oracle(cs_dist, num_envs=E, num_scenarios=E,
                                  vehicle_config=dict(side_detector=dict(num_lasers=12, distance=50),
                                                      lane_line_detector=dict(num_lasers=6, distance=20),
                                                      lidar=dict(num_lasers=120, distance=50)))
    assert eng.obs_dim == 12 + 6 + 6 + 10 + 120
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="detector reset")
    for t in range(60):
        a = scripted_actions(E, 1, t, seed=5)
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
    st = eng.download_state()
    assert_state_equal(st, orc.state, where="detector final")
    side = st["obs"][:, :12]
    assert (side < 1.0).any() and (side >= 0).all()           # the road border is within 50 m of a car on the road
    ll = st["obs"][:, 18:24]
    assert (ll < 1.0).any()
    # the step above took both clouds from ONE launch (md_line_detectors); the single-fan entry point, called once per detector on
    # the same poses, gives the same bits
    both = eng.obs.clone()
    h = eng.host
    vc = eng.cfg["vehicle_config"]
    eng.obs[:, :, :12] = -1.0
    eng.obs[:, :, 18:24] = -1.0
    eng.line_detector(eng._side_beams, h.n_side, float(vc["side_detector"]["distance"]), eng.SIDE_MASK, eng.state_dev["obs"], h.obs_dim, h.obs_base)
    eng.line_detector(eng._ll_beams, h.n_ll, float(vc["lane_line_detector"]["distance"]), eng.LANE_LINE_MASK, eng.state_dev["obs"], h.obs_dim,
                      h.obs_base + h.n_side + 6)
    assert torch.equal(eng.obs.view(torch.int32), both.view(torch.int32))


def test_default_distribution_maps_rollout_parity():
    """The reference's default config (map=3, BLOCK_TYPE_DISTRIBUTION_V2: ramps, intersections, roundabouts):
    big lane tables take the non-staged kernel variant."""
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    import oracle_binding as ob
    E = 64
    cfg = make_config(dict(num_envs=E, num_scenarios=E, horizon=250))
    eng = BatchedEngine(cfg)
    assert eng.w.max_lanes > 64
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="v2 reset")
    for t in range(300):
        a = scripted_actions(E, 1, t, seed=21)
        a[:, :, 0] *= 0.4
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 25 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="v2 step %d" % t)
    assert_state_equal(eng.download_state(), orc.state, where="v2 final")


@pytest.mark.parametrize("mode", ["respawn", "hybrid"])
def test_traffic_respawn_modes_rollout_parity(mode):
    """traffic_mode respawn / hybrid (manager/traffic_manager.py:94-122): vehicles that run off the end of
    their route re-enter on a respawn lane; the slot rewrite (pose, route, PID, IDM timer) is bit-exact and the
    per-env RNG streams stay in step."""
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    import oracle_binding as ob
    E = 32
    cfg = make_config(dict(num_envs=E, num_scenarios=E, traffic_mode=mode, traffic_density=0.2, horizon=400))
    eng = BatchedEngine(cfg)
    orc = ob.OracleWorld(eng.host)
    rng0 = eng.host.state["rng"].copy()
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where=mode + " reset")
    for t in range(700):
        a = scripted_actions(E, 1, t, seed=2)
        a[:, :, 0] *= 0.05                   # agents mostly keep their lane, so episodes last and traffic cycles
        a[:, :, 1] = 0.3
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 50 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="%s step %d" % (mode, t))
    st = eng.download_state()
    assert_state_equal(st, orc.state, where=mode + " final")
    assert (st["rng"] != rng0).sum() >= 4, "respawns must have happened in several envs"


@pytest.mark.parametrize("navi", [False, True])
def test_others_block_and_detected_sets_parity(navi):
    """lidar num_others > 0: the detected sets (which body each beam hits first, united over the four sector
    waves through LDS atomics) and the others block computed from them equal the oracle's brute force."""
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    import oracle_binding as ob
    E = 40
    cfg = make_config(dict(num_envs=E, num_scenarios=E, traffic_density=0.3, horizon=200, accident_prob=0.5,
                           crash_vehicle_done=False, crash_object_done=False,
                           vehicle_config=dict(lidar=dict(num_others=4, add_others_navi=navi))))
    eng = BatchedEngine(cfg)
    assert eng.obs_dim == 19 + (32 if navi else 16) + 240
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="others reset")
    for t in range(240):
        a = scripted_actions(E, 1, t, seed=4)
        a[:, :, 0] *= 0.2
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 30 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="others step %d" % t)
    st = eng.download_state()
    assert_state_equal(st, orc.state, where="others final")
    assert (st["detected"] != 0).any()


def test_random_lanes_and_inverse_traffic_rollout_parity():
    """random_lane_width + random_lane_num + need_inverse_traffic: 2- and 3-lane maps of odd widths with oncoming
    traffic in one batch (different lane tables, LDS-staged and not, in the same launch)."""
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    import oracle_binding as ob
    E = 48
    cfg = make_config(dict(num_envs=E, num_scenarios=E, random_lane_width=True, random_lane_num=True,
                           need_inverse_traffic=True, traffic_density=0.2, horizon=250))
    eng = BatchedEngine(cfg)
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="random lanes reset")
    for t in range(300):
        a = scripted_actions(E, 1, t, seed=13)
        a[:, :, 0] *= 0.3
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 30 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="random lanes step %d" % t)
    assert_state_equal(eng.download_state(), orc.state, where="random lanes final")


@pytest.mark.parametrize("name,cfg_kw,steps", [
    # BASELINE configs[0] shape: one env, map 'S', no traffic, lidar off -> 19-dim obs
    ("c1_single_env", dict(num_envs=1, num_scenarios=1, start_seed=1010, map="S", traffic_density=0.0,
                           vehicle_config=dict(lidar=dict(num_lasers=0, distance=0))), 120),
    # maximum mover capacity (128 slots: two 64-wide chunks everywhere), dense traffic on 5-block maps
    ("max_capacity", dict(num_envs=6, num_scenarios=6, map=5, traffic_density=0.6, mover_capacity=128, horizon=300), 260),
    # maximum beam count, and beam counts that are not a multiple of the 64-wide sector
    ("beams_1024", dict(num_envs=4, num_scenarios=4, vehicle_config=dict(lidar=dict(num_lasers=1024, distance=50))), 60),
    ("beams_30", dict(num_envs=4, num_scenarios=4, vehicle_config=dict(lidar=dict(num_lasers=30, distance=50))), 60),
    ("beams_100_short_range", dict(num_envs=4, num_scenarios=4, vehicle_config=dict(lidar=dict(num_lasers=100, distance=12.5))), 60),
    # explicit capacity smaller than the auto choice would be; no traffic at all; long straight-only map
    ("no_traffic_tight_cap", dict(num_envs=8, num_scenarios=8, traffic_density=0.0, mover_capacity=8, map="SSS"), 150),
    # enable_reverse: a negative throttle is a negative engine force (the scripted actions brake / reverse half the time)
    ("reverse", dict(num_envs=16, num_scenarios=16, traffic_density=0.1, vehicle_config=dict(enable_reverse=True)), 200),
    # ParkingLot block on a one-lane-per-direction road: ~150 short roads, right-angle bends of radius 4 m; reverse allowed
    ("parking_lot", dict(num_envs=6, num_scenarios=6, start_seed=421, map="SPS", map_config=dict(lane_num=1), traffic_density=0.2,
                         vehicle_config=dict(enable_reverse=True)), 160),
    # TollGate block: toll booths (buildings) on the odd lanes; the traffic queues behind them, an agent that hits one ends
    ("toll_gate", dict(num_envs=12, num_scenarios=12, start_seed=521, map="S$S", traffic_density=0.2, random_spawn_lane_index=True), 200),
    # VaryingDynamicsEnv: extreme agent dynamics (80 deg steering, 300 kg / 3000 N, friction 0.1 ...)
    ("varying_dynamics", dict(num_envs=24, num_scenarios=24, vehicle_config=dict(vehicle_model="varying_dynamics"),
                              random_dynamics=dict(max_engine_force=(100, 3000), max_brake_force=(20, 600),
                                                   wheel_friction=(0.1, 2.5), max_steering=(10, 80), mass=(300, 3000))), 150),
])
def test_edge_configurations_parity(name, cfg_kw, steps):
    """Sizes at the limits of the ABI (MD_MAX_CAP, MD_MAX_BEAMS), degenerate ones (one env, lidar off, no traffic)
    and ragged ones (beam counts that leave a partial sector): HIP == oracle, bit for bit."""
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    import oracle_binding as ob
    cfg = make_config(cfg_kw)
    eng = BatchedEngine(cfg)
    E = eng.E
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where=name + " reset")
    for t in range(steps):
        a = scripted_actions(E, 1, t, seed=23)
        a[:, :, 0] *= 0.3
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 40 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="%s step %d" % (name, t))
    assert_state_equal(eng.download_state(), orc.state, where=name + " final")
    if name == "max_capacity":
        fl = eng.host.state["shape0"]["flags"].reshape(E, -1)
        assert ((fl & 0xF) == 1).sum(1).max() > 64, "the scene must actually use slots beyond the first 64-wide chunk"


def test_random_agent_model_rollout_parity():
    """random_agent_model with side + lane-line detectors and the others block: every optional obs section at once
    (offsets: 2 size dims | side cloud | 5 + yaw | lane-line cloud | navi 10 | others 16 | lidar)."""
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    import oracle_binding as ob
    E = 32
    cfg = make_config(dict(num_envs=E, num_scenarios=E, random_agent_model=True, traffic_density=0.2, horizon=200,
                           vehicle_config=dict(side_detector=dict(num_lasers=8, distance=50),
                                               lane_line_detector=dict(num_lasers=4, distance=20),
                                               lidar=dict(num_lasers=120, distance=50, num_others=4))))
    eng = BatchedEngine(cfg)
    assert eng.obs_dim == 2 + 8 + 6 + 4 + 10 + 16 + 120
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="agent model reset")
    for t in range(220):
        a = scripted_actions(E, 1, t, seed=31)
        a[:, :, 0] *= 0.3
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 40 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="agent model step %d" % t)
    st = eng.download_state()
    assert_state_equal(st, orc.state, where="agent model final")
    assert len(np.unique(st["obs"][:, 0])) >= 3           # several vehicle classes in the batch


@pytest.mark.parametrize("M", [4, 16, 64])
def test_lidar_microbench_cases_bit_exact(M):
    """The synthetic shape tables of tools/lidar_microbench.py (M boxes in a 100 m square around each ego, 240
    beams): md_lidar on a bare MdShape table == ref_lidar, bit for bit."""
    import ctypes as C
    import os
    import sys
    import torch
    from metadrive_ped_amd import _lib, abi
    import oracle_binding as ob
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import lidar_microbench as lm
    E, B = 512, 240
    shape, beams, cap = lm.make_case(E, M, B)
    lib = _lib.load()
    t_shape = torch.from_numpy(shape.view(np.uint8)).cuda()
    t_beams = torch.from_numpy(beams).cuda()
    out = torch.empty(E, B, device="cuda")
    w, s, k = lm.structs(abi, E, cap, B, t_shape.data_ptr(), t_beams.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.md_lidar(C.byref(w), C.byref(s), C.byref(k), C.c_void_p(out.data_ptr()), B, 0, st), "md_lidar")
    torch.cuda.synchronize()
    ref = ob.lidar_raw(shape, beams, E, cap, B, 50.0)
    assert np.array_equal(ref.view(np.uint32), out.cpu().numpy().view(np.uint32))
    assert (ref < 1.0).mean() > 0.05


@pytest.mark.parametrize("variant", ["trigger", "respawn", "marl"])
def test_spawned_participants_rollout_parity(variant):
    """Pedestrians / cyclists spawned through the engine API in mid-rollout (participants.py): every kernel variant
    (lean single-agent, respawn-mode, multi-agent) moves them, sees them with the lidar and reports crash_human exactly
    like the oracle."""
    import torch
    from metadrive_ped_amd import participants as P
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    import oracle_binding as ob
    E = 24
    if variant == "marl":
        from metadrive_ped_amd.envs import BatchedMultiAgentRoundaboutEnv
        cfg = BatchedMultiAgentRoundaboutEnv(dict(num_envs=E, num_scenarios=E, num_agents=12, mover_capacity=16, horizon=200)).config
    else:
        cfg = make_config(dict(num_envs=E, num_scenarios=E, traffic_density=0.2, traffic_mode=variant, mover_capacity=40,
                               horizon=300, crash_human_done=(variant == "trigger")))
    eng = BatchedEngine(cfg)
    A = eng.A
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    hits = 0
    for t in range(160):
        if t in (5, 60):
            # one pedestrian 12 m ahead of the first agent of every env, walking towards it; one cyclist crossing
            st = eng.download_state()
            sh = st["shape"].reshape(E, -1)
            ahead = np.stack([sh["cx"][:, 0] + 12.0 * sh["c"][:, 0], sh["cy"][:, 0] + 12.0 * sh["s"][:, 0]], 1)
            side = np.stack([sh["cx"][:, 0] + 25.0 * sh["c"][:, 0] + 6.0 * sh["s"][:, 0],
                             sh["cy"][:, 0] + 25.0 * sh["s"][:, 0] - 6.0 * sh["c"][:, 0]], 1)
            hd = np.arctan2(sh["s"][:, 0], sh["c"][:, 0])
            for world, who in ((eng, "gpu"), (orc, "cpu")):
                if who == "gpu":
                    p = eng.spawn_object("pedestrian", ahead, hd + np.pi)
                    c = eng.spawn_object("cyclist", side, hd + np.pi / 2)
                    eng.set_velocity(p, [1, 0], 1.2, in_local_frame=True)
                    eng.set_velocity(c, [1, 0], 4.0, in_local_frame=True)
                else:
                    p2 = P.spawn(orc.state, E, eng.cap, A, "pedestrian", ahead, hd + np.pi)
                    c2 = P.spawn(orc.state, E, eng.cap, A, "cyclist", side, hd + np.pi / 2)
                    P.set_velocity(orc.state, E, eng.cap, p2, [1, 0], 1.2, in_local_frame=True)
                    P.set_velocity(orc.state, E, eng.cap, c2, [1, 0], 4.0, in_local_frame=True)
            assert (p, c) == (p2, c2)
            assert_state_equal(eng.download_state(), orc.state, where="%s after spawn at %d" % (variant, t))
        a = scripted_actions(E, A, t, seed=31)
        a[:, :, 0] *= 0.2
        a[:, :, 1] = np.abs(a[:, :, 1]) * 0.6 + 0.2
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 20 == 0 or t in (6, 61):
            assert_state_equal(eng.download_state(), orc.state, where="%s step %d" % (variant, t))
        hits += int(((orc.state["flags"].reshape(E, -1)[:, :A] & 0x4) != 0).sum())
    assert_state_equal(eng.download_state(), orc.state, where=variant + " final")
    assert hits > 0, "no agent ever touched a participant: the scenario does not test crash_human"


@pytest.mark.parametrize("mode", ["trigger", "hybrid"])
def test_agent_policy_idm_rollout_parity(mode):
    """agent_policy = IDMPolicy: the agents are planned by the traffic's policy in the reference's order (decide, move,
    observe) -- fused step bit-exact with the oracle, with auto-resets, in trigger and hybrid traffic modes."""
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    import oracle_binding as ob
    E = 32
    cfg = make_config(dict(num_envs=E, num_scenarios=E, traffic_density=0.2, traffic_mode=mode, agent_policy="IDMPolicy",
                           horizon=250))
    eng = BatchedEngine(cfg)
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="idm agent reset")
    arrived = 0
    for t in range(400):
        eng.step(None)
        orc.step(None)
        if t % 25 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="idm agent %s step %d" % (mode, t))
        arrived += int(((orc.state["flags"].reshape(E, -1)[:, 0] & 0x40) != 0).sum())
    assert_state_equal(eng.download_state(), orc.state, where="idm agent final")
    sp = orc.state["dyn"]["speed"].reshape(E, -1)[:, 0]
    assert (sp > 1.0).sum() > E // 2, "the agents are not driving"


@pytest.mark.parametrize("step_kernel", ["wave", "wg"])
@pytest.mark.parametrize("name,cfg_kw,steps", [
    ("default_maps", dict(num_envs=48, num_scenarios=48, horizon=200), 260),
    ("safe", dict(num_envs=32, num_scenarios=32, accident_prob=0.8, traffic_density=0.05, crash_vehicle_done=False,
                  crash_object_done=False, horizon=200), 240),
    ("dense_many_awake", dict(num_envs=12, num_scenarios=12, map=5, traffic_density=0.5, horizon=300), 300),
    ("hybrid_idm_agent", dict(num_envs=16, num_scenarios=16, traffic_density=0.2, traffic_mode="hybrid", agent_policy="IDMPolicy",
                              horizon=250), 300),
])
def test_alternative_step_kernels_parity(step_kernel, name, cfg_kw, steps):
    """The other machine mappings of md_step for single-agent envs (MdConfig.step_kernel): "wave" = one wave per env,
    "pm" = one launch per phase with vehicles as work items (lean trigger-mode path; other configs fall back to the
    default kernel).  Same wave-level device functions, same results: bit-exact against the oracle, auto-resets included."""
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    import oracle_binding as ob
    cfg = make_config(dict(cfg_kw, step_kernel=step_kernel))
    eng = BatchedEngine(cfg)
    E = eng.E
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="%s/%s reset" % (step_kernel, name))
    for t in range(steps):
        a = scripted_actions(E, 1, t, seed=31)
        a[:, :, 0] *= 0.4
        act = None if cfg["agent_policy"] == "IDMPolicy" else torch.from_numpy(a).to(eng.device)
        eng.step(act)
        orc.step(None if act is None else a)
        if t % 40 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="%s/%s step %d" % (step_kernel, name, t))
    assert_state_equal(eng.download_state(), orc.state, where="%s/%s final" % (step_kernel, name))

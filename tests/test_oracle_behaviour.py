"""Behavioural invariants of the step path restated from the reference's integration tests
(metadrive/tests/test_functionality, test_env), run on the CPU oracle; plus analytic checks of the
geometry primitives (reference tests/test_component/test_detector_mask.py:125-147)."""
import ctypes as C
import math

import numpy as np
import pytest

from helpers import make_cfg, scripted_actions
from metadrive_ped_amd import abi
import oracle_binding as ob


def _world(cs_dist, **kw):
    from metadrive_ped_amd.engine import HostScene
    host = HostScene(make_cfg(cs_dist, **kw))
    o = ob.OracleWorld(host)
    o.reset()
    return host, o


def test_obs_in_space_and_info_contract(cs_dist):
    """obs in Box(0,1,(259,)), scalar reward per env (test_env/test_metadrive_env.py:12-71)."""
    host, o = _world(cs_dist, num_envs=8, num_scenarios=8)
    for t in range(30):
        o.step(scripted_actions(8, 1, t))
        obs = o.obs
        assert obs.shape == (8, 259) and obs.dtype == np.float32
        assert np.isfinite(obs).all() and (obs >= 0).all() and (obs <= 1).all()
        assert np.isfinite(o.state["reward"]).all()


def test_spawn_heading_diff_is_half(cs_dist):
    """heading_diff(lane) == 0.5 for a freshly spawned vehicle (test_component/test_ego_vehicle.py:21)."""
    host, o = _world(cs_dist, num_envs=4, num_scenarios=4)
    np.testing.assert_allclose(o.obs[:, 2], 0.5, atol=1e-3)


def test_full_throttle_drives_forward_and_engine_cuts_at_max_speed(cs_dist):
    host, o = _world(cs_dist, num_envs=2, num_scenarios=2, map="S", traffic_density=0.0, auto_reset=False)
    x0 = o.state["shape"]["cx"].reshape(2, -1)[:, 0].copy()
    for t in range(120):
        o.step(np.tile(np.array([0.0, 1.0], np.float32), (2, 1, 1)))
    x = o.state["shape"]["cx"].reshape(2, -1)[:, 0]
    v = o.state["dyn"]["speed"].reshape(2, -1)[:, 0]
    assert (x > x0 + 50).all()
    assert (v * 3.6 <= 80.0 + 1.5).all() and (v * 3.6 > 75).all()      # engine force is 0 above max_speed_km_h


def test_brake_stops_without_reversing(cs_dist):
    host, o = _world(cs_dist, num_envs=1, num_scenarios=1, map="S", traffic_density=0.0, auto_reset=False)
    for t in range(40):
        o.step(np.array([[[0.0, 1.0]]], np.float32))
    for t in range(60):
        o.step(np.array([[[0.0, -1.0]]], np.float32))
    assert o.state["dyn"]["speed"][0] == 0.0                                 # enable_reverse=False


def test_steering_off_road_terminates_with_penalty(cs_dist):
    """out_of_road ends the episode with -out_of_road_penalty (test_reward_cost_done.py:54-74)."""
    host, o = _world(cs_dist, num_envs=1, num_scenarios=1, map="S", traffic_density=0.0, auto_reset=False)
    done_at = None
    for t in range(200):
        o.step(np.array([[[1.0, 0.6]]], np.float32))
        fl = int(o.state["flags"][0])
        if fl & abi.FL_TERMINATED:
            done_at = t
            break
    assert done_at is not None
    assert fl & abi.FL_OUT_OF_ROAD
    assert o.state["reward"][0] == -5.0 and o.state["cost"][0] == 1.0


def test_arrive_destination_gives_success_reward(cs_dist):
    host, o = _world(cs_dist, num_envs=1, num_scenarios=1, map="S", traffic_density=0.0, auto_reset=False,
                     random_spawn_lane_index=False)
    got = False
    for t in range(400):
        o.step(np.array([[[0.0, 0.5]]], np.float32))
        fl = int(o.state["flags"][0])
        if fl & abi.FL_ARRIVE_DEST:
            assert o.state["reward"][0] == 10.0 and (fl & abi.FL_TERMINATED)
            got = True
            break
        assert not (fl & abi.FL_TERMINATED), abi.__dict__
    assert got


@pytest.mark.parametrize("truncate_as_terminate", [False, True])
def test_horizon_semantics(cs_dist, truncate_as_terminate):
    """horizon -> truncated at step == horizon; terminated only with truncate_as_terminate
    (test_functionality/test_horizon_termination.py:14-63)."""
    host, o = _world(cs_dist, num_envs=1, num_scenarios=1, map="S", traffic_density=0.0, auto_reset=False, horizon=7,
                     truncate_as_terminate=truncate_as_terminate)
    for t in range(1, 8):
        o.step(np.array([[[0.0, 0.1]]], np.float32))
        fl = int(o.state["flags"][0])
        assert bool(fl & abi.FL_TRUNCATED) == (t >= 7)
        assert bool(fl & abi.FL_TERMINATED) == (t >= 7 and truncate_as_terminate)


def test_auto_reset_restores_snapshot(cs_dist):
    host, o = _world(cs_dist, num_envs=1, num_scenarios=1, map="S", traffic_density=0.0, auto_reset=True, horizon=5)
    first_obs = o.obs.copy()
    for t in range(5):
        o.step(np.array([[[0.0, 1.0]]], np.float32))
    assert o.state["need_reset"][0] == 1
    o.step(np.array([[[0.0, 1.0]]], np.float32))     # the step after `done` returns the reset observation
    np.testing.assert_array_equal(o.obs, first_obs)
    assert o.state["reward"][0] == 0.0 and int(o.state["flags"][0]) & abi.FL_TERMINATED == 0
    assert o.state["nav"]["steps"][0] == 0


def test_trigger_traffic_wakes_up_and_moves(cs_dist):
    """Trigger-mode traffic is parked until the agent enters the block's trigger road, then drives
    (test_functionality/test_traffic_mode.py:4-32, test_policy/test_idm_policy.py:57-76)."""
    host, o = _world(cs_dist, num_envs=6, num_scenarios=6, traffic_density=0.3, auto_reset=False, crash_vehicle_done=False,
                     mover_capacity=96)
    f0 = o.state["shape"]["flags"].copy()
    x0 = o.state["shape"]["cx"].copy()
    assert ((f0 & abi.F_PENDING) != 0).sum() > 10
    woke = np.zeros_like(f0, dtype=bool)
    for t in range(150):
        o.step(np.tile(np.array([0.0, 0.7], np.float32), (6, 1, 1)))
        woke |= ((f0 & abi.F_PENDING) != 0) & ((o.state["shape"]["flags"] & abi.F_PENDING) == 0)
    assert woke.sum() > 5
    moved = np.abs(o.state["shape"]["cx"] - x0) + 0  # x is enough: blocks extend along +x first
    still_pending = (o.state["shape"]["flags"] & abi.F_PENDING) != 0
    assert (moved[still_pending] == 0).all()                                   # parked traffic never moves
    assert (np.hypot(o.state["shape"]["cx"] - x0, o.state["shape"]["cy"] - o.state["shape"]["cy"])[woke] > 0.5).mean() > 0.5


def test_dense_traffic_eventually_crashes(cs_dist):
    """crash with a vehicle happens within 500 steps at high density (test_collision.py:4-19)."""
    host, o = _world(cs_dist, num_envs=4, num_scenarios=4, map="SSS", traffic_density=1.0, auto_reset=False,
                     mover_capacity=128)
    crashed = np.zeros(4, bool)
    for t in range(500):
        o.step(np.tile(np.array([0.0, 1.0], np.float32), (4, 1, 1)))
        crashed |= (o.state["flags"].reshape(4, -1)[:, 0] & abi.FL_CRASH_VEHICLE) != 0
        if crashed.all():
            break
    assert crashed.any()


def test_determinism_same_seed_same_rollout(cs_dist):
    """Same config + seed => identical trajectories (test_scenario_randomness.py:10-67: 1e-5; here bitwise)."""
    outs = []
    for _ in range(2):
        host, o = _world(cs_dist, num_envs=4, num_scenarios=4, start_seed=7)
        for t in range(80):
            o.step(scripted_actions(4, 1, t))
        outs.append({k: v.copy() for k, v in o.state.items()})
    for k in outs[0]:
        assert outs[0][k].tobytes() == outs[1][k].tobytes(), k


# ---- analytic geometry ---------------------------------------------------------------------------
def _line_intersect(theta, pos, a, b, maximum=10000.0):
    """Ray from `pos` at angle theta against segment a-b: distance or `maximum` (analytic form of the
    reference's test helper, tests/test_component/test_detector_mask.py:125-147, restated)."""
    x0, y0 = pos
    dx, dy = math.cos(theta), math.sin(theta)
    ex, ey = b[0] - a[0], b[1] - a[1]
    den = dx * ey - dy * ex
    if abs(den) < 1e-12:
        return maximum
    t = ((a[0] - x0) * ey - (a[1] - y0) * ex) / den
    u = ((a[0] - x0) * dy - (a[1] - y0) * dx) / den
    if t >= 0 and 0 <= u <= 1:
        return t
    return maximum


def test_ray_box_matches_analytic_segments():
    lib = ob.load()
    rng = np.random.RandomState(5)
    n_hit = 0
    for _ in range(3000):
        sh = np.zeros(1, dtype=abi.SHAPE_DT)
        cx, cy = rng.uniform(-25, 25, 2)
        th = rng.uniform(-math.pi, math.pi)
        hl, hw = rng.uniform(1.5, 3), rng.uniform(0.7, 1.2)
        sh["cx"], sh["cy"], sh["c"], sh["s"], sh["hl"], sh["hw"] = cx, cy, math.cos(th), math.sin(th), hl, hw
        sh["flags"] = abi.KIND_VEHICLE | abi.F_ALIVE
        beam = rng.uniform(-math.pi, math.pi)
        got = lib.ref_ray_shape(C.c_float(0), C.c_float(0), C.c_float(50 * math.cos(beam)), C.c_float(50 * math.sin(beam)),
                                sh.ctypes.data)
        u = np.array([math.cos(th), math.sin(th)])
        v = np.array([-u[1], u[0]])
        c = np.array([cx, cy])
        corners = [c + hl * u + hw * v, c - hl * u + hw * v, c - hl * u - hw * v, c + hl * u - hw * v]
        inside = abs(np.dot(-c, u)) <= hl and abs(np.dot(-c, v)) <= hw
        d = min(_line_intersect(beam, (0, 0), corners[i], corners[(i + 1) % 4]) for i in range(4))
        want = d / 50.0 if (d <= 50.0 and not inside) else 2.0
        if want <= 1.0:
            n_hit += 1
            assert abs(got - want) < 2e-5, (got, want)
        else:
            assert got >= 1.0 or abs(d - 50.0) < 1e-3, (got, want, d)
    assert n_hit > 100, n_hit


def test_obb_overlap_symmetry_and_containment():
    lib = ob.load()
    rng = np.random.RandomState(9)
    for _ in range(2000):
        a, b = np.zeros(1, dtype=abi.SHAPE_DT), np.zeros(1, dtype=abi.SHAPE_DT)
        for s in (a, b):
            th = rng.uniform(-math.pi, math.pi)
            s["cx"], s["cy"] = rng.uniform(-6, 6, 2)
            s["c"], s["s"], s["hl"], s["hw"] = math.cos(th), math.sin(th), rng.uniform(1.5, 3), rng.uniform(0.7, 1.2)
        ab = lib.ref_obb_obb(a.ctypes.data, b.ctypes.data)
        ba = lib.ref_obb_obb(b.ctypes.data, a.ctypes.data)
        assert ab == ba
        dist = math.hypot(float(a["cx"][0] - b["cx"][0]), float(a["cy"][0] - b["cy"][0]))
        if dist < min(float(a["hw"][0]), float(b["hw"][0])):
            assert ab == 1
        if dist > math.hypot(float(a["hl"][0]), float(a["hw"][0])) + math.hypot(float(b["hl"][0]), float(b["hw"][0])):
            assert ab == 0


@pytest.mark.parametrize("mode", ["respawn", "hybrid"])
def test_traffic_respawn_modes(mode):
    """PGTrafficManager in respawn / hybrid mode (manager/traffic_manager.py:51-72,94-122,213-228): respawn mode
    starts ceil(density * slots) driving vehicles per respawn lane; a vehicle that leaves the lanes comes back
    on a respawn lane in the first half of it, at rest, with a route; the count of vehicles is conserved."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.mapgen.tables import respawn_lanes
    E = 6
    cfg = make_config(dict(num_envs=E, num_scenarios=E, traffic_mode=mode, traffic_density=0.2, horizon=5000))
    host = HostScene(cfg)
    fl0 = host.state["shape0"]["flags"].reshape(E, -1)
    is_traffic = ((fl0 & abi.KIND_MASK) == abi.KIND_VEHICLE) & ((fl0 & abi.F_AGENT) == 0)
    if mode == "respawn":
        assert not (fl0 & abi.F_PENDING).any()
        for e, seed in enumerate(host.seeds):
            lanes = respawn_lanes(host.scenes[seed].tables.pg_map)
            want = sum(int(np.ceil(0.2 * int(l.length / 10))) for l in lanes)
            assert is_traffic[e].sum() == want
    else:
        assert ((fl0 & abi.F_PENDING) != 0)[is_traffic].all()          # hybrid starts like trigger mode
    o = ob.OracleWorld(host)
    o.reset()
    st = o.state
    n0 = is_traffic.sum(1)
    last = st["rng"].copy()
    seen = 0
    for t in range(900):
        o.step(np.tile(np.array([0.0, 0.25 if mode == "hybrid" else 0.0], np.float32), (E, 1, 1)))
        changed = np.nonzero(st["rng"] != last)[0]
        last = st["rng"].copy()
        fl = st["shape"]["flags"].reshape(E, -1)
        for e in changed:
            seen += 1
            m = host.world.arrays
            p0, p1 = m["spawn_off"][e], m["spawn_off"][e + 1]
            ok = False
            nav = st["nav"].reshape(E, -1)
            dyn = st["dyn"].reshape(E, -1)
            for j in range(1, host.cap):
                if dyn["speed"][e, j] == 0.0 and nav["steps"][e, j] == 0 and (fl[e, j] & abi.F_ALIVE) and \
                        nav["lane"][e, j] in m["spawn_lane"][p0:p1] and nav["route_len"][e, j] >= 2:
                    ok = True
            assert ok, "env %d: a respawned vehicle must sit at rest on a respawn lane" % e
        alive = ((fl & abi.F_ALIVE) != 0) & ((fl & abi.KIND_MASK) == abi.KIND_VEHICLE) & ((fl & abi.F_AGENT) == 0) & \
            ((fl & abi.F_STATIC) == 0)
        if not st["need_reset"].any():
            assert (alive.sum(1) == n0).all(), "traffic count is conserved in %s mode" % mode
    assert seen > 0


def test_others_block_of_the_observation(cs_dist):
    """lidar num_others (Lidar.get_surrounding_vehicles_info, component/sensors/lidar.py:93-138): the nearest
    DETECTED vehicles, nearest first, described in the ego frame; absent entries are zeros; the cloud follows."""
    from metadrive_ped_amd.engine import HostScene
    E, K = 6, 4
    cfg = make_cfg(cs_dist, num_envs=E, num_scenarios=E, traffic_density=0.3,
                   vehicle_config=dict(lidar=dict(num_lasers=240, distance=50, num_others=K, add_others_navi=True)))
    host = HostScene(cfg)
    assert host.obs_dim == 19 + 8 * K + 240
    o = ob.OracleWorld(host)
    o.reset()
    seen = 0
    for t in range(150):
        o.step(np.tile(np.array([0.0, 0.6], np.float32), (E, 1, 1)))
        obs = o.obs
        assert (obs >= 0).all() and (obs <= 1).all()
        others = obs[:, 19:19 + 8 * K].reshape(E, K, 8)
        det = o.state["detected"].reshape(E, 2)
        sh = o.state["shape"].reshape(E, -1)
        for e in range(E):
            ids = [j for j in range(host.cap) if (int(det[e, j >> 6]) >> (j & 63)) & 1]
            assert 0 not in ids                                           # the ego's own chassis is filtered out
            veh = [j for j in ids if (sh["flags"][e, j] & abi.KIND_MASK) == abi.KIND_VEHICLE]
            d = sorted(math.hypot(sh["cx"][e, j] - sh["cx"][e, 0], sh["cy"][e, j] - sh["cy"][e, 0]) for j in veh)
            n = min(K, len(veh))
            assert (others[e, n:] == 0).all()
            for k in range(n):
                seen += 1
                fwd = (others[e, k, 0] * 2 - 1) * 50.0
                left = (others[e, k, 1] * 2 - 1) * 50.0
                assert abs(math.hypot(fwd, left) - min(d[k], 50 * math.sqrt(2))) < 1e-2 or d[k] > 50.0
            # every detected body really is within lidar reach
            for j in ids:
                r = math.hypot(sh["hl"][e, j], sh["hw"][e, j])
                assert math.hypot(sh["cx"][e, j] - sh["cx"][e, 0], sh["cy"][e, j] - sh["cy"][e, 0]) <= 50.0 + r + 1e-3
    assert seen > 50


def test_replay_traffic_mode_reproduces_a_recorded_episode():
    """traffic_mode 'replay' (ReplayTrafficParticipantPolicy, policy/replay_policy.py:43-67): the traffic takes its
    poses from recorded frames.  Replaying with the SAME agent actions reproduces the recorded observations
    exactly; with other actions the traffic still follows the recording (no reaction)."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    E, T = 6, 120
    base = dict(num_envs=E, num_scenarios=E, traffic_density=0.3, horizon=1000, auto_reset=False)   # no restarts: frame k = step k
    rec_host = HostScene(make_config(dict(base)))
    o = ob.OracleWorld(rec_host)
    o.reset()
    n = E * rec_host.cap
    shape = np.zeros((T + 1, n), dtype=abi.SHAPE_DT)
    dyn = np.zeros((T + 1, n, 2), np.float32)
    obs_rec = []

    def grab(k):
        shape[k] = o.state["shape"]
        dyn[k, :, 0] = o.state["dyn"]["heading"]
        dyn[k, :, 1] = o.state["dyn"]["speed"]

    grab(0)
    acts = [np.tile(np.array([0.02 * math.sin(0.1 * t), 0.5], np.float32), (E, 1, 1)) for t in range(T)]
    for t in range(T):
        o.step(acts[t])
        grab(t + 1)
        obs_rec.append(o.obs.copy())
    assert not (o.state["nav"]["steps"].reshape(E, -1)[:, 0] < T).any(), "episodes must not have restarted while recording"
    moved = np.abs(shape["cx"][T] - shape["cx"][0]).reshape(E, -1)[:, 1:]
    assert (moved > 1.0).any()                                            # some traffic did drive

    rp_host = HostScene(make_config(dict(base, traffic_mode="replay")))
    assert rp_host.cap == rec_host.cap and rp_host.md_config.traffic_mode == 3
    r = ob.OracleWorld(rp_host)
    r.set_tracks(shape, dyn)
    r.reset()
    for t in range(T):
        r.step(acts[t])
        assert r.obs.tobytes() == obs_rec[t].tobytes(), "replay diverged at step %d" % t
    # other actions: traffic poses still equal the recording, step for step
    r2 = ob.OracleWorld(HostScene(make_config(dict(base, traffic_mode="replay"))))
    r2.set_tracks(shape, dyn)
    r2.reset()
    for t in range(60):
        r2.step(np.tile(np.array([0.0, 0.1], np.float32), (E, 1, 1)))
        got = r2.state["shape"].reshape(E, -1)
        want = shape[t + 1].reshape(E, -1)
        for f in ("cx", "cy", "c", "s"):
            assert np.array_equal(got[f][:, 1:], want[f][:, 1:])


def test_random_agent_model_obs_and_types():
    """random_agent_model (manager/agent_manager.py:41, obs/state_obs.py:24-27,70-75): the agent's class is drawn
    per scenario among s / m / l / xl / default and its length / 10, width / 2.5 lead the observation (21 + 240)."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.pg_space import VEHICLE_TYPES
    E = 24
    host = HostScene(make_config(dict(num_envs=E, num_scenarios=E, random_agent_model=True, traffic_density=0.0)))
    assert host.obs_dim == 21 + 240 and host.md_config.random_agent_model == 1
    types = [host.scenes[s].vehicle_cfgs[0]["type"] for s in host.seeds]
    assert set(types) <= {"s", "m", "l", "xl", "default"} and len(set(types)) >= 3
    o = ob.OracleWorld(host)
    o.reset()
    o.step(np.tile(np.array([0.0, 0.4], np.float32), (E, 1, 1)))
    obs = o.obs
    for e, t in enumerate(types):
        assert obs[e, 0] == pytest.approx(VEHICLE_TYPES[t]["length"] / 10.0, abs=1e-6)
        assert obs[e, 1] == pytest.approx(VEHICLE_TYPES[t]["width"] / 2.5, abs=1e-6)
    plain = ob.OracleWorld(HostScene(make_config(dict(num_envs=E, num_scenarios=E, traffic_density=0.0))))
    plain.reset()
    plain.step(np.tile(np.array([0.0, 0.4], np.float32), (E, 1, 1)))
    # same scenarios without the option: default car everywhere; where the random draw was "default" the rest of
    # the observation agrees (the spawn-lane draw precedes the class draw; the vehicle's own seed is the engine's)
    assert "default" in types
    for e, t in enumerate(types):
        if t == "default":
            assert np.array_equal(obs[e, 2:], plain.obs[e, :])



@pytest.mark.parametrize("steering", [-0.01, 0.01])
@pytest.mark.parametrize("distance", [10, 50, 100])
def test_out_of_road_coincides_with_side_detector(steering, distance):
    """tests/test_functionality/test_out_of_road.py:7-39: drifting off a long straight road, the episode ends when
    the side detector's nearest return is closer than the chassis diagonal (in units of the detector range)."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    cfg = make_config(dict(num_envs=1, num_scenarios=1, map="SSSSSSSSSSS", auto_reset=False,
                           vehicle_config=dict(side_detector=dict(num_lasers=120, distance=distance))))
    host = HostScene(cfg)
    o = ob.OracleWorld(host)
    o.reset()
    sh = o.state["shape"][0]
    tolerance = math.sqrt((2 * sh["hw"]) ** 2 + (2 * sh["hl"]) ** 2) / distance
    act = np.array([[[steering, 1.0]]], np.float32)
    for t in range(3000):
        o.step(act)
        fl = int(o.state["flags"][0])
        if fl & (abi.FL_TERMINATED | abi.FL_TRUNCATED):
            side = o.obs[0, :120]
            assert side.min() < tolerance, (side.min(), tolerance)
            assert fl & abi.FL_OUT_OF_ROAD
            break
    else:
        raise AssertionError("the episode never ended")


def test_sidewalk_and_line_contacts_with_hard_left():
    """tests/test_functionality/test_collision.py:22-49: steering -0.5 at full throttle from the spawn point crosses a
    broken line, a continuous white line and hits the sidewalk within 100 steps."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    host = HostScene(make_config(dict(num_envs=1, num_scenarios=1, traffic_density=0.0, auto_reset=False)))
    o = ob.OracleWorld(host)
    o.reset()
    seen = 0
    for t in range(100):
        o.step(np.array([[[-0.5, 1.0]]], np.float32))
        seen |= int(o.state["flags"][0])
    assert seen & abi.FL_CRASH_SIDEWALK
    assert seen & abi.FL_ON_BROKEN and seen & abi.FL_ON_WHITE_CONT


def test_bicycle_substep_rotation_matches_exact_trigonometry():
    """The integrator carries (cos, sin) of the travel direction and rotates it per sub-step with a short series
    (include/md_geom.h: md_bicycle_substep); against a float64 restatement of the same model (grip-limited slip angle,
    yaw rate with grip-limited angular acceleration) with exact sin / cos the pose after a 0.1 s step differs by less
    than 2e-5 m / 2e-6 rad even at full lock and top speed."""
    lib = ob.load()
    P = np.zeros(1, dtype=abi.PARAM_DT)
    P["max_steer"], P["accel_gain"], P["brake_gain"], P["roll_decel"] = math.radians(40.0), 3.0, 5.0, 0.3
    P["max_speed_kmh"], P["lf"], P["lr"], P["fric_decel"] = 80.0, 1.05, 1.42, 8.8
    hl, hw = 2.2575, 0.926                                   # ref_bicycle's box (the default car)
    slew = 8.8 * 1.05 * 1.5 / (hl * hl + hw * hw) * 0.02 * 0.02
    rng = np.random.RandomState(0)
    capped = slewed = 0
    for _ in range(200):
        steer, thr = float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))
        st = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(-3.1, 3.1), rng.uniform(0, 22),
                       rng.uniform(-0.01, 0.01)], np.float32)
        x, y, psi, v, yaw = [float(t) for t in st]
        lib.ref_bicycle(st.ctypes.data, steer, thr, P.ctypes.data, 0.02, 5)
        beta = math.atan(1.42 / (1.05 + 1.42) * math.tan(steer * math.radians(40.0)))
        grip = 8.8 * 1.42 / max(v * v, 1e-3)            # tyre grip bounds v^2 sin(beta) / lr by wheel_friction * g
        capped += abs(math.sin(beta)) > grip
        beta = math.copysign(math.asin(min(abs(math.sin(beta)), grip)), beta)
        th = psi + beta                                 # travel direction
        for k in range(5):
            acc = 3.0 * thr if (thr > 0 and not abs(v) * 3.6 > 80.0) else 0.0
            dec = 0.0 if acc > 0 or (thr > 0 and not abs(v) * 3.6 > 80.0) else (0.3 if thr >= 0 else min(-thr * 5.0, 8.8))
            vn = v + acc * 0.02
            if vn > 0:
                vn = max(vn - dec * 0.02, 0.0)
            vm = 0.5 * (v + vn)
            x += vm * math.cos(th) * 0.02
            y += vm * math.sin(th) * 0.02
            want = vm * math.sin(beta) / 1.42 * 0.02
            slewed += abs(want - yaw) > slew
            d = yaw + min(max(want - yaw, -slew), slew)
            d = min(max(d, -abs(vm) * 0.5 / 1.42 * 0.02), abs(vm) * 0.5 / 1.42 * 0.02)
            yaw = d
            psi += d
            th += d
            v = vn
        psi = (psi + math.pi) % (2 * math.pi) - math.pi
        assert abs(st[0] - x) < 2e-5 and abs(st[1] - y) < 2e-5, (st, x, y)
        assert abs(((st[2] - psi) + math.pi) % (2 * math.pi) - math.pi) < 2e-6
        assert abs(st[4] - yaw) < 1e-7
    assert 40 < capped < 180 and slewed > 100       # every regime occurs: free / grip-limited slip, free / slewing yaw


def test_traffic_steering_settles():
    """The reference's PID steering (PID_controller.py; heading gains 1.7 / 0.01 / 3.5 per step) on this repo's vehicle
    model: traffic at its cruise speed holds its lane with small steering commands.  On a purely kinematic bicycle
    (instant yaw response, no tyre grip limit) the same controller limit-cycles between about +-0.5 of full lock at the
    step rate -- more than a quarter of all steps flipped sign at |steer| > 0.5."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    E = 8
    h = HostScene(make_config(dict(num_envs=E, num_scenarios=E, map="SCS", traffic_density=0.2, traffic_mode="respawn",
                                   horizon=1500, auto_reset=False)))
    o = ob.OracleWorld(h)
    o.reset()
    st, sp = [], []
    for t in range(300):
        o.step(np.zeros((E, 1, 2), np.float32))
        d = o.state["dyn"].reshape(E, -1)
        st.append(d["steering"][:, 1:8].copy())
        sp.append(d["speed"][:, 1:8].copy())
    st, sp = np.stack(st), np.stack(sp)
    assert (sp[-1] > 7.0).sum() >= 10                            # plenty of vehicles at cruise speed
    flips = (np.sign(st[1:]) != np.sign(st[:-1])) & (np.abs(st[1:]) > 0.5) & (np.abs(st[:-1]) > 0.5)
    assert flips.mean() < 0.002
    cruising = sp[-100:] > 7.0
    jitter = np.abs(np.diff(st[-101:], axis=0))[cruising]
    assert np.percentile(jitter, 90) < 0.1                       # step-to-step steering change of cruising traffic


def test_agent_policy_idm_drives_the_agent_to_its_destination():
    """agent_policy = IDMPolicy (envs/base_env.py:53, manager/agent_manager.py:37-70; used throughout the reference's
    export / randomness tests): the agent is planned by the traffic's IDM + PID policy.  Whatever env.step() is given
    is ignored; the agent follows its route, keeps to the lanes, does not run into the traffic ahead and arrives."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    E = 6
    cfg = make_config(dict(num_envs=E, num_scenarios=E, start_seed=300, map="SCS", traffic_density=0.1, agent_policy="IDMPolicy",
                           horizon=2000, auto_reset=False))
    assert cfg["agent_policy"] == "IDMPolicy"
    h = HostScene(cfg)
    assert h.md_config.agent_idm == 1
    o = ob.OracleWorld(h)
    o.reset()
    junk = np.tile(np.array([1.0, -1.0], np.float32), (E, 1, 1))      # full right lock + full brake: must be ignored
    arrived = np.zeros(E, bool)
    crashed = np.zeros(E, bool)
    top = np.zeros(E, np.float32)
    for t in range(1500):
        o.step(junk)
        fl = o.state["flags"].reshape(E, -1)[:, 0]
        crashed |= ((fl & (abi.FL_CRASH_VEHICLE | abi.FL_OUT_OF_ROAD | abi.FL_CRASH_SIDEWALK)) != 0) & ~arrived
        arrived |= ((fl & abi.FL_ARRIVE_DEST) != 0) & ~crashed     # (no auto-reset: afterwards the car just drives on)
        top = np.maximum(top, o.state["dyn"]["speed"].reshape(E, -1)[:, 0])
        if (arrived | crashed).all():
            break
    assert arrived.sum() >= E - 1 and crashed.sum() <= 1, (arrived, crashed)
    assert (top > 6.0).all() and (top < 30.0 / 3.6 + 1.5).all()       # cruises near NORMAL_SPEED = 30 km/h, never beyond
    # the action the observation reports is the one the policy applied in this step, not the junk
    act = o.state["action"].reshape(E, -1, 2)[:, 0]
    assert not np.allclose(act, junk[:, 0])
    with pytest.raises(NotImplementedError):
        make_config(dict(agent_policy="ReplayEgoCarPolicy"))

    class IDMPolicy:            # the reference passes the class itself
        pass
    assert make_config(dict(agent_policy=IDMPolicy))["agent_policy"] == "IDMPolicy"


def test_enable_reverse_drives_the_agent_backwards():
    """vehicle_config.enable_reverse (base_vehicle.py:476-484): with it a negative throttle is a negative engine force
    (brake released), so the agent slows down and then backs up; without it the same action only brakes."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    E = 2
    out = {}
    for rev in (False, True):
        h = HostScene(make_config(dict(num_envs=E, num_scenarios=E, map="SSS", traffic_density=0.0, auto_reset=False,
                                       horizon=1000, vehicle_config=dict(enable_reverse=rev))))
        assert h.md_config.enable_reverse == int(rev)
        o = ob.OracleWorld(h)
        o.reset()
        for _ in range(30):
            o.step(np.tile(np.array([0.0, 1.0], np.float32), (E, 1, 1)))       # get rolling
        v0 = o.state["dyn"]["speed"].reshape(E, -1)[:, 0].copy()
        x0 = o.state["shape"]["cx"].reshape(E, -1)[:, 0].copy()
        speeds = []
        for _ in range(80):
            o.step(np.tile(np.array([0.0, -1.0], np.float32), (E, 1, 1)))
            speeds.append(o.state["dyn"]["speed"].reshape(E, -1)[:, 0].copy())
        out[rev] = (v0, x0, np.stack(speeds), o.state["shape"]["cx"].reshape(E, -1)[:, 0].copy())
    v0, x0, sp, x1 = out[False]
    assert (v0 > 5.0).all() and (sp[-1] == 0.0).all() and (sp >= 0.0).all() and (x1 > x0).all()    # brakes to a stop
    v0, x0, sp_r, x1_r = out[True]
    assert (sp_r[-1] < -3.0).all()                                   # ... backs up
    assert (sp_r[5] > sp[5]).all()                                   # and sheds speed more slowly than the brake does
    assert (x1_r < out[False][3]).all()


def test_single_agent_env_on_a_map_with_a_bidirection_block():
    """map="yBY" / "BC": the two-way single lane inside an ordinary procedurally generated map (pinned lane for lane by
    tests/golden/pg_maps_v4.json).  The block narrows the road to ONE lane placed on the centre line -- in the reference as
    here it is meant to follow a Merge -- so this only checks that scenes build, traffic spawns on it and a rollout runs;
    the multi-agent env built around it is tested in test_marl.py."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    for seed, mp in ((412, "yBY"), (411, "BC")):
        E = 3
        h = HostScene(make_config(dict(num_envs=E, num_scenarios=E, start_seed=seed, map=mp, traffic_density=0.3, horizon=300)))
        assert ((h.state["shape0"]["flags"] & abi.KIND_MASK) == abi.KIND_VEHICLE).sum() > 2 * E      # traffic on it too
        o = ob.OracleWorld(h)
        o.reset()
        for t in range(200):
            o.step(np.tile(np.array([0.0, 0.6], np.float32), (E, 1, 1)))
        obs = o.obs.reshape(E, -1)
        assert np.isfinite(obs).all() and (obs >= 0).all() and (obs <= 1).all()


def test_single_agent_env_on_a_map_with_a_parking_lot():
    """map="SPS" on a one-lane-per-direction road (ParkingLot, pgblock/parking_lot.py; pinned lane for lane by
    tests/golden/pg_maps_v5.json): scenes build -- lane tables, grid, line / sidewalk quads of ~150 roads --, traffic spawns,
    the IDM-driven agent passes the lot and arrives; with enable_reverse the agent can back up inside it."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    E = 3
    h = HostScene(make_config(dict(num_envs=E, num_scenarios=E, start_seed=421, map="SPS", traffic_density=0.1,
                                   map_config=dict(lane_num=1), agent_policy="IDMPolicy", horizon=2000, auto_reset=False)))
    assert max(len(t.lane_objs) for t in h.map_tables) > 100
    o = ob.OracleWorld(h)
    o.reset()
    arrived = np.zeros(E, bool)
    bad = np.zeros(E, bool)
    for t in range(900):
        o.step(None)
        fl = o.state["flags"].reshape(E, -1)[:, 0]
        bad |= ((fl & (abi.FL_OUT_OF_ROAD | abi.FL_CRASH_SIDEWALK)) != 0) & ~arrived
        arrived |= ((fl & abi.FL_ARRIVE_DEST) != 0) & ~bad
        if (arrived | bad).all():
            break
    assert arrived.sum() >= E - 1, (arrived, bad)


def test_reward_cost_done_known_answers():
    """tests/test_functionality/test_reward_cost_done.py (its live case and the three it keeps commented out): the terminal
    step of an episode reports the configured reward / cost of its cause -- success 1111 / 0, out of road -2222 / 5555,
    crash vehicle -3333 / 6666, crash object -4444 / 7777."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    rewards = dict(success_reward=1111, out_of_road_penalty=2222, crash_vehicle_penalty=3333, crash_object_penalty=4444,
                   out_of_road_cost=5555, crash_vehicle_cost=6666, crash_object_cost=7777)

    def run(extra, action, want_flag, steps=1000):
        h = HostScene(make_config(dict(rewards, num_envs=1, auto_reset=False, horizon=5000, **extra)))
        o = ob.OracleWorld(h)
        o.reset()
        for _ in range(steps):
            o.step(np.array([[action]], np.float32))
            fl = int(o.state["flags"][0])
            if fl & (abi.FL_TERMINATED | abi.FL_TRUNCATED):
                break
        assert fl & want_flag, hex(fl)
        return float(o.state["reward"][0]), float(o.state["cost"][0])

    assert run(dict(map="S", traffic_density=0.0), [0.0, 1.0], abi.FL_ARRIVE_DEST) == (1111.0, 0.0)
    assert run(dict(map="S", traffic_density=0.0), [1.0, 1.0], abi.FL_OUT_OF_ROAD) == (-2222.0, 5555.0)
    assert run(dict(map="SSS", traffic_density=1.0, start_seed=1), [0.0, 1.0], abi.FL_CRASH_VEHICLE) == (-3333.0, 6666.0)
    r, c = run(dict(map="SSS", traffic_density=0.0, accident_prob=1.0, start_seed=5), [0.0, 1.0],
               abi.FL_CRASH_OBJECT | abi.FL_CRASH_VEHICLE)
    assert (r, c) in ((-4444.0, 7777.0), (-3333.0, 6666.0))      # a cone / barrier, or the broken-down car of the scene


def test_nav_road_cache_follows_the_route_cursors(cs_dist):
    """MdNav.road0 / road1 (ABI v7) are route_roads[ck0] / route_roads[ck1] at all times: after checkpoint advances,
    traffic respawns (hybrid mode) and auto-resets.  The step logic reads only the cached ids."""
    from helpers import make_cfg, scripted_actions
    from metadrive_ped_amd.engine import HostScene
    for mode in ("trigger", "hybrid"):
        cfg = make_cfg(cs_dist, num_envs=12, num_scenarios=12, start_seed=70, traffic_density=0.2, traffic_mode=mode,
                       horizon=150, auto_reset=True)
        h = HostScene(cfg)
        o = ob.OracleWorld(h)
        o.reset()
        advanced = 0
        for t in range(260):
            o.step(scripted_actions(h.E, 1, t))
            if t % 20 == 19:
                nav, rr = o.state["nav"], o.state["route_roads"].reshape(-1, abi.MD_ROUTE_LEN)
                rows = np.arange(len(nav))
                live = nav["route_len"] >= 2
                assert (nav["road0"][live] == rr[rows, nav["ck0"]][live]).all()
                assert (nav["road1"][live] == rr[rows, nav["ck1"]][live]).all()
                advanced += int((nav["ck0"][live] > 0).sum())
        assert advanced > 0


def test_toll_gate_booths_block_the_odd_lanes():
    """TollGate block (pgblock/tollgate.py, buildings/tollgate_building.py): a booth on every odd lane of both directions;
    an agent that keeps lane 1 runs into it -> crash_building, which terminates whatever the crash_*_done switches say
    (envs/metadrive_env.py:170-175); on lane 0 it passes; the lidar sees the booth; traffic on lane 1 stops behind it."""
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    out = {}
    for lane_idx in (0, 1):
        cfg = make_config(dict(num_envs=1, num_scenarios=1, start_seed=521, map="$S", traffic_density=0.0, auto_reset=False,
                               crash_vehicle_done=False, crash_object_done=False, horizon=2000, random_spawn_lane_index=False,
                               agent_configs=dict(default_agent=dict(spawn_lane_index=(">", ">>", lane_idx)))))
        h = HostScene(cfg)
        kinds = h.state["shape0"]["flags"] & abi.KIND_MASK
        booths = np.nonzero(kinds == abi.KIND_BUILDING)[0]
        assert len(booths) == 2          # lane 1 of the positive and of the negative road (3 lanes each)
        assert np.allclose(h.state["shape0"]["hl"][booths], 5.0) and np.allclose(h.state["shape0"]["hw"][booths], 1.75)
        o = ob.OracleWorld(h)
        o.reset()
        saw, hit, done = False, False, False
        o_mid = 2
        for t in range(260):
            ob_ = o.obs[0]
            steer = float(np.clip(4.0 * (ob_[o_mid] - 0.5) + 2.0 * (ob_[8] - 0.5), -1, 1))
            o.step(np.array([[[steer, 0.5 if ob_[o_mid + 1] < 0.3 else 0.0]]], np.float32))
            fl = int(o.state["flags"][0])
            saw |= bool(o.obs[0][19] < 0.5)                     # beam 0: something within 25 m straight ahead
            hit |= bool(fl & abi.FL_CRASH_BUILDING)
            done |= bool(fl & abi.FL_TERMINATED)
            if done:
                break
        out[lane_idx] = (saw, hit, done, float(o.state["shape"]["cx"][0]))
    assert out[1][0] and out[1][1] and out[1][2]                # lane 1: sees it, hits it, episode over
    assert not out[0][1]                                        # lane 0: no booth
    assert out[0][3] > out[1][3]                                # ... and it got further

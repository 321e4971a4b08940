"""Scenario mode (BASELINE configs[4]: ScenarioEnv + reactive TrajectoryIDMPolicy traffic).

CPU: the host tables + the oracle against the reference's own classes (tests/golden/scenario.json: InterpolatingLine /
PointLane, TrajectoryNavigation, ScenarioEnv reward / cost / done, TrajectoryIDMPolicy) and behaviour of the traffic
lifecycle.  GPU: scenario_step_kernel bit-exact against the oracle on synthetic scenes and on scenes exported from PG
rollouts; a 2048-scene property test (BASELINE configs[4]'s batch)."""
import json
import math
import os

import numpy as np
import pytest

import oracle_binding as ob
from metadrive_ped_amd import abi
from metadrive_ped_amd.scenario import PolyLine, ScenarioHostScene, make_scenario_config, synthetic_scenarios

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SC_KEYS = ["shape", "dyn", "nav", "pid", "action", "flags", "obs", "reward", "cost", "step_info", "need_reset", "next_agent_id"]
ROUTE_KEYS = ["route_n", "route_segs", "route_verts", "route_aux"]     # routes cut at a later spawn frame (reactive_traffic only)


def _oracle(host):
    o = ob.OracleWorld(host)
    o.set_tracks(host.tracks["shape"], host.tracks["dyn"])
    return o


def _follow(obs, n_side=12, gain_h=6.0, gain_l=2.0, throttle=0.3):
    """a small route follower on the navigation dims (lateral, heading error)"""
    o_navi = (n_side or 2) + 6 + 1
    a = np.zeros((len(obs), 1, 2), np.float32)
    a[:, 0, 0] = np.clip(gain_h * (obs[:, o_navi + 19] - 0.5) + gain_l * (obs[:, o_navi + 18] - 0.5), -1, 1)
    a[:, 0, 1] = throttle
    return a


def test_scenario_rollout_on_the_oracle_lifecycle_and_reactive_traffic():
    E = 8
    cfg = make_scenario_config(dict(num_envs=E, num_scenarios=E, reactive_traffic=True, horizon=400, auto_reset=False))
    host = ScenarioHostScene(cfg, synthetic_scenarios(E, 300))
    assert host.obs_dim == 12 + 6 + 1 + 22 + 120
    o = _oracle(host)
    o.reset()
    nav = o.state["nav"].reshape(E, host.cap)
    sh = o.state["shape"].reshape(E, host.cap)
    meta = host.world.arrays["track_meta"].reshape(E, host.cap, 4)
    # after the reset: every spawnable track valid at frame 0 is in the world, the ones behind the ego that move are reactive
    # with policy indices 0, 1, 2, 3, 4, 0, ... in slot order (scenario_traffic_manager.py:231-236)
    for e in range(E):
        idm = np.nonzero(nav["ck0"][e] == abi.SC_IDM)[0]
        assert len(idm) >= 3
        assert list(nav["timer"][e][idm]) == [i % 5 for i in range(len(idm))]
        assert o.state["next_agent_id"][e] == len(idm)
        ego = sh[e, 0]
        for j in idm:
            rx, ry = sh["cx"][e, j] - ego["cx"], sh["cy"][e, j] - ego["cy"]
            assert rx * ego["c"] + ry * ego["s"] < -1.0                     # behind the ego
            assert o.state["dyn"]["speed"].reshape(E, host.cap)[e, j] == 0.0   # spawned at rest
        never = (meta[e, :, 2] & abi.TM_NEVER) != 0
        assert (nav["ck0"][e][never] == abi.SC_ABSENT).all()
        late = (meta[e, :, 0] > 0) & ~never
        assert (nav["ck0"][e][late] == abi.SC_ABSENT).all()
    frames = host.tracks["shape"].reshape(host.T, E, host.cap)
    succeeded = np.zeros(E, bool)
    for t in range(1, 230):
        o.step(_follow(o.obs))
        k = t
        if k < host.T:
            # replayed movers sit exactly on their recorded frame; absent ones are not alive
            rep = nav["ck0"] == abi.SC_REPLAY
            kinds = sh["flags"] & abi.KIND_MASK
            moving_rep = rep & ~np.isin(kinds, [abi.KIND_CONE, abi.KIND_BARRIER])
            assert np.array_equal(sh["cx"][moving_rep], frames["cx"][k][moving_rep])
            assert ((frames["flags"][k][moving_rep] & abi.F_ALIVE) != 0).all()
            absent = (nav["ck0"] == abi.SC_ABSENT)
            absent[:, 0] = False
            assert ((sh["flags"][absent] & abi.F_ALIVE) == 0).all()
        fl = o.state["flags"].reshape(E, host.cap)[:, 0]
        succeeded |= (fl & abi.FL_ARRIVE_DEST) != 0
    # after the data are over only reactive vehicles and static objects remain
    rep = nav["ck0"] == abi.SC_REPLAY
    kinds = sh["flags"] & abi.KIND_MASK
    assert np.isin(kinds[rep], [abi.KIND_CONE, abi.KIND_BARRIER]).all()
    assert succeeded.sum() >= E // 2          # the follower completes most routes
    assert (o.state["step_info"][:, 6] > 0.5).all()


def test_reactive_vehicle_brakes_behind_a_stopped_ego():
    """The point of reactive_traffic: a TrajectoryIDMPolicy vehicle following the ego slows down when the ego stops, while
    the same track replayed (reactive_traffic off) drives through it."""
    E = 4
    scs = synthetic_scenarios(E, 900, n_vehicles=10)
    out = {}
    for reactive in (True, False):
        cfg = make_scenario_config(dict(num_envs=E, num_scenarios=E, reactive_traffic=reactive, horizon=400, auto_reset=False))
        host = ScenarioHostScene(cfg, scs)
        o = _oracle(host)
        o.reset()
        crashed = np.zeros(E, bool)
        for t in range(150):
            a = _follow(o.obs, throttle=-1.0)          # the ego brakes to a halt and stays
            o.step(a)
            crashed |= (o.state["flags"].reshape(E, host.cap)[:, 0] & abi.FL_CRASH_VEHICLE) != 0
        nav = o.state["nav"].reshape(E, host.cap)
        sp = o.state["dyn"]["speed"].reshape(E, host.cap)
        out[reactive] = (crashed.copy(), nav["ck0"].copy(), sp.copy())
    # same-lane followers: replayed ones run into the halted ego in some scene, reactive ones queue up behind it
    assert out[False][0].sum() > out[True][0].sum()
    idm = out[True][1] == abi.SC_IDM
    assert idm.sum() >= 4
    assert (out[True][2][idm] < 12.0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("reactive", [True, False])
def test_scenario_step_gpu_parity(reactive):
    import torch
    from metadrive_ped_amd.engine import BatchedEngine
    E = 24
    cfg = make_scenario_config(dict(num_envs=E, num_scenarios=E, reactive_traffic=reactive, horizon=260, auto_reset=True))
    host = ScenarioHostScene(cfg, synthetic_scenarios(E, 40))
    eng = BatchedEngine(cfg, host=host)
    o = _oracle(host)
    eng.reset()
    o.reset()
    from helpers import assert_state_equal
    keys = SC_KEYS + (ROUTE_KEYS if reactive else [])
    assert_state_equal(eng.download_state(), o.state, keys=keys, where="scenario reset")
    rng = np.random.RandomState(3)
    for t in range(320):
        a = _follow(o.obs, throttle=0.35 if (t // 60) % 2 == 0 else -0.4)
        a[:, 0, 0] += rng.uniform(-0.05, 0.05, size=E).astype(np.float32)
        a[::5, 0, 0] += 0.4 * math.sin(t * 0.05)          # some egos wander off the route
        eng.step(torch.from_numpy(a).to(eng.device))
        o.step(a)
        if t % 20 == 0 or t > 300:
            assert_state_equal(eng.download_state(), o.state, keys=keys, where="scenario step %d" % t)
    st = eng.download_state()
    assert_state_equal(st, o.state, keys=keys, where="scenario final")
    nav = st["nav"].reshape(E, host.cap)
    if reactive:
        assert (nav["ck0"] == abi.SC_IDM).any() or st["next_agent_id"].sum() > 0
    assert np.isfinite(st["obs"]).all() and st["obs"].min() >= 0.0 and st["obs"].max() <= 1.0


# ------------------------------------------------------------------------------------------------
# golden vectors from the reference's own classes (oracle/gen/gen_golden.py::section_scenario)
def _golden():
    with open(os.path.join(GOLDEN, "scenario.json")) as fh:
        return json.load(fh)


def f32(x):
    return float(np.float32(x))


def _tie_zone(pl, p, eps=2e-3):
    p = np.asarray(p, np.float64)
    a = ((pl.start - p) * pl.direction).sum(1)
    b = ((p - pl.end) * pl.direction).sum(1)
    h = np.maximum.reduce([a, b, np.zeros(len(a))])
    dpa = p - pl.start
    c = dpa[:, 0] * pl.direction[:, 1] - dpa[:, 1] * pl.direction[:, 0]
    d = np.sort(np.hypot(h, c))
    return len(d) > 1 and d[1] - d[0] < eps


def test_polyline_against_interpolating_line_and_point_lane():
    g = _golden()
    lib = ob.load()
    n_q = 0
    for rec in g["polylines"]:
        pl = PolyLine(np.asarray(rec["points"]))
        assert len(pl.seg_len) == len(rec["segments"])
        for i, sg in enumerate(rec["segments"]):
            np.testing.assert_allclose(pl.start[i], sg["start"], atol=1e-9)
            np.testing.assert_allclose(pl.end[i], sg["end"], atol=1e-9)
            assert abs(pl.seg_len[i] - sg["length"]) < 1e-9 and abs(pl.heading[i] - sg["heading"]) < 1e-9
            np.testing.assert_allclose(pl.direction[i], sg["direction"], atol=1e-9)
            if len(rec["segments"]) > 1 or sg["length"] != 0.1:      # (the static one-piece line: lateral hard-wired to +y)
                np.testing.assert_allclose([pl.direction[i][1], -pl.direction[i][0]], sg["lateral_direction"], atol=1e-9)
        assert abs(pl.length - rec["length"]) < 1e-9
        np.testing.assert_allclose(pl.outline(rec["width"]), rec["polygon"], atol=1e-9)        # PointLane.auto_generate_polygon
        np.testing.assert_allclose(pl.position(0.0), rec["start"], atol=1e-9)
        np.testing.assert_allclose(pl.position(pl.length), rec["end"], atol=1e-9)
        segs = pl.records()
        out4 = np.zeros(4, np.float32)
        out2 = np.zeros(2, np.float32)
        for q in rec["queries"]:
            lg, lt = pl.local_coordinates(q["point"])
            assert abs(lg - q["long"]) < 1e-9 and abs(lt - q["lat"]) < 1e-9
            np.testing.assert_allclose(pl.position(q["s"], q["lateral"]), q["position"], atol=1e-9)
            assert abs(pl.heading_at(lg) - q["heading_at_long"]) < 1e-12
            # the float32 oracle: coordinates of a few hundred metres -> millimetres.  Outside a convex corner the
            # nearest point of BOTH adjacent pieces is their shared vertex: an exact tie that rounding decides, in the
            # reference as well -- such points say nothing about the arithmetic
            if _tie_zone(pl, q["point"]) or (len(rec["segments"]) == 1 and rec["segments"][0]["length"] == 0.1):
                continue      # (the never-moving line: its hard-wired lateral direction is not modelled on the device)
            lib.ref_poly_local(segs.ctypes.data, len(segs), f32(q["point"][0]), f32(q["point"][1]), out4.ctypes.data)
            assert abs(out4[0] - q["long"]) < 3e-3 and abs(out4[1] - q["lat"]) < 3e-3, (out4, q)
            assert abs(out4[3] - rec["length"]) < 1e-3
            lib.ref_poly_position(segs.ctypes.data, len(segs), f32(q["s"]), f32(q["lateral"]), out2.ctypes.data)
            np.testing.assert_allclose(out2, q["position"], atol=3e-3)
            n_q += 1
    assert n_q > 150


def _one_scene_host(points, T=160, extra_tracks=None, **cfg_kw):
    """a ScenarioHostScene whose SDC track follows `points`"""
    from metadrive_ped_amd.scenario import _track_dict
    pts = np.asarray(points, np.float64)
    n = len(pts)
    h = np.arctan2(np.gradient(pts[:, 1]), np.gradient(pts[:, 0]))
    tracks = {"0": _track_dict("0", "VEHICLE", n, np.ones(n, bool), pts[:, 0], pts[:, 1], h, np.zeros(n), 4.5, 1.85, 1.5)}
    tracks.update(extra_tracks or {})
    sc = {"id": "t", "version": "t", "length": n, "metadata": {"sdc_id": "0", "ts": np.arange(n) * 0.1}, "tracks": tracks,
          "dynamic_map_states": {}, "map_features": {}}
    cfg = make_scenario_config(dict(dict(num_envs=1, num_scenarios=1, auto_reset=False), **cfg_kw))
    return ScenarioHostScene(cfg, [sc])


def test_trajectory_navigation_reward_cost_done_against_reference():
    """TrajectoryNavigation.update_localization (22 dims, unpacking quirk, route completion), the state observation with a
    PointLane, ScenarioEnv.reward_function / cost_function / done_function: the oracle on posed agents."""
    g = _golden()
    FLAG = dict(crash_vehicle=abi.FL_CRASH_VEHICLE, crash_object=abi.FL_CRASH_OBJECT, crash_building=abi.FL_CRASH_BUILDING,
                crash_human=abi.FL_CRASH_HUMAN, crash_sidewalk=abi.FL_CRASH_SIDEWALK,
                on_yellow_continuous_line=abi.FL_ON_YELLOW_CONT, on_white_continuous_line=abi.FL_ON_WHITE_CONT)
    n = 0
    for case in g["agent"]:
        host = _one_scene_host(case["points"], vehicle_config=dict(side_detector=dict(num_lasers=0, distance=50),
                                                                    lidar=dict(num_lasers=0, distance=0)))
        np.testing.assert_allclose(host.world.arrays["ckpt_xy"], case["checkpoints"], atol=2e-4)
        ref_line = PolyLine(np.asarray(case["points"]))
        assert host.obs_dim == 2 + 6 + 1 + 22
        host.world.arrays["track_meta"].reshape(-1, 4)[0, 1] = case["scenario_length"]   # data_manager.current_scenario_length of the case
        for smp in case["samples"]:
            if abs((smp["long"] / 2.0) - round(smp["long"] / 2.0)) < 2e-3 or _tie_zone(ref_line, smp["pos"]):
                continue        # float32 vs float64 right at a checkpoint boundary / in a corner's tie zone
            o = ob.OracleWorld(host, host.clone_state())
            st, k = o.state, o.k
            st["need_reset"][:] = 0
            sh, dy = st["shape"], st["dyn"]
            sh["cx"][0], sh["cy"][0] = smp["pos"]
            sh["c"][0], sh["s"][0] = math.cos(smp["heading"]), math.sin(smp["heading"])
            dy["heading"][0], dy["speed"][0] = smp["heading"], smp["speed"]
            dy["steering"][0] = smp["action"][0]
            dy["last_c"][0], dy["last_s"][0] = math.cos(smp["last_heading"]), math.sin(smp["last_heading"])
            dy["last_x"][0], dy["last_y"][0] = smp["pos"]
            st["action"][0] = smp["action"]
            st["pid"]["hp"][0] = smp["long_before"]
            st["nav"]["steps"][0] = smp["steps"]
            fl = 0
            for name, bit in FLAG.items():
                if smp["flags"][name]:
                    fl |= bit
            st["flags"][0] = fl
            cf = smp["config"]
            k.horizon = smp["horizon"]
            k.truncate_as_terminate = int(cf["truncate_as_terminate"])
            k.relax_out_of_road_done = int(cf["relax_out_of_road_done"])
            k.out_of_route_done = int(cf["out_of_route_done"])
            k.crash_vehicle_done, k.crash_object_done, k.crash_human_done = int(cf["crash_vehicle_done"]), int(cf["crash_object_done"]), int(cf["crash_human_done"])
            k.no_negative_reward = int(cf["no_negative_reward"])
            k.allowed_more_steps = int(cf["allowed_more_steps"] or 0)
            k.scenario_length = case["scenario_length"]
            o.call("ref_scenario_observe")
            obs = st["obs"][0]
            np.testing.assert_allclose(obs[:7], smp["state9"][:7], atol=2e-5, err_msg=str(smp))     # borders, heading_diff, speed, steering, actions
            assert abs(obs[7] - smp["state9"][7]) < 4e-3                                           # yaw rate: acos of a float32 cosine near 1
            assert abs(obs[8] - smp["state9"][8]) < 5e-4                                           # lateral / 4.5 (float32 coordinates)
            np.testing.assert_allclose(obs[9:31], smp["navi"], atol=2e-4, err_msg=str(smp))
            assert obs[9] == obs[10] and obs[25] == obs[26]                                        # the unpacking quirk: both slots alike
            assert abs(st["step_info"][0, 6] - smp["route_completion"]) < 1e-4
            out = int(st["flags"][0])
            di = smp["done_info"]
            edge = abs(abs(smp["lat"]) - 4.0) < 5e-3 or abs(smp["route_completion"] - 0.95) < 1e-4 or \
                abs(smp["route_completion"] + 0.1) < 1e-4 or abs(abs(smp["lat"]) - 10.0) < 5e-3
            if not edge:
                assert bool(out & abi.FL_ARRIVE_DEST) == di["arrive_dest"], smp
                assert bool(out & abi.FL_OUT_OF_ROAD) == di["out_of_road"], smp
                assert bool(out & abi.FL_MAX_STEP) == di["max_step"], smp
                assert bool(out & abi.FL_TERMINATED) == smp["done"], smp
                assert abs(st["reward"][0] - smp["reward"]) < 3e-3, (st["reward"][0], smp)
                assert abs(st["step_info"][0, 0] - smp["step_reward"]) < 3e-3
                assert abs(st["cost"][0] - smp["cost"]) < 1e-6
                n += 1
    assert n >= 100


def test_trajectory_idm_policy_act_against_reference():
    """TrajectoryIDMPolicy.act: heading PID (1.2, 0.1, 3.5) carried over the sequence, speed control only when
    step % 5 == policy_index (else the last acceleration), single-lane front search within 20 m over objects that have a
    chassis corner on the route's outline, arrival inside 2 m of the route's end."""
    from metadrive_ped_amd.scenario import _track_dict
    g = _golden()
    lib = ob.load()
    n_speed = n_front = 0
    for case in g["traj_idm"]:
        pts = np.asarray(case["points"], np.float64)
        n = len(pts)
        hh = np.arctan2(np.gradient(pts[:, 1]), np.gradient(pts[:, 0]))
        extra = {"1": _track_dict("1", "VEHICLE", n, np.ones(n, bool), pts[:, 0], pts[:, 1], hh, np.full(n, 5.0), 4.6, 1.9, 1.5)}
        for q in range(2, 6):
            extra[str(q)] = _track_dict(str(q), "VEHICLE", n, np.ones(n, bool), pts[:, 0] + 500.0, pts[:, 1], hh, np.full(n, 5.0), 4.6, 1.9, 1.5)
        host = _one_scene_host(pts + np.array([0.0, 300.0]), T=n, extra_tracks=extra, reactive_traffic=True)
        a = host.world.arrays
        np.testing.assert_allclose(a["polyv"][a["polyv_off"][1]:a["polyv_off"][2]], case["polygon"], atol=2e-4)
        o = ob.OracleWorld(host, host.clone_state())
        o.set_tracks(host.tracks["shape"], host.tracks["dyn"])
        st = o.state
        cap = host.cap
        sh, dy, nav = st["shape"], st["dyn"], st["nav"]
        nav["ck0"][1], nav["timer"][1] = abi.SC_IDM, case["policy_index"]
        for smp in case["sequence"]:
            sh["flags"][1] = abi.KIND_VEHICLE | abi.F_ALIVE
            sh["cx"][1], sh["cy"][1] = smp["pos"]
            sh["c"][1], sh["s"][1] = math.cos(smp["heading"]), math.sin(smp["heading"])
            dy["heading"][1], dy["speed"][1] = smp["heading"], smp["speed"]
            sh["flags"][2:cap] = 0
            for j, ob_ in enumerate(smp["objs"], start=2):
                sh["flags"][j] = abi.KIND_VEHICLE | abi.F_ALIVE | abi.F_STATIC
                sh["cx"][j], sh["cy"][j] = ob_["pos"]
                sh["c"][j], sh["s"][j] = math.cos(ob_["heading"]), math.sin(ob_["heading"])
                sh["hl"][j], sh["hw"][j] = ob_["length"] / 2, ob_["width"] / 2
                dy["speed"][j], dy["heading"][j] = ob_["speed"], ob_["heading"]
            nav["ck0"][1] = abi.SC_IDM
            lib.ref_tidm_vehicle(o.w, o.s, o.k, 0, 1, smp["step"])
            assert (nav["ck0"][1] == abi.SC_ARRIVED) == smp["arrived"]
            if smp["arrived"]:
                continue            # the manager removes it; the reference's act() is not reached
            assert ((smp["step"] % 5) == case["policy_index"]) == smp["do_speed_control"]
            act = st["action"][1]
            assert abs(act[0] - smp["action"][0]) < 2e-3 * max(1.0, abs(smp["action"][0])), (act, smp["action"])
            assert abs(act[1] - smp["action"][1]) < 2e-3 * max(1.0, abs(smp["action"][1])), (act, smp)
            n_speed += smp["do_speed_control"]
            n_front += smp["do_speed_control"] and len(smp["objs"]) > 0
    assert n_speed >= 20 and n_front >= 10


def test_track_bookkeeping_against_reference():
    """first valid run (get_max_valid_indicis), static-car test (std of the valid positions > 3 m), minimum length for a
    reactive policy, noise objects (< 20 valid frames): ScenarioHostScene's track_meta."""
    from metadrive_ped_amd.scenario import synthetic_scenario
    g = _golden()
    by_seed = {}
    for b in g["bookkeeping"]:
        by_seed.setdefault(b["seed"], []).append(b)
    n = 0
    for seed, rows in by_seed.items():
        sc = synthetic_scenario(seed, T=120)
        cfg = make_scenario_config(dict(num_envs=1, num_scenarios=1))
        host = ScenarioHostScene(cfg, [sc])
        meta = host.world.arrays["track_meta"]
        for b in rows:
            j = host.track_ids[0].index(b["oid"])
            if j == 0:
                continue
            assert [int(meta[j, 0]), int(meta[j, 1])] == b["run"], b
            if b["type"] == "VEHICLE":
                assert bool(meta[j, 2] & abi.TM_MOVING) == b["moving"] and bool(meta[j, 2] & abi.TM_LENGTH_OK) == b["length_ok"], b
            if b["type"] == "TRAFFIC_CONE":
                assert bool(meta[j, 2] & abi.TM_NEVER) == b["noise"], b
            n += 1
    assert n > 40


@pytest.mark.gpu
def test_scenarios_exported_from_pg_rollouts_gpu_parity():
    """record -> export_scenarios() -> ScenarioEnv: scenes recorded on PG maps with the IDM-driven agent as the SDC, loaded
    as scenario descriptions; HIP and oracle agree bit for bit, and the follower completes the recorded routes."""
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine, HostScene
    from metadrive_ped_amd.scenario_export import tracks_to_scenarios
    E = 12
    pg = HostScene(make_config(dict(num_envs=E, num_scenarios=E, start_seed=5, traffic_density=0.2, agent_policy="IDMPolicy",
                                    horizon=1000, auto_reset=False, map="SCS")))
    rec = ob.OracleWorld(pg)
    rec.reset()
    scs = tracks_to_scenarios(ob.record_episode(rec, [None] * 150), pg)
    cfg = make_scenario_config(dict(num_envs=E, num_scenarios=E, reactive_traffic=True, horizon=300, auto_reset=True))
    host = ScenarioHostScene(cfg, scs)
    eng = BatchedEngine(cfg, host=host)
    o = _oracle(host)
    eng.reset()
    o.reset()
    done = np.zeros(E, bool)
    for t in range(200):
        a = _follow(o.obs, throttle=0.4)
        eng.step(torch.from_numpy(a).to(eng.device))
        o.step(a)
        done |= (o.state["flags"].reshape(E, host.cap)[:, 0] & abi.FL_ARRIVE_DEST) != 0
        if t % 25 == 0:
            assert_state_equal(eng.download_state(), o.state, keys=SC_KEYS, where="pg scenario step %d" % t)
    assert_state_equal(eng.download_state(), o.state, keys=SC_KEYS, where="pg scenario final")
    assert done.sum() >= E // 2


@pytest.mark.gpu
def test_scenario_full_size_properties_and_env_api():
    """BASELINE configs[4]'s batch: 2048 scenes with reactive traffic through BatchedScenarioEnv -- observation bounds,
    run-to-run bit determinism, the oracle on a sampled slice of the scenes, episodes that end and restart."""
    import torch
    from metadrive_ped_amd.envs.scenario_env import BatchedScenarioEnv
    E = 2048
    user = dict(num_envs=E, num_scenarios=E, reactive_traffic=True, horizon=250, auto_reset=True)
    scs = synthetic_scenarios(E, 7000)
    runs = []
    idx = list(range(0, E, 64))
    used = []          # run 0's actions of the sampled scenes (torch and numpy may round the controller differently)
    for rep in range(2):
        env = BatchedScenarioEnv(user, scenarios=scs)
        if rep == 0:
            host = ScenarioHostScene(env.config, scs)
        env.lazy_init(host=host)
        obs, info = env.reset()
        assert tuple(obs.shape) == (E, env.observation_space.shape[0]) and obs.dtype == torch.float32
        ended = torch.zeros(E, dtype=torch.bool, device=obs.device)
        for t in range(120):
            a = torch.zeros(E, 2, device=obs.device)
            o_navi = 12 + 6 + 1
            a[:, 0] = (6.0 * (obs[:, o_navi + 19] - 0.5) + 2.0 * (obs[:, o_navi + 18] - 0.5)).clamp(-1, 1)
            a[:, 1] = 0.5
            if rep == 0:
                used.append(a[idx].cpu().numpy().reshape(len(idx), 1, 2))
            obs, rew, term, trunc, info = env.step(a)
            ended |= term | trunc
        assert float(obs.min()) >= 0.0 and float(obs.max()) <= 1.0 and bool(torch.isfinite(rew).all())
        assert float(info["route_completion"].max()) > 0.3
        runs.append((obs.clone().cpu().numpy(), env.engine.download_state()))
        env.close()
    assert runs[0][0].tobytes() == runs[1][0].tobytes()
    for key in ("shape", "nav", "flags", "reward"):
        assert runs[0][1][key].tobytes() == runs[1][1][key].tobytes(), key
    # the oracle on every 64th scene with the very same actions
    sub = ScenarioHostScene(make_scenario_config(dict(user, num_envs=len(idx), num_scenarios=len(idx))), [scs[i] for i in idx])
    o = _oracle(sub)
    o.reset()
    for t in range(120):
        o.step(used[t])
    got = runs[0][0][idx]
    assert got.tobytes() == o.obs.tobytes()


def test_road_line_pieces_against_reference():
    """ScenarioBlock.construct_continuous_line / construct_broken_line (component/scenario_block/scenario_block.py:74-99) on
    seven polylines, and MetaDriveType's classification of every map-feature type: tests/golden/scenario_lines.json."""
    import json
    import os
    from metadrive_ped_amd import abi
    from metadrive_ped_amd import scenario as S
    with open(os.path.join(os.path.dirname(__file__), "golden", "scenario_lines.json")) as f:
        g = json.load(f)
    assert g["stripe_length"] == S.STRIPE_LENGTH
    n = 0
    for c in g["cases"]:
        for fn, broken in (("construct_continuous_line", False), ("construct_broken_line", True)):
            mine = S.line_pieces(np.asarray(c["polyline"]), broken)
            assert len(mine) == len(c[fn])
            for (a, b), ref in zip(mine, c[fn]):
                np.testing.assert_allclose([*a, *b], [*ref[0], *ref[1]], atol=1e-9)
                assert ref[3] == ("broken" if broken else "continuous")
                n += 1
    assert n > 150
    for name, t in g["types"].items():
        quads, kinds = S.scene_line_quads({"f": {"type": name, "polyline": [[0.0, 0.0], [10.0, 0.0]]}}, 512)
        if t["road_line"] and t["broken"]:
            assert kinds == [abi.Q_LINE_BROKEN] * 3
        elif t["road_line"]:
            assert kinds == [abi.Q_LINE_YELLOW_CONT if t["yellow"] else abi.Q_LINE_WHITE_CONT] * 6
        elif t["boundary"]:
            assert kinds == [abi.Q_LINE_WHITE_CONT] * 6          # road boundaries are continuous grey lines (:69-70)
        else:
            assert kinds == []
    # a piece whose middle lies outside the map region is not built (block/base_block.py:481)
    assert S.scene_line_quads({"f": {"type": "ROAD_LINE_SOLID_SINGLE_WHITE", "polyline": [[300.0, 0.0], [310.0, 0.0]]}}, 512)[1] == []
    assert len(S.scene_line_quads({"f": {"type": "ROAD_LINE_SOLID_SINGLE_WHITE", "polyline": [[250.0, 0.0], [262.0, 0.0]]}}, 512)[1]) == 4


def test_road_lines_raise_flags_and_end_episodes_on_oracle():
    """The scene's road lines are bodies: an agent steered off the road touches the solid edge line -> on_white / on_yellow
    continuous line -> with relax_out_of_road_done off that is ScenarioEnv's out of road (scenario_env.py:380-401) and the
    step's reward is -on_lane_line_penalty... then -out_of_road_penalty; the side detector sees the lines."""
    from metadrive_ped_amd import abi
    from metadrive_ped_amd.scenario import ScenarioHostScene, make_scenario_config, synthetic_scenarios
    import oracle_binding as ob
    E = 4
    cfg = make_scenario_config(dict(num_envs=E, num_scenarios=E, relax_out_of_road_done=False, auto_reset=False, no_traffic=True,
                                    out_of_road_penalty=7.0, on_lane_line_penalty=3.0, horizon=400))
    host = ScenarioHostScene(cfg, synthetic_scenarios(E, 100))
    w = host.world.arrays
    assert host.world.n_maps == E and (np.diff(w["quad_off"]) > 300).all()
    kinds = set(np.unique(w["quad_kind"]).tolist())
    assert {abi.Q_LINE_WHITE_CONT, abi.Q_LINE_BROKEN, abi.Q_LINE_YELLOW_CONT} <= kinds
    o = _oracle(host)
    o.reset()
    obs0 = o.state["obs"].reshape(E, -1)
    assert (obs0[:, :12] < 1.0).any(axis=1).all()          # the side detector (12 beams, 50 m) sees the edge lines
    done_step = np.full(E, -1)
    for t in range(120):
        a = np.zeros((E, 1, 2), np.float32)
        a[:, 0, 0] = np.where(np.arange(E) % 2 == 0, -0.35, 0.35)      # half of them to each side
        a[:, 0, 1] = 0.4
        o.step(a)
        fl = o.state["flags"].reshape(E, -1)[:, 0]
        for e in range(E):
            if done_step[e] < 0 and fl[e] & abi.FL_TERMINATED:
                done_step[e] = t
                assert fl[e] & abi.FL_OUT_OF_ROAD and fl[e] & (abi.FL_ON_WHITE_CONT | abi.FL_ON_YELLOW_CONT)
                assert float(o.state["reward"].reshape(E, -1)[e, 0]) == -7.0
                assert float(o.state["step_info"].reshape(E, -1, 8)[e, 0, 0]) == -3.0          # step_reward = -on_lane_line_penalty
    assert (done_step >= 0).all() and (done_step > 5).all()
    # crossing a BROKEN separator earlier raised no termination: every env drove more than one lane width before it ended


# ------------------------------------------------------------------------------------------------
# routes cut at a later spawn frame (scenario_traffic_manager.py:216-236: get_max_valid_indicis(track, episode_step), get_idm_route)
# ------------------------------------------------------------------------------------------------
def test_route_builder_against_reference_point_lane():
    """md_build_route (what md_step runs when a reactive policy is created at a frame other than its first run's start) against
    the reference's PointLane on the golden polylines, and -- on every suffix of them -- against the host's PolyLine, which the
    golden test above pins to the reference: pieces, outline polygon, end point."""
    g = _golden()
    n = 0
    for pl in g["polylines"]:
        pts = np.asarray(pl["points"], np.float64)[:, :2]
        segs, verts, aux, rc = ob.build_route(pts)
        assert rc == 0 and len(segs) == len(pl["segments"]) and len(verts) == len(pl["polygon"])
        ref = pl["segments"]
        tol = 4e-7 * max(1.0, float(np.abs(pts).max()))      # float32 storage of coordinates up to ~900 m
        np.testing.assert_allclose(np.c_[segs["sx"], segs["sy"]], [r["start"] for r in ref], atol=tol)
        np.testing.assert_allclose(np.c_[segs["ex"], segs["ey"]], [r["end"] for r in ref], atol=tol)
        np.testing.assert_allclose(np.c_[segs["dx"], segs["dy"]], [r["direction"] for r in ref], atol=1e-6)
        np.testing.assert_allclose(segs["len"], [r["length"] for r in ref], atol=1e-6)
        np.testing.assert_allclose(segs["heading"], [r["heading"] for r in ref], atol=1e-6)
        np.testing.assert_allclose(verts, np.asarray(pl["polygon"])[:, :2], atol=tol)
        np.testing.assert_allclose(aux[:2], np.asarray(pl["end"])[:2], atol=tol)
        assert np.allclose(aux[2:6], [verts[:, 0].min(), verts[:, 1].min(), verts[:, 0].max(), verts[:, 1].max()])
        for k in range(1, len(pts) - 1, max(1, len(pts) // 9)):       # suffixes: the route of a spawn at frame k
            sub = pts[k:].astype(np.float32).astype(np.float64)      # (the device reads float32 frames)
            host_line = PolyLine(sub)
            segs, verts, aux, rc = ob.build_route(sub)
            rec = host_line.records()
            assert rc == 0 and len(segs) == len(rec)
            for f in ("sx", "sy", "ex", "ey", "dx", "dy", "len", "cum"):
                np.testing.assert_allclose(segs[f], rec[f], atol=tol, err_msg=f)
            np.testing.assert_allclose(verts, host_line.outline(), atol=tol)
            n += 1
    assert n > 40
    # buffers too small: refused, nothing usable left
    segs, verts, aux, rc = ob.build_route(np.asarray(g["polylines"][0]["points"]), seg_cap=5)
    assert rc == -1 and len(segs) == 0


def test_vehicle_spawned_late_gets_a_route_cut_at_the_spawn_frame():
    """A reactive vehicle that reaches the end of its route is removed; while its track is still valid it is spawned again at the
    recorded position, and -- being behind the ego, moving, with more than 5 m left -- gets a NEW policy whose route starts at that
    frame and ends with the run.  Same for a track that appears a second time after a gap."""
    E = 16
    cfg = make_scenario_config(dict(num_envs=E, num_scenarios=E, reactive_traffic=True, horizon=400, auto_reset=False))
    host = ScenarioHostScene(cfg, synthetic_scenarios(E, 300))
    runs, run_off = host.world.arrays["runs"], host.world.arrays["run_off"]
    assert host.md_config.route_seg_cap >= int((runs[:, 1] - runs[:, 0]).max())
    o = _oracle(host)
    o.reset()
    frames = host.tracks["shape"].reshape(host.T, E, host.cap)
    rn = o.state["route_n"].reshape(E, host.cap, 4)
    segs = o.state["route_segs"].reshape(E, host.cap, -1)
    verts = o.state["route_verts"].reshape(E, host.cap, -1, 2)
    nav = o.state["nav"].reshape(E, host.cap)
    meta = host.world.arrays["track_meta"].reshape(E, host.cap, 4)
    assert (rn == 0).all()
    seen = {}
    for t in range(1, 200):
        o.step(_follow(o.obs))
        for e, j in np.argwhere(rn[:, :, 0] > 0):
            k = int(rn[e, j, 2])
            if (e, j, k) in seen:
                continue
            seen[(e, j, k)] = t
            assert k == t and k != meta[e, j, 0] and nav["ck0"][e, j] == abi.SC_IDM     # built in the step of the spawn
            ng = e * host.cap + j
            mine = runs[run_off[ng]:run_off[ng + 1]]
            t1 = int(mine[(mine[:, 0] <= k) & (k < mine[:, 1])][0, 1])
            pts = np.c_[frames["cx"][k:t1, e, j], frames["cy"][k:t1, e, j]].astype(np.float64)
            want = PolyLine(pts)
            got = segs[e, j, :rn[e, j, 0]]
            assert len(got) == len(want.seg_len)
            np.testing.assert_allclose(np.c_[got["sx"], got["sy"]], want.start, atol=1e-4)
            np.testing.assert_allclose(got["cum"], want.cum, atol=1e-4)
            assert np.hypot(*(pts[0] - pts[-1])) > 5.0                                    # IDM_CREATE_MIN_LENGTH on the cut route
            np.testing.assert_allclose(verts[e, j, :rn[e, j, 1]], want.outline(), atol=1e-4)
            assert abs(got["sx"][0] - frames["cx"][k, e, j]) < 1e-6                       # the route starts where the vehicle was put
    assert len(seen) >= 4
    # an env reset leaves no cut route behind
    o.state["need_reset"][:] = 1
    o.step(_follow(o.obs))
    assert (rn == 0).all()
    # without the route buffers (MdState.route_n NULL) such spawns are replayed, as before ABI 9
    st = host.clone_state()
    for key in ("route_n", "route_segs", "route_verts", "route_aux"):
        st.pop(key)
    o2 = ob.OracleWorld(host, st)
    o2.set_tracks(host.tracks["shape"], host.tracks["dyn"])
    o2.reset()
    first = min(seen.values())
    e, j, k = [key for key, t in seen.items() if t == first][0]
    for t in range(1, first + 1):
        o2.step(_follow(o2.obs))
    assert o2.state["nav"].reshape(E, host.cap)["ck0"][e, j] == abi.SC_REPLAY


def test_contact_flags_see_replayed_bodies_one_frame_late():
    """BaseVehicle.after_step (contact test at the bodies' present poses) runs before ScenarioTrafficManager.after_step moves the
    replayed bodies (both managers have priority 10, the agent manager is registered first: envs/base_env.py:744,
    envs/scenario_env.py:118-126): a body teleported onto the ego in frame 5 is hit in step 6, when it is already gone again;
    the lidar (observation time) sees it in step 5."""
    from metadrive_ped_amd.scenario import _track_dict
    T = 40
    pts = np.c_[np.linspace(0.0, 60.0, T), np.zeros(T)]
    x = np.full(T, 300.0)
    y = np.full(T, 50.0) + np.arange(T) * 0.5        # far away, drifting (so it is not a "static" car)
    x[5], y[5] = 4.0, 0.0                              # one frame into the nose of the (standing) ego
    other = _track_dict("9", "VEHICLE", T, np.ones(T, bool), x, y, np.zeros(T), np.zeros(T), 4.5, 1.85, 1.5)
    host = _one_scene_host(pts, extra_tracks={"9": other}, reactive_traffic=False, crash_vehicle_done=False,
                           vehicle_config=dict(lidar=dict(num_lasers=60, distance=50), side_detector=dict(num_lasers=0, distance=50)))
    o = _oracle(host)
    o.reset()
    crash, near = [], []
    for t in range(1, 9):
        a = np.zeros((1, 1, 2), np.float32)
        a[0, 0, 1] = -1.0
        o.step(a)
        crash.append(bool(o.state["flags"][0] & abi.FL_CRASH_VEHICLE))
        near.append(float(o.obs[0, -60:].min()) < 0.2)
    assert crash == [False, False, False, False, False, True, False, False]
    assert near == [False, False, False, False, True, False, False, False]


def test_spawn_decisions_against_reference_spawn_vehicle():
    """ScenarioTrafficManager.spawn_vehicle itself (tests/golden/scenario_spawn.json: the reference's method with Bullet-side calls
    recorded): created or not (validity, static cars, the 8 m x 2 m overlap filter), replay or reactive policy (behind the ego
    by > 1 m, within 15 m sideways, heading within 90 degrees, moving, > 5 m of path left), the policy index, and the route of the
    reactive policy -- cut at the spawn frame when that is not the start of the track's first valid run."""
    import ctypes as C
    from metadrive_ped_amd.scenario import _track_dict
    with open(os.path.join(GOLDEN, "scenario_spawn.json")) as f:
        g = json.load(f)
    T = g["frames"]
    sdc_pts = np.c_[np.linspace(0.0, 59.0, T), np.zeros(T)]
    n_kind = dict(none=0, replay=0, idm=0, idm_late=0)
    for case in g["cases"]:
        tr, k = case["track"], case["step"]
        t = np.arange(T)
        px = tr["xk"] + (t - k) * 0.1 * tr["speed"] * math.cos(tr["heading"])
        py = tr["yk"] + (t - k) * 0.1 * tr["speed"] * math.sin(tr["heading"])
        valid = np.array([ch == "1" for ch in tr["valid"]])
        other = _track_dict("9", "VEHICLE", T, valid, px, py, np.full(T, tr["heading"]), np.full(T, tr["speed"]), tr["length"], 1.9, 1.6)
        cf = case["config"]
        host = _one_scene_host(sdc_pts, T=T, extra_tracks={"9": other}, reactive_traffic=cf["reactive_traffic"],
                               no_static_vehicles=cf["no_static_vehicles"], filter_overlapping_car=cf["filter_overlapping_car"],
                               vehicle_config=dict(lidar=dict(num_lasers=0, distance=0), side_detector=dict(num_lasers=0, distance=50)))
        assert host.track_ids[0][1] == "9"
        meta = host.world.arrays["track_meta"].reshape(-1, 4)
        if valid.any():
            assert bool(meta[1, 2] & abi.TM_MOVING) == case["moving"], case
        o = ob.OracleWorld(host, host.clone_state())
        o.set_tracks(host.tracks["shape"], host.tracks["dyn"])
        st = o.state
        st["need_reset"][:] = 0
        eh = case["ego_heading"]
        st["shape"]["cx"][0], st["shape"]["cy"][0] = case["ego_position"]
        st["shape"]["c"][0], st["shape"]["s"][0] = math.cos(eh), math.sin(eh)
        st["dyn"]["heading"][0] = eh
        st["next_agent_id"][0] = case["idm_count"]
        assert o.lib.ref_scenario_after_step(C.byref(o.w), C.byref(o.s), C.byref(o.k), 0, k) == 0
        nav, sh, dy = st["nav"][1], st["shape"][1], st["dyn"][1]
        where = "case ego=%s step=%d track=%s cfg=%s" % (case["ego_position"], k, tr, cf)
        if case["spawned"] is None:
            assert nav["ck0"] == abi.SC_ABSENT and not (sh["flags"] & abi.F_ALIVE), where
            n_kind["none"] += 1
            continue
        assert sh["flags"] & abi.F_ALIVE, where
        assert abs(sh["cx"] - case["spawned"]["position"][0]) < 1e-4 and abs(sh["cy"] - case["spawned"]["position"][1]) < 1e-4, where
        pol = case["policy"]
        if pol["cls"] == "ReplayTrafficParticipantPolicy":
            assert nav["ck0"] == abi.SC_REPLAY, where
            assert st["next_agent_id"][0] == case["idm_count"]
            n_kind["replay"] += 1
            continue
        assert nav["ck0"] == abi.SC_IDM and nav["timer"] == pol["policy_index"] and dy["speed"] == 0.0, where
        assert st["next_agent_id"][0] == case["idm_count_after"]
        rn = st["route_n"][1]
        if rn[0] > 0:
            segs, aux = st["route_segs"][1][:rn[0]], st["route_aux"][1]
            n_kind["idm_late"] += 1
            assert k != meta[1, 0]
        else:
            a, b = host.world.arrays["poly_off"][1], host.world.arrays["poly_off"][2]
            segs, aux = host.world.arrays["segs"][a:b], host.world.arrays["poly_aux"][1]
            assert k == meta[1, 0]
        assert len(segs) == pol["route_pieces"], where
        np.testing.assert_allclose([segs["sx"][0], segs["sy"][0]], pol["route_start"], atol=1e-4, err_msg=where)
        np.testing.assert_allclose(aux[:2], pol["route_end"], atol=1e-3, err_msg=where)
        assert abs(float(segs["cum"][-1] + segs["len"][-1]) - pol["route_length"]) < 1e-3, where
        n_kind["idm"] += 1
    assert n_kind["none"] > 80 and n_kind["replay"] > 150 and n_kind["idm"] > 40 and n_kind["idm_late"] > 10, n_kind


# ------------------------------------------------------------------------------------------------
# agent_policy = ReplayEgoCarPolicy (policy/replay_policy.py:70-82; tests/benchmark_FPS/benchmark_waymo.py runs with it)
# ------------------------------------------------------------------------------------------------
def test_replayed_ego_follows_the_sdc_track_on_the_oracle():
    E = 6
    cfg = make_scenario_config(dict(num_envs=E, num_scenarios=E, reactive_traffic=True, horizon=0, auto_reset=False,
                                    agent_policy="ReplayEgoCarPolicy"))
    assert cfg["agent_policy"] == "ReplayEgoCarPolicy"
    scs = synthetic_scenarios(E, 700)
    host = ScenarioHostScene(cfg, scs)
    assert host.md_config.ego_replay == 1
    o = _oracle(host)
    o.reset()
    sh = o.state["shape"].reshape(E, host.cap)
    dy = o.state["dyn"].reshape(E, host.cap)
    done_at = np.full(E, -1)
    prev = None
    for t in range(1, host.T):
        a = np.random.RandomState(t).uniform(-1, 1, size=(E, 1, 2)).astype(np.float32)      # ignored
        live = done_at < 0
        o.step(a)
        for e in range(E):
            if not live[e]:
                continue
            sdc = scs[e]["tracks"][scs[e]["metadata"]["sdc_id"]]["state"]
            assert abs(sh["cx"][e, 0] - sdc["position"][t, 0]) < 1e-3 and abs(sh["cy"][e, 0] - sdc["position"][t, 1]) < 1e-3
            assert abs(dy["heading"][e, 0] - sdc["heading"][t]) < 1e-5
            assert abs(dy["speed"][e, 0] - np.hypot(*sdc["velocity"][t])) < 1e-4
            # last actions in the observation: before_step([0, 0]) -> 0.5, 0.5; steering 0 -> 0.5
            mid = 12
            assert np.allclose(o.obs[e, mid + 2:mid + 5], 0.5)
        fl = o.state["flags"].reshape(E, host.cap)[:, 0]
        ended = (fl & (abi.FL_TERMINATED | abi.FL_TRUNCATED)) != 0
        done_at[(done_at < 0) & ended] = t
        if (done_at >= 0).all():
            break
    fl = o.state["flags"].reshape(E, host.cap)[:, 0]
    # the recorded drive completes its own route in most scenes (an episode may end earlier: a replayed neighbour is hit, or
    # the route's 95 % are reached before the last frame)
    assert ((fl & abi.FL_ARRIVE_DEST) != 0).sum() >= E // 2
    assert (o.state["step_info"][:, 6][(fl & abi.FL_ARRIVE_DEST) != 0] > 0.95).all()
    # no overlap filter with a replayed ego (scenario_traffic_manager.py:188-192): config rejected elsewhere
    with pytest.raises(NotImplementedError):
        from metadrive_ped_amd.config import make_config
        make_config(dict(agent_policy="ReplayEgoCarPolicy"))


@pytest.mark.gpu
def test_replayed_ego_gpu_parity():
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.envs.scenario_env import BatchedScenarioEnv
    E = 16
    scs = synthetic_scenarios(E, 720)
    env = BatchedScenarioEnv(dict(num_envs=E, num_scenarios=E, reactive_traffic=True, horizon=0, auto_reset=True,
                                  agent_policy="ReplayEgoCarPolicy", allowed_more_steps=5, truncate_as_terminate=True), scenarios=scs)
    obs, info = env.reset()
    host = env.host
    o = _oracle(host)
    o.reset()
    keys = SC_KEYS + ROUTE_KEYS
    assert_state_equal(env.engine.download_state(), o.state, keys=keys, where="ego replay reset")
    for t in range(260):
        obs, rew, term, trunc, info = env.step(None)
        o.step(np.zeros((E, 1, 2), np.float32))
        if t % 20 == 0 or t > 250:
            assert_state_equal(env.engine.download_state(), o.state, keys=keys, where="ego replay step %d" % t)
    assert np.isfinite(obs.cpu().numpy()).all()


@pytest.mark.gpu
def test_scenario_checkpoint_resumes_bit_identically():
    """get_state / set_state of BatchedScenarioEnv: a checkpoint taken after the device has cut routes replays the same future."""
    import torch
    from metadrive_ped_amd.envs.scenario_env import BatchedScenarioEnv
    E = 16
    env = BatchedScenarioEnv(dict(num_envs=E, num_scenarios=E, reactive_traffic=True, horizon=0, auto_reset=True),
                             scenarios=synthetic_scenarios(E, 300))
    obs, _ = env.reset()
    def act(o, t):
        return torch.from_numpy(_follow(o.cpu().numpy(), throttle=0.35 if (t // 50) % 2 == 0 else -0.3)[:, 0]).cuda()
    for t in range(160):
        obs, *_ = env.step(act(obs, t))
    ck = env.get_state()
    assert (ck["route_n"][:, 0] > 0).any()          # some vehicle follows a route cut at its spawn frame
    o0 = obs.clone()
    first = []
    for t in range(160, 260):
        obs, r, term, trunc, _ = env.step(act(obs, t))
        first.append((obs.cpu().numpy().copy(), r.cpu().numpy().copy(), term.cpu().numpy().copy()))
    env.set_state(ck)
    obs = o0
    for i, t in enumerate(range(160, 260)):
        obs, r, term, trunc, _ = env.step(act(obs, t))
        assert obs.cpu().numpy().tobytes() == first[i][0].tobytes(), "obs diverged at step %d" % t
        assert r.cpu().numpy().tobytes() == first[i][1].tobytes() and (term.cpu().numpy() == first[i][2]).all()
    bad = dict(ck)
    bad["__seeds__"] = ck["__seeds__"] + 1
    with pytest.raises(ValueError):
        env.set_state(bad)


def test_scene_identity_follows_the_scenario_index():
    """ScenarioHostScene.seeds / scenario_ids name WHICH scenarios a batch holds (start_scenario_index-based), so that a checkpoint
    or track set of scenarios 0..3 cannot be loaded into a batch of scenarios 100..103."""
    from metadrive_ped_amd.scenario import ScenarioHostScene, make_scenario_config
    mk = lambda start, off=0: make_scenario_config(dict(num_envs=4, num_scenarios=4, start_scenario_index=start, env_seed_offset=off,
                                                        build_workers=1))
    h0 = ScenarioHostScene(mk(0), synthetic_scenarios(4, 0))
    h100 = ScenarioHostScene(mk(100), synthetic_scenarios(4, 100))
    assert h0.seeds == [0, 1, 2, 3] and h100.seeds == [100, 101, 102, 103]
    assert h0.scenario_ids != h100.scenario_ids and len(set(h100.scenario_ids)) == 4
    assert ScenarioHostScene(mk(100, off=2), synthetic_scenarios(4, 102)).seeds == [102, 103, 100, 101]


@pytest.mark.gpu
def test_scenario_checkpoint_refuses_other_scenarios():
    from metadrive_ped_amd.envs.scenario_env import BatchedScenarioEnv
    E = 8
    a = BatchedScenarioEnv(dict(num_envs=E, num_scenarios=E, start_scenario_index=0, horizon=0))
    b = BatchedScenarioEnv(dict(num_envs=E, num_scenarios=E, start_scenario_index=100, horizon=0))
    a.reset()
    b.reset()
    ck = a.get_state()
    assert int(ck["__abi__"][0]) == abi.MD_ABI_VERSION
    with pytest.raises(ValueError, match="another scenario assignment"):
        b.set_state(ck)
    same_index_other_data = dict(ck)
    same_index_other_data["__scenario_ids__"] = np.asarray(["x%d" % i for i in range(E)])
    with pytest.raises(ValueError, match="other scenarios"):
        a.set_state(same_index_other_data)
    old = dict(ck)
    old["__abi__"] = np.asarray([abi.MD_ABI_VERSION - 1])
    with pytest.raises(ValueError, match="ABI"):
        a.set_state(old)
    a.set_state(ck)


def test_piece_group_circles_contain_their_pieces():
    """MdWorld.poly_ball (the projections' exact cull): every group's circle contains its pieces' end points with >= 1e-3 m to spare,
    one group per MD_POLY_GROUP pieces of every slot's polyline."""
    E = 6
    cfg = make_scenario_config(dict(num_envs=E, num_scenarios=E, reactive_traffic=True))
    host = ScenarioHostScene(cfg, synthetic_scenarios(E, 11))
    a = host.world.arrays
    po, segs, bo, ball = a["poly_off"], a["segs"], a["poly_ball_off"], a["poly_ball"]
    G = abi.MD_POLY_GROUP
    assert len(bo) == len(po) and bo[-1] == len(ball)
    n_checked = 0
    for k in range(len(po) - 1):
        n = po[k + 1] - po[k]
        assert bo[k + 1] - bo[k] == (n + G - 1) // G
        for gi in range(bo[k + 1] - bo[k]):
            g = segs[po[k] + gi * G:min(po[k] + gi * G + G, po[k + 1])]
            c = ball[bo[k] + gi].astype(np.float64)
            far = max(np.hypot(g["sx"] - c[0], g["sy"] - c[1]).max(), np.hypot(g["ex"] - c[0], g["ey"] - c[1]).max())
            assert c[2] - far >= 1.0e-3 * 0.999 and c[3] == 0.0
            n_checked += 1
    assert n_checked > 100


@pytest.mark.gpu
def test_piece_cull_changes_nothing_gpu():
    """The same scenes stepped with and without MdWorld.poly_ball (the kernel then walks every piece): identical states.  (Both are
    compared with the oracle, which always walks every piece, in test_scenario_step_gpu_parity.)"""
    import torch
    from metadrive_ped_amd.engine import BatchedEngine
    E = 16
    cfg = make_scenario_config(dict(num_envs=E, num_scenarios=E, reactive_traffic=True, horizon=200, auto_reset=True))
    h1 = ScenarioHostScene(cfg, synthetic_scenarios(E, 70))
    h2 = ScenarioHostScene(cfg, synthetic_scenarios(E, 70))
    del h2.world.arrays["poly_ball"], h2.world.arrays["poly_ball_off"]
    e1, e2 = BatchedEngine(cfg, host=h1), BatchedEngine(cfg, host=h2)
    e1.reset()
    e2.reset()
    from helpers import assert_state_equal
    rng = np.random.RandomState(5)
    for t in range(150):
        a = rng.uniform(-1, 1, (E, 1, 2)).astype(np.float32)
        a[..., 0] *= 0.2
        a[..., 1] = np.abs(a[..., 1])
        ta = torch.from_numpy(a).to(e1.device)
        e1.step(ta)
        e2.step(ta)
        if t % 25 == 24:
            assert_state_equal(e1.download_state(), e2.download_state(), keys=SC_KEYS + ROUTE_KEYS, where="cull vs full walk, step %d" % t)


@pytest.mark.gpu
@pytest.mark.parametrize("n_vehicles", [0, 1, 3])
def test_tiny_scenes_gpu_parity(n_vehicles):
    """Scenes of one to four movers (the LDS image's regions shrink to a few words each: the decision stage's scratch, the pair
    list, the route records and the detectors' lists must not meet)."""
    import torch
    from metadrive_ped_amd.engine import BatchedEngine
    from metadrive_ped_amd.scenario import synthetic_scenario
    from helpers import assert_state_equal
    E = 8
    scenes = [synthetic_scenario(200 + i, T=80, n_vehicles=n_vehicles, n_parked=0, n_pedestrians=0, n_cones=0) for i in range(E)]
    cfg = make_scenario_config(dict(num_envs=E, num_scenarios=E, reactive_traffic=True, horizon=70, auto_reset=True))
    host = ScenarioHostScene(cfg, scenes)
    assert host.cap <= 8
    eng = BatchedEngine(cfg, host=host)
    o = _oracle(host)
    eng.reset()
    o.reset()
    keys = SC_KEYS + ROUTE_KEYS
    for t in range(100):
        a = _follow(o.obs, throttle=0.4)
        eng.step(torch.from_numpy(a).to(eng.device))
        o.step(a)
        if t % 10 == 9:
            assert_state_equal(eng.download_state(), o.state, keys=keys, where="tiny scenes (%d vehicles) step %d" % (n_vehicles, t))

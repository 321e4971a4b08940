"""GPU: the reset()/step() surface and size-independent properties at BASELINE size (4096 envs)."""
import numpy as np
import pytest

from helpers import make_cfg, scripted_actions

pytestmark = pytest.mark.gpu


def test_env_api_contract(cs_dist):
    import torch
    from metadrive_ped_amd.envs import BatchedMetaDriveEnv
    E = 12
    env = BatchedMetaDriveEnv(dict(num_envs=E, num_scenarios=E, block_dist_config=cs_dist, horizon=40))
    obs, info = env.reset()
    assert tuple(obs.shape) == (E, 259) and obs.dtype == torch.float32 and obs.is_cuda
    o = obs.cpu().numpy()
    assert env.observation_space.contains(o[0]) and (o >= 0).all() and (o <= 1).all()
    for k in ("velocity", "steering", "acceleration", "step_energy", "episode_energy", "step_reward", "episode_reward",
              "episode_length", "cost", "crash_vehicle", "crash_object", "crash_building", "crash_human", "crash_sidewalk",
              "out_of_road", "arrive_dest", "max_step", "crash", "env_seed", "raw_action", "action"):
        assert k in info and info[k].shape[0] == E, k
    seen_trunc = False
    for t in range(45):
        obs, r, term, trunc, info = env.step(torch.from_numpy(scripted_actions(E, 1, t)[:, 0]).cuda())
        assert tuple(r.shape) == (E, ) and term.dtype == torch.bool and trunc.dtype == torch.bool
        seen_trunc |= bool(trunc.any())
    assert seen_trunc                                   # horizon=40 truncates, next step auto-resets
    assert int(info["episode_length"].max()) <= 40
    with pytest.raises(ValueError):
        env.step(torch.zeros(E + 1, 2))
    # terminated / truncated come straight from the kernel (MdState.done_out): they are the two bits of the flag word
    from metadrive_ped_amd import abi
    for t in range(45):
        o_, r_, tm_, tr_, inf_ = env.step(torch.from_numpy(scripted_actions(E, 1, t)[:, 0]).cuda())
        fl_ = env.engine.flags[:, 0]
        assert tm_.dtype == torch.bool and tr_.dtype == torch.bool
        assert torch.equal(tm_, (fl_ & abi.FL_TERMINATED) != 0) and torch.equal(tr_, (fl_ & abi.FL_TRUNCATED) != 0)
        # ... and bytes 2-3 of the same word are the agent's whole step flag word (ABI v10)
        assert torch.equal(env.engine.step_flags[:, 0].to(torch.int32) & 0xFFFF, fl_ & 0xFFFF)
    # obs | reward | done_out are ONE allocation (the slab the multi-GPU gather moves with one collective): the typed views of the
    # slab are the tensors step() returned
    from metadrive_ped_amd.sharding import slab_layout, split_step_slab
    eng_ = env.engine
    layout, total = slab_layout(E, 1, eng_.obs_dim)
    assert eng_.out_slab.numel() == total and {k: tuple(v) for k, v in eng_.out_layout.items()} == layout
    assert eng_.obs.data_ptr() == eng_.out_slab.data_ptr()
    parts = split_step_slab(eng_.out_slab.view(1, -1), E, 1, eng_.obs_dim)
    assert torch.equal(parts["obs"][:, 0], o_) and torch.equal(parts["reward"][:, 0], r_)
    assert torch.equal(parts["terminated"][:, 0], tm_) and torch.equal(parts["truncated"][:, 0], tr_)
    assert torch.equal(parts["flags"][:, 0].to(torch.int32) & 0xFFFF, eng_.flags[:, 0] & 0xFFFF)
    obs2, _ = env.reset(seed=5)                          # re-base scenarios: maps regenerate
    assert env.current_seeds[0] == 5 and env.current_seed == 5 and env.num_scenarios == E
    assert tuple(env.episode_step.shape) == (E, ) and int(env.episode_step.max()) == 0
    env.seed(3)
    with pytest.raises(NotImplementedError):
        env.render()
    env.close()
    # old gym API (envs/gym_wrapper.py of the reference): 4-tuple step, reset -> obs, attributes pass through
    from metadrive_ped_amd.envs.gym_wrapper import createGymWrapper
    from metadrive_ped_amd.envs import BatchedVaryingDynamicsEnv
    genv = createGymWrapper(BatchedVaryingDynamicsEnv)(dict(num_envs=E, num_scenarios=E, block_dist_config=cs_dist, horizon=30))
    o = genv.reset()
    assert tuple(o.shape) == (E, 259) and genv.num_envs == E and genv.observation_space.shape == (259, )
    assert len(genv.dynamics_parameters()) == E and len({d["mass"] for d in genv.dynamics_parameters()}) == E
    done_seen = False
    for t in range(35):
        out = genv.step(torch.from_numpy(scripted_actions(E, 1, t)[:, 0]).cuda())
        assert len(out) == 4 and out[2].dtype == torch.bool
        done_seen |= bool(out[2].any())
    assert done_seen                                    # horizon 30: done = terminated | truncated shows up
    genv.close()


def test_full_size_properties(cs_dist):
    """4096 envs x 240 beams (BASELINE configs[1]): run-to-run bit determinism on the GPU, obs bounds,
    episode bookkeeping, and agreement with the oracle on a sampled subset of envs."""
    import torch
    from metadrive_ped_amd.engine import BatchedEngine, HostScene
    import oracle_binding as ob
    E = 4096
    cfg = make_cfg(cs_dist, num_envs=E, num_scenarios=256, mover_capacity=32, horizon=120)
    host = HostScene(cfg)
    acts = [torch.from_numpy(scripted_actions(E, 1, t)).cuda() for t in range(150)]
    finals = []
    for run in range(2):
        eng = BatchedEngine(cfg, host=host)
        eng.reset()
        for t in range(150):
            eng.step(acts[t])
        torch.cuda.synchronize()
        finals.append(eng.download_state())
    for k in finals[0]:
        assert finals[0][k].tobytes() == finals[1][k].tobytes(), k
    obs = finals[0]["obs"]
    assert np.isfinite(obs).all() and (obs >= 0).all() and (obs <= 1).all()
    steps = finals[0]["nav"]["steps"].reshape(E, -1)[:, 0]
    assert steps.max() <= 120 and steps.min() >= 0
    # envs sharing a scenario seed and fed the same actions stay identical (256 scenarios over 4096 envs)
    # -> compare a 64-env slice against the oracle stepping the same slice
    sub = 64
    sub_cfg = make_cfg(cs_dist, num_envs=sub, num_scenarios=256, mover_capacity=32, horizon=120)
    o = ob.OracleWorld(HostScene(sub_cfg))
    o.reset()
    for t in range(150):
        o.step(acts[t][:sub].cpu().numpy())
    for k in ("obs", "reward", "flags"):
        a = finals[0][k].reshape(E, -1)[:sub]
        b = o.state[k].reshape(sub, -1)
        assert a.tobytes() == b.tobytes(), k


def test_state_checkpoint_resumes_bit_identically():
    """get_state / set_state: a checkpoint taken mid-episode and restored later replays the same future
    (the contract of the reference's record / replay and set_state: same state + same actions -> same result)."""
    import torch
    from metadrive_ped_amd.envs import BatchedMetaDriveEnv
    E = 24
    env = BatchedMetaDriveEnv(dict(num_envs=E, num_scenarios=E, horizon=200, traffic_mode="hybrid", traffic_density=0.15))
    env.reset()
    acts = [torch.from_numpy(scripted_actions(E, 1, t, seed=9)[:, 0]).cuda() for t in range(120)]
    for t in range(40):
        env.step(acts[t])
    ck = env.get_state()
    first = []
    for t in range(40, 120):
        obs, r, term, trunc, _ = env.step(acts[t])
        first.append((obs.cpu().numpy().copy(), r.cpu().numpy().copy(), term.cpu().numpy().copy()))
    env.set_state(ck)
    for i, t in enumerate(range(40, 120)):
        obs, r, term, trunc, _ = env.step(acts[t])
        assert obs.cpu().numpy().tobytes() == first[i][0].tobytes(), "obs diverged at step %d" % t
        assert r.cpu().numpy().tobytes() == first[i][1].tobytes()
        assert (term.cpu().numpy() == first[i][2]).all()
    bad = dict(ck)
    bad["__seeds__"] = ck["__seeds__"] + 1
    with pytest.raises(ValueError):
        env.set_state(bad)


def test_discrete_actions_and_lidar_noise():
    """Discrete action grid drives the same path as the continuous action it maps to; lidar noise / dropout touch
    only the cloud dims of the observation (obs/state_obs.py:225-244)."""
    import torch
    from metadrive_ped_amd.envs import BatchedMetaDriveEnv
    E = 8
    base = dict(num_envs=E, num_scenarios=E, horizon=100)
    a = BatchedMetaDriveEnv(dict(base, discrete_action=True, discrete_steering_dim=3, discrete_throttle_dim=5))
    b = BatchedMetaDriveEnv(dict(base))
    a.reset()
    b.reset()
    idx = torch.tensor([13] * E)                     # 13 -> steering 13 % 3 = 1 -> 0.0 ; throttle 13 // 3 = 4 -> 1.0
    for t in range(30):
        oa, ra, *_ = a.step(idx)
        ob_, rb, *_ = b.step(torch.tensor([[0.0, 1.0]] * E))
    assert oa.cpu().numpy().tobytes() == ob_.cpu().numpy().tobytes()
    n = BatchedMetaDriveEnv(dict(base, vehicle_config=dict(lidar=dict(gaussian_noise=0.05, dropout_prob=0.1))))
    n.reset()
    for t in range(30):
        on, *_ = n.step(torch.tensor([[0.0, 1.0]] * E))
    on, ob_ = on.cpu().numpy(), ob_.cpu().numpy()
    assert np.array_equal(on[:, :19], ob_[:, :19])                       # state + navi dims untouched
    cloud = on[:, 19:]
    assert (cloud >= 0).all() and (cloud <= 1).all()
    assert 0.05 < (cloud == 0.0).mean() < 0.2                            # ~10 % dropped
    assert not np.array_equal(cloud, ob_[:, 19:])
    # the side / lane-line detector clouds take their own detectors' noise settings (state_obs.py:82-85,134-137)
    det = dict(side_detector=dict(num_lasers=8, distance=50), lane_line_detector=dict(num_lasers=4, distance=20))
    noisy = dict(side_detector=dict(num_lasers=8, distance=50, dropout_prob=0.5),
                 lane_line_detector=dict(num_lasers=4, distance=20, gaussian_noise=0.1))
    c, d = BatchedMetaDriveEnv(dict(base, vehicle_config=det)), BatchedMetaDriveEnv(dict(base, vehicle_config=noisy))
    c.reset()
    d.reset()
    for t in range(20):
        oc, *_ = c.step(torch.tensor([[0.0, 0.6]] * E))
        od, *_ = d.step(torch.tensor([[0.0, 0.6]] * E))
    oc, od = oc.cpu().numpy(), od.cpu().numpy()
    side, ll = slice(0, 8), slice(14, 18)                                # [side 8 | 6 state dims | lane-line 4 | navi ...]
    assert 0.2 < (od[:, side] == 0.0).mean() < 0.8 and (oc[:, side] > 0.0).all()
    assert not np.allclose(od[:, ll], oc[:, ll]) and np.abs(od[:, ll] - oc[:, ll]).max() < 0.6
    rest = np.r_[8:14, 18:oc.shape[1]]
    assert np.array_equal(od[:, rest], oc[:, rest])                      # nothing else is touched (lidar noise is off)


def test_record_then_replay_traffic():
    """env.start_recording / stop_recording / load_tracks + traffic_mode='replay': replaying a recorded rollout with
    the same agent actions gives the same observations bit for bit; the replay path equals the oracle's."""
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.envs import BatchedMetaDriveEnv
    import oracle_binding as ob
    from metadrive_ped_amd.scenario_export import tracks_to_scenarios
    E, T = 16, 150
    base = dict(num_envs=E, num_scenarios=E, traffic_density=0.25, horizon=1000, auto_reset=False)
    acts = [torch.from_numpy(scripted_actions(E, 1, t, seed=17)[:, 0] * np.array([0.1, 1.0], np.float32)).cuda() for t in range(T)]
    rec = BatchedMetaDriveEnv(dict(base))
    rec.reset()
    rec.start_recording(T)
    obs_rec = []
    for t in range(T):
        o, *_ = rec.step(acts[t])
        obs_rec.append(o.cpu().numpy().copy())
    tracks = rec.stop_recording()
    assert tracks["shape"].shape[0] == T + 1
    # export_scenarios of the device recording == the export of the same rollout recorded from the oracle
    o_rec = ob.OracleWorld(rec.engine.host, rec.engine.host.clone_state())
    o_rec.reset()
    sc_gpu = rec.export_scenarios(tracks, envs=[0, E - 1])
    sc_cpu = tracks_to_scenarios(ob.record_episode(o_rec, [a.cpu().numpy()[:, None, :] for a in acts]), rec.engine.host, [0, E - 1])
    for a, b in zip(sc_gpu, sc_cpu):
        assert a["id"] == b["id"] and set(a["tracks"]) == set(b["tracks"])
        for oid in a["tracks"]:
            for k, v in a["tracks"][oid]["state"].items():
                assert v.tobytes() == b["tracks"][oid]["state"][k].tobytes(), (oid, k)
    # scenario descriptions -> replay tracks gives the poses of the device recording back
    rp_sd = BatchedMetaDriveEnv(dict(base, traffic_mode="replay", mover_capacity=rec.engine.cap))
    rp_sd.load_scenarios(rec.export_scenarios(tracks))
    back = rp_sd.engine._tracks["shape"].cpu().numpy().view(ob.abi.SHAPE_DT).reshape(T + 1, -1)
    orig = tracks["shape"].cpu().numpy().view(ob.abi.SHAPE_DT).reshape(T + 1, -1)
    on = (orig["flags"] & ob.abi.F_ALIVE) != 0
    assert np.array_equal(back["cx"][on], orig["cx"][on]) and np.array_equal(back["cy"][on], orig["cy"][on])
    rp_sd.close()
    rp = BatchedMetaDriveEnv(dict(base, traffic_mode="replay"))
    rp.load_tracks(tracks)
    rp.reset()
    orc = ob.OracleWorld(rp.engine.host)
    orc.set_tracks(tracks["shape"].cpu().numpy().view(ob.abi.SHAPE_DT).reshape(T + 1, -1), tracks["dyn"].cpu().numpy())
    orc.reset()
    for t in range(T):
        o, *_ = rp.step(acts[t])
        orc.step(acts[t].cpu().numpy()[:, None, :])
        assert o.cpu().numpy().tobytes() == obs_rec[t].tobytes(), "replay diverged from the recording at step %d" % t
        if t % 30 == 0:
            assert_state_equal(rp.engine.download_state(), orc.state, where="replay step %d" % t)
    assert_state_equal(rp.engine.download_state(), orc.state, where="replay final")
    with pytest.raises(ValueError):
        BatchedMetaDriveEnv(dict(base)).load_tracks(tracks)


def test_double_buffered_sub_batches_equal_the_whole_batch():
    """envs/pipeline.py: S sub-batches stepped on S HIP streams give, env for env, the observations / rewards / flags
    of the whole batch stepped in one launch (same scenarios, same actions), bit for bit."""
    import torch
    from metadrive_ped_amd.envs import BatchedMetaDriveEnv
    from metadrive_ped_amd.envs.pipeline import SubBatchedEnvs
    E, S, T = 48, 3, 120
    user = dict(num_envs=E, num_scenarios=E, traffic_density=0.2, horizon=80, mover_capacity=48)
    whole = BatchedMetaDriveEnv(dict(user))
    sub = SubBatchedEnvs(BatchedMetaDriveEnv, user, sub_batches=S)
    sub.build_host()
    o_w, _ = whole.reset()
    o_s = [o for o, _ in sub.reset()]
    sub.synchronize()
    assert torch.equal(torch.cat(o_s), o_w)
    n = E // S
    for t in range(T):
        a = torch.from_numpy(scripted_actions(E, 1, t, seed=5)[:, 0]).cuda()
        torch.cuda.synchronize()                      # `a` was made on the default stream
        ow, rw, tw, cw, _ = whole.step(a)
        res = sub.step([a[k * n:(k + 1) * n] for k in range(S)])
        sub.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(torch.cat([r[0] for r in res]), ow), "obs differ at step %d" % t
        assert torch.equal(torch.cat([r[1] for r in res]), rw) and torch.equal(torch.cat([r[2] for r in res]), tw)
        assert torch.equal(torch.cat([r[3] for r in res]), cw)
    sub.close()
    whole.close()


def test_env_spawn_object_pedestrian_known_answer_and_crash():
    """env.spawn_object / set_velocity (the reference's engine.spawn_object(Pedestrian, ...) + set_velocity): the walking
    schedule of the reference's test_pedestrian.py ends at x = 160 +- 1 on the device too; driving into a standing
    pedestrian sets info['crash_human'] and terminates."""
    import torch
    from metadrive_ped_amd.envs import BatchedMetaDriveEnv
    E = 4
    env = BatchedMetaDriveEnv(dict(num_envs=E, num_scenarios=E, traffic_density=0.0, map="X", start_seed=22,
                                   random_lane_width=True, mover_capacity=8, horizon=5000, auto_reset=False))
    env.reset()
    ped = env.spawn_object("pedestrian", [30.0, 0.0], 0.0)
    env.set_velocity(ped, [1, 0], 1, in_local_frame=True)
    stand = torch.zeros(E, 2, device="cuda")
    for s in range(1, 1000):
        env.step(stand)
        if s == 300:
            env.set_velocity(ped, [1, 0], 0, in_local_frame=True)
        elif s == 500:
            env.set_velocity(ped, [1, 0], 2, in_local_frame=True)
    x = env.engine.object_positions(ped)[:, 0].cpu().numpy()
    assert (np.abs(x - 160.0) < 1.0).all(), "Pedestrian movement error!"
    env.close()

    env = BatchedMetaDriveEnv(dict(num_envs=E, num_scenarios=E, traffic_density=0.0, map="SS", mover_capacity=8, horizon=400))
    env.reset()
    sf = env.engine.shape_f[:, 0].cpu().numpy()
    ahead = np.stack([sf[:, 0] + 18.0 * sf[:, 2], sf[:, 1] + 18.0 * sf[:, 3]], 1)
    env.spawn_object("pedestrian", ahead, 0.0)
    go = torch.tensor([[0.0, 1.0]], device="cuda").repeat(E, 1)
    hit = torch.zeros(E, dtype=torch.bool, device="cuda")
    for t in range(120):
        _, _, term, _, info = env.step(go)
        assert bool((term | ~info["crash_human"]).all())
        hit |= info["crash_human"]
    assert bool(hit.all())
    with pytest.raises(ValueError):
        env.spawn_object("truck", [0.0, 0.0])
    env.close()


def test_env_with_idm_agent_policy():
    """BatchedMetaDriveEnv(agent_policy='IDMPolicy'): step() ignores its argument, the agents drive, episodes end by
    arrival or horizon and restart."""
    import torch
    from metadrive_ped_amd.envs import BatchedMetaDriveEnv
    E = 16
    env = BatchedMetaDriveEnv(dict(num_envs=E, num_scenarios=E, map="SCS", traffic_density=0.1, agent_policy="IDMPolicy", horizon=1500))
    obs, _ = env.reset()
    arrived = torch.zeros(E, dtype=torch.bool, device="cuda")
    bad = torch.zeros(E, dtype=torch.bool, device="cuda")
    for t in range(1200):
        obs, r, tm, tc, info = env.step(None if t % 2 else torch.ones(E, 2, device="cuda"))
        arrived |= info["arrive_dest"]
        bad |= info["out_of_road"] | info["crash_vehicle"]
    assert int(arrived.sum()) >= E - 2 and int(bad.sum()) <= 2
    env.close()


def test_step_is_capturable_in_a_hip_graph():
    """A closed rollout loop -- policy ops on the observation, then the engine's step -- captured once with
    torch.cuda.graph and replayed gives the state of the same loop run eagerly, bit for bit: md_step allocates nothing,
    synchronises nothing and touches only persistent buffers (tools/graph_probe.py times it)."""
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    E, K = 64, 8

    def make():
        eng = BatchedEngine(make_config(dict(num_envs=E, num_scenarios=E, traffic_density=0.2, horizon=300)))
        eng.reset()
        return eng, torch.zeros(E, 1, 2, device="cuda")

    def drive(eng, act):
        ob_ = eng.obs[:, 0, :]
        act[:, 0, 0] = (4.0 * (ob_[:, 2] - 0.5) + 2.0 * (ob_[:, 8] - 0.5)).clamp_(-1.0, 1.0)
        act[:, 0, 1] = (ob_[:, 3] < 0.35).to(torch.float32) * 0.5
        eng.step(act)

    e1, a1 = make()
    e2, a2 = make()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            drive(e1, a1)
            drive(e2, a2)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(K):
            drive(e2, a2)
    torch.cuda.synchronize()
    for _ in range(30):                 # 240 steps: auto-resets included
        for _ in range(K):
            drive(e1, a1)
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(e1.obs, e2.obs) and torch.equal(e1.reward, e2.reward)
    for k in ("shape", "dyn", "nav", "flags"):
        assert torch.equal(e1.state_dev[k], e2.state_dev[k]), k


def _run_twice_and_sample(cfg, host, acts, sub_cfg_fn, sub, A, keys=("obs", "reward", "flags")):
    """two runs on the GPU must agree bit for bit; the first `sub` envs must agree with the oracle stepping that slice"""
    import torch
    from metadrive_ped_amd.engine import BatchedEngine, HostScene
    import oracle_binding as ob
    finals = []
    for run in range(2):
        eng = BatchedEngine(cfg, host=host)
        eng.reset()
        for a in acts:
            eng.step(a)
        torch.cuda.synchronize()
        finals.append(eng.download_state())
        del eng
    for k in finals[0]:
        assert finals[0][k].tobytes() == finals[1][k].tobytes(), k
    sub_host = HostScene(sub_cfg_fn(sub))
    o = ob.OracleWorld(sub_host)
    o.reset()
    for a in acts:
        o.step(a[:sub].cpu().numpy())
    E = host.E
    for k in keys:
        got = finals[0][k].reshape(E, -1)[:sub]
        want = o.state[k].reshape(sub, -1)
        assert got.tobytes() == want.tobytes(), k
    return finals[0]


def test_full_size_default_maps_4096():
    """BASELINE configs[1] as the bench runs it: 4096 envs, one scenario seed each, the reference's DEFAULT block
    distribution (ramps, intersections, roundabouts: the non-staged kernel variant), 240 beams, density 0.1."""
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    E = 4096
    mk = lambda n: make_config(dict(num_envs=n, num_scenarios=E, map=3, traffic_density=0.1, horizon=100, mover_capacity=24))
    cfg = mk(E)
    host = HostScene(cfg)
    assert host.world.arrays["lane_off"][1:].max() - 0 > 64 and int(np.diff(host.world.arrays["lane_off"]).max()) > 64
    acts = [torch.from_numpy(scripted_actions(E, 1, t)).cuda() for t in range(130)]
    st = _run_twice_and_sample(cfg, host, acts, mk, 48, 1)
    obs = st["obs"]
    assert np.isfinite(obs).all() and (obs >= 0).all() and (obs <= 1).all()
    assert st["nav"]["steps"].reshape(E, -1)[:, 0].max() <= 100


def test_full_size_safe_env_8192():
    """BASELINE configs[3]'s per-GPU shard: 8192 SafeMetaDriveEnv (accident scenes, crashes cost but do not end the
    episode), 240 beams."""
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.envs.metadrive_env import BatchedSafeMetaDriveEnv
    from metadrive_ped_amd import abi
    E = 8192
    mk = lambda n: make_config(dict(BatchedSafeMetaDriveEnv.SAFE_DEFAULTS, num_envs=n, num_scenarios=512, map=3, horizon=120,
                                    mover_capacity=64))
    cfg = mk(E)
    host = HostScene(cfg)
    kinds = host.state["shape0"]["flags"] & abi.KIND_MASK
    assert (kinds == abi.KIND_CONE).sum() > 10000
    acts = []
    for t in range(140):
        a = scripted_actions(E, 1, t, seed=4)
        a[:, :, 0] *= 0.3
        acts.append(torch.from_numpy(a).cuda())
    st = _run_twice_and_sample(cfg, host, acts, mk, 48, 1, keys=("obs", "reward", "cost", "flags"))
    obs = st["obs"]
    assert np.isfinite(obs).all() and (obs >= 0).all() and (obs <= 1).all()
    fl = st["flags"].reshape(E, -1)[:, 0]
    assert st["step_info"][:, 5].max() > 0          # total_cost accumulates somewhere
    assert ((fl & abi.FL_CRASH_OBJECT) != 0).sum() >= 0


def test_full_size_roundabout_1024x40():
    """BASELINE configs[2]: 1024 MultiAgentRoundaboutEnv x 40 agents, 240 beams, respawn on."""
    import torch
    from metadrive_ped_amd.engine import BatchedEngine, HostScene
    from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentRoundaboutEnv
    import oracle_binding as ob
    E, A = 1024, 40
    mk = lambda n: BatchedMultiAgentRoundaboutEnv(dict(num_envs=n, num_scenarios=E, horizon=120,
                                                       vehicle_config=dict(lidar=dict(num_lasers=240, distance=50)))).config
    cfg = mk(E)
    host = HostScene(cfg)
    assert host.A == A and host.n_beams == 240
    acts = [torch.from_numpy(scripted_actions(E, A, t, seed=9)).cuda() for t in range(100)]
    finals = []
    for run in range(2):
        eng = BatchedEngine(cfg, host=host)
        eng.reset()
        for a in acts:
            eng.step(a)
        torch.cuda.synchronize()
        finals.append(eng.download_state())
        del eng
    for k in finals[0]:
        assert finals[0][k].tobytes() == finals[1][k].tobytes(), k
    obs = finals[0]["obs"]
    assert obs.shape == (E * A, 19 + 240) and np.isfinite(obs).all() and (obs >= 0).all() and (obs <= 1).all()
    assert (finals[0]["next_agent_id"] > A).sum() > E // 2       # respawns happened in most envs
    sub = 12
    o = ob.OracleWorld(HostScene(mk(sub)))
    o.reset()
    for a in acts:
        o.step(a[:sub].cpu().numpy())
    for k in ("obs", "reward", "flags", "agent_id"):
        per = finals[0][k].size // E
        assert finals[0][k].reshape(E, -1)[:sub].tobytes() == o.state[k].reshape(sub, -1).tobytes(), k


def test_batched_lidar_sensor_plug_matches_the_oracle():
    """BatchedLidar.perceive (the reference's Lidar.perceive signature and return value) through md_lidar_detect: cloud
    points bit-equal to the oracle's brute-force lidar on the same shape table, detected_objects = the bodies some beam
    hit first, a known geometric answer, beam masks."""
    import oracle_binding as ob
    from metadrive_ped_amd import abi
    from metadrive_ped_amd.mapgen.tables import beam_table
    from metadrive_ped_amd.sensors import BatchedLidar

    class SimpleNamespace:          # hashable by identity, like the reference's objects
        def __init__(self, **kw):
            self.__dict__.update(kw)
    rng = np.random.RandomState(5)
    lidar = BatchedLidar("cuda:0")
    ego = SimpleNamespace(position=(10.0, -3.0), heading_theta=0.3, LENGTH=4.515, WIDTH=1.852)
    ahead = SimpleNamespace(position=(10.0 + 20.0 * np.cos(0.3), -3.0 + 20.0 * np.sin(0.3)), heading_theta=0.3, LENGTH=4.0, WIDTH=1.8)
    cone = SimpleNamespace(position=(10.0 - 8.0 * np.sin(0.3), -3.0 + 8.0 * np.cos(0.3)), heading_theta=0.0, RADIUS=0.2)
    far = SimpleNamespace(position=(200.0, 200.0), heading_theta=0.0, LENGTH=4.0, WIDTH=1.8)
    cloud, found = lidar.perceive(ego, [ego, ahead, cone, far], num_lasers=240, distance=50)
    assert len(cloud) == 240 and found == {ahead, cone}
    assert abs(cloud[0] - (20.0 - 2.0) / 50.0) < 1e-4                 # beam 0 along the heading hits the rear of `ahead`
    assert abs(cloud[60] - (8.0 - 0.2) / 50.0) < 1e-4                 # beam 60 = 90 deg to the left hits the cone
    assert cloud[120] == 1.0
    # a batch of random worlds against the oracle on the very same shape table
    E, M, B = 16, 12, 120
    vehicles, worlds = [], []
    for e in range(E):
        v = SimpleNamespace(position=tuple(rng.uniform(-20, 20, 2)), heading_theta=float(rng.uniform(-3, 3)), LENGTH=4.5, WIDTH=1.85)
        objs = [SimpleNamespace(position=tuple(np.asarray(v.position) + rng.uniform(-45, 45, 2)), heading_theta=float(rng.uniform(-3, 3)),
                                LENGTH=float(rng.uniform(3.5, 6)), WIDTH=float(rng.uniform(1.6, 2.1))) for _ in range(M)]
        vehicles.append(v)
        worlds.append([v] + objs)
    mask = [rng.rand(B) < 0.7 for _ in range(E)]
    clouds, sets = lidar.perceive_batch(vehicles, worlds, B, 50.0, detector_mask=mask)
    cap = M + 1
    shape = np.zeros((E, cap), dtype=abi.SHAPE_DT)
    for e in range(E):
        for j, o in enumerate(worlds[e]):
            shape[e, j] = (o.position[0], o.position[1], np.cos(o.heading_theta), np.sin(o.heading_theta), o.LENGTH / 2, o.WIDTH / 2,
                           abi.KIND_VEHICLE | abi.F_ALIVE, -1)
    want = ob.lidar_raw(shape.reshape(-1), beam_table(B), E, cap, B, 50.0)
    for e in range(E):
        w_e = np.where(mask[e], want[e], np.float32(1.0))
        assert np.asarray(clouds[e], np.float32).tobytes() == w_e.astype(np.float32).tobytes()
        assert all(o in worlds[e][1:] for o in sets[e])
    assert any(len(s_) > 0 for s_ in sets)


@pytest.mark.gpu
def test_random_traffic_gives_every_auto_reset_episode_another_draw():
    """random_traffic=True with envs that reset themselves (manager/traffic_manager.py:335-337: the traffic stream is not re-seeded at
    reset): traffic_draws host-built draws are staged on the device, md_swap_draw hands an env the next one when its episode ends.
    The oracle is stepped beside the engine with the same swap done in numpy: every state array stays bit-identical, and the
    traffic an env starts an episode with changes from episode to episode."""
    import torch
    import oracle_binding as ob
    from helpers import assert_state_equal
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    E, K = 12, 3
    cfg = make_config(dict(num_envs=E, num_scenarios=4, map=3, traffic_density=0.2, start_seed=40, random_traffic=True, traffic_draws=K,
                           horizon=40, auto_reset=True, mover_capacity=0))
    eng = BatchedEngine(cfg)
    hosts = eng.draw_hosts_
    assert len(hosts) == K and len({h.cap for h in hosts}) == 1
    orc = ob.OracleWorld(eng.host)
    eng.reset()
    orc.reset()
    cap = eng.cap
    idx = np.zeros(E, np.int64)
    twins = dict(param="param0", route_nodes="route_nodes0", route_roads="route_roads0", final_lane="final_lane0")
    starts = [[] for _ in range(E)]
    rng = np.random.RandomState(0)
    for t in range(170):
        a = rng.uniform(-1, 1, (E, 1, 2)).astype(np.float32)
        a[..., 0] *= 0.2
        a[..., 1] = np.abs(a[..., 1])
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        for e in np.nonzero(orc.state["need_reset"])[0]:            # what md_swap_draw does
            idx[e] = (idx[e] + 1) % K
            src = hosts[idx[e]].state
            rows = slice(e * cap, (e + 1) * cap)
            for k in BatchedEngine.DRAW_ARRAYS:
                if k in orc.state:
                    orc.state[k][rows] = src[k][rows]
                    if k in twins and twins[k] in orc.state:
                        orc.state[twins[k]][rows] = src[k][rows]
            starts[e].append(int(idx[e]))
        if t % 10 == 9:
            assert_state_equal(eng.download_state(), orc.state, where="random_traffic step %d" % t)
    assert np.array_equal(eng.draw_idx.cpu().numpy(), idx)
    assert all(len(s) >= 3 for s in starts)                         # horizon 40: every env went through several episodes
    sh = [h.state["shape0"].reshape(E, cap) for h in hosts]
    assert all(not np.array_equal(sh[0][e].view(np.uint8), sh[1][e].view(np.uint8)) for e in range(E))   # other traffic in the next episode


@pytest.mark.gpu
def test_random_traffic_checkpoint_resumes_bit_identically():
    """get_state / set_state of an env with staged traffic draws: the snapshot an env is on and its draw index travel with the
    checkpoint, the continuation repeats bit for bit (incl. the episodes that start after it)."""
    import torch
    from metadrive_ped_amd.envs.metadrive_env import BatchedMetaDriveEnv
    E = 8
    env = BatchedMetaDriveEnv(dict(num_envs=E, num_scenarios=E, map=2, traffic_density=0.2, start_seed=7, random_traffic=True,
                                   traffic_draws=3, horizon=30))
    env.reset()
    g = torch.Generator().manual_seed(1)
    acts = torch.rand(200, E, 2, generator=g) * 2 - 1
    acts[..., 1] = acts[..., 1].abs()
    acts[..., 0] *= 0.2
    acts = acts.to(env.engine.device)
    for t in range(70):
        env.step(acts[t])
    ck = env.get_state()
    assert "__draw_idx__" in ck and ck["__draw_idx__"].max() >= 1
    first = []
    for t in range(70, 170):
        o, r, tm, tr, _ = env.step(acts[t])
        first.append((o.clone(), r.clone(), tm.clone()))
    env.set_state(ck)
    for i, t in enumerate(range(70, 170)):
        o, r, tm, tr, _ = env.step(acts[t])
        assert torch.equal(o, first[i][0]) and torch.equal(r, first[i][1]) and torch.equal(tm, first[i][2]), t
    env.close()

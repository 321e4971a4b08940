#!/usr/bin/env python
"""bench.py -- agent-steps/s of the batched MetaDrive step() on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one env.step() of the whole batch: ONE md_step launch that advances every env by 0.1 s
of simulated time (IDM traffic -> kinematic integration -> lane localisation -> contacts -> obs /
reward / done -> 240-beam lidar), with auto-reset of finished envs inside the same launch.
Workload at N=1 is BASELINE.json configs[1]: 4096 batched MetaDriveEnv, 3-block PG map drawn from the
reference's default block distribution, 240-beam lidar, traffic_density 0.1, one env per scenario seed.
For N>1 every rank owns 4096 envs of the global batch (weak scaling, no data-path collective: envs
are independent worlds); rank 0 prints ONE JSON line.  Inputs (state, maps, actions) are resident in
HBM when the timed region starts.

Extra objects in the JSON line:
  roofline     the fused step kernel: algorithmic bytes per launch / average launch duration
               (HIP events on the launch stream) against the HBM peak
  lidar        the stand-alone md_lidar kernel measured the same way (bytes = 16 + 24*M + 4*B per agent)
  cpu_baseline the CPU oracle (oracle/md_oracle.c, "port") on the host cores, bounded sample
  with_gather  (N>1) the same K steps followed by an RCCL all_gather of (obs, reward, flags)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3   # vector FP32, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=300)
    p.add_argument("--warmup", type=int, default=30)
    p.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    p.add_argument("--cap", type=int, default=0, help="mover slots per env (0 = smallest that fits)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-lane-follow", action="store_true", help="skip the scripted-driver operating point")
    p.add_argument("--sub-batches", type=int, default=2,
                   help="double-buffered leg (N=1): the same envs as S sub-batches on S HIP streams; 0 or 1 = skip")
    p.add_argument("--workload", default="metadrive", choices=["metadrive", "safe", "marl", "replay"],
                   help="metadrive = BASELINE configs[1] (the headline line); safe = configs[3] per-GPU shard (8192 "
                        "SafeMetaDriveEnv); marl = configs[2] (1024 x 40-agent roundabout, 240 beams)")
    p.add_argument("--cpu-envs", type=int, default=2048)
    p.add_argument("--cpu-steps", type=int, default=100)
    return p.parse_args()


def cs_dist():
    from metadrive_ped_amd.mapgen.pg import BLOCK_TYPE_DISTRIBUTION_V2
    d = OrderedDict((k, 0.0) for k in BLOCK_TYPE_DISTRIBUTION_V2)
    d["Curve"], d["Straight"] = 0.6, 0.4
    return d


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Rehearsal of the N > 1 control flow on a ONE-GPU box (every rank on cuda:0, gloo instead of RCCL, host tensors
    # for the two collectives): exercises rank / seed / barrier / reduction / printing logic, measures nothing.
    rehearse = os.environ.get("MD_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    import torch
    import torch.distributed as dist
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    E = args.envs
    common = dict(num_envs=E, num_scenarios=E * max(world, 1), env_seed_offset=rank * E, start_seed=0,
                  mover_capacity=args.cap, auto_reset=True, device="cuda:%d" % local_rank)
    if args.workload == "metadrive":
        cfg = make_config(dict(common, map=3, traffic_density=0.1, horizon=1000))
        label = ("BASELINE configs[1]: %d batched MetaDriveEnv per GPU, 3-block PG map (reference default block "
                 "distribution: curves, straights, ramps, X/T intersections, roundabouts), 240-beam lidar, "
                 "traffic_density=0.1, trigger traffic, auto-reset" % E)
    elif args.workload == "safe":
        if args.envs == 4096:
            E = common["num_envs"] = 8192
            common["num_scenarios"], common["env_seed_offset"] = E * max(world, 1), rank * E
        from metadrive_ped_amd.envs.metadrive_env import BatchedSafeMetaDriveEnv
        cfg = make_config(dict(BatchedSafeMetaDriveEnv.SAFE_DEFAULTS, **dict(common, map=3, horizon=1000)))
        label = "BASELINE configs[3] shard: %d SafeMetaDriveEnv per GPU (accident_prob 0.8, density 0.05), 240 beams" % E
    elif args.workload == "replay":
        if args.envs == 4096:
            E = common["num_envs"] = 2048
            common["num_scenarios"], common["env_seed_offset"] = E * max(world, 1), rank * E
        cfg = make_config(dict(common, map=3, traffic_density=0.1, horizon=200, traffic_mode="replay"))
        label = ("BASELINE configs[4] stand-in: %d envs replaying recorded traffic tracks (200 frames, non-reactive) on 3-block "
                 "PG maps, 240-beam lidar -- ScenarioNet data is not available here" % E)
    else:
        if args.envs == 4096:
            E = common["num_envs"] = 1024
            common["num_scenarios"], common["env_seed_offset"] = E * max(world, 1), rank * E
        from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentRoundaboutEnv
        cfg = BatchedMultiAgentRoundaboutEnv(dict(common, vehicle_config=dict(lidar=dict(num_lasers=240, distance=50)))).config
        label = "BASELINE configs[2]: %d MultiAgentRoundaboutEnv x 40 agents per GPU, 240-beam lidar, respawn on" % E
    # host-side scene generation happens BEFORE the GPU / process group are touched (fork pool inside)
    t0 = time.time()
    from metadrive_ped_amd.engine import HostScene
    host = HostScene(cfg)
    rhost = None
    if args.workload == "replay":   # the recording run's scenes: also built before the GPU is touched (fork pool)
        rcfg = make_config(dict(common, map=3, traffic_density=0.1, horizon=200, traffic_mode="trigger"))
        rhost = HostScene(rcfg)
    # double-buffered leg: the same environments as S sub-batches (envs/pipeline.py); their scenes too are generated
    # before the GPU is touched
    sub = None
    if world == 1 and args.sub_batches > 1 and args.workload in ("metadrive", "safe", "marl") and E % args.sub_batches == 0:
        from metadrive_ped_amd.envs.metadrive_env import BatchedMetaDriveEnv
        from metadrive_ped_amd.envs.pipeline import SubBatchedEnvs
        if args.workload == "marl":
            sub = SubBatchedEnvs(BatchedMultiAgentRoundaboutEnv,
                                 dict(common, vehicle_config=dict(lidar=dict(num_lasers=240, distance=50))), sub_batches=args.sub_batches)
        else:
            user = dict(common, map=3, horizon=1000, mover_capacity=host.cap)
            user.update(dict(traffic_density=0.1) if args.workload == "metadrive" else BatchedSafeMetaDriveEnv.SAFE_DEFAULTS)
            user.update(num_scenarios=common["num_scenarios"])
            sub = SubBatchedEnvs(BatchedMetaDriveEnv, user, sub_batches=args.sub_batches)
        sub.build_host()
    torch.cuda.set_device(local_rank)
    tracks = None
    if args.workload == "replay":
        # record the tracks first: the same scenarios with reacting (trigger-mode IDM) traffic, 200 steps
        reng = BatchedEngine(rcfg, host=rhost)
        reng.reset()
        reng.start_recording(200)
        g0 = torch.Generator(device="cpu")
        g0.manual_seed(1000 + rank)
        for i in range(200):
            a0 = torch.rand(E, 1, 2, generator=g0) * 2 - 1
            a0[..., 1] = a0[..., 1].abs() * 0.9 + 0.1
            a0[..., 0] *= 0.25
            reng.step(a0.to(reng.device))
        tracks = reng.stop_recording()
        del reng
    eng = BatchedEngine(cfg, host=host)
    if tracks is not None:
        eng.set_tracks(tracks)
    build_s = time.time() - t0
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = eng.device
    A, cap, B = eng.A, eng.cap, eng.n_beams

    gen = torch.Generator(device="cpu")
    gen.manual_seed(rank)
    n_act = 64 if E * A <= 8192 else 16
    actions = (torch.rand(n_act, E, A, 2, generator=gen) * 2 - 1)
    actions[..., 1] = actions[..., 1].abs() * 0.9 + 0.1  # mostly forward, so that envs meet traffic and curves
    actions[..., 0] *= 0.25
    actions = actions.to(dev)

    eng.reset()
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    def run(k, base=0):
        for i in range(k):
            eng.step(actions[(base + i) % n_act])

    run(args.warmup)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    # multi-agent: only slots holding a live agent count as agent-steps (dying / free slots do not)
    sf = eng.shape_f.view(torch.int32)[:, :A, 6]
    active_frac = float((((sf & 0x10) != 0) & ((sf & 0x80) == 0)).float().mean().item()) if A > 1 else 1.0
    total_agent_steps = args.steps * E * A * world * active_frac
    value = total_agent_steps / elapsed

    # ---- per-launch duration of the fused step kernel with HIP events on the launch stream ----
    # One event pair around a batch of back-to-back launches (a pair per launch adds ~8 us of its own to a 100 us
    # kernel); the quotient contains the ~1.5 us dependent-launch gaps, which is what rocprofv3's per-kernel
    # average agrees with to a few per cent.
    # (Fresh actions every launch, like the timed loop: with one action repeated the envs fall into short, cheap
    # episodes and the kernel looks 25 % faster than it is.)
    n_ev = 50
    torch.cuda.synchronize()
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev_a.record()
    for i in range(n_ev):
        eng.step(actions[i % n_act])
    ev_b.record()
    torch.cuda.synchronize()
    step_ms_avg = ev_a.elapsed_time(ev_b) / n_ev
    # present movers per env right now (alive, kind != none)
    flags = eng.shape_f.view(torch.int32)[..., 6]
    present = ((flags & 0x10) != 0) & ((flags & 0xF) != 0)
    M = float(present.sum().item()) / E           # movers per env (agents + traffic)
    T = M - A
    # whole step per agent ~ 2.6 KB + 136 B * T/A  (state R/W, action, navi/route, obs, flags)
    bytes_step = (2600.0 + 136.0 * T / A) * E * A
    achieved = bytes_step / (step_ms_avg * 1e-3) / 1e9
    # HBM bytes per launch from the committed PMC passes (tools/profile_round.sh; counters cannot be read
    # from inside this process), same workload only
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        if pmc.get("workload") == args.workload and E == 4096:
            traffic, traffic_src = pmc["bytes_per_launch"].get("env_kernel<511>"), "profiles/" + pmc["source"]
    except (OSError, ValueError, KeyError):
        pass
    roofline = dict(bound="hbm", kernel="env_kernel<511> (fused md_step)", achieved=round(achieved, 2), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_src,
                    bytes_per_launch=int(bytes_step), avg_launch_us=round(step_ms_avg * 1e3, 2),
                    movers_per_env=round(M, 2))

    # ---- stand-alone lidar kernel ----
    out = torch.empty(E * A, B, device=dev)
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.lidar(out, B, 0)
    torch.cuda.synchronize()
    ev_a.record()
    for i in range(n_ev):
        eng.lidar(out, B, 0)
    ev_b.record()
    torch.cuda.synchronize()
    lid_ms = ev_a.elapsed_time(ev_b) / n_ev
    bytes_lidar = (16.0 + 24.0 * (M - 1) + 4.0 * B) * E * A
    lidar = dict(kernel="env_kernel<128> (md_lidar)", avg_launch_us=round(lid_ms * 1e3, 2),
                 achieved=round(bytes_lidar / (lid_ms * 1e-3) / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                 frac=round(bytes_lidar / (lid_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), bytes_per_launch=int(bytes_lidar),
                 flops_per_launch=int(30.0 * B * (M - 1) * E * A))

    # ---- attainable HBM bandwidth on this box: HIP stream-copy kernel through the same library ----
    import ctypes as C
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).fill_(1)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(3):
        eng._check(eng.lib.md_probe_stream_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), nbytes, st), "copy")
    ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ca.record()
    for _ in range(10):
        eng._check(eng.lib.md_probe_stream_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), nbytes, st), "copy")
    cb.record()
    torch.cuda.synchronize()
    copy_gbs = 2.0 * nbytes * 10 / (ca.elapsed_time(cb) * 1e-3) / 1e9
    del src, dst
    roofline["peak_attainable"] = round(copy_gbs, 1)            # measured stream copy (read + write), GB/s
    roofline["frac_attainable"] = round(achieved / copy_gbs, 5)
    lidar["fp32_tflops"] = round(lidar["flops_per_launch"] / (lid_ms * 1e-3) / 1e12, 2)
    lidar["fp32_peak_tflops"] = FP32_PEAK_TFLOPS
    lidar["fp32_frac"] = round(lidar["fp32_tflops"] / FP32_PEAK_TFLOPS, 4)

    # ---- second operating point (N=1, single-agent workloads): a scripted lane-following driver instead of random
    #      actions.  Episodes last several hundred steps, more traffic blocks get triggered (about twice the driving
    #      vehicles per env), so the step is heavier: reported next to `value`, never instead of it. ----
    lane_follow = None
    if rank == 0 and world == 1 and A == 1 and args.workload in ("metadrive", "safe") and not args.no_lane_follow:
        o_hd, o_v, o_lat = eng.host.obs_base + (eng.host.n_side or 2), eng.host.obs_base + (eng.host.n_side or 2) + 1, \
            eng.host.obs_base + (eng.host.n_side or 2) + 6
        act_buf = torch.zeros(E, 1, 2, device=dev)

        def drive():
            ob_ = eng.obs[:, 0, :]
            act_buf[:, 0, 0] = (4.0 * (ob_[:, o_hd] - 0.5) + 2.0 * (ob_[:, o_lat] - 0.5)).clamp_(-1.0, 1.0)
            act_buf[:, 0, 1] = (ob_[:, o_v] < 0.35).to(torch.float32) * 0.5
            eng.step(act_buf)

        eng.reset()
        for i in range(300):            # let the batch reach its steady mix of episode phases
            drive()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_lf = 200
        for i in range(n_lf):
            drive()
        torch.cuda.synchronize()
        dt_lf = time.perf_counter() - t0
        f_lf = eng.shape_f.view(torch.int32)[..., 6]
        drv = (((f_lf & 0x10) != 0) & ((f_lf & 0x40) == 0) & ((f_lf & 0xF) == 1) & ((f_lf & 0x80) == 0)).sum().item() / E
        lane_follow = dict(value=round(n_lf * E / dt_lf, 1), unit="agent-steps/s", ms_per_step=round(dt_lf / n_lf * 1e3, 4),
                           driving_vehicles_per_env=round(drv, 2),
                           policy="steer = clip(4 (heading_diff - .5) + 2 (lateral - .5)), throttle .5 below 28 km/h; "
                                  "computed from the observation with four torch element-wise ops per step (included)")

    # ---- double-buffered stepping (N=1): the same environments as S sub-batches, each on its own HIP stream, so
    #      that the draining tail of one launch overlaps the body of another (envs/pipeline.py).  A rollout loop that
    #      runs its policy per sub-batch gets this rate; `value` above stays the single-launch rate. ----
    double_buffered = None
    if sub is not None:
        S = sub.sub_batches
        sub.reset()
        sub.synchronize()
        sub_acts = [actions[:, k * (E // S):(k + 1) * (E // S)].contiguous() for k in range(S)]
        torch.cuda.synchronize()

        def run_sub(k_steps, base=0):
            for i in range(k_steps):
                for k, env in enumerate(sub.envs):
                    with sub.on(k):
                        env.engine.step(sub_acts[k][(base + i) % n_act])

        run_sub(args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_sub(args.steps, args.warmup)
        torch.cuda.synchronize()
        dt_db = time.perf_counter() - t0
        sf_db = torch.cat([e_.engine.shape_f.view(torch.int32)[:, :A, 6] for e_ in sub.envs])
        act_db = float((((sf_db & 0x10) != 0) & ((sf_db & 0x80) == 0)).float().mean().item()) if A > 1 else 1.0
        double_buffered = dict(value=round(args.steps * E * A * act_db / dt_db, 1), unit="agent-steps/s", sub_batches=S,
                               envs_per_sub_batch=E // S, ms_per_step=round(dt_db / args.steps * 1e3, 4),
                               note="same %d envs, same actions; one step = one md_step launch per sub-batch, each on "
                                    "its own HIP stream; no cross-stream wait inside the timed region" % E)
        sub.close()

    # ---- optional gather (N>1): obs + reward + flags to every rank over RCCL ----
    with_gather = None
    if world > 1:
        obs_g = torch.empty(world * E * A * eng.obs_dim, device=dev)
        rew_g = torch.empty(world * E * A, device=dev)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            eng.step(actions[i % n_act])
            if rehearse:   # gloo: host copies stand in for the RCCL gather
                o_cpu, r_cpu = eng.obs.reshape(-1).cpu(), eng.reward.reshape(-1).cpu()
                dist.all_gather([torch.empty_like(o_cpu) for _ in range(world)], o_cpu)
                dist.all_gather([torch.empty_like(r_cpu) for _ in range(world)], r_cpu)
            else:
                dist.all_gather_into_tensor(obs_g, eng.obs.reshape(-1))
                dist.all_gather_into_tensor(rew_g, eng.reward.reshape(-1))
        torch.cuda.synchronize()
        barrier()
        eg = torch.tensor([time.perf_counter() - t0], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(eg, op=dist.ReduceOp.MAX)
        with_gather = dict(value=round(total_agent_steps / float(eg.item()), 1), unit="agent-steps/s",
                           collective="all_gather_into_tensor(obs,reward)")

    # ---- CPU baseline: the oracle on the host cores, bounded sample (rank 0, N=1 only) ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "metadrive":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import numpy as np
        import oracle_binding as ob
        from metadrive_ped_amd.engine import HostScene
        n_cpu = min(args.cpu_envs, E)
        ccfg = dict(cfg)
        ccfg["num_envs"] = n_cpu
        host = HostScene(ccfg)
        orc = ob.OracleWorld(host)
        # the GPU box gives one GPU's job a 16-CPU share, whatever os.cpu_count() says
        cores = min(len(os.sched_getaffinity(0)), 16)
        orc.reset()
        acts = actions[:, :n_cpu].cpu().numpy()
        t0 = time.perf_counter()
        for i in range(args.cpu_steps):
            orc.step(acts[i % n_act], threads=cores)
        dt = time.perf_counter() - t0
        # single thread, on a slice of the same envs (SURVEY 8d asks for both figures)
        n1 = max(1, n_cpu // 16)
        ccfg1 = dict(cfg)
        ccfg1["num_envs"] = n1
        orc1 = ob.OracleWorld(HostScene(ccfg1))
        orc1.reset()
        t1 = time.perf_counter()
        for i in range(args.cpu_steps):
            orc1.step(acts[i % n_act][:n1], threads=1)
        dt1 = time.perf_counter() - t1
        try:
            with open("/proc/cpuinfo") as fh:
                cpu_model = [l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")][0]
        except (OSError, IndexError):
            cpu_model = "unknown"
        cpu_baseline = dict(value=round(n_cpu * A * args.cpu_steps / dt, 1), unit="agent-steps/s", cores=cores, kind="port",
                            sample="%d envs x %d steps of the same workload, oracle/md_oracle.c ref_step_mt on %d threads"
                                   % (n_cpu, args.cpu_steps, cores),
                            single_thread_value=round(n1 * A * args.cpu_steps / dt1, 1),
                            single_thread_sample="%d envs x %d steps, 1 thread" % (n1, args.cpu_steps),
                            cpu_model=cpu_model, nproc=os.cpu_count())

    if rank == 0:
        line = OrderedDict(
            metric="agent-steps/sec at 4096 envs x 240-beam lidar" if args.workload == "metadrive" else
            "agent-steps/sec (%s workload)" % args.workload, value=round(value, 1), unit="agent-steps/s",
            n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4),
            higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
            config=dict(workload=label,
                        envs_per_gpu=E, active_agent_fraction=round(active_frac, 3), agents_per_env=A, mover_capacity=cap, n_beams=B, sharding="env-range per rank"),
            roofline=roofline, lidar=lidar, cpu_baseline=cpu_baseline, lane_follow_policy=lane_follow, double_buffered=double_buffered, with_gather=with_gather,
            host_build_s=round(build_s, 1))
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

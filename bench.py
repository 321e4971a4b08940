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

Launching: `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N
rank processes ITSELF (fresh children of a parent that never touches the GPU); under torchrun
(WORLD_SIZE set) it is one rank and WORLD_SIZE must equal --gpus.

Steady state: the timed region starts after an UNTIMED pre-roll (--preroll, default 300 steps)
whatever --warmup says, so that the batch holds its stationary mix of episode phases (traffic blocks
triggered, envs resetting at the stationary rate), and lasts at least 50 ms (the K steps are repeated
R times; `timed_steps` = R*K).  The per-launch kernel duration in `roofline` is taken with HIP events
over that SAME region, and the line is refused (exit 3) if the two disagree.

Extra objects in the JSON line:
  roofline     the fused step kernel: algorithmic bytes per launch / average launch duration
               (HIP events on the launch stream over the timed region) against the HBM peak
  lidar        the stand-alone md_lidar kernel measured the same way (bytes = 16 + 24*M + 4*B per agent)
  cpu_baseline the CPU oracle (oracle/md_oracle.c, "port", rebuilt -O3 -march=native on this host) on the
               host cores, bounded sample
  env_api      the same steps through BatchedMetaDriveEnv.step (the Gymnasium-shaped boundary: done flags,
               lazy info dict)
  shared_maps  the same envs on 1 / 64 distinct maps (the reference's default num_scenarios is 1): md_step's kernel as
               config["step_kernel"] = "auto" picks it (one wave per env on few maps)
  with_gather  (N>1) the same steps, each followed by ONE RCCL all_gather of the packed slab obs | reward | done+flags
"""
import argparse
import json
import math
import os
import pickle
import socket
import subprocess
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3   # vector FP32, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate
MIN_TIMED_S = 0.05
SHARED_MAPS = (1, 64)   # distinct maps of the shared-maps operating points


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=300)
    p.add_argument("--warmup", type=int, default=30)
    p.add_argument("--preroll", type=int, default=300,
                   help="untimed steps after reset() before --warmup: the batch reaches its stationary episode mix")
    p.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    p.add_argument("--cap", type=int, default=0, help="mover slots per env (0 = smallest that fits)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-lane-follow", action="store_true", help="skip the scripted-driver operating point")
    p.add_argument("--no-env-api", action="store_true", help="skip the BatchedMetaDriveEnv.step leg")
    p.add_argument("--no-shared-maps", action="store_true", help="skip the shared-maps operating points (num_scenarios 1 / 64)")
    p.add_argument("--sub-batches", type=int, default=2,
                   help="double-buffered leg (N=1): the same envs as S sub-batches on S HIP streams; 0 or 1 = skip")
    p.add_argument("--workload", default="metadrive", choices=["metadrive", "safe", "marl", "replay", "scenario"],
                   help="metadrive = BASELINE configs[1] (the headline line); safe = configs[3] per-GPU shard (8192 "
                        "SafeMetaDriveEnv); marl = configs[2] (1024 x 40-agent roundabout, 240 beams); scenario = "
                        "configs[4] (2048 ScenarioEnv scenes, reactive TrajectoryIDM traffic, synthetic scenario data)")
    p.add_argument("--cpu-envs", type=int, default=2048)
    p.add_argument("--cpu-steps", type=int, default=100)
    p.add_argument("--host-cache", default="",
                   help="pickle of the host-side scenes (written if absent, read if present): profiled runs load it "
                        "instead of forking map-builder workers from a process whose GPU the profiler has initialised")
    return p.parse_args()


def cs_dist():
    from metadrive_ped_amd.mapgen.pg import BLOCK_TYPE_DISTRIBUTION_V2
    d = OrderedDict((k, 0.0) for k in BLOCK_TYPE_DISTRIBUTION_V2)
    d["Curve"], d["Straight"] = 0.6, 0.4
    return d


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as fresh child processes.  This parent has
    not imported torch and never touches the GPU; it waits and exits with the first non-zero child code."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = env.get("MASTER_PORT") or str(_free_port())
    env["WORLD_SIZE"] = str(args.gpus)
    env["MD_BENCH_CHILD"] = "1"
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    if rc:   # a dead rank leaves the others in a collective: end them (exact PIDs, our own children)
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def cpu_threads():
    """Host threads this process may really use: the cgroup CPU quota when there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(math.ceil(float(quota) / float(period)))))
    except (OSError, ValueError):
        pass
    return n


def build_fast_oracle():
    """The CPU baseline's build of the oracle: -O3 -march=native for THIS host (the parity tests keep the -O2 build;
    -ffp-contract=off stays, so the numbers do not change).  Built at bench time: -march=native code must not travel."""
    out = os.path.join(ROOT, "oracle", "_build", "libmdoracle_native.so")
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "native"])
    return out


def host_cache_key(args, rank, world):
    """What a cached set of host scenes was built for: ABI, workload, sizes, rank, and the sources that shape the tables."""
    import hashlib
    from metadrive_ped_amd import abi
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "metadrive_ped_amd")
    for rel in ("abi.py", "engine.py", "scene.py", "scenario.py", "marl.py", "config.py", "pg_space.py", "rng.py",
                os.path.join("mapgen", "pg.py"), os.path.join("mapgen", "tables.py"), os.path.join("mapgen", "lanes.py")):
        with open(os.path.join(pkg, rel), "rb") as fh:
            h.update(fh.read())
    return dict(abi=abi.MD_ABI_VERSION, workload=args.workload, envs=args.envs, cap=args.cap, sub=args.sub_batches, rank=rank,
                world=world, cpu=(args.cpu_envs, bool(args.no_cpu_baseline)), src=h.hexdigest()[:16],
                step_kernel=os.environ.get("MD_STEP_KERNEL", ""), shared=bool(args.no_shared_maps))


def load_or_build_hosts(path, build, key):
    """--host-cache: our own pickle of the host scenes, valid only for the key it was written with (a stale one is rebuilt)."""
    if path and os.path.exists(path):
        with open(path, "rb") as fh:   # our own file, written by the branch below
            blob = pickle.load(fh)
        if isinstance(blob, dict) and blob.get("key") == key:
            return blob["hosts"]
        print("bench.py: host cache %s was built for %r, not %r: rebuilding" % (path, blob.get("key") if isinstance(blob, dict) else None, key),
              file=sys.stderr)
    hosts = build()
    if path:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "wb") as fh:
            pickle.dump(dict(key=key, hosts=hosts), fh, protocol=4)
    return hosts


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%d but --gpus %d: launch with `python bench.py --gpus N` (it starts the ranks "
                 "itself) or with torchrun --nproc-per-node N ... --gpus N" % (world, args.gpus))
    # Rehearsal of the N > 1 control flow on a ONE-GPU box (every rank on cuda:0, gloo instead of RCCL, host tensors
    # for the two collectives): exercises rank / seed / barrier / reduction / printing logic, measures nothing.
    rehearse = os.environ.get("MD_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # every rank builds its own 4096 maps: share the host cores
        os.environ.setdefault("MD_BUILD_WORKERS", str(max(1, min(32, (os.cpu_count() or 8) // world))))
    # the host build workers first: plain child interpreters, started while this process has no GPU context, kept to the end
    from metadrive_ped_amd import hostpool
    if not (args.host_cache and os.path.exists(args.host_cache)):
        hostpool.start()
    import torch
    import torch.distributed as dist
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine, HostScene

    E = args.envs
    common = dict(num_envs=E, num_scenarios=E * max(world, 1), env_seed_offset=rank * E, start_seed=0,
                  mover_capacity=args.cap, auto_reset=True, device="cuda:%d" % local_rank,
                  build_cache=True)   # the sub-batch leg and the CPU baseline step the same scenario seeds: built once
    env_cls = None
    if args.workload == "metadrive":
        user = dict(common, map=3, traffic_density=0.1, horizon=1000)
        cfg = make_config(user)
        label = ("BASELINE configs[1]: %d batched MetaDriveEnv per GPU, 3-block PG map (reference default block "
                 "distribution: curves, straights, ramps, X/T intersections, roundabouts), 240-beam lidar, "
                 "traffic_density=0.1, trigger traffic, auto-reset" % E)
    elif args.workload == "safe":
        if args.envs == 4096:
            E = common["num_envs"] = 8192
            common["num_scenarios"], common["env_seed_offset"] = E * max(world, 1), rank * E
        from metadrive_ped_amd.envs.metadrive_env import BatchedSafeMetaDriveEnv
        user = dict(BatchedSafeMetaDriveEnv.SAFE_DEFAULTS, **dict(common, map=3, horizon=1000))
        cfg = make_config(user)
        label = "BASELINE configs[3] shard: %d SafeMetaDriveEnv per GPU (accident_prob 0.8, density 0.05), 240 beams" % E
    elif args.workload == "replay":
        if args.envs == 4096:
            E = common["num_envs"] = 2048
            common["num_scenarios"], common["env_seed_offset"] = E * max(world, 1), rank * E
        cfg = make_config(dict(common, map=3, traffic_density=0.1, horizon=200, traffic_mode="replay"))
        label = ("BASELINE configs[4] stand-in: %d envs replaying recorded traffic tracks (200 frames, non-reactive) on 3-block "
                 "PG maps, 240-beam lidar -- ScenarioNet data is not available here" % E)
    elif args.workload == "scenario":
        if args.envs == 4096:
            E = common["num_envs"] = 2048
            common["num_scenarios"], common["env_seed_offset"] = E * max(world, 1), rank * E
        from metadrive_ped_amd.envs.scenario_env import scenario_bench_config
        cfg = scenario_bench_config(common)
        label = ("BASELINE configs[4]: %d ScenarioEnv scenes per GPU (synthetic scenario descriptions: polyline lanes, "
                 "recorded tracks), reactive TrajectoryIDMPolicy traffic, 240-beam lidar" % E)
    else:
        if args.envs == 4096:
            E = common["num_envs"] = 1024
            common["num_scenarios"], common["env_seed_offset"] = E * max(world, 1), rank * E
        from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentRoundaboutEnv
        cfg = BatchedMultiAgentRoundaboutEnv(dict(common, vehicle_config=dict(lidar=dict(num_lasers=240, distance=50)))).config
        label = "BASELINE configs[2]: %d MultiAgentRoundaboutEnv x 40 agents per GPU, 240-beam lidar, respawn on" % E

    # ---- host-side scene generation: ALL of it happens BEFORE the GPU / process group are touched (fork pools
    #      inside HostScene), including the scenes of the CPU baseline and of the extra legs ----
    t0 = time.time()
    want_sub = (world == 1 and args.sub_batches > 1 and args.workload in ("metadrive", "safe", "marl")
                and E % args.sub_batches == 0)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "metadrive"
    want_shared = world == 1 and args.workload == "metadrive" and not args.no_shared_maps
    n_cpu = min(args.cpu_envs, E)
    n_cpu1 = max(1, n_cpu // 16)

    def build_hosts():
        if args.workload == "scenario":
            from metadrive_ped_amd.scenario import ScenarioHostScene, synthetic_scenarios
            return dict(main=ScenarioHostScene(cfg, synthetic_scenarios(E, rank * E)))
        h = dict(main=HostScene(cfg))
        if args.workload == "replay":   # the recording run's scenes
            h["rcfg"] = make_config(dict(common, map=3, traffic_density=0.1, horizon=200, traffic_mode="trigger"))
            h["record"] = HostScene(h["rcfg"])
        if want_sub:
            if args.workload == "marl":
                sub_user = dict(common, vehicle_config=dict(lidar=dict(num_lasers=240, distance=50)))
            else:
                sub_user = dict(common, map=3, horizon=1000, mover_capacity=h["main"].cap)
                sub_user.update(dict(traffic_density=0.1) if args.workload == "metadrive" else BatchedSafeMetaDriveEnv.SAFE_DEFAULTS)
                sub_user.update(num_scenarios=common["num_scenarios"])
            h["sub_user"] = sub_user
            from metadrive_ped_amd.envs.pipeline import SubBatchedEnvs
            from metadrive_ped_amd.envs.metadrive_env import BatchedMetaDriveEnv
            sub_ = SubBatchedEnvs(BatchedMultiAgentRoundaboutEnv if args.workload == "marl" else BatchedMetaDriveEnv,
                                  sub_user, sub_batches=args.sub_batches)
            h["sub_hosts"] = sub_.build_host()
        if want_shared:
            # the same envs on FEW distinct maps (the reference's default is num_scenarios = 1): step_kernel "auto" picks the
            # wave-per-env kernel there
            h["shared"] = {n: HostScene(make_config(dict(user, num_scenarios=n, env_seed_offset=0))) for n in SHARED_MAPS}
        if want_cpu:
            for key, n in (("cpu", n_cpu), ("cpu1", n_cpu1)):
                c = dict(cfg)
                c["num_envs"] = n
                h[key] = HostScene(c)
        return h

    hosts = load_or_build_hosts(args.host_cache, build_hosts, host_cache_key(args, rank, world))
    hostpool.clear_memo()
    host = hosts["main"]
    cpu_lib = build_fast_oracle() if want_cpu else None
    build_s = time.time() - t0

    torch.cuda.set_device(local_rank)
    sub = None
    if want_sub and "sub_hosts" in hosts:
        from metadrive_ped_amd.envs.metadrive_env import BatchedMetaDriveEnv
        from metadrive_ped_amd.envs.pipeline import SubBatchedEnvs
        sub = SubBatchedEnvs(BatchedMultiAgentRoundaboutEnv if args.workload == "marl" else BatchedMetaDriveEnv,
                             hosts["sub_user"], sub_batches=args.sub_batches)
        sub._hosts = hosts["sub_hosts"]
    tracks = None
    if args.workload == "replay":
        # record the tracks first: the same scenarios with reacting (trigger-mode IDM) traffic, 200 steps
        reng = BatchedEngine(hosts["rcfg"], host=hosts["record"])
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
    if world > 1:
        # the communication libraries print connection notes on fd 1 ("[Gloo] Rank 0 is connected to ..."): stdout of this
        # program is the ONE JSON line, so fd 1 points at stderr while the process group comes up (and for the first collective)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearse:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
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

    cursor = [0]

    def run(k):
        for _ in range(k):
            eng.step(actions[cursor[0] % n_act])
            cursor[0] += 1

    # ---- untimed pre-roll to the stationary episode mix, whatever --warmup is ----
    tp = time.perf_counter()
    run(max(args.preroll, 0))
    torch.cuda.synchronize()
    est = (time.perf_counter() - tp) / max(args.preroll, 1) if args.preroll > 0 else 1e-4
    repeats = max(1, int(math.ceil(MIN_TIMED_S / max(args.steps * est, 1e-9))))
    rep_t = torch.tensor([repeats], dtype=torch.int64, device="cpu" if (rehearse or world == 1) else dev)
    if world > 1:
        dist.all_reduce(rep_t, op=dist.ReduceOp.MAX)    # every rank times the same number of steps
    repeats = int(rep_t.item())
    timed_steps = args.steps * repeats

    run(args.warmup)
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_a.record()
    run(timed_steps)
    ev_b.record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # one event pair around the whole timed region (a pair per launch adds ~8 us of its own to each kernel): the
    # quotient contains the ~1.5 us dependent-launch gaps; rocprofv3's per-kernel average agrees to a few per cent
    step_ms_avg = ev_a.elapsed_time(ev_b) / timed_steps
    el = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    # multi-agent: only slots holding a live agent count as agent-steps (dying / free slots do not)
    sf = eng.shape_f.view(torch.int32)[:, :A, 6]
    active_frac = float((((sf & 0x10) != 0) & ((sf & 0x80) == 0)).float().mean().item()) if A > 1 else 1.0
    total_agent_steps = timed_steps * E * A * world * active_frac
    value = total_agent_steps / elapsed
    ms_per_step = elapsed / timed_steps * 1e3
    if world == 1 and ms_per_step < step_ms_avg / 1.05:
        sys.stderr.write("bench.py: inconsistent timing: wall %.4f ms/step < HIP-event %.4f ms/launch over the same %d "
                         "steps; refusing to print a line\n" % (ms_per_step, step_ms_avg, timed_steps))
        sys.exit(3)

    # present movers per env right now (alive, kind != none)
    flags = eng.shape_f.view(torch.int32)[..., 6]
    present = ((flags & 0x10) != 0) & ((flags & 0xF) != 0)
    M = float(present.sum().item()) / E           # movers per env (agents + traffic)
    T = M - A
    drv_now = float((((flags & 0x10) != 0) & ((flags & 0x40) == 0) & ((flags & 0xF) == 1) & ((flags & 0x80) == 0)).sum().item()) / E
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
            traffic, traffic_src = (pmc["bytes_per_launch"].get("step_kernel") or pmc["bytes_per_launch"].get("env_kernel<511>")), "profiles/" + pmc["source"]
    except (OSError, ValueError, KeyError):
        pass
    roofline = dict(bound="hbm", kernel="fused md_step kernel", achieved=round(achieved, 2), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_src,
                    bytes_per_launch=int(bytes_step), avg_launch_us=round(step_ms_avg * 1e3, 2),
                    movers_per_env=round(M, 2), driving_vehicles_per_env=round(drv_now, 2),
                    limiter="instruction issue / dependent-chain latency, not HBM: the path moves ~15 MB per launch "
                            "(SURVEY 8d); frac is the BASELINE metric, not a claim that HBM bounds the kernel")

    # ---- stand-alone lidar kernel ----
    n_ev = 50
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
    lidar = dict(kernel="lidar_kernel (md_lidar)", avg_launch_us=round(lid_ms * 1e3, 2),
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

    # ---- the Gymnasium-shaped boundary (N=1, single-agent): BatchedMetaDriveEnv.step on the same scenes and the
    #      same actions: engine.step + the (terminated, truncated) flags + the lazy info dict, per step ----
    env_api = None
    if rank == 0 and world == 1 and A == 1 and args.workload in ("metadrive", "safe") and not args.no_env_api:
        from metadrive_ped_amd.envs.metadrive_env import BatchedMetaDriveEnv
        env = BatchedMetaDriveEnv(dict(user, mover_capacity=host.cap) if args.workload == "metadrive" else
                                  dict(user, mover_capacity=host.cap))
        env.lazy_init(host=host)
        env.reset()
        acts2 = actions[:, :, 0, :].contiguous()
        for i in range(max(args.preroll, 0)):
            env.step(acts2[i % n_act])
        torch.cuda.synchronize()
        n_api = max(args.steps, int(math.ceil(MIN_TIMED_S / max(est, 1e-9))))
        t0 = time.perf_counter()
        for i in range(n_api):
            o_, r_, te_, tr_, info_ = env.step(acts2[i % n_act])
        torch.cuda.synchronize()
        dt_api = time.perf_counter() - t0
        env_api = dict(value=round(n_api * E / dt_api, 1), unit="agent-steps/s", ms_per_step=round(dt_api / n_api * 1e3, 4),
                       steps=n_api, surface="BatchedMetaDriveEnv.step -> (obs, reward, terminated, truncated, LazyInfo): "
                                            "one md_step launch per step and no other device op (terminated / truncated are written by the kernel: MdState.done_out), info values on demand")
        env.close()

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

        run_sub(max(args.preroll, 0) + args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_sub(timed_steps, args.preroll + args.warmup)
        torch.cuda.synchronize()
        dt_db = time.perf_counter() - t0
        sf_db = torch.cat([e_.engine.shape_f.view(torch.int32)[:, :A, 6] for e_ in sub.envs])
        act_db = float((((sf_db & 0x10) != 0) & ((sf_db & 0x80) == 0)).float().mean().item()) if A > 1 else 1.0
        double_buffered = dict(value=round(timed_steps * E * A * act_db / dt_db, 1), unit="agent-steps/s", sub_batches=S,
                               envs_per_sub_batch=E // S, ms_per_step=round(dt_db / timed_steps * 1e3, 4),
                               note="same %d envs, same actions; one step = one md_step launch per sub-batch, each on "
                                    "its own HIP stream; no cross-stream wait inside the timed region" % E)
        sub.close()

    # ---- shared maps (N=1): the same number of envs on 1 / 64 distinct maps; md_step's kernel is chosen by the engine
    #      (step_kernel "auto": one wave per env when a large batch shares few maps) ----
    shared_maps = None
    if want_shared and "shared" in hosts:
        shared_maps = []
        for n_maps in SHARED_MAPS:
            se = BatchedEngine(hosts["shared"][n_maps].cfg, host=hosts["shared"][n_maps])
            se.reset()
            for i in range(max(args.preroll, 0)):
                se.step(actions[i % n_act])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(timed_steps):
                se.step(actions[i % n_act])
            torch.cuda.synchronize()
            dt_s = time.perf_counter() - t0
            shared_maps.append(dict(num_scenarios=n_maps, step_kernel=se.host.step_kernel, ms_per_step=round(dt_s / timed_steps * 1e3, 4),
                                    value=round(timed_steps * E * A / dt_s, 1), unit="agent-steps/s"))
            del se

    # ---- gather leg (N>1): the rank's whole step output -- obs | reward | terminated, truncated, flags, one allocation
    #      (BatchedEngine.out_slab) -- to every rank with ONE RCCL collective per step, timed over the same number of steps
    #      as the main leg (>= MIN_TIMED_S) ----
    with_gather = None
    if world > 1:
        from metadrive_ped_amd.sharding import gather_step_slab
        slab = eng.out_slab
        gathered = torch.empty((world, slab.numel()), dtype=torch.uint8, device="cpu" if rehearse else dev)
        cpu_slab = torch.empty(slab.numel(), dtype=torch.uint8).pin_memory() if rehearse else None

        def gather_once():
            if rehearse:          # gloo rehearsal on one GPU: a host copy stands in for the device-side RCCL gather
                cpu_slab.copy_(slab)
                gather_step_slab(cpu_slab, out=gathered)
            else:
                gather_step_slab(slab, out=gathered)
        for i in range(min(args.warmup, 5)):
            eng.step(actions[i % n_act])
            gather_once()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(timed_steps):
            eng.step(actions[i % n_act])
            gather_once()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        eg = torch.tensor([time.perf_counter() - t0], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(eg, op=dist.ReduceOp.MAX)
        dt_g = float(eg.item())
        with_gather = dict(value=round(timed_steps * E * A * world * active_frac / dt_g, 1), unit="agent-steps/s",
                           ms_per_step=round(dt_g / timed_steps * 1e3, 4), steps=timed_steps,
                           replicas_only_ms_per_step=round(ms_per_step, 4),
                           collective="one all_gather_into_tensor per step of the packed slab obs | reward | (terminated, truncated, flags)",
                           bytes_per_rank_per_step=int(slab.numel()), bytes_gathered_per_step=int(slab.numel()) * world,
                           backend="gloo (one-GPU rehearsal, host copies)" if rehearse else "nccl (RCCL)")

    # ---- CPU baseline: the oracle on the host cores, bounded sample (rank 0, N=1 only) ----
    cpu_baseline = None
    if want_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding as ob
        cores = cpu_threads()
        lib = ob.load(cpu_lib)
        orc = ob.OracleWorld(hosts["cpu"], lib=lib)
        orc.reset()
        acts = actions[:, :n_cpu].cpu().numpy()
        for i in range(20):               # untimed: first touches, and the traffic starts to wake up
            orc.step(acts[i % n_act], threads=cores)
        t0 = time.perf_counter()
        for i in range(args.cpu_steps):
            orc.step(acts[i % n_act], threads=cores)
        dt = time.perf_counter() - t0
        # single thread, on a slice of the same envs (SURVEY 8d asks for both figures)
        orc1 = ob.OracleWorld(hosts["cpu1"], lib=lib)
        orc1.reset()
        t1 = time.perf_counter()
        for i in range(args.cpu_steps):
            orc1.step(acts[i % n_act][:n_cpu1], threads=1)
        dt1 = time.perf_counter() - t1
        try:
            with open("/proc/cpuinfo") as fh:
                cpu_model = [l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")][0]
        except (OSError, IndexError):
            cpu_model = "unknown"
        cpu_baseline = dict(value=round(n_cpu * A * args.cpu_steps / dt, 1), unit="agent-steps/s", cores=cores, kind="port",
                            sample="%d envs x %d steps of the same workload, oracle/md_oracle.c (gcc -O3 -march=native) "
                                   "ref_step_mt on %d threads" % (n_cpu, args.cpu_steps, cores),
                            single_thread_value=round(n_cpu1 * A * args.cpu_steps / dt1, 1),
                            single_thread_sample="%d envs x %d steps, 1 thread" % (n_cpu1, args.cpu_steps),
                            cpu_model=cpu_model, nproc=os.cpu_count())

    if rank == 0:
        line = OrderedDict(
            metric="agent-steps/sec at 4096 envs x 240-beam lidar" if args.workload == "metadrive" else
            "agent-steps/sec (%s workload)" % args.workload, value=round(value, 1), unit="agent-steps/s",
            n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 4),
            higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
            config=dict(workload=label,
                        envs_per_gpu=E, active_agent_fraction=round(active_frac, 3), agents_per_env=A, mover_capacity=cap, n_beams=B, sharding="env-range per rank"),
            preroll=args.preroll, timed_steps=timed_steps,
            roofline=roofline, lidar=lidar, cpu_baseline=cpu_baseline, env_api=env_api, lane_follow_policy=lane_follow,
            double_buffered=double_buffered, shared_maps=shared_maps, with_gather=with_gather, host_build_s=round(build_s, 1))
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Phase stamps of scenario_step_kernel (diagnostic -DMD_STAMP build) on the scenario bench workload: shader cycles per phase,
workgroup life in microseconds, by number of reactive vehicles.  Read SHARES from it, never the run time of this build.

    ENVS=2048 python tools/scenario_stamps.py
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ["stage-in", "decide (TrajectoryIDMPolicy, wave per vehicle)", "integrate", "traffic manager after_step (wave 0; contacts on wave 1, detectors on waves 2-3 beside it)",
          "agent: trajectory projection, then the barrier (contacts / detectors done)", "agent: observation / reward / done (one lane)", "lidar", "write-back"]


def main():
    import numpy as np
    out = os.path.join(ROOT, "gpurun_out", "libmdstep_stamp.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared",
                           "-fvisibility=hidden", "-std=c++17", "-DMD_STAMP"] + os.environ.get("MD_EXTRA_FLAGS", "").split() +
                          ["-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "metadrive_ped_amd", "csrc", "mdstep.hip"), "-o", out])
    from metadrive_ped_amd import _lib
    _lib.LIB_PATH = out
    from metadrive_ped_amd.envs.scenario_env import scenario_bench_config
    from metadrive_ped_amd.scenario import ScenarioHostScene, synthetic_scenarios
    from metadrive_ped_amd.engine import BatchedEngine
    E = int(os.environ.get("ENVS", "2048"))
    cfg = scenario_bench_config(dict(num_envs=E, num_scenarios=E, env_seed_offset=0, start_seed=0, mover_capacity=0,
                                     auto_reset=True, device="cuda:0"))
    host = ScenarioHostScene(cfg, synthetic_scenarios(E, 0))
    import torch
    eng = BatchedEngine(cfg, host=host)
    eng.reset()
    g = torch.Generator().manual_seed(0)
    acts = (torch.rand(16, E, 1, 2, generator=g) * 2 - 1)
    acts[..., 1] = acts[..., 1].abs() * 0.9 + 0.1
    acts[..., 0] *= 0.25
    acts = acts.cuda()
    for i in range(int(os.environ.get("PREROLL", "60"))):
        eng.step(acts[i % 16])
    buf = torch.zeros(E * 32, dtype=torch.int64, device="cuda")
    eng.lib.md_debug_set_stamp_buffer.argtypes = [C.c_void_p]
    assert eng.lib.md_debug_set_stamp_buffer(buf.data_ptr()) == 0
    torch.cuda.synchronize()
    eng.step(acts[0])
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(E, 32)
    st = raw[:, :12].astype(np.int64)
    for i in range(1, 12):
        st[:, i] = np.where(st[:, i] == 0, st[:, i - 1], st[:, i])
    cols = [0, 1, 2, 3, 4, 5, 6, 7, 11]
    d = np.diff(st[:, cols], axis=1)
    tot = st[:, 11] - st[:, 0]
    print("per-scene cycles: mean %.0f  p50 %.0f  p99 %.0f  max %.0f" % (tot.mean(), np.median(tot), np.percentile(tot, 99), tot.max()))
    for i, name in enumerate(PHASES):
        x = d[:, i]
        print("%-52s mean %8.0f  p50 %8.0f  p99 %8.0f   share %5.1f%%" % (name, x.mean(), np.median(x), np.percentile(x, 99), 100.0 * x.sum() / tot.sum()))
    fine = raw[:, 16:].astype(np.int64)
    for off, name in ((0, "ordinary step"), (8, "speed-control step (every fifth)")):
        f = fine[:, off:off + 6]
        ok = (f[:, 0] > 0) & (f[:, 5] > f[:, 0])
        if ok.any():
            dd = np.diff(f[ok], axis=1)
            print("first reactive vehicle, %-34s (%4d scenes): arrival check %.0f | own projection %.0f | front search %.0f | heading "
                  "ahead %.0f | decide (lane 0) %.0f   cycles, p50" % ((name, ok.sum()) + tuple(np.median(dd, axis=0))))
            seg = eng.host.world.arrays["poly_off"]
    po = np.asarray(eng.host.world.arrays["poly_off"])
    npiece = np.diff(po)
    print("polyline pieces per slot: mean %.0f  p50 %.0f  max %d; outline vertices per slot: mean %.0f" % (
        npiece[npiece > 0].mean(), np.median(npiece[npiece > 0]), npiece.max(), np.diff(np.asarray(eng.host.world.arrays["polyv_off"])).mean()))
    life = (raw[:, 13] - raw[:, 12]) / 100.0
    t0 = (raw[:, 12] - raw[:, 12].min()) / 100.0
    print("workgroup life: mean %.1f  p50 %.1f  p99 %.1f  max %.1f us; launch span %.1f us; started after 5 us: %d" % (
        life.mean(), np.median(life), np.percentile(life, 99), life.max(), ((raw[:, 13] - raw[:, 12].min()) / 100.0).max(), (t0 > 5).sum()))
    nav = eng.nav_i[:, :, 1].cpu().numpy()                       # MdNav.ck0 = MD_SC_* state of the slot
    flags = eng.shape_f.view(torch.int32)[..., 6].cpu().numpy()
    idm = ((nav == 2) & ((flags & 0x10) != 0)).sum(axis=1)
    for k_ in range(int(idm.max()) + 1):
        m = idm == k_
        if m.sum() >= 8:
            print("reactive vehicles %2d: %5d scenes, mean life %.1f us, decide %.0f cycles" % (k_, m.sum(), life[m].mean(), d[m, 1].mean()))
    # the scenes that end the launch, and the phases that make them slow (against the batch's mean)
    order = np.argsort(-life)[:8]
    mean_d = d.mean(axis=0)
    print("slowest scenes (phase cycles; mean of the batch: %s)" % " ".join("%d" % x for x in mean_d))
    for e_ in order:
        print("  scene %4d life %.1f us, %d reactive: %s" % (e_, life[e_], idm[e_], " ".join("%d" % x for x in d[e_])))
    top = life >= np.percentile(life, 97)
    print("slowest 3 %% of the scenes minus the mean, per phase: %s" % " ".join("%+d" % x for x in (d[top].mean(axis=0) - mean_d)))


if __name__ == "__main__":
    main()

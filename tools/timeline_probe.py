"""Occupancy timeline of ONE md_step launch on the bench workload (diagnostic build, -DMD_STAMP): every workgroup stamps
s_memrealtime (100 MHz, chip-wide time base) at its start and end plus the CU it ran on.  Prints how many workgroups were
resident over time, how long a workgroup lives, how busy the CU slots were, and the gap between a workgroup's end and the next
start on the same CU.  Read SHAPES from it (ramp, plateau, tail), never the run time of this build.

    ENVS=4096 MD_STEP_KERNEL=wg python tools/timeline_probe.py
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    out = os.path.join(ROOT, "gpurun_out", "libmdstep_stamp.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared",
                           "-fvisibility=hidden", "-std=c++17", "-DMD_STAMP"] + os.environ.get("MD_EXTRA_FLAGS", "").split() +
                          ["-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "metadrive_ped_amd", "csrc", "mdstep.hip"), "-o", out])
    from metadrive_ped_amd import _lib
    _lib.LIB_PATH = out
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine, HostScene
    E = int(os.environ.get("ENVS", "4096"))
    cfg = make_config(dict(num_envs=E, num_scenarios=min(E, int(os.environ.get("SCEN", str(E)))), mover_capacity=0, horizon=1000,
                           step_kernel=os.environ.get("MD_STEP_KERNEL", "wg")))
    host = HostScene(cfg)              # before the GPU is initialised: the map builders fork
    import torch
    eng = BatchedEngine(cfg, host=host)
    eng.reset()
    g = torch.Generator().manual_seed(0)
    acts = torch.rand(16, E, 1, 2, generator=g) * 2 - 1
    acts[..., 1] = acts[..., 1].abs() * 0.9 + 0.1     # as bench.py: mostly forward, so that envs meet traffic and curves
    acts[..., 0] *= 0.25
    acts = acts.cuda()
    for i in range(int(os.environ.get("PREROLL", "300"))):
        eng.step(acts[i % 16])
    def driving():
        flags = eng.shape_f.view(torch.int32)[..., 6]
        return (((flags & 0x10) != 0) & ((flags & 0x40) == 0) & ((flags & 0xF) == 1)).sum(dim=1)

    order = None
    if os.environ.get("ORDER", "") == "heavy":       # heaviest envs first (by the number of driving vehicles before the launch)
        order = torch.argsort(driving(), descending=True, stable=True).to(torch.int32).contiguous()
        eng.lib.md_debug_set_env_order.argtypes = [C.c_void_p]
        assert eng.lib.md_debug_set_env_order(order.data_ptr()) == 0
    buf = torch.zeros(E * 32, dtype=torch.int64, device="cuda")
    eng.lib.md_debug_set_stamp_buffer.argtypes = [C.c_void_p]
    assert eng.lib.md_debug_set_stamp_buffer(buf.data_ptr()) == 0
    reps = []
    for rep in range(3):
        buf.zero_()
        torch.cuda.synchronize()
        eng.step(acts[rep])
        torch.cuda.synchronize()
        reps.append(buf.cpu().numpy().reshape(E, 32).copy())
    raw = reps[-1]
    t0 = raw[:, 12].astype(np.int64)
    t1 = raw[:, 13].astype(np.int64)
    ok = (t0 > 0) & (t1 >= t0)
    print("workgroups with both stamps: %d of %d" % (ok.sum(), E))
    base = t0[ok].min()
    a = (t0 - base) / 100.0            # us
    b = (t1 - base) / 100.0
    life = b - a
    print("launch span (first start -> last end) %.1f us; workgroup life: mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us" % (
        b.max(), life.mean(), *np.percentile(life, [10, 50, 90, 99]), life.max()))
    hw = raw[:, 14].astype(np.uint64)
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ncu = len(np.unique(cuid))
    per_cu = np.bincount(np.unique(cuid, return_inverse=True)[1])
    print("distinct CUs seen: %d; workgroups per CU: min %d  mean %.1f  max %d; per XCC: %s" % (
        ncu, per_cu.min(), per_cu.mean(), per_cu.max(), np.bincount(xcc).tolist()))
    span = b.max()
    print("mean resident workgroups over the launch: %.0f (= %.2f per CU);  sum of lives / (CUs x span) " % (
        life.sum() / span, life.sum() / span / ncu))
    step = float(os.environ.get("BIN_US", "4"))
    print("time [us]   resident WGs   started   finished   (start order: mean blockIdx of the WGs started in the bin)")
    t = 0.0
    while t < span:
        res = ((a <= t) & (b > t)).sum()
        st = (a >= t) & (a < t + step)
        fin = ((b >= t) & (b < t + step)).sum()
        print("%8.1f   %8d   %8d   %8d   %s" % (t, res, st.sum(), fin, ("%.0f" % np.nonzero(st)[0].mean()) if st.any() else "-"))
        t += step
    # the gap between a workgroup's end and the next start on the same CU: dispatcher latency
    gaps = []
    peak = []
    for c in np.unique(cuid):
        m = cuid == c
        ends = np.sort(b[m])
        starts = np.sort(a[m])
        ev = sorted([(x, 1) for x in a[m]] + [(x, -1) for x in b[m]])
        lvl = 0
        mx = 0
        for _, d in ev:
            lvl += d
            mx = max(mx, lvl)
        peak.append(mx)
        late = starts[starts > 0.5]
        for s_ in late:
            j = np.searchsorted(ends, s_, side="right") - 1
            if j >= 0:
                gaps.append(s_ - ends[j])
    gaps = np.array(gaps)
    print("peak resident workgroups on one CU: min %d  median %d  max %d" % (min(peak), int(np.median(peak)), max(peak)))
    if len(gaps):
        print("refill gap (a later start minus the latest end before it on that CU): p10 %.2f  p50 %.2f  p90 %.2f us" % tuple(
            np.percentile(gaps, [10, 50, 90])))
    # life against start time: do later workgroups (emptier chip) run faster?
    for lo, hi in ((0, 5), (5, 20), (20, 40), (40, 60), (60, 1e9)):
        m = (a >= lo) & (a < hi)
        if m.any():
            print("workgroups started in [%g, %g) us: %5d, mean life %.1f us" % (lo, hi, m.sum(), life[m].mean()))
    drv = driving().cpu().numpy()
    print("driving vehicles per env: mean %.2f" % drv.mean())
    if order is not None:
        drv = drv[order.cpu().numpy()]              # stamps are per workgroup
    for k in range(0, int(drv.max()) + 1):
        m = drv == k
        if m.sum() >= 8:
            print("driving vehicles %d: %5d envs, mean life %.1f us" % (k, m.sum(), life[m].mean()))
    spans = [((r[:, 13].astype(np.int64)).max() - (r[:, 12][r[:, 12] > 0].astype(np.int64)).min()) / 100.0 for r in reps]
    print("span of the three probed launches: %s us" % ", ".join("%.1f" % x for x in spans))


if __name__ == "__main__":
    main()

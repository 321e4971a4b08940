"""In-kernel phase stamps of the fused step kernel (diagnostic build, -DMD_STAMP).  Builds a separate
library, runs the bench workload, prints per-phase cycle shares (mean / p50 / max over envs).  Read
SHARES from it, never the run time of this build."""
import ctypes as C
import os
import subprocess
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# stamps 2 and 5 exist only in the kernel variants that run the IDM before the integration / localisation apart from
# the contacts (multi-agent order); where a stamp was not written its phase reads 0 and the next one gets the interval
PHASES = ["stage-in", "trigger (idm-first variants)", "idm before integrate", "integrate", "localize (own stage)",
          "locate: localize + contacts", "traffic + next trigger", "observe || idm for the next step", "lidar",
          "wait for the idm waves", "write-back"]


WAVE_PHASES = ["stage-in", "trigger + idm (idm-first variants)", "-", "integrate", "-", "locate: localize + contacts",
               "traffic + next trigger", "idm for the next step", "observe", "lidar", "write-back"]


def main():
    import numpy as np
    global PHASES
    if os.environ.get("MD_STEP_KERNEL", "wg") != "wg":
        PHASES = WAVE_PHASES     # wave_step_kernel (build with MD_EXTRA_FLAGS=-DMD_WAVE_ENVS=1: one env per workgroup)
    out = os.path.join(ROOT, "gpurun_out", "libmdstep_stamp.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared",
                           "-fvisibility=hidden", "-std=c++17", "-DMD_STAMP"] + os.environ.get("MD_EXTRA_FLAGS", "").split() + [ "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "metadrive_ped_amd", "csrc", "mdstep.hip"), "-o", out])
    from metadrive_ped_amd import _lib
    _lib.LIB_PATH = out
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine, HostScene
    from metadrive_ped_amd.mapgen.pg import BLOCK_TYPE_DISTRIBUTION_V2
    d = OrderedDict((k, 0.0) for k in BLOCK_TYPE_DISTRIBUTION_V2)
    d["Curve"], d["Straight"] = 0.6, 0.4
    E = int(os.environ.get("ENVS", "4096"))
    cfg = make_config(dict(num_envs=E, num_scenarios=min(E, int(os.environ.get("SCEN", "512"))), mover_capacity=0, horizon=1000))
    workload = os.environ.get("WORKLOAD", "metadrive")
    if workload == "marl":        # BASELINE configs[2]: the 40-agent roundabout (ENVS=1024)
        from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentRoundaboutEnv
        cfg = BatchedMultiAgentRoundaboutEnv(dict(num_envs=E, num_scenarios=E, vehicle_config=dict(lidar=dict(num_lasers=240, distance=50)))).config
        PHASES = ["stage-in", "lifecycle", "idm before integrate", "integrate", "localize (own stage)", "contacts (own stage)",
                  "traffic", "observe", "lidar", "-", "write-back"]
    elif workload == "safe":      # BASELINE configs[3]
        cfg = make_config(dict(num_envs=E, num_scenarios=E, mover_capacity=0, horizon=1000, accident_prob=0.8, traffic_density=0.1,
                               crash_vehicle_done=False, crash_object_done=False, out_of_road_done=True))
    eng = BatchedEngine(cfg, host=HostScene(cfg))
    eng.reset()
    A_ = eng.A
    g = torch.Generator().manual_seed(0)
    acts = torch.rand(16, E, A_, 2, generator=g) * 2 - 1
    acts[..., 1] = acts[..., 1].abs() * 0.9 + 0.1
    acts[..., 0] *= 0.25
    acts = acts.cuda()
    lane_follow = os.environ.get("POLICY", "random") == "lane"   # scripted driver: longer episodes, more traffic awake

    def next_action(i):
        if not lane_follow:
            return acts[i % 16]
        ob_ = eng.obs[:, 0, :]
        a = torch.zeros(E, A_, 2, device="cuda")
        a[:, 0, 0] = (4.0 * (ob_[:, 2] - 0.5) + 2.0 * (ob_[:, 8] - 0.5)).clamp_(-1.0, 1.0)
        a[:, 0, 1] = (ob_[:, 3] < 0.35).to(torch.float32) * 0.5
        return a

    for i in range(400 if lane_follow else 80):
        eng.step(next_action(i))
    buf = torch.zeros(E * 32, dtype=torch.int64, device="cuda")
    eng.lib.md_debug_set_stamp_buffer.argtypes = [C.c_void_p]
    assert eng.lib.md_debug_set_stamp_buffer(buf.data_ptr()) == 0
    eng.step(next_action(0))
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(E, 32)
    st = raw[:, :12].astype(np.int64)
    for i in range(1, 11):                          # a stamp that was not written takes the one before it
        st[:, i] = np.where(st[:, i] == 0, st[:, i - 1], st[:, i])
    fine = raw[:, 16:].astype(np.int64)
    d = np.diff(st, axis=1)
    tot = st[:, 11] - st[:, 0]
    print("per-env cycles: mean %.0f  p50 %.0f  p99 %.0f  max %.0f" % (tot.mean(), np.median(tot), np.percentile(tot, 99), tot.max()))
    print("kernel span (first start -> last end): %.0f cycles" % (st[:, 11].max() - st[:, 0].min()))
    for i, name in enumerate(PHASES):
        x = d[:, i]
        print("%-34s mean %8.0f  p50 %8.0f  p99 %8.0f  max %8.0f   share %5.1f%%" %
              (name, x.mean(), np.median(x), np.percentile(x, 99), x.max(), 100.0 * x.sum() / tot.sum()))
    ok = (fine[:, 0] > 0) & (fine[:, 3] > 0)
    lf = fine[ok]
    if len(lf):
      print("localize(agent) fine: grid+cell loads %.0f | items+AABB %.0f | hull tests+frenet %.0f  (p50 cycles); candidates p50 %d, cell items p50 %d" % (
        np.median(lf[:, 1] - lf[:, 0]), np.median(lf[:, 2] - lf[:, 1]), np.median(lf[:, 3] - lf[:, 2]),
        np.median(lf[:, 15] & 0xffffffff), np.median(lf[:, 15] >> 32)))
    ok = (fine[:, 4] > 0) & (fine[:, 7] > fine[:, 4])
    li = fine[ok]
    if len(li):
        print("idm(first traffic slot) fine over %d envs: plan %.0f | scan %.0f | decide %.0f (p50 cycles)" % (
            len(li), np.median(li[:, 5] - li[:, 4]), np.median(li[:, 6] - li[:, 5]), np.median(li[:, 7] - li[:, 6])))
    ok = (fine[:, 8] > 0) & (fine[:, 11] > fine[:, 8])
    lo = fine[ok]
    if len(lo):
        print("observe(agent) fine: context %.0f | nine tasks %.0f | combine + stores %.0f (p50 cycles)" % (
            np.median(lo[:, 9] - lo[:, 8]), np.median(lo[:, 10] - lo[:, 9]), np.median(lo[:, 11] - lo[:, 10])))
    flags = eng.shape_f.view(torch.int32)[..., 6]
    drv = (((flags & 0x10) != 0) & ((flags & 0x40) == 0) & ((flags & 0xF) == 1)).sum(dim=1).cpu().numpy()
    print("driving vehicles/env: mean %.2f max %d" % (drv.mean(), drv.max()))
    life = (raw[:, 13].astype(np.int64) - raw[:, 12].astype(np.int64)) / 100.0
    t_start = (raw[:, 12].astype(np.int64) - raw[:, 12].astype(np.int64).min()) / 100.0
    print("workgroup life: mean %.1f  p50 %.1f  p99 %.1f  max %.1f us; launch span %.1f us; workgroups started after 5 us: %d of %d" % (
        life.mean(), np.median(life), np.percentile(life, 99), life.max(),
        (raw[:, 13].astype(np.int64).max() - raw[:, 12].astype(np.int64).min()) / 100.0, (t_start > 5).sum(), E))
    for k_ in range(1, int(drv.max()) + 1):
        m_ = drv == k_
        if m_.sum() >= 5:
            print("driving %2d: %5d envs  life %.1f us | locate %6.0f | observe||idm %6.0f | wait idm %6.0f | total %6.0f cycles (means)" % (
                k_, m_.sum(), life[m_].mean(), d[m_, 5].mean(), d[m_, 7].mean(), d[m_, 9].mean(), tot[m_].mean()))
    idx = np.argsort(tot)[-3:]
    for i in idx:
        print("slow env", i, "drv", drv[i], "phases", d[i].tolist())


if __name__ == "__main__":
    main()

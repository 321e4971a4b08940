"""Per-phase launch times of the step path (each md_* entry point on its own), HIP events.
Usage: python tools/phase_profile.py [--envs 4096] [--cap 32] [--warm 60]"""
import argparse
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--cap", type=int, default=0)
    ap.add_argument("--warm", type=int, default=80)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine, HostScene
    from metadrive_ped_amd.mapgen.pg import BLOCK_TYPE_DISTRIBUTION_V2
    d = OrderedDict((k, 0.0) for k in BLOCK_TYPE_DISTRIBUTION_V2)
    d["Curve"], d["Straight"] = 0.6, 0.4
    E = args.envs
    cfg = make_config(dict(num_envs=E, num_scenarios=min(E, 512), mover_capacity=args.cap,
                           auto_reset=True, horizon=1000))
    host = HostScene(cfg)
    eng = BatchedEngine(cfg, host=host)
    eng.reset()
    g = torch.Generator().manual_seed(0)
    acts = torch.rand(16, E, 1, 2, generator=g) * 2 - 1
    acts[..., 1] = acts[..., 1].abs() * 0.9 + 0.1
    acts[..., 0] *= 0.25
    acts = acts.cuda()
    for i in range(args.warm):
        eng.step(acts[i % 16])
    torch.cuda.synchronize()
    flags = eng.shape_f.view(torch.int32)[..., 6]
    alive = ((flags & 0x10) != 0).sum().item() / E
    pending = ((flags & 0x40) != 0).sum().item() / E
    print("movers/env %.2f  pending/env %.2f" % (alive, pending))
    names = ["md_idm", "md_integrate", "md_localize", "md_contacts", "md_traffic_after_step", "md_observe"]

    def time_fn(fn):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        return ts[len(ts) // 2] * 1e3

    for n in names:
        print("%-24s %9.1f us" % (n, time_fn(lambda: eng.call(n))))
    out = torch.empty(E, eng.n_beams, device="cuda")
    print("%-24s %9.1f us" % ("md_lidar", time_fn(lambda: eng.lidar(out, eng.n_beams, 0))))
    print("%-24s %9.1f us" % ("md_step (fused)", time_fn(eng.step_raw)))


if __name__ == "__main__":
    main()

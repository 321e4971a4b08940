#!/usr/bin/env python
"""Per-kernel times of the phase-per-launch step (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine, HostScene
    E = 4096
    cfg = make_config(dict(num_envs=E, num_scenarios=E, horizon=1000, mover_capacity=24, step_kernel=os.environ.get("KERNEL", "pm"),
                           build_workers=int(os.environ.get("WORKERS", "0"))))
    host = HostScene(cfg)
    import torch
    g = torch.Generator().manual_seed(0)
    acts = torch.rand(64, E, 1, 2, generator=g) * 2 - 1
    acts[..., 1] = acts[..., 1].abs() * 0.9 + 0.1
    acts[..., 0] *= 0.25
    acts = acts.cuda()
    eng = BatchedEngine(cfg, host=host)
    eng.reset()
    for i in range(int(os.environ.get("STEPS", "400"))):
        eng.step(acts[i % 64])
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()

"""Sub-batches in flight on separate HIP streams (double-buffered stepping): S engines of E/S envs each, every one
on its own torch stream; one 'step' = one md_step launch per engine.  The tail of one sub-batch's launch overlaps
the body of the other's.  Usage: python tools/pipeline_probe.py [--envs 4096] [--streams 1 2 4] [--steps 300]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=60)
    args = ap.parse_args()
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    E = args.envs
    for S in args.streams:
        n = E // S
        engines, streams, acts = [], [], []
        for k in range(S):
            cfg = make_config(dict(num_envs=n, num_scenarios=E, env_seed_offset=k * n, horizon=1000))
            eng = BatchedEngine(cfg)
            eng.reset()
            engines.append(eng)
            streams.append(torch.cuda.Stream())
            g = torch.Generator().manual_seed(k)
            a = torch.rand(64, n, 1, 2, generator=g) * 2 - 1
            acts.append(a.cuda())
        torch.cuda.synchronize()

        def run(steps, base):
            for t in range(steps):
                for k in range(S):
                    with torch.cuda.stream(streams[k]):
                        engines[k].step(acts[k][(base + t) % 64])
        run(args.warmup, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps, args.warmup)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("streams %d x %d envs: %.1f us per %d agent-steps, %.2f M agent-steps/s" % (S, n, dt / args.steps * 1e6, E, E * args.steps / dt / 1e6), flush=True)
        del engines


if __name__ == "__main__":
    main()

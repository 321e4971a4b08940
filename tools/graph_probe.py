"""HIP graph replay of a closed rollout loop (policy ops + md_step) for small, launch-bound batches: K steps of
`policy(obs) -> env step` are captured once with torch.cuda.graph (md_step is launched on the capturing stream like any
other kernel; every buffer it touches is persistent) and replayed.  Prints eager vs graph time per step and checks
that both give the same state.  Usage: python tools/graph_probe.py [--envs 256] [--k 16]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[64, 256, 1024, 4096])
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--reps", type=int, default=40)
    args = ap.parse_args()
    import torch
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    for E in args.envs:
        def make():
            eng = BatchedEngine(make_config(dict(num_envs=E, num_scenarios=min(E, 512), horizon=1000)))
            eng.reset()
            return eng, torch.zeros(E, 1, 2, device="cuda")

        def drive(eng, act):
            ob_ = eng.obs[:, 0, :]
            act[:, 0, 0] = (4.0 * (ob_[:, 2] - 0.5) + 2.0 * (ob_[:, 8] - 0.5)).clamp_(-1.0, 1.0)
            act[:, 0, 1] = (ob_[:, 3] < 0.35).to(torch.float32) * 0.5
            eng.step(act)

        e1, a1 = make()
        e2, a2 = make()
        for _ in range(3 * args.k):                 # warm both the same way (eager)
            drive(e1, a1)
            drive(e2, a2)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            drive(e2, a2)                            # one eager step on the side stream before capturing
            drive(e1, a1)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(args.k):
                drive(e2, a2)
        torch.cuda.synchronize()
        # the capture itself executes nothing: e1 and e2 are level here
        t0 = time.perf_counter()
        for _ in range(args.reps):
            for _ in range(args.k):
                drive(e1, a1)
        torch.cuda.synchronize()
        t_eager = (time.perf_counter() - t0) / (args.reps * args.k)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            g.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / (args.reps * args.k)
        same = torch.equal(e1.obs, e2.obs) and torch.equal(e1.state_dev["shape"], e2.state_dev["shape"])
        print("envs %5d: eager %.1f us/step, graph of %d steps %.1f us/step, identical state: %s" %
              (E, t_eager * 1e6, args.k, t_graph * 1e6, same), flush=True)


if __name__ == "__main__":
    main()

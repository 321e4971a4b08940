"""Per-phase launch times of the multi-agent step (1024 roundabout envs x 40 agents, 240 beams), HIP events.
Each md_* entry point alone (they share the stage-in / write-back cost of ~10 us)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from metadrive_ped_amd.engine import BatchedEngine
    from metadrive_ped_amd.envs.marl_env import BatchedMultiAgentRoundaboutEnv
    E = int(os.environ.get("ENVS", "1024"))
    cfg = BatchedMultiAgentRoundaboutEnv(dict(num_envs=E, num_scenarios=E,
                                              vehicle_config=dict(lidar=dict(num_lasers=240, distance=50)))).config
    eng = BatchedEngine(cfg)
    eng.reset()
    g = torch.Generator().manual_seed(0)
    acts = torch.rand(16, E, 40, 2, generator=g) * 2 - 1
    acts[..., 1] = acts[..., 1].abs() * 0.9 + 0.1
    acts[..., 0] *= 0.25
    acts = acts.cuda()
    for i in range(100):
        eng.step(acts[i % 16])
    torch.cuda.synchronize()

    def time_fn(fn, reps=20):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        return ts[len(ts) // 2] * 1e3

    for n in ["md_traffic_after_step", "md_lifecycle", "md_idm", "md_integrate", "md_localize", "md_contacts", "md_observe"]:
        print("%-24s %9.1f us" % (n, time_fn(lambda: eng.call(n))))
    out = torch.empty(E * 40, 240, device="cuda")
    print("%-24s %9.1f us" % ("md_lidar", time_fn(lambda: eng.lidar(out, 240, 0))))
    print("%-24s %9.1f us" % ("md_step (fused)", time_fn(eng.step_raw)))


if __name__ == "__main__":
    main()

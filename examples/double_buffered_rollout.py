#!/usr/bin/env python
"""Two sub-batches in flight on two HIP streams: the policy works on one while the engine steps the other."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from metadrive_ped_amd.envs import BatchedMetaDriveEnv
    from metadrive_ped_amd.envs.pipeline import SubBatchedEnvs
    E, S, steps = 4096, 2, 300
    envs = SubBatchedEnvs(BatchedMetaDriveEnv, dict(num_envs=E, num_scenarios=E, map=3, traffic_density=0.1, horizon=1000),
                          sub_batches=S)
    envs.build_host()                                   # maps and scenes of both sub-batches (host, fork pool)
    obs = [o for o, _ in envs.reset()]
    envs.synchronize()

    def policy(o):
        return torch.stack([(4.0 * (o[:, 2] - 0.5) + 2.0 * (o[:, 8] - 0.5)).clamp(-1, 1), (o[:, 3] < 0.35).float() * 0.5], 1)

    for t in range(150):                                # warm-up: lazy code loading, clocks
        for k, env in enumerate(envs.envs):
            with envs.on(k):
                obs[k], *_ = env.step(policy(obs[k]))
    envs.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        for k, env in enumerate(envs.envs):
            with envs.on(k):                            # policy ops and the step launch go to sub-batch k's stream
                obs[k], reward, terminated, truncated, info = env.step(policy(obs[k]))
    envs.synchronize()
    dt = time.perf_counter() - t0
    print("%d x %d envs, %d steps: %.1f M agent-steps/s" % (S, E // S, steps, E * steps / dt / 1e6))
    envs.close()


if __name__ == "__main__":
    main()

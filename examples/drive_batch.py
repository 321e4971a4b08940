#!/usr/bin/env python
"""The reference's drive_in_single_agent_env loop over a batch of environments."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--idm", action="store_true", help="agent_policy=IDMPolicy: the agents drive themselves")
    args = ap.parse_args()
    import torch
    from metadrive_ped_amd.envs import BatchedMetaDriveEnv
    E = args.envs
    env = BatchedMetaDriveEnv(dict(num_envs=E, num_scenarios=E, map=3, traffic_density=0.1, horizon=1000,
                                   agent_policy="IDMPolicy" if args.idm else "EnvInputPolicy"))
    obs, info = env.reset()
    ret = torch.zeros(E, device="cuda")
    done_episodes, arrived = torch.zeros((), dtype=torch.int64, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")
    act = torch.zeros(E, 2, device="cuda")
    weight = torch.zeros(obs.shape[1], 1, device="cuda")
    weight[2], weight[8] = 4.0, 2.0                      # steer = 4 (heading_diff - .5) + 2 (lateral - .5)
    bias = torch.full((1, ), -3.0, device="cuda")
    for t in range(150):                                 # the first ~100 calls pay for lazy code loading / clocks: not timed
        obs, *_ = env.step(torch.zeros(E, 2, device="cuda"))
    torch.cuda.synchronize()
    t0 = t_chunk = time.perf_counter()
    for t in range(args.steps):
        # a scripted lane-keeping driver written on the observation: heading difference (dim 2), speed (dim 3), lateral (dim 8)
        # (few device ops on purpose: in an eager loop every small op costs about a tenth of the whole step on this GPU)
        act[:, 0] = torch.addmm(bias, obs, weight).clamp_(-1, 1).squeeze(1)
        act[:, 1] = (obs[:, 3] < 0.35) * 0.5
        obs, reward, terminated, truncated, info = env.step(act)
        ret += reward
        done = terminated | truncated                    # finished envs restart by themselves at the next step
        done_episodes += done.sum()                      # counters stay on the device: no host sync inside the loop
        arrived += info["arrive_dest"].sum()
        if t % 100 == 99:
            torch.cuda.synchronize()
            print("  steps %4d-%4d: %.0f us per step" % (t - 99, t, (time.perf_counter() - t_chunk) / 100 * 1e6), flush=True)
            t_chunk = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d envs x %d steps in %.2f s = %.1f M agent-steps/s; %d episodes ended, %d at the destination; mean reward per step %.3f"
          % (E, args.steps, dt, E * args.steps / dt / 1e6, int(done_episodes), int(arrived), float(ret.mean()) / args.steps))
    env.close()


if __name__ == "__main__":
    main()

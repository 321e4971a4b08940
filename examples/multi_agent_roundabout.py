#!/usr/bin/env python
"""MultiAgentRoundaboutEnv, batched: 40 agent slots per env, respawn, the reference's dict view of one env."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from metadrive_ped_amd.envs import BatchedMultiAgentRoundaboutEnv
    E = 64
    env = BatchedMultiAgentRoundaboutEnv(dict(num_envs=E, num_scenarios=E))
    obs, info = env.reset()
    A = env.num_agents
    o_hd = 2
    for t in range(300):
        act = torch.stack([(4.0 * (obs[..., o_hd] - 0.5) + 2.0 * (obs[..., o_hd + 6] - 0.5)).clamp(-1, 1),
                           (obs[..., o_hd + 1] < 0.3).float() * 0.5], -1)
        obs, reward, terminated, truncated, info = env.step(act)
    o, r, tm, tc = env.to_dicts(0, obs, reward, terminated, truncated, info)
    print("env 0: %d active agents (%s ...), %d agents created so far, all done: %s" %
          (len(o), ", ".join(sorted(o)[:4]), int(info["agent_id"][0].max()) + 1, tm["__all__"]))
    print("active agents per env: mean %.1f of %d slots" % (float(info["active"].float().sum(1).mean()), A))
    env.close()


if __name__ == "__main__":
    main()

"""MultiAgentRacingEnv driven by the agents' own IDMPolicy (what the reference's tests/test_env/test_ma_racing.py does), batched:
64 tracks x 12 vehicles, no actions needed.  Prints how far the field got and who went idle.

    python examples/marl_racing_idm.py            (needs an MI355X; builds the library on first use)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from metadrive_ped_amd import hostpool
    hostpool.start()                                     # host build workers: before the first GPU call
    import torch
    from metadrive_ped_amd import abi
    from metadrive_ped_amd.envs import BatchedMultiAgentRacingEnv
    E = 64
    env = BatchedMultiAgentRacingEnv(dict(num_envs=E, num_scenarios=E, agent_policy="IDMPolicy", map_config=dict(exit_length=60)))
    obs, info = env.reset()
    print("obs", tuple(obs.shape), "agents per env", env.num_agents)
    total = torch.zeros(E, env.num_agents, device=obs.device)
    for t in range(600):
        obs, reward, terminated, truncated, info = env.step(None)       # IDMPolicy ignores the actions
        total += reward
        if t % 100 == 99:
            fl = env.engine.flags[:, :env.num_agents]
            print("step %3d: mean return %.1f, crashed %d, idle %d, arrived %d" % (
                t + 1, float(total.mean()), int(((fl & abi.FL_CRASH_VEHICLE) != 0).sum()), int(((fl & abi.FL_IDLE) != 0).sum()),
                int(((fl & abi.FL_ARRIVE_DEST) != 0).sum())))
    env.close()


if __name__ == "__main__":
    main()

import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from metadrive_ped_amd.envs import BatchedMultiAgentTollgateEnv, BatchedMultiAgentBottleneckEnv
from metadrive_ped_amd.engine import HostScene
import torch
for cls, E in ((BatchedMultiAgentTollgateEnv, 512), (BatchedMultiAgentBottleneckEnv, 1024)):
    env = cls(dict(num_envs=E, num_scenarios=E))
    env.lazy_init(HostScene(env.config))
    env.reset()
    A = env.num_agents
    a = torch.zeros(E, A, 2, device="cuda"); a[..., 1] = 0.5
    for _ in range(100): env.step(a)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(200): env.step(a)
    torch.cuda.synchronize(); dt = (time.time() - t) / 200
    print(cls.__name__, E, "x", A, "agents: %.1f us per step" % (dt * 1e6), flush=True)

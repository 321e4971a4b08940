import time, torch, sys
sys.path.insert(0, "/root/repo")
from metadrive_ped_amd.config import make_config
from metadrive_ped_amd.engine import BatchedEngine
cfg = make_config(dict(num_envs=64, num_scenarios=64))
eng = BatchedEngine(cfg); eng.reset()
acts = (torch.rand(16, 64, 1, 2) * 2 - 1).cuda()
for i in range(100): eng.step(acts[i % 16])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2000): eng.step(acts[i % 16])
torch.cuda.synchronize()
print("per step (64 envs): %.1f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
t0 = time.perf_counter()
for i in range(2000): eng.step_raw()
torch.cuda.synchronize()
print("per step_raw (64 envs): %.1f us" % ((time.perf_counter() - t0) / 2000 * 1e6))

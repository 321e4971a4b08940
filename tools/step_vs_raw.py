"""Loop time of eng.step(actions) (zero-copy agent actions) against eng.step_raw() (actions already in the slot array)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metadrive_ped_amd.config import make_config
from metadrive_ped_amd.engine import BatchedEngine, HostScene
E = int(os.environ.get("ENVS", "4096"))
cfg = make_config(dict(num_envs=E, num_scenarios=E, horizon=1000))
eng = BatchedEngine(cfg, host=HostScene(cfg))
eng.reset()
g = torch.Generator().manual_seed(0)
acts = torch.rand(64, E, 1, 2, generator=g) * 2 - 1
acts[..., 1] = acts[..., 1].abs() * 0.9 + 0.1
acts[..., 0] *= 0.25
acts = acts.cuda()
for i in range(60):
    eng.step(acts[i % 64])
torch.cuda.synchronize()
for name in ("step", "raw_same_actions", "step", "raw_new_actions"):
    t0 = time.perf_counter()
    for i in range(300):
        if name == "step":
            eng.step(acts[i % 64])
        elif name == "raw_new_actions":
            eng.action[:, :1, :] = acts[i % 64]
            eng.step_raw()
        else:
            eng.step_raw()
    torch.cuda.synchronize()
    print("%-18s %.1f us per step" % (name, (time.perf_counter() - t0) / 300 * 1e6))

"""How much of the fused step is map-data latency, and which step kernel wins where?  Same envs, but sharing 1 / 8 / 64 / 512 /
4096 distinct maps (with few maps the lane / grid tables are L2-resident, with 4096 every workgroup reads cold lines), stepped by
the workgroup-per-env and the wave-per-env kernel: the table behind engine.WAVE_KERNEL_MAX_MAPS / WAVE_KERNEL_MIN_ENVS.

    ENVS=4096 MAPS=1,8,16,64,512,4096 KERNELS=wg,wave python tools/locality_probe.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine, HostScene
    E = int(os.environ.get("ENVS", "4096"))
    sizes = [int(x) for x in os.environ.get("MAPS", "1,8,64,512,4096").split(",")]
    kernels = os.environ.get("KERNELS", "wg,wave").split(",")
    from metadrive_ped_amd import hostpool
    hostpool.start()                                   # host-side generation before the GPU is touched
    cfgs = {(S, k): make_config(dict(num_envs=E, num_scenarios=min(S, E), horizon=1000, mover_capacity=32, step_kernel=k, build_cache=True))
            for S in sizes for k in kernels}
    hosts = {key: HostScene(c) for key, c in cfgs.items()}
    import torch
    g = torch.Generator().manual_seed(0)
    acts = torch.rand(64, E, 1, 2, generator=g) * 2 - 1
    acts[..., 1] = acts[..., 1].abs() * 0.9 + 0.1
    acts[..., 0] *= 0.25
    acts = acts.cuda()
    for S, kern in cfgs:
        eng = BatchedEngine(cfgs[(S, kern)], host=hosts[(S, kern)])
        eng.reset()
        for i in range(60):
            eng.step(acts[i % 64])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(300):
            eng.step(acts[i % 64])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 300
        fl = eng.shape_f.view(torch.int32)[..., 6]
        drv = (((fl & 0x10) != 0) & ((fl & 0x40) == 0) & ((fl & 0xF) == 1) & ((fl & 0x80) == 0)).sum().item() / E
        print("envs %5d distinct maps %5d kernel %-4s: %.1f us per step (%.2f driving vehicles per env)" % (E, S, kern, dt * 1e6, drv), flush=True)
        del eng


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Pedestrian + cyclist, recording, export in the scenario-description format, replay of the exported episode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    from metadrive_ped_amd.envs import BatchedMetaDriveEnv
    E, T = 8, 120
    base = dict(num_envs=E, num_scenarios=E, map="SCS", traffic_density=0.1, mover_capacity=32, horizon=1000, auto_reset=False)
    env = BatchedMetaDriveEnv(dict(base))
    obs, _ = env.reset()
    ego = env.engine.shape_f[:, 0].cpu().numpy()                       # cx, cy, cos, sin, ...
    ahead = np.stack([ego[:, 0] + 30 * ego[:, 2], ego[:, 1] + 30 * ego[:, 3]], 1)
    ped = env.spawn_object("pedestrian", ahead + [0.0, 5.0], -np.pi / 2)    # five metres to the left, walking across
    cyc = env.spawn_object("cyclist", ahead + [20.0, 0.0], 0.0)
    env.set_velocity(ped, [1, 0], 1.2, in_local_frame=True)
    env.set_velocity(cyc, [1, 0], 4.0, in_local_frame=True)
    env.start_recording(T)
    acts = []
    for t in range(T):
        a = torch.stack([(4.0 * (obs[:, 2] - 0.5) + 2.0 * (obs[:, 8] - 0.5)).clamp(-1, 1), (obs[:, 3] < 0.3).float() * 0.5], 1)
        acts.append(a)
        obs, r, tm, tc, info = env.step(a)
    tracks = env.stop_recording()
    scenarios = env.export_scenarios(tracks)
    s0 = scenarios[0]
    print("scenario", s0["id"], "frames", s0["length"], s0["metadata"]["number_summary"]["num_objects_each_type"],
          "crash_human in", int(info["crash_human"].sum()), "envs")
    rp = BatchedMetaDriveEnv(dict(base, traffic_mode="replay"))
    rp.load_scenarios(scenarios)                                        # the exported episode as replay traffic
    o2, _ = rp.reset()
    for t in range(T):
        o2, *_ = rp.step(acts[t])
    print("replayed: max |obs - recorded obs| =", float((o2 - obs).abs().max()))
    env.close()
    rp.close()


if __name__ == "__main__":
    main()

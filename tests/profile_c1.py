#!/usr/bin/env python
"""BASELINE configs[0] (plumbing, no GPU): the shape of the reference's examples/profile_metadrive.py:9-43 on the
CPU oracle -- ONE MetaDriveEnv, map 'S', traffic_density 0, lidar off (19-dim obs), action [0, 1], 10 000
steps, reset on done -- reported as steps/s of one host thread.  The reference's README quotes "+1000 FPS"
for its own engine (README.md:38; published, unverified here, different physics).  TEST INFRASTRUCTURE:
this script drives the oracle, not the product path.

Usage: python tests/profile_c1.py [-n 10000]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # lives under tests/: only test infrastructure may drive the oracle


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-steps", "-n", default=10_000, type=int)
    args = ap.parse_args()
    import oracle_binding as ob
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    cfg = make_config(dict(num_envs=1, num_scenarios=1, start_seed=1010, map="S", traffic_density=0.0,
                           vehicle_config=dict(lidar=dict(num_lasers=0, distance=0)), auto_reset=True))
    host = HostScene(cfg)
    assert host.obs_dim == 19
    o = ob.OracleWorld(host)
    o.reset()
    act = np.array([[[0.0, 1.0]]], np.float32)
    t0 = time.perf_counter()
    for s in range(args.num_steps):
        o.step(act)
    dt = time.perf_counter() - t0
    steps = o.state["nav"]["steps"][0]
    print("C1 plumbing: %d steps of 1 env (map 'S', no traffic, lidar off) in %.3f s = %.0f steps/s on one host thread "
          "(python call overhead included); obs dim %d; last episode length %d" % (args.num_steps, dt, args.num_steps / dt,
                                                                                 host.obs_dim, steps))


if __name__ == "__main__":
    main()

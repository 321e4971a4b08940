"""Long parity soak (not collected by pytest: run by hand on the GPU box): thousands of steps of HIP against the
oracle on a few hundred envs per configuration, comparing every state array at the end of each 250-step chunk.
Usage: python tests/soak_parity.py [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import oracle_binding as ob
    from helpers import assert_state_equal
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    E = 256
    configs = dict(
        default=dict(),
        hybrid=dict(traffic_mode="hybrid", traffic_density=0.2),
        respawn=dict(traffic_mode="respawn", traffic_density=0.15),
        safe=dict(accident_prob=0.8, traffic_density=0.05, crash_vehicle_done=False, crash_object_done=False),
        dense5=dict(map=5, traffic_density=0.3),
        varying=dict(vehicle_config=dict(vehicle_model="varying_dynamics"),
                     random_dynamics=dict(max_engine_force=(100, 3000), max_brake_force=(20, 600), wheel_friction=(0.1, 2.5),
                                          max_steering=(10, 80), mass=(300, 3000))),
        walkers=dict(mover_capacity=40, traffic_density=0.15),   # + a pedestrian and a cyclist spawned every 400 steps
        idm_agent=dict(agent_policy="IDMPolicy", traffic_density=0.15),
        wave_shared=dict(num_scenarios=8, step_kernel="wave"),     # the wave-per-env kernel (what "auto" picks for large batches on few maps)
    )
    for name, extra in configs.items():
        cfg = make_config(dict(dict(num_envs=E, num_scenarios=E, horizon=1000), **extra))
        eng = BatchedEngine(cfg)
        assert eng.host.step_kernel == cfg.get("step_kernel", "auto").replace("auto", "wg")
        orc = ob.OracleWorld(eng.host)
        eng.reset()
        orc.reset()
        rng = np.random.RandomState(7)
        t0 = time.time()
        for t in range(steps):
            if name == "walkers" and t % 400 == 10:
                from metadrive_ped_amd import participants as P
                sh = orc.state["shape"].reshape(E, -1)
                hd = np.arctan2(sh["s"][:, 0], sh["c"][:, 0])
                spots = [np.stack([sh["cx"][:, 0] + d * sh["c"][:, 0] + q * sh["s"][:, 0],
                                   sh["cy"][:, 0] + d * sh["s"][:, 0] - q * sh["c"][:, 0]], 1) for d, q in ((15.0, 0.0), (30.0, 5.0))]
                free = ((sh["flags"] & 0xF) == 0) & ((orc.state["shape0"].reshape(E, -1)["flags"] & 0xF) == 0)
                if free[:, eng.A:].all(0).sum() >= 2:        # earlier walkers are gone wherever the env has reset since
                    p = eng.spawn_object("pedestrian", spots[0], hd + np.pi)
                    c = eng.spawn_object("cyclist", spots[1], hd + np.pi / 2)
                    eng.set_velocity(p, [1, 0], 1.0, in_local_frame=True)
                    eng.set_velocity(c, [1, 0], 3.0, in_local_frame=True)
                    assert p == P.spawn(orc.state, E, eng.cap, eng.A, "pedestrian", spots[0], hd + np.pi)
                    assert c == P.spawn(orc.state, E, eng.cap, eng.A, "cyclist", spots[1], hd + np.pi / 2)
                    P.set_velocity(orc.state, E, eng.cap, p, [1, 0], 1.0, in_local_frame=True)
                    P.set_velocity(orc.state, E, eng.cap, c, [1, 0], 3.0, in_local_frame=True)
            # a mix of random and lane-keeping actions so that both short and long episodes occur
            obs = orc.obs
            steer = np.clip(4.0 * (obs[:, 2] - 0.5) + 2.0 * (obs[:, 8] - 0.5), -1, 1)
            rnd = rng.uniform(-1, 1, (E, 2)).astype(np.float32)
            use_rnd = (np.arange(E) % 3 == 0)
            a = np.stack([np.where(use_rnd, rnd[:, 0] * 0.3, steer), np.where(use_rnd, np.abs(rnd[:, 1]), (obs[:, 3] < 0.35) * 0.5)], 1)
            a = a.astype(np.float32)[:, None, :]
            eng.step(torch.from_numpy(a).to(eng.device))
            orc.step(a, threads=8)
            if (t + 1) % 250 == 0:
                assert_state_equal(eng.download_state(), orc.state, where="%s step %d" % (name, t + 1))
        fl = orc.state["shape"]["flags"].reshape(E, -1)
        drv = (((fl & 0x10) != 0) & ((fl & 0x40) == 0) & ((fl & 0xF) == 1) & ((fl & 0x80) == 0)).sum(1).mean()
        print("%-8s %d steps x %d envs bit-exact (%.0f s, %.1f driving vehicles per env at the end)" % (name, steps, E, time.time() - t0, drv), flush=True)


def main_marl():
    import torch
    import oracle_binding as ob
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    from metadrive_ped_amd.envs.marl_env import (BatchedMultiAgentBottleneckEnv, BatchedMultiAgentIntersectionEnv,
                                                 BatchedMultiAgentParkingLotEnv, BatchedMultiAgentRacingEnv, BatchedMultiAgentRoundaboutEnv,
                                                 BatchedMultiAgentTinyInter, BatchedMultiAgentTollgateEnv)
    steps_all = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    only = os.environ.get("SOAK_ONLY", "")          # e.g. SOAK_ONLY=Tollgate,ParkingLot
    for cls, extra in ((BatchedMultiAgentRoundaboutEnv, {}), (BatchedMultiAgentIntersectionEnv, {}), (BatchedMultiAgentBottleneckEnv, {}),
                       (BatchedMultiAgentRoundaboutEnv, dict(num_agents=-1, map_config=dict(exit_length=40, lane_num=2))),
                       (BatchedMultiAgentTollgateEnv, {}), (BatchedMultiAgentParkingLotEnv, {}), (BatchedMultiAgentTinyInter, {}),
                       (BatchedMultiAgentRacingEnv, dict(map_config=dict(exit_length=60), horizon=700))):
        if only and not any(k in cls.__name__ for k in only.split(",")):
            continue
        racing = cls is BatchedMultiAgentRacingEnv      # 72 + 72 beams x 3256 line pieces per agent on the CPU oracle: a smaller batch
        E, steps = (6, min(steps_all, 800)) if racing else (48, steps_all)
        cfg = cls(dict(dict(num_envs=E, num_scenarios=E), **extra)).config
        eng = BatchedEngine(cfg)
        A = eng.A
        orc = ob.OracleWorld(eng.host)
        eng.reset()
        orc.reset()
        rng = np.random.RandomState(3)
        t0 = time.time()
        o_hd = (eng.host.n_side or 2)
        for t in range(steps):
            obs = orc.obs.reshape(E, A, -1)
            steer = np.clip(4.0 * (obs[..., o_hd] - 0.5) + 2.0 * (obs[..., o_hd + 6] - 0.5), -1, 1)
            rnd = rng.uniform(-1, 1, (E, A, 2)).astype(np.float32)
            use_rnd = (np.arange(A)[None, :] % 4 == 0)
            a = np.stack([np.where(use_rnd, rnd[..., 0] * 0.3, steer), np.where(use_rnd, np.abs(rnd[..., 1]), (obs[..., o_hd + 1] < 0.3) * 0.5)], -1)
            a = a.astype(np.float32)
            eng.step(torch.from_numpy(a).to(eng.device))
            orc.step(a, threads=8)
            if (t + 1) % 250 == 0:
                assert_state_equal(eng.download_state(), orc.state, where="%s step %d" % (cls.__name__, t + 1))
        print("%-36s %d steps x %d envs x %d agents bit-exact (%.0f s, %d agents created per env)" %
              (cls.__name__ + (" infinite" if extra else ""), steps, E, A, time.time() - t0, int(orc.state["next_agent_id"].mean())), flush=True)


def main_scenario():
    """Scenario mode (BASELINE configs[4]): 192 synthetic scenes of three different lengths, reactive traffic, auto reset; the ego
    driven by a route follower with excursions, and replayed (agent_policy=ReplayEgoCarPolicy).  Every state array incl. the routes
    cut on the device is compared every 100 steps."""
    import torch
    import oracle_binding as ob
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    from metadrive_ped_amd.scenario import ScenarioHostScene, make_scenario_config, synthetic_scenario
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    E = 192
    scs = [synthetic_scenario(5000 + i, T=(200, 150, 110)[i % 3]) for i in range(E)]
    keys = ["shape", "dyn", "nav", "pid", "action", "flags", "obs", "reward", "cost", "step_info", "need_reset", "next_agent_id",
            "route_n", "route_segs", "route_verts", "route_aux", "done_out"]
    for name, extra in (("scenario", dict()), ("scenario ego replay", dict(agent_policy="ReplayEgoCarPolicy"))):
        cfg = make_scenario_config(dict(dict(num_envs=E, num_scenarios=E, reactive_traffic=True, horizon=0, auto_reset=True,
                                             allowed_more_steps=20, truncate_as_terminate=True,
                                             vehicle_config=dict(lidar=dict(num_lasers=240, distance=50))), **extra))
        host = ScenarioHostScene(cfg, scs)
        eng = BatchedEngine(cfg, host=host)
        orc = ob.OracleWorld(host)
        orc.set_tracks(host.tracks["shape"], host.tracks["dyn"])
        eng.reset()
        orc.reset()
        rng = np.random.RandomState(11)
        t0 = time.time()
        cut = 0
        o_navi = (host.n_side or 2) + 6 + 1
        for t in range(steps):
            obs = orc.obs
            a = np.zeros((E, 1, 2), np.float32)
            a[:, 0, 0] = np.clip(6.0 * (obs[:, o_navi + 19] - 0.5) + 2.0 * (obs[:, o_navi + 18] - 0.5), -1, 1)
            a[:, 0, 1] = 0.35 if (t // 70) % 3 else -0.5
            a[::7, 0, 0] += 0.5 * np.sin(t * 0.04)
            a[:, 0, 0] += rng.uniform(-0.05, 0.05, E)
            before = orc.state["route_n"][:, 2].copy()
            eng.step(torch.from_numpy(a).to(eng.device))
            orc.step(a, threads=8)
            rn = orc.state["route_n"]
            cut += int(((rn[:, 0] > 0) & (rn[:, 2] != before)).sum())
            if (t + 1) % 100 == 0:
                assert_state_equal(eng.download_state(), orc.state, keys=keys, where="%s step %d" % (name, t + 1))
        print("%-20s %d steps x %d scenes bit-exact (%.0f s, %d routes cut at a later spawn frame)" % (name, steps, E, time.time() - t0, cut), flush=True)


if __name__ == "__main__":
    from metadrive_ped_amd import hostpool
    hostpool.start()                                   # before the first GPU call
    only = os.environ.get("SOAK_ONLY", "")
    if only == "scenario":
        main_scenario()
    else:
        if not only:
            main()
        main_marl()
        if not only:
            main_scenario()

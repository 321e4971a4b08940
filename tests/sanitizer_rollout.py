"""Run by tests/test_sanitizers.py in a child process with libasan preloaded: short rollouts of every env family on the oracle built
with -fsanitize=address,undefined (oracle/Makefile: asan).  Any report aborts the process (exit code != 0)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rollout(host, steps, A, tracks=False):
    import oracle_binding as ob
    lib = ob.load(os.path.join(ob.ORACLE_DIR, "_build", "libmdoracle_asan.so"))
    o = ob.OracleWorld(host, lib=lib)
    if tracks:
        o.set_tracks(host.tracks["shape"], host.tracks["dyn"])
    o.reset()
    rng = np.random.RandomState(0)
    for t in range(steps):
        a = rng.uniform(-1, 1, (host.E, A, 2)).astype(np.float32)
        a[..., 0] *= 0.3
        if t % 5:
            a[..., 1] = np.abs(a[..., 1])
        o.step(a)
    assert np.isfinite(o.state["obs"]).all()
    assert "libmdoracle_asan" in open("/proc/self/maps").read()


def main():
    maps = open("/proc/self/maps").read()
    assert "libasan" in maps, "run with LD_PRELOAD=libasan.so (tests/test_sanitizers.py does)"
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.envs import marl_env as M
    from metadrive_ped_amd.scenario import ScenarioHostScene, make_scenario_config, synthetic_scenarios
    common = dict(num_envs=6, num_scenarios=6, build_workers=1)
    for user in (dict(map=3, traffic_density=0.2, accident_prob=0.5, horizon=120),
                 dict(map="XTO", traffic_density=0.3, traffic_mode="respawn", horizon=120),
                 dict(map="rRC", traffic_density=0.2, traffic_mode="hybrid", need_inverse_traffic=True, horizon=120,
                      vehicle_config=dict(lidar=dict(num_lasers=72, distance=40, num_others=4), side_detector=dict(num_lasers=8, distance=50),
                                          lane_line_detector=dict(num_lasers=4, distance=20))),
                 dict(map=2, traffic_density=0.1, agent_policy="IDMPolicy", horizon=120)):
        h = HostScene(make_config(dict(common, mover_capacity=0, **user)))
        rollout(h, 250, h.A)
        print("ok", user.get("map"), user.get("traffic_mode", "trigger"), flush=True)
    for cls in (M.BatchedMultiAgentRoundaboutEnv, M.BatchedMultiAgentIntersectionEnv, M.BatchedMultiAgentBottleneckEnv,
                M.BatchedMultiAgentTollgateEnv, M.BatchedMultiAgentParkingLotEnv, M.BatchedMultiAgentTinyInter,
                M.BatchedMultiAgentRacingEnv, M.BatchedMultiAgentMetaDrive):
        extra = dict(map_config=dict(exit_length=60, lane_num=2), num_agents=8) if cls is M.BatchedMultiAgentRacingEnv else {}
        cfg = cls(dict(num_envs=2, num_scenarios=2, horizon=150, build_workers=1, **extra)).config
        h = HostScene(cfg)
        rollout(h, 200 if cls is not M.BatchedMultiAgentRacingEnv else 60, h.A)
        print("ok", cls.__name__, flush=True)
    for reactive, policy in ((True, "EnvInputPolicy"), (False, "EnvInputPolicy"), (True, "ReplayEgoCarPolicy")):
        cfg = make_scenario_config(dict(num_envs=4, num_scenarios=4, reactive_traffic=reactive, horizon=150, agent_policy=policy))
        h = ScenarioHostScene(cfg, synthetic_scenarios(4, 30))
        rollout(h, 220, 1, tracks=True)
        print("ok scenario", reactive, policy, flush=True)


if __name__ == "__main__":
    main()

"""Random combinations of the config options (not collected by pytest: run by hand on the GPU box): for each draw a
short rollout of HIP against the oracle, every state array compared.  Usage: python tests/fuzz_parity.py [n] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def draw(rng):
    pick = lambda *xs: xs[int(rng.randint(len(xs)))]
    cfg = dict(num_envs=int(pick(1, 5, 16, 33)), start_seed=int(rng.randint(0, 2000)), horizon=int(pick(40, 150, 1000)),
               map=pick(1, 2, 3, 4, "SCS", "XT", "rRO", "yY", "CrX", "yBY", "BS"), traffic_density=float(pick(0.0, 0.05, 0.1, 0.3)),
               traffic_mode=pick("trigger", "trigger", "respawn", "hybrid"), accident_prob=float(pick(0.0, 0.0, 0.5, 1.0)),
               random_lane_width=bool(rng.randint(2)), random_lane_num=bool(rng.randint(2)),
               need_inverse_traffic=bool(rng.randint(2)), random_agent_model=bool(rng.randint(2)),
               auto_reset=bool(rng.randint(4) > 0), crash_vehicle_done=bool(rng.randint(2)), crash_object_done=bool(rng.randint(2)),
               out_of_route_done=bool(rng.randint(4) == 0), on_continuous_line_done=bool(rng.randint(2)),
               use_lateral_reward=bool(rng.randint(2)), enable_idm_lane_change=bool(rng.randint(4) > 0),
               agent_policy=pick("EnvInputPolicy", "EnvInputPolicy", "EnvInputPolicy", "IDMPolicy"),
               step_kernel=pick("auto", "wg", "wave"))
    cfg["num_scenarios"] = int(pick(1, cfg["num_envs"], max(1, cfg["num_envs"] // 2)))
    if rng.randint(6) == 0:                      # other traffic in every episode: staged draws, md_swap_draw
        cfg["random_traffic"], cfg["traffic_draws"] = True, int(pick(2, 3))
    vc = dict(enable_reverse=bool(rng.randint(3) == 0))
    beams = int(pick(0, 30, 72, 240))
    vc["lidar"] = dict(num_lasers=beams, distance=float(pick(20, 50)) if beams else 0, num_others=int(pick(0, 0, 2, 4)) if beams else 0,
                       add_others_navi=bool(rng.randint(2)))
    if rng.randint(3) == 0:
        vc["side_detector"] = dict(num_lasers=int(pick(2, 8)), distance=50)
    if rng.randint(3) == 0:
        vc["lane_line_detector"] = dict(num_lasers=int(pick(2, 4)), distance=20)
    if rng.randint(4) == 0:
        vc["vehicle_model"] = "varying_dynamics"
        cfg["random_agent_model"] = False
        cfg["random_dynamics"] = dict(max_engine_force=(100, 3000), max_brake_force=(20, 600), wheel_friction=(0.1, 2.5),
                                      max_steering=(10, 80), mass=(300, 3000))
    if rng.randint(4) == 0:
        vc["spawn_velocity"] = [float(rng.uniform(0, 15)), 0.0]
    cfg["vehicle_config"] = vc
    return cfg


def draw_marl(rng):
    pick = lambda *xs: xs[int(rng.randint(len(xs)))]
    kind = pick("roundabout", "intersection", "bottleneck", "bidirection", "pg", "tollgate", "parking_lot", "tinyinter", "racing")
    cfg = dict(num_envs=int(pick(1, 4, 9)), start_seed=int(rng.randint(0, 500)), horizon=int(pick(60, 200, 1000)),
               num_agents=int(pick(1, 3, 8, 12, -1)), delay_done=int(pick(0, 5, 25)), allow_respawn=bool(rng.randint(4) > 0),
               crash_done=bool(rng.randint(2)), out_of_road_done=bool(rng.randint(4) > 0), random_agent_model=bool(rng.randint(4) == 0),
               map_config=dict(exit_length=int(pick(30, 50, 60)), lane_num=int(pick(2, 3))))
    cfg["num_scenarios"] = int(pick(1, cfg["num_envs"]))
    beams = int(pick(0, 30, 72))
    cfg["vehicle_config"] = dict(lidar=dict(num_lasers=beams, distance=float(pick(20, 40)) if beams else 0,
                                            num_others=int(pick(0, 0, 4)) if beams else 0, add_others_navi=bool(rng.randint(2))))
    if kind == "pg":
        cfg["map"] = pick(2, 3, "SCS", "XT")
        cfg["num_agents"] = min(cfg["num_agents"], 12) if cfg["num_agents"] > 0 else -1
    if kind in ("bottleneck", "bidirection"):
        cfg["map_config"]["lane_num"] = int(pick(3, 4))
    if kind == "tollgate":
        cfg["map_config"] = dict(exit_length=int(pick(40, 70)), lane_num=int(pick(2, 3)), toll_lane_num=int(pick(6, 8)))
        cfg["vehicle_config"]["min_pass_steps"] = int(pick(5, 30))
        cfg["cross_yellow_line_done"] = bool(rng.randint(2))
        cfg["overspeed_penalty"] = float(pick(0.5, 2.0))
    if rng.randint(5) == 0 and kind != "tinyinter":
        cfg["agent_policy"] = "IDMPolicy"            # every agent driven by its own IDMPolicy (round 3)
    if kind == "tinyinter":
        cfg["map_config"] = dict(exit_length=int(pick(30, 40)), lane_num=1, lane_width=4.0, radius=pick(None, 20.0, 50.0))
        cfg["num_agents"] = int(pick(1, 3, 8))
        cfg.pop("delay_done")
    if kind == "racing":
        cfg["map_config"] = dict(exit_length=60, lane_num=2)
        cfg["num_agents"] = int(pick(1, 3, 8, 12))
        cfg["allow_respawn"] = False
        cfg["idle_done"] = bool(rng.randint(2))
        cfg["crash_sidewalk_done"] = bool(rng.randint(2))
        cfg["vehicle_config"]["side_detector"] = dict(num_lasers=int(pick(4, 72)), distance=50)
    if kind == "parking_lot":
        cfg["map_config"] = dict(exit_length=int(pick(20, 30)), lane_num=1)
        cfg["parking_space_num"] = int(pick(4, 8, 12))
        cfg["num_agents"] = int(pick(1, 3, 6, 10))
    return kind, cfg


def draw_scenario(rng):
    pick = lambda *xs: xs[int(rng.randint(len(xs)))]
    scene = dict(T=int(pick(40, 90, 200)), n_vehicles=int(pick(0, 2, 9, 18, 26)), n_parked=int(pick(0, 3)),
                 n_pedestrians=int(pick(0, 2)), n_cones=int(pick(0, 4)))
    beams = int(pick(0, 30, 120, 240))
    cfg = dict(num_envs=int(pick(1, 6, 17)), reactive_traffic=bool(rng.randint(4) > 0), horizon=int(pick(50, 150, 400)),
               auto_reset=bool(rng.randint(4) > 0), no_traffic=bool(rng.randint(8) == 0), no_static_vehicles=bool(rng.randint(4) == 0),
               filter_overlapping_car=bool(rng.randint(4) > 0), crash_vehicle_done=bool(rng.randint(2)), out_of_route_done=bool(rng.randint(3) == 0),
               relax_out_of_road_done=bool(rng.randint(2)), no_negative_reward=bool(rng.randint(2)),
               vehicle_config=dict(lidar=dict(num_lasers=beams, distance=float(pick(30, 50)) if beams else 0),
                                   side_detector=dict(num_lasers=int(pick(0, 12, 40)), distance=50),
                                   lane_line_detector=dict(num_lasers=int(pick(0, 0, 4)), distance=20)))
    if rng.randint(4) == 0:
        cfg["agent_policy"] = "ReplayEgoCarPolicy"
    cfg["num_scenarios"] = cfg["num_envs"]
    return scene, cfg


def run_scenario(it, rng):
    """One scenario-mode draw: synthetic scenes of a random shape, every state array (and the routes cut on the device) compared."""
    import torch
    import oracle_binding as ob
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    from metadrive_ped_amd.scenario import ScenarioHostScene, make_scenario_config, synthetic_scenario
    scene, user = draw_scenario(rng)
    cfg = make_scenario_config(user)
    E = cfg["num_envs"]
    host = ScenarioHostScene(cfg, [synthetic_scenario(int(rng.randint(0, 10000)), **scene) for _ in range(E)])
    eng = BatchedEngine(cfg, host=host)
    orc = ob.OracleWorld(host)
    orc.set_tracks(host.tracks["shape"], host.tracks["dyn"])
    eng.reset()
    orc.reset()
    keys = ["shape", "dyn", "nav", "pid", "action", "flags", "obs", "reward", "cost", "step_info", "need_reset", "next_agent_id"]
    if "route_n" in orc.state:
        keys += ["route_n", "route_segs", "route_verts", "route_aux"]
    where = "fuzz %d scenario %r %r" % (it, scene, user)
    arng = np.random.RandomState(it)
    for t in range(120):
        a = arng.uniform(-1, 1, (E, 1, 2)).astype(np.float32)
        a[..., 0] *= 0.25
        if t % 9:
            a[..., 1] = np.abs(a[..., 1])
        eng.step(torch.from_numpy(a).to(eng.device))
        orc.step(a)
        if t % 30 == 29:
            assert_state_equal(eng.download_state(), orc.state, keys=keys, where=where + " step %d" % t)


def main():
    import torch
    import oracle_binding as ob
    from helpers import assert_state_equal
    from metadrive_ped_amd import participants as P
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    done = skipped = n_scn = 0
    t0 = time.time()
    for it in range(n):
        if rng.randint(6) == 0:
            try:
                run_scenario(it, rng)
                done += 1
                n_scn += 1
            except (NotImplementedError, ValueError) as ex:
                skipped += 1
                print("skip %d: %s" % (it, str(ex)[:90]), flush=True)
            continue
        marl = rng.randint(4) == 0
        try:
            if marl:
                from metadrive_ped_amd.envs import marl_env as M
                kind, user = draw_marl(rng)
                cls = dict(roundabout=M.BatchedMultiAgentRoundaboutEnv, intersection=M.BatchedMultiAgentIntersectionEnv,
                           bottleneck=M.BatchedMultiAgentBottleneckEnv, bidirection=M.BatchedMultiAgentBidirectionEnv,
                           pg=M.BatchedMultiAgentMetaDrive, tollgate=M.BatchedMultiAgentTollgateEnv,
                           parking_lot=M.BatchedMultiAgentParkingLotEnv, tinyinter=M.BatchedMultiAgentTinyInter,
                           racing=M.BatchedMultiAgentRacingEnv)[kind]
                cfg = cls(user).config
                user = dict(user, marl_map=kind)
            else:
                user = draw(rng)
                cfg = make_config(dict(user, mover_capacity=0))
            eng = BatchedEngine(cfg)
        except (NotImplementedError, ValueError) as ex:      # combinations the config layer rejects (e.g. too many movers)
            skipped += 1
            print("skip %d: %s" % (it, str(ex)[:90]), flush=True)
            continue
        E, A = eng.E, eng.A
        orc = ob.OracleWorld(eng.host)
        eng.reset()
        orc.reset()
        where = "fuzz %d %r" % (it, user)
        assert_state_equal(eng.download_state(), orc.state, where=where + " reset")
        arng = np.random.RandomState(it)
        draw_of = np.zeros(E, np.int64)
        for t in range(120):
            if t == 20 and not marl and eng.cap > eng.host.state["shape0"].reshape(E, -1)["flags"].astype(bool).sum(1).max() + 1 and rng.randint(2):
                sh = orc.state["shape"].reshape(E, -1)
                spot = np.stack([sh["cx"][:, 0] + 14.0 * sh["c"][:, 0], sh["cy"][:, 0] + 14.0 * sh["s"][:, 0]], 1)
                try:
                    p = eng.spawn_object("pedestrian", spot, 0.3)
                    assert p == P.spawn(orc.state, E, eng.cap, A, "pedestrian", spot, 0.3)
                    eng.set_velocity(p, [0.5, 0.5], None)
                    P.set_velocity(orc.state, E, eng.cap, p, [0.5, 0.5], None)
                except RuntimeError:
                    pass
            a = arng.uniform(-1, 1, (E, A, 2)).astype(np.float32)
            a[..., 0] *= 0.3
            if t % 7:
                a[..., 1] = np.abs(a[..., 1])
            eng.step(torch.from_numpy(a).to(eng.device))
            orc.step(a)
            if getattr(eng, "_staged", None) is not None:          # what md_swap_draw does, on the oracle's arrays
                K = len(eng.draw_hosts_)
                for e in np.nonzero(orc.state["need_reset"])[0]:
                    draw_of[e] = (draw_of[e] + 1) % K
                    rows = slice(e * eng.cap, (e + 1) * eng.cap)
                    for k in BatchedEngine.DRAW_ARRAYS:
                        if k in orc.state:
                            orc.state[k][rows] = eng.draw_hosts_[draw_of[e]].state[k][rows]
                            if k + "0" in orc.state and not k.endswith("0"):
                                orc.state[k + "0"][rows] = eng.draw_hosts_[draw_of[e]].state[k][rows]
            if t % 30 == 29:
                assert_state_equal(eng.download_state(), orc.state, where=where + " step %d" % t)
        done += 1
        del eng
    print("%d random configurations bit-exact over 120 steps each (%d of them scenario mode; %d rejected by the config layer), %.0f s" % (done, n_scn, skipped, time.time() - t0))


if __name__ == "__main__":
    from metadrive_ped_amd import hostpool
    hostpool.start()                                   # before the first GPU call
    main()

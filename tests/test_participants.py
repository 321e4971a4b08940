"""User-spawned pedestrians / cyclists (the reference fork's namesake: component/traffic_participants/*.py).

CPU part: the host edit (participants.py) + the oracle's step.  The known answer of the reference's own
tests/test_functionality/test_pedestrian.py:38-68 is reproduced: a pedestrian spawned at x = 30 walking 1 m/s for 300
steps, standing for 200, walking 2 m/s for 499 ends at x = 160 +- 1.
"""
import numpy as np
import pytest

import oracle_binding as ob
from metadrive_ped_amd import abi, participants as P
from metadrive_ped_amd.config import make_config
from metadrive_ped_amd.engine import HostScene


def _world(**kw):
    cfg = dict(num_envs=2, num_scenarios=2, traffic_density=0.0, map="X", start_seed=22, random_lane_width=True,
               traffic_mode="hybrid", mover_capacity=8, horizon=5000, auto_reset=False)
    cfg.update(kw)
    h = HostScene(make_config(cfg))
    o = ob.OracleWorld(h)
    o.reset()
    return h, o


def test_reference_pedestrian_known_answer():
    h, o = _world()
    E, cap = h.E, h.cap
    s1 = P.spawn(o.state, E, cap, h.A, "pedestrian", [30.0, 0.0], 0.0)
    s2 = P.spawn(o.state, E, cap, h.A, "pedestrian", [30.0, 6.0], 0.0)
    assert s1 == cap - 1 and s2 == cap - 2                      # top-down, distinct
    P.set_velocity(o.state, E, cap, s1, [1, 0], 1, in_local_frame=True)
    P.set_velocity(o.state, E, cap, s2, [1, 0], 0, in_local_frame=True)
    stand = np.zeros((E, 1, 2), np.float32)
    for s in range(1, 1000):
        o.step(stand)
        if s == 300:
            P.set_velocity(o.state, E, cap, s1, [1, 0], 0, in_local_frame=True)
        elif s == 500:
            P.set_velocity(o.state, E, cap, s1, [1, 0], 2, in_local_frame=True)
    sh = o.state["shape"].reshape(E, cap)
    assert (np.abs(sh["cx"][:, s1] - 160.0) < 1.0).all(), "Pedestrian movement error!"
    assert (np.abs(sh["cy"][:, s1]) < 1e-4).all()
    assert (np.abs(sh["cx"][:, s2] - 30.0) < 1e-3).all() and (np.abs(sh["cy"][:, s2] - 6.0) < 1e-4).all()
    assert ((sh["flags"][:, s1] & abi.KIND_MASK) == abi.KIND_PEDESTRIAN).all()


def test_agent_sees_and_hits_a_pedestrian_and_reset_clears_it():
    h, o = _world(map="SS", random_lane_width=False, auto_reset=True, horizon=400)
    E, cap = h.E, h.cap
    sh = o.state["shape"].reshape(E, cap)
    ego = np.stack([sh["cx"][:, 0], sh["cy"][:, 0]], 1)
    ahead = ego + 20.0 * np.stack([sh["c"][:, 0], sh["s"][:, 0]], 1)
    slot = P.spawn(o.state, E, cap, h.A, "pedestrian", ahead, 0.0)
    o.step(np.zeros((E, 1, 2), np.float32))
    B = h.n_beams
    obs = o.obs.reshape(E, -1)
    lidar0 = obs[:, h.obs_dim - B]                              # beam 0 points along the heading
    assert np.allclose(lidar0, (20.0 - 0.35) / 50.0, atol=2e-3)
    hit = np.zeros(E, bool)
    for t in range(120):
        o.step(np.tile(np.array([0.0, 1.0], np.float32), (E, 1, 1)))
        fl = o.state["flags"].reshape(E, cap)[:, 0]
        now = (fl & abi.FL_CRASH_HUMAN) != 0
        assert (((fl & abi.FL_TERMINATED) != 0) >= now).all()   # crash_human_done: terminal
        hit |= now
        if hit.all():
            break
    assert hit.all()
    o.step(np.zeros((E, 1, 2), np.float32))                     # auto-reset: back to the snapshot, spawned objects gone
    sh = o.state["shape"].reshape(E, cap)
    assert ((sh["flags"][:, slot] & abi.KIND_MASK) == 0).all()


def test_cyclist_velocity_frames_slots_and_errors():
    h, o = _world(map="SS", random_lane_width=False)
    E, cap = h.E, h.cap
    c = P.spawn(o.state, E, cap, h.A, "cyclist", [[40.0, 3.0], [50.0, -2.0]], np.pi / 2, envs=[0, 1])
    sh = o.state["shape"].reshape(E, cap)
    assert np.allclose(sh["hl"][:, c], 0.875) and np.allclose(sh["hw"][:, c], 0.2)
    P.set_velocity(o.state, E, cap, c, [1, 0], 3.0, in_local_frame=True)        # forward = +y for heading pi/2
    d = o.state["dyn"].reshape(E, cap)
    assert np.allclose(d["steering"][:, c], 0.0, atol=1e-6) and np.allclose(d["throttle"][:, c], 3.0, atol=1e-5)
    P.set_velocity(o.state, E, cap, c, [3.0, 4.0], None, envs=[1])               # world frame, taken as is
    assert np.allclose(d["speed"][1, c], 5.0)
    y0 = sh["cy"][:, c].copy()
    for _ in range(10):
        o.step(np.zeros((E, 1, 2), np.float32))
    sh = o.state["shape"].reshape(E, cap)
    assert abs(sh["cy"][0, c] - y0[0] - 3.0) < 1e-3 and abs(sh["cy"][1, c] - y0[1] - 4.0) < 1e-3
    # the record / export path names it by its type
    from metadrive_ped_amd.scenario_export import tracks_to_scenarios
    tracks = ob.record_episode(o, [np.zeros((E, 1, 2), np.float32)] * 3)
    sc = tracks_to_scenarios(tracks, h, envs=[0])[0]
    assert sc["tracks"][str(c)]["type"] == "CYCLIST" and sc["metadata"]["number_summary"]["num_objects_each_type"]["CYCLIST"] == 1
    P.clear(o.state, E, cap, c)
    assert ((o.state["shape"].reshape(E, cap)["flags"][:, c] & abi.KIND_MASK) == 0).all()
    with pytest.raises(ValueError):
        P.set_velocity(o.state, E, cap, c, [1, 0], 1.0)        # nothing there any more
    with pytest.raises(ValueError):
        P.clear(o.state, E, cap, 0)                             # the agent is not a participant
    with pytest.raises(ValueError):
        P.spawn(o.state, E, cap, h.A, "truck", [0, 0])
    for _ in range(cap - h.A):
        P.spawn(o.state, E, cap, h.A, "pedestrian", [100.0, 100.0])
    with pytest.raises(RuntimeError):
        P.spawn(o.state, E, cap, h.A, "pedestrian", [100.0, 100.0])


def test_traffic_near_a_pedestrian_takes_the_bare_except_fallback(cs_dist):
    """IDMPolicy.act with a pedestrian inside the 50 m ghost cylinder (policy/idm_policy.py:110,254-260; golden cases in
    tests/golden/idm_policy.json): no lead object, distance 5 -- so a traffic vehicle queueing behind another one stops
    braking for it -- while vehicles farther than 50 m from the pedestrian keep following their lead."""
    from helpers import make_cfg
    cfg = make_cfg(cs_dist, num_envs=6, num_scenarios=6, start_seed=40, traffic_density=0.3, auto_reset=False, horizon=5000)
    h = HostScene(cfg)
    a, b = ob.OracleWorld(h), ob.OracleWorld(h)
    a.reset(), b.reset()
    E, cap = h.E, h.cap
    go = np.tile(np.array([0.0, 0.6], np.float32), (E, 1, 1))
    for _ in range(40):            # let the first traffic block wake up
        a.step(go), b.step(go)
    sh = a.state["shape"].reshape(E, cap)
    fl = sh["flags"]
    drives = ((fl & abi.KIND_MASK) == abi.KIND_VEHICLE) & ((fl & abi.F_ALIVE) != 0) & ((fl & (abi.F_PENDING | abi.F_STATIC | abi.F_AGENT)) == 0)
    assert drives.any()
    # the pedestrian stands 20 m off to the side of the first driving traffic vehicle of every env that has one
    pos = np.zeros((E, 2), np.float32) + 1.0e4
    first = np.full(E, -1)
    for e in range(E):
        js = np.nonzero(drives[e])[0]
        if len(js):
            first[e] = js[0]
            pos[e] = [sh["cx"][e, js[0]] - 20.0 * sh["s"][e, js[0]], sh["cy"][e, js[0]] + 20.0 * sh["c"][e, js[0]]]
    P.spawn(b.state, E, cap, h.A, "pedestrian", pos, 0.0)
    a.step(go), b.step(go)
    act_a, act_b = a.state["action"].reshape(E, cap, 2), b.state["action"].reshape(E, cap, 2)
    shb = b.state["shape"].reshape(E, cap)
    near = far = changed = 0
    for e in range(E):
        for j in np.nonzero(drives[e])[0]:
            d = np.hypot(shb["cx"][e, j] - pos[e, 0], shb["cy"][e, j] - pos[e, 1])
            if d < 45.0:
                near += 1
                # fallback: the IDM term uses "no front object" -> acceleration 1 - (v/v0)^10 >= what car following gives
                assert act_b[e, j, 1] >= act_a[e, j, 1] - 1e-6
                changed += act_b[e, j, 1] != act_a[e, j, 1]
            elif d > 56.0:
                far += 1
                assert act_b[e, j].tobytes() == act_a[e, j].tobytes()
    assert near >= 3, (near, far)

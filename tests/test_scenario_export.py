"""Episode export in the reference's scenario-description format (record / replay).

tests/golden/scenario_export.json was written by oracle/gen/gen_golden.py `scenario_export`: the SAME oracle-driven
episode exported by metadrive_ped_amd.scenario_export, passed through the reference's own
ScenarioDescription.sanity_check(valid_check=True) (accepted) and summarised by its update_summaries.  Here the
export is rebuilt and (i) its summaries must equal the reference's, (ii) the sanity rules are restated and checked,
(iii) the tracks must equal the recorded frames.
"""
import json
import math
import os
import sys

import numpy as np
import pytest

from metadrive_ped_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))

FIRST_LEVEL = {"tracks", "version", "id", "dynamic_map_states", "map_features", "length", "metadata"}


def _case():
    import oracle_binding as ob
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.scenario_export import tracks_to_scenarios
    E, T = 3, 80
    host = HostScene(make_config(dict(num_envs=E, num_scenarios=E, start_seed=20, traffic_density=0.2, accident_prob=1.0,
                                      traffic_mode="respawn", horizon=1000, auto_reset=False)))
    o = ob.OracleWorld(host)
    o.reset()
    acts = [np.tile(np.array([0.03 * math.sin(0.07 * t), 0.6], np.float32), (E, 1, 1)) for t in range(T)]
    tracks = ob.record_episode(o, acts)
    return host, tracks, tracks_to_scenarios(tracks, host)


def _sanity(sc):
    """ScenarioDescription.sanity_check(valid_check=True), restated (scenario_description.py:199-321)."""
    assert FIRST_LEVEL.issubset(sc)
    T = sc["length"]
    assert {"metadrive_processed", "coordinate", "ts"}.issubset(sc["metadata"]) and sc["metadata"]["ts"].shape == (T, )
    for oid, tr in sc["tracks"].items():
        assert {"type", "state", "metadata"}.issubset(tr) and tr["type"] != "UNSET"
        assert "position" in tr["state"] and "heading" in tr["state"]
        valid = tr["state"]["valid"]
        assert valid.sum() >= 1
        for k, arr in tr["state"].items():
            assert len(arr) == T and arr.ndim in (1, 2)
            a = arr[..., :2] if k == "position" else arr
            assert abs(np.sum(a[~valid])) < 1e-2, (oid, k)
        assert tr["metadata"]["object_id"] == oid and "type" in tr["metadata"]
    for f in sc["map_features"].values():
        assert f["type"].startswith(("LANE_", "ROAD_LINE_", "UNKNOWN_LINE")) and isinstance(f["polyline"], np.ndarray)
        assert f["polyline"].shape[1] == 2


def test_export_matches_the_reference_summaries_of_the_same_episode():
    gold = json.load(open(os.path.join(HERE, "golden", "scenario_export.json")))
    assert gold["accepted_by_reference_sanity_check"] is True
    host, tracks, scs = _case()
    assert len(scs) == len(gold["scenarios"]) == host.E
    for sc, g in zip(scs, gold["scenarios"]):
        _sanity(sc)
        assert sc["id"] == g["id"] and sc["length"] == g["length"]
        ours = sc["metadata"]["object_summary"]
        assert set(ours) == set(g["object_summary"])
        for oid, go in g["object_summary"].items():
            o = ours[oid]
            for k in ("type", "object_id", "track_length", "valid_length", "continuous_valid_length"):
                assert o[k] == go[k], (oid, k)
            assert abs(o["moving_distance"] - go["moving_distance"]) <= 1e-3 * max(1.0, go["moving_distance"])
        ns = dict(sc["metadata"]["number_summary"])
        for k in ("object_types", "num_traffic_light_types"):
            ns[k] = sorted(ns[k])
        assert ns == g["number_summary"]
        sdc = sc["metadata"]["sdc_id"]
        assert abs(ours[sdc]["moving_distance"] - g["sdc_moving_dist"]) < 1e-3 * g["sdc_moving_dist"]


def test_tracks_equal_the_recorded_frames_and_lanes_cover_the_map():
    host, tracks, scs = _case()
    T = tracks["shape"].shape[0]
    shape = tracks["shape"].reshape(T, host.E, host.cap)
    dyn = tracks["dyn"].reshape(T, host.E, host.cap, 2)
    for e, sc in enumerate(scs):
        ego = sc["tracks"]["0"]
        assert ego["type"] == "VEHICLE" and ego["state"]["valid"].all()
        assert np.array_equal(ego["state"]["position"][:, 0], shape["cx"][:, e, 0])
        assert np.array_equal(ego["state"]["position"][:, 1], shape["cy"][:, e, 0])
        assert np.array_equal(ego["state"]["heading"], dyn[:, e, 0, 0])
        sp = np.hypot(ego["state"]["velocity"][:, 0], ego["state"]["velocity"][:, 1])
        assert np.allclose(sp, dyn[:, e, 0, 1], atol=1e-4)
        assert np.allclose(ego["state"]["length"], 2.0 * shape["hl"][:, e, 0])
        # every alive slot-frame is in exactly one track
        n_alive = int((((shape["flags"][:, e] & abi.F_ALIVE) != 0) & ((shape["flags"][:, e] & abi.KIND_MASK) != 0)).sum())
        assert sum(int(t["state"]["valid"].sum()) for t in sc["tracks"].values()) == n_alive
        # the ego's recorded positions lie on some exported lane polygon's bounding box and near a centre line
        polys = np.concatenate([f["polyline"] for f in sc["map_features"].values()])
        for t in (0, T // 2, T - 1):
            d = np.hypot(polys[:, 0] - ego["state"]["position"][t, 0], polys[:, 1] - ego["state"]["position"][t, 1]).min()
            assert d < 4.0
        mt = host.map_tables[int(host.world.arrays["env_map"][e])]
        lanes_f = {k: f for k, f in sc["map_features"].items() if f["type"].startswith("LANE_")}
        lines_f = {k: f for k, f in sc["map_features"].items() if not f["type"].startswith("LANE_")}
        assert len(lanes_f) == len(mt.lane_objs)
        # PGMap.get_boundary_line_vector: the left line of every lane that has one, plus the right line of each road's last lane
        assert len(lines_f) >= len(mt.roads) and {f["type"] for f in lines_f.values()} >= {"ROAD_LINE_SOLID_SINGLE_WHITE",
                                                                                         "ROAD_LINE_BROKEN_SINGLE_WHITE"}
        for k, f in lanes_f.items():
            for other in f["exit_lanes"] + f["entry_lanes"] + f["left_neighbor"] + f["right_neighbor"]:
                assert other in sc["map_features"]
            seg = np.hypot(*np.diff(f["polyline"], axis=0).T)
            assert seg.max() <= 1.0 + 1e-3
            assert abs(seg.sum() - float(mt.lanes[int(k[5:])]["length"])) < 0.05 * max(1.0, seg.sum())


def test_a_refilled_slot_becomes_a_new_object():
    """Respawned traffic / a new agent in the same slot is another object (the reference names it anew)."""
    from metadrive_ped_amd.scenario_export import _segments
    valid = np.array([1, 1, 1, 0, 0, 1, 1, 1, 1], bool)
    x = np.array([0, 1, 2, 0, 0, 50, 51, 90, 91], np.float32)
    assert _segments(valid, x, np.zeros_like(x)) == [(0, 3), (5, 7), (7, 9)]
    assert _segments(np.zeros(4, bool), x[:4], x[:4]) == []


def test_exported_scenarios_load_back_as_replay_tracks():
    """record -> export_scenarios -> scenarios_to_tracks -> traffic_mode='replay': the traffic of the replayed episode is
    where the scenario descriptions say, frame by frame, and the agent driven by the recorded actions sees the recorded
    observations (the replayed bodies are kinematic: same poses, same lidar)."""
    import oracle_binding as ob
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.scenario_export import scenarios_to_tracks, tracks_to_scenarios
    E, T = 3, 60
    base = dict(num_envs=E, num_scenarios=E, start_seed=20, traffic_density=0.2, horizon=1000, auto_reset=False)
    host = HostScene(make_config(dict(base)))
    o = ob.OracleWorld(host)
    o.reset()
    acts = [np.tile(np.array([0.03 * math.sin(0.07 * t), 0.6], np.float32), (E, 1, 1)) for t in range(T)]
    obs_rec = []
    frames = dict(shape=np.zeros((T + 1, E * host.cap), abi.SHAPE_DT), dyn=np.zeros((T + 1, E * host.cap, 2), np.float32))
    for k in range(T + 1):
        if k:
            o.step(acts[k - 1])
            obs_rec.append(o.obs.copy())
        frames["shape"][k] = o.state["shape"]
        frames["dyn"][k, :, 0], frames["dyn"][k, :, 1] = o.state["dyn"]["heading"], o.state["dyn"]["speed"]
    scs = tracks_to_scenarios(dict(frames, seeds=list(host.seeds), cap=host.cap), host)
    rp_host = HostScene(make_config(dict(base, traffic_mode="replay", mover_capacity=host.cap)))
    tracks = scenarios_to_tracks(scs, rp_host)
    alive = (frames["shape"]["flags"] & abi.F_ALIVE) != 0
    for f in ("cx", "cy", "hl", "hw"):
        assert np.array_equal(tracks["shape"][f][alive], frames["shape"][f][alive]), f
    assert np.array_equal(tracks["shape"]["flags"][alive] & abi.KIND_MASK, frames["shape"]["flags"][alive] & abi.KIND_MASK)
    r = ob.OracleWorld(rp_host)
    r.set_tracks(tracks["shape"], tracks["dyn"])
    r.reset()
    for t in range(T):
        r.step(acts[t])
        sh = r.state["shape"].reshape(E, -1)
        want = frames["shape"][t + 1].reshape(E, -1)
        on = (want["flags"][:, 1:] & abi.F_ALIVE) != 0
        assert np.array_equal(sh["cx"][:, 1:][on], want["cx"][:, 1:][on]) and np.array_equal(sh["cy"][:, 1:][on], want["cy"][:, 1:][on])
        assert np.abs(r.obs - obs_rec[t]).max() < 2e-3, t          # headings pass through float32 cos / sin once more
    with pytest.raises(ValueError):
        scenarios_to_tracks(scs[:2], rp_host)
    bad = [dict(s, metadata=dict(s["metadata"], seed=999)) for s in scs]
    with pytest.raises(ValueError):
        scenarios_to_tracks(bad, rp_host)

"""ScenarioNet dataset directories (scenario_data.py: what the reference's ScenarioDataManager reads, scenario/utils.py:324-395)
and batches of scenes of different lengths.  CPU only: host tables + oracle."""
import os
import pickle

import numpy as np
import pytest

import oracle_binding as ob
from metadrive_ped_amd import abi, scenario_data as sd
from metadrive_ped_amd.envs.scenario_env import BatchedScenarioEnv
from metadrive_ped_amd.scenario import ScenarioHostScene, make_scenario_config, synthetic_scenario, synthetic_scenarios

SC_KEYS = ["shape", "dyn", "nav", "pid", "action", "flags", "obs", "reward", "cost", "step_info"]


def _oracle(host):
    o = ob.OracleWorld(host)
    o.set_tracks(host.tracks["shape"], host.tracks["dyn"])
    return o


def test_scenario_file_names():
    # scenario_description.py:382-396
    assert sd.is_scenario_file("sd_nuscenes_v1.0-mini_scene-0061.pkl")
    assert sd.is_scenario_file("/some/where/12.pkl")
    assert not sd.is_scenario_file("dataset_summary.pkl")
    assert not sd.is_scenario_file("sd_x.json")
    assert not sd.is_scenario_file("a12.pkl")


def test_dataset_round_trip_and_index_walk(tmp_path):
    scs = synthetic_scenarios(5, 40, T=60, n_vehicles=4, n_parked=1, n_pedestrians=1, n_cones=1)
    names = sd.write_dataset(str(tmp_path), scs, dataset_name="syn")
    assert all(sd.is_scenario_file(n) for n in names) and len(set(names)) == 5
    summary, order, mapping = sd.read_dataset_summary(str(tmp_path))
    assert order == names and set(mapping) == set(names) and sd.get_number_of_scenarios(str(tmp_path)) == 5
    for n, sc in zip(names, scs):
        assert summary[n]["sdc_id"] == sc["metadata"]["sdc_id"]
    # scene e <-> scenario start + (offset + e) % num_scenarios; each file is read once
    cfg = make_scenario_config(dict(num_envs=7, data_directory=str(tmp_path), start_scenario_index=1, num_scenarios=3, env_seed_offset=2))
    assert sd.scenario_indices(cfg) == [3, 1, 2, 3, 1, 2, 3]
    got = sd.load_scenarios(cfg)
    assert got[0] is got[3] is got[6] and got[1] is got[4]
    for g, i in zip(got, sd.scenario_indices(cfg)):
        sdc = scs[i]["metadata"]["sdc_id"]
        assert np.array_equal(g["tracks"][sdc]["state"]["position"], scs[i]["tracks"][sdc]["state"]["position"])
        assert set(g["map_features"]) == set(scs[i]["map_features"])
    # the reference's own assertions on an insufficient dataset (scenario_data_manager.py:45-49)
    with pytest.raises(ValueError, match="Insufficient scenarios"):
        sd.load_scenarios(make_scenario_config(dict(num_envs=2, data_directory=str(tmp_path), start_scenario_index=3, num_scenarios=3)))
    with pytest.raises(ValueError, match="Insufficient scenarios"):
        sd.load_scenarios(make_scenario_config(dict(num_envs=2, data_directory=str(tmp_path), start_scenario_index=5, num_scenarios=1)))
    os.remove(os.path.join(str(tmp_path), names[2]))
    with pytest.raises(FileNotFoundError):
        sd.read_dataset_summary(str(tmp_path))
    with pytest.raises(FileNotFoundError):
        sd.read_dataset_summary(str(tmp_path / "nowhere"))


def test_old_layout_without_summary_sorts_by_number(tmp_path):
    scs = synthetic_scenarios(3, 70, T=40, n_vehicles=2, n_parked=0, n_pedestrians=0, n_cones=0)
    for name, sc in zip(("10.pkl", "2.pkl", "0.pkl"), scs):
        with open(os.path.join(str(tmp_path), name), "wb") as f:
            pickle.dump(sc, f)
    with open(os.path.join(str(tmp_path), "notes.txt"), "w") as f:
        f.write("not a scenario")
    summary, order, mapping = sd.read_dataset_summary(str(tmp_path))
    assert order == ["0.pkl", "2.pkl", "10.pkl"] and all(v == "" for v in mapping.values())
    assert summary["10.pkl"]["sdc_id"] == scs[0]["metadata"]["sdc_id"]
    with open(os.path.join(str(tmp_path), "3.pkl"), "wb") as f:
        pickle.dump({"tracks": {}}, f)
    with pytest.raises(KeyError):
        sd.read_scenario_data(os.path.join(str(tmp_path), "3.pkl"))


def test_env_from_data_directory_builds_the_same_host_tables(tmp_path):
    scs = synthetic_scenarios(3, 11, T=80, n_vehicles=6)
    sd.write_dataset(str(tmp_path), scs)
    common = dict(num_envs=3, num_scenarios=3, reactive_traffic=True)
    env_a = BatchedScenarioEnv(dict(common, data_directory=str(tmp_path)))
    env_b = BatchedScenarioEnv(dict(common), scenarios=scs)
    ha, hb = ScenarioHostScene(env_a.config, env_a.scenarios), ScenarioHostScene(env_b.config, env_b.scenarios)
    for k, v in hb.world.arrays.items():
        assert np.array_equal(np.asarray(ha.world.arrays[k]), np.asarray(v)), k
    for k in ("shape", "dyn"):
        assert np.array_equal(ha.tracks[k], hb.tracks[k])


def test_scenes_of_different_lengths_in_one_batch():
    """A batch's frame tables have the longest scene's length; a shorter scene ends with invalid frames, which is what the
    reference does after `current_scenario_length` (replayed movers are cleaned, static objects and reactive vehicles stay:
    scenario_traffic_manager.py:106-137), and `allowed_more_steps` counts from the scene's OWN length (scenario_env.py:177)."""
    kw = dict(n_vehicles=8, n_parked=2, n_pedestrians=1, n_cones=2)
    short, long_ = synthetic_scenario(5, T=60, **kw), synthetic_scenario(6, T=110, **kw)
    base = dict(num_scenarios=1, reactive_traffic=True, horizon=0, auto_reset=False, allowed_more_steps=15)
    mixed = ScenarioHostScene(make_scenario_config(dict(base, num_envs=2, num_scenarios=2)), [short, long_])
    alone = ScenarioHostScene(make_scenario_config(dict(base, num_envs=1)), [short])
    assert mixed.T == 110 and alone.T == 60
    meta = mixed.world.arrays["track_meta"].reshape(2, mixed.cap, 4)
    assert meta[0, 0, 1] == 60 and meta[1, 0, 1] == 110
    om, oa = _oracle(mixed), _oracle(alone)
    om.reset()
    oa.reset()
    cap_m, cap_a = mixed.cap, alone.cap
    assert cap_m == cap_a
    max_step_at = [None, None]
    for t in range(1, 100):
        act = np.zeros((2, 1, 2), np.float32)
        act[:, 0, 1] = -1.0          # the ego brakes and stands: neither arrival nor out-of-road ends the episode early
        om.step(act)
        oa.step(act[:1])
        # the short scene behaves exactly as in a batch of its own
        for k in SC_KEYS:
            a, b = np.asarray(om.state[k]), np.asarray(oa.state[k])
            if k in ("obs", "reward", "cost", "step_info"):     # per agent
                assert a[0].tobytes() == b[0].tobytes(), (k, t)
            else:                                                # per mover slot
                xm = a.reshape(2, cap_m, -1)[0] if a.dtype.names is None else a.reshape(2, cap_m)[0]
                xa = b.reshape(1, cap_a, -1)[0] if b.dtype.names is None else b.reshape(1, cap_a)[0]
                assert xm.tobytes() == xa.tobytes(), (k, t)
        fl = om.state["flags"].reshape(2, cap_m)[:, 0]
        for e in range(2):
            if max_step_at[e] is None and (fl[e] & abi.FL_MAX_STEP):
                max_step_at[e] = t
        if t == 70:   # the short scene's data are over: no replayed mover left in it, the long scene still replays
            nav = om.state["nav"].reshape(2, cap_m)
            kinds = om.state["shape"].reshape(2, cap_m)["flags"] & abi.KIND_MASK
            rep = nav["ck0"] == abi.SC_REPLAY
            assert np.isin(kinds[0][rep[0]], [abi.KIND_CONE, abi.KIND_BARRIER]).all()
            assert (~np.isin(kinds[1][rep[1]], [abi.KIND_CONE, abi.KIND_BARRIER])).any()
    assert max_step_at[0] == 60 + 15 and max_step_at[1] is None


@pytest.mark.gpu
def test_mixed_length_batch_gpu_parity(tmp_path):
    """scenario_step_kernel == oracle on a batch read from a dataset directory whose scenes have three different lengths,
    with `allowed_more_steps` (each scene counts from its own length) and auto reset."""
    import torch
    from helpers import assert_state_equal
    from metadrive_ped_amd.engine import BatchedEngine
    kw = dict(n_vehicles=10, n_parked=2, n_pedestrians=1, n_cones=2)
    scs = [synthetic_scenario(20 + i, T=(70, 100, 130)[i % 3], **kw) for i in range(6)]
    sd.write_dataset(str(tmp_path), scs)
    E = 12
    cfg = make_scenario_config(dict(num_envs=E, data_directory=str(tmp_path), num_scenarios=6, reactive_traffic=True, horizon=0,
                                    allowed_more_steps=10, truncate_as_terminate=True, auto_reset=True))
    host = ScenarioHostScene(cfg, sd.load_scenarios(cfg))
    assert host.T == 130
    eng = BatchedEngine(cfg, host=host)
    o = _oracle(host)
    eng.reset()
    o.reset()
    keys = SC_KEYS + ["need_reset", "next_agent_id", "route_n", "route_segs", "route_verts", "route_aux"]
    saw_max_step = np.zeros(E, bool)
    for t in range(180):
        a = np.zeros((E, 1, 2), np.float32)
        a[:, 0, 1] = -1.0 if t % 90 < 60 else 0.3
        eng.step(torch.from_numpy(a).to(eng.device))
        o.step(a)
        saw_max_step |= (o.state["flags"].reshape(E, host.cap)[:, 0] & abi.FL_MAX_STEP) != 0
        if t % 15 == 0 or t > 170:
            assert_state_equal(eng.download_state(), o.state, keys=keys, where="mixed lengths step %d" % t)
    assert saw_max_step[[0, 1, 3, 4, 6, 7, 9, 10]].all()      # the 70- and 100-frame scenes ran past their length + 10

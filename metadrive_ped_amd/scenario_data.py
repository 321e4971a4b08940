"""ScenarioNet dataset directories: what the reference's ScenarioDataManager reads (manager/scenario_data_manager.py:11-60
through scenario/utils.py:324-395) -- `dataset_summary.pkl` (scenario file name -> its metadata, in dataset order),
`dataset_mapping.pkl` (file name -> sub-folder) and one pickled ScenarioDescription dict per scenario (`sd_*.pkl`, or the old
`0.pkl, 1.pkl, ...` layout without a summary).  `load_scenarios` returns the description dicts `BatchedScenarioEnv` takes,
scene e of the batch = scenario `start_scenario_index + (env_seed_offset + e) % num_scenarios` of the dataset;
`write_dataset` writes such a directory (e.g. from `BatchedMetaDriveEnv.export_scenarios()`), readable by the reference.

The format is pickle -- a file of it can run code when it is read.  Read only datasets you would also hand to the reference.
"""
import os
import pickle

SUMMARY_FILE = "dataset_summary.pkl"   # ScenarioDescription.DATASET (scenario/scenario_description.py:195-197)
MAPPING_FILE = "dataset_mapping.pkl"
_REQUIRED = ("tracks", "dynamic_map_states", "metadata", "map_features")   # ScenarioDescription.sanity_check's top level


def is_scenario_file(name):
    """`sd_*.pkl`, or all digits + `.pkl` (scenario_description.py:382-396)."""
    name = os.path.basename(str(name))
    if not name.endswith(".pkl"):
        return False
    stem = name[:-len(".pkl")]
    return stem[:3] == "sd_" or (len(stem) > 0 and all(ch.isdigit() for ch in stem))


def read_scenario_data(path):
    if not is_scenario_file(path):
        raise ValueError("File: {} is not scenario file".format(path))
    with open(path, "rb") as f:
        data = pickle.load(f)
    data = dict(data)
    missing = [k for k in _REQUIRED if k not in data]
    if missing:
        raise KeyError("scenario {} lacks {}".format(path, missing))
    return data


def read_dataset_summary(folder, check_file_existence=True):
    """-> (summary: file name -> metadata, file names in dataset order, mapping: file name -> sub-folder)"""
    folder = str(folder)
    if not os.path.isdir(folder):
        raise FileNotFoundError("data_directory {!r} does not exist".format(folder))
    summary_path = os.path.join(folder, SUMMARY_FILE)
    if os.path.isfile(summary_path):
        with open(summary_path, "rb") as f:
            summary = dict(pickle.load(f))
    else:   # the old layout: every scenario file of the folder, by number where the names are numbers
        names = [n for n in os.listdir(folder) if is_scenario_file(n)]
        try:
            names.sort(key=lambda n: int(n[:-len(".pkl")]))
        except ValueError:
            names.sort(key=lambda n: n[:-len(".pkl")])
        summary = {n: read_scenario_data(os.path.join(folder, n))["metadata"] for n in names}
    mapping = None
    mapping_path = os.path.join(folder, MAPPING_FILE)
    if os.path.exists(mapping_path):
        with open(mapping_path, "rb") as f:
            mapping = pickle.load(f)
    if not mapping:
        mapping = {k: "" for k in summary}
    if check_file_existence:
        for name in summary:
            if name not in mapping:
                raise KeyError("FileName in mapping mismatch with summary: {}".format(name))
            if not is_scenario_file(name):
                raise ValueError("File:{} is not sd scenario file".format(name))
            p = os.path.join(folder, mapping[name], name)
            if not os.path.exists(p):
                raise FileNotFoundError("Can not find file: {}".format(p))
    return summary, list(summary.keys()), mapping


def get_number_of_scenarios(folder):
    return len(read_dataset_summary(folder)[1])


def scenario_indices(cfg, num_envs=None):
    """dataset index of every scene of this shard's batch (ScenarioDataManager.available_scenario_indices walked by
    seed: scene e <-> seed start_scenario_index + (env_seed_offset + e) % num_scenarios)"""
    n = int(cfg["num_scenarios"])
    E = int(cfg["num_envs"] if num_envs is None else num_envs)
    off = int(cfg.get("env_seed_offset", 0))
    return [int(cfg["start_scenario_index"]) + ((off + e) % max(1, n)) for e in range(E)]


def load_scenarios(cfg, num_envs=None):
    """The description dicts of this batch from cfg['data_directory'] (each distinct scenario is read once)."""
    folder = cfg["data_directory"]
    summary, names, mapping = read_dataset_summary(folder, check_file_existence=False)
    start, n = int(cfg["start_scenario_index"]), int(cfg["num_scenarios"])
    if not start < len(names):
        raise ValueError("Insufficient scenarios!")
    if start + n > len(names):
        raise ValueError("Insufficient scenarios! Need: {} Has: {}".format(n, len(names) - start))
    cache = {}
    out = []
    for i in scenario_indices(cfg, num_envs):
        if i not in cache:
            name = names[i]
            p = os.path.join(folder, mapping[name], name)
            if not os.path.exists(p):
                raise FileNotFoundError("No Data at path: {}".format(p))
            cache[i] = read_scenario_data(p)
        out.append(cache[i])
    return out


def write_dataset(folder, scenarios, dataset_name="mdamd"):
    """One `sd_<dataset>_<id>.pkl` per description + summary + mapping; returns the file names in dataset order."""
    os.makedirs(folder, exist_ok=True)
    summary, mapping = {}, {}
    for k, sc in enumerate(scenarios):
        sid = str(sc.get("id", k))
        name = "sd_{}_{}.pkl".format(dataset_name, sid)
        if name in summary:
            name = "sd_{}_{}_{}.pkl".format(dataset_name, sid, k)
        with open(os.path.join(folder, name), "wb") as f:
            pickle.dump(dict(sc), f)
        summary[name] = sc["metadata"]
        mapping[name] = ""
    with open(os.path.join(folder, SUMMARY_FILE), "wb") as f:
        pickle.dump(summary, f)
    with open(os.path.join(folder, MAPPING_FILE), "wb") as f:
        pickle.dump(mapping, f)
    return list(summary.keys())

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


def _unpickle(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def read_scenario_data(path):
    """One scenario description as a plain dict; refuses files that are not named like scenario files or lack a top-level part."""
    if not is_scenario_file(path):
        raise ValueError("File: {} is not scenario file".format(path))
    data = dict(_unpickle(path))
    absent = [k for k in _REQUIRED if k not in data]
    if absent:
        raise KeyError("scenario {} lacks {}".format(path, absent))
    return data


class DatasetIndex:
    """What a dataset folder holds, resolved once: the scenario files in dataset order, the metadata of each, and where each file
    lives.  Two layouts exist (scenario/utils.py:342-392): a `dataset_summary.pkl` (+ optional `dataset_mapping.pkl` of
    sub-folders), or bare `0.pkl, 1.pkl, ...` / `sd_*.pkl` files, whose order is numeric where the names are numbers."""
    def __init__(self, folder):
        self.folder = str(folder)
        if not os.path.isdir(self.folder):
            raise FileNotFoundError("data_directory {!r} does not exist".format(self.folder))
        self.summary = self._summary()
        self.names = list(self.summary)
        self.subfolder = self._subfolders()

    def _summary(self):
        listed = os.path.join(self.folder, SUMMARY_FILE)
        if os.path.isfile(listed):
            return dict(_unpickle(listed))
        stems = {n: n[:-len(".pkl")] for n in os.listdir(self.folder) if is_scenario_file(n)}
        numeric = all(st.isdigit() for st in stems.values())
        order = sorted(stems, key=(lambda n: int(stems[n])) if numeric else (lambda n: stems[n]))
        return {n: read_scenario_data(os.path.join(self.folder, n))["metadata"] for n in order}

    def _subfolders(self):
        listed = os.path.join(self.folder, MAPPING_FILE)
        found = _unpickle(listed) if os.path.exists(listed) else None
        return found if found else dict.fromkeys(self.summary, "")

    def path_of(self, name):
        return os.path.join(self.folder, self.subfolder[name], name)

    def verify(self):
        """every listed scenario has a sub-folder entry, a scenario-file name and a file"""
        for name in self.names:
            if name not in self.subfolder:
                raise KeyError("FileName in mapping mismatch with summary: {}".format(name))
            if not is_scenario_file(name):
                raise ValueError("File:{} is not sd scenario file".format(name))
            if not os.path.exists(self.path_of(name)):
                raise FileNotFoundError("Can not find file: {}".format(self.path_of(name)))
        return self


def read_dataset_summary(folder, check_file_existence=True):
    """-> (summary: file name -> metadata, file names in dataset order, mapping: file name -> sub-folder): the triple the
    reference's function of this name returns (scenario/utils.py:342-392)"""
    index = DatasetIndex(folder)
    if check_file_existence:
        index.verify()
    return index.summary, index.names, index.subfolder


def get_number_of_scenarios(folder):
    return len(DatasetIndex(folder).verify().names)


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

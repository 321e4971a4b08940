"""BatchedScenarioEnv: the reset()/step() surface of the reference's ScenarioEnv (envs/scenario_env.py) for E
lock-stepped scenes on one MI355X.  Scene e replays scenario description e: read from `data_directory` (a ScenarioNet
dataset folder, as the reference's ScenarioDataManager reads it: metadrive_ped_amd/scenario_data.py), or handed in as dicts
of the same format (e.g. from BatchedMetaDriveEnv.export_scenarios()), or metadrive_ped_amd.scenario.synthetic_scenarios().

Returns like the single-agent env: obs [E, obs_dim] (side cloud | state | 22 navigation dims | lidar), reward [E],
terminated [E], truncated [E], info = dict of [E] tensors with ScenarioEnv's keys (route_completion, cost, crash_*,
out_of_road, arrive_dest, max_step, ...)."""
import numpy as np

from metadrive_ped_amd import abi
from metadrive_ped_amd.envs.spaces import Box, LazyInfo
from metadrive_ped_amd.scenario import ScenarioHostScene, make_scenario_config, synthetic_scenarios


def scenario_bench_config(common):
    """bench.py --workload scenario: reactive traffic, 240-beam lidar, 200-frame synthetic scenes"""
    cfg = make_scenario_config(dict(common, reactive_traffic=True, horizon=400,
                                    vehicle_config=dict(lidar=dict(num_lasers=240, distance=50))))
    return cfg


class BatchedScenarioEnv:
    metadata = {"render_modes": []}

    @classmethod
    def default_config(cls):
        return make_scenario_config({})

    def __init__(self, config=None, scenarios=None):
        self.config = make_scenario_config(config)
        self.num_envs = self.config["num_envs"]
        if scenarios is None and self.config["data_directory"] is not None:
            from metadrive_ped_amd.scenario_data import load_scenarios
            scenarios = load_scenarios(self.config)   # scene e = scenario start_scenario_index + (offset + e) % num_scenarios
        if scenarios is None:
            scenarios = synthetic_scenarios(self.num_envs, self.config["start_scenario_index"] + self.config["env_seed_offset"])
        self.scenarios = scenarios
        self.host = None
        self.engine = None
        self.action_space = Box(-1.0, 1.0, (2, ), np.float32)
        vc = self.config["vehicle_config"]
        n = vc["lidar"]["num_lasers"] if vc["lidar"]["distance"] > 0 else 0
        n_s = vc["side_detector"]["num_lasers"] if vc["side_detector"]["distance"] > 0 else 0
        n_l = vc["lane_line_detector"]["num_lasers"] if vc["lane_line_detector"]["distance"] > 0 else 0
        self._obs_dim = (n_s or 2) + 6 + (n_l or 1) + 22 + n
        self.observation_space = Box(-0.0, 1.0, (self._obs_dim, ), np.float32)

    def lazy_init(self, host=None):
        if self.engine is None:
            from metadrive_ped_amd.engine import BatchedEngine
            self.host = host or ScenarioHostScene(self.config, self.scenarios)
            self.engine = BatchedEngine(self.config, host=self.host)

    def reset(self, seed=None):
        self.lazy_init()
        self.engine.reset()
        return self.engine.obs[:, 0, :], self._info()

    def step(self, actions):
        if self.engine is None:
            raise RuntimeError("call reset() before step()")
        torch = self.engine.torch
        if actions is None and self.config["agent_policy"] == "ReplayEgoCarPolicy":
            actions = np.zeros((self.num_envs, 2), np.float32)      # the agent replays the SDC track: actions are ignored
        a = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions, dtype=np.float32))
        if a.dim() == 1:
            a = a.unsqueeze(0).expand(self.num_envs, 2)
        if tuple(a.shape) != (self.num_envs, 2):
            raise ValueError("actions must have shape [{}, 2], got {}".format(self.num_envs, tuple(a.shape)))
        self.engine.step(a)
        fl = self.engine.flags[:, 0]
        if self.engine.done_tt is not None:      # (terminated, truncated) written by md_step itself
            return self.engine.obs[:, 0, :], self.engine.reward[:, 0], self.engine.done_tt[:, 0, 0], self.engine.done_tt[:, 0, 1], \
                self._info()
        return self.engine.obs[:, 0, :], self.engine.reward[:, 0], (fl & abi.FL_TERMINATED) != 0, (fl & abi.FL_TRUNCATED) != 0, \
            self._info()

    # -- checkpoints (envs/base_env.py:775-836 get_state / set_state through the managers): a dict of numpy arrays; the routes the
    #    device cut at later spawn frames are state too and travel with it -----------------------------------------------------
    def get_state(self):
        if self.engine is None:
            raise RuntimeError("call reset() before get_state()")
        st = self.engine.download_state()
        st["__seeds__"] = np.asarray(self.engine.host.seeds, dtype=np.int64)
        st["__scenario_ids__"] = np.asarray(self.engine.host.scenario_ids)
        st["__abi__"] = np.asarray([abi.MD_ABI_VERSION], dtype=np.int64)
        return st

    def set_state(self, state):
        if self.engine is None:
            raise RuntimeError("call reset() before set_state()")
        if "__abi__" in state and int(np.asarray(state["__abi__"])[0]) != abi.MD_ABI_VERSION:
            raise ValueError("the checkpoint was written by ABI v{}, this library is v{}: the record layouts differ".format(
                int(np.asarray(state["__abi__"])[0]), abi.MD_ABI_VERSION))
        if np.asarray(state["__seeds__"]).tolist() != list(self.engine.host.seeds):
            raise ValueError("the checkpoint was taken with another scenario assignment (start_scenario_index / num_scenarios / "
                             "env_seed_offset differ): tracks and routes would not match")
        if "__scenario_ids__" in state and [str(x) for x in np.asarray(state["__scenario_ids__"]).tolist()] != list(self.engine.host.scenario_ids):
            raise ValueError("the checkpoint was taken on other scenarios (their ids differ): tracks and routes would not match")
        arrays = {k: v for k, v in state.items() if not k.startswith("__")}
        ref = self.engine.host.state
        for k, v in arrays.items():
            if k not in ref or np.asarray(v).nbytes != ref[k].nbytes:
                raise ValueError("checkpoint array {!r} does not fit this batch".format(k))
        self.engine.upload_state(arrays)

    def _info(self):
        e = self.engine
        fl = e.flags[:, 0]
        si = e.step_info[:, 0, :]
        bit = lambda m: (lambda: (fl & m) != 0)
        eager = {"velocity": si[:, 1], "step_energy": si[:, 2], "episode_energy": si[:, 3], "step_reward": si[:, 0],
                 "episode_reward": si[:, 4], "episode_length": e.nav_i[:, 0, 8], "cost": e.cost[:, 0], "total_cost": si[:, 5],
                 "route_completion": si[:, 6], "action": e.action[:, 0, :], "raw_action": e.action[:, 0, :]}
        lazy = {"crash_vehicle": bit(abi.FL_CRASH_VEHICLE), "crash_object": bit(abi.FL_CRASH_OBJECT),
                "crash_human": bit(abi.FL_CRASH_HUMAN), "crash_building": bit(abi.FL_CRASH_BUILDING),
                "crash_sidewalk": bit(abi.FL_CRASH_SIDEWALK), "out_of_road": bit(abi.FL_OUT_OF_ROAD),
                "arrive_dest": bit(abi.FL_ARRIVE_DEST), "max_step": bit(abi.FL_MAX_STEP),
                "crash": bit(abi.FL_CRASH_VEHICLE | abi.FL_CRASH_OBJECT | abi.FL_CRASH_BUILDING | abi.FL_CRASH_SIDEWALK |
                             abi.FL_CRASH_HUMAN)}
        return LazyInfo(eager, lazy)

    def close(self):
        self.engine = None

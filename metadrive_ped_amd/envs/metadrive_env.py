"""BatchedMetaDriveEnv: the reset()/step() surface of the reference's MetaDriveEnv for E lock-stepped
environments on one MI355X.

Mirrors (same names, argument meaning, return structure and error behaviour, with a leading env
dimension): BaseEnv.reset / step / close / observation_space / action_space
(metadrive/envs/base_env.py:269,426-431,502-537,678-700), MetaDriveEnv.default_config
(envs/metadrive_env.py:16-99).  Single-agent returns are unwrapped like the reference's
(_wrap_as_single_agent, base_env.py:618-623): obs [E, 259], reward [E], terminated [E], truncated [E],
info = dict of [E] tensors with the reference's info keys (base_vehicle.py:243-252,
metadrive_env.py:132-152,203-211,269; base_env.py:614-616).

Everything returned is a torch tensor on the engine's device (views of the engine's buffers: copy
them if you keep them across steps).  With config auto_reset=True an env that terminated or was
truncated at step t is restored from its reset snapshot during step t+1, which then returns the
reset observation with reward 0 (gymnasium's NEXT_STEP autoreset convention).
"""
import numpy as np

from metadrive_ped_amd import abi
from metadrive_ped_amd.config import make_config
from metadrive_ped_amd.envs.spaces import Box, LazyInfo, Discrete, MultiDiscrete


def make_action_space(cfg):
    """EnvInputPolicy.get_input_space (policy/env_input_policy.py:50-69)."""
    if not cfg["discrete_action"]:
        return Box(-1.0, 1.0, (2, ), np.float32)
    if cfg["use_multi_discrete"]:
        return MultiDiscrete([cfg["discrete_steering_dim"], cfg["discrete_throttle_dim"]])
    return Discrete(cfg["discrete_steering_dim"] * cfg["discrete_throttle_dim"])


def discrete_to_continuous(torch, cfg, actions, lead_shape, device):
    """EnvInputPolicy.convert_to_continuous_action (policy/env_input_policy.py:40-48) for a batch: Discrete
    index -> (index % steering_dim, index // steering_dim), MultiDiscrete -> (a[0], a[1]); each grid index i
    maps to i * 2/(dim-1) - 1.  Returns float32 [*lead_shape, 2] on `device`."""
    sd, td = int(cfg["discrete_steering_dim"]), int(cfg["discrete_throttle_dim"])
    a = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions))
    if a.is_floating_point():
        raise TypeError("discrete_action=True expects integer actions, got dtype {}".format(a.dtype))
    a = a.to(device)
    if cfg["use_multi_discrete"]:
        if tuple(a.shape) == (2, ):
            a = a.expand(*lead_shape, 2)
        if tuple(a.shape) != tuple(lead_shape) + (2, ):
            raise ValueError("actions must have shape {}, got {}".format(tuple(lead_shape) + (2, ), tuple(a.shape)))
        si, ti = a[..., 0], a[..., 1]
    else:
        if a.dim() == 0:
            a = a.expand(*lead_shape)
        if tuple(a.shape) != tuple(lead_shape):
            raise ValueError("actions must have shape {}, got {}".format(tuple(lead_shape), tuple(a.shape)))
        si, ti = a % sd, torch.div(a, sd, rounding_mode="floor")
    if cfg["action_check"]:
        ok = (si >= 0) & (si < sd) & (ti >= 0) & (ti < td)
        assert bool(ok.all()), "Input is not compatible with action space {}!".format(make_action_space(cfg))
    steering = si.to(torch.float32) * (2.0 / (sd - 1)) - 1.0
    throttle = ti.to(torch.float32) * (2.0 / (td - 1)) - 1.0
    return torch.stack([steering, throttle], dim=-1)


class BatchedMetaDriveEnv:
    metadata = {"render_modes": []}

    @classmethod
    def default_config(cls):
        return make_config({})

    def __init__(self, config=None):
        self.config = make_config(config)
        if self.config["num_agents"] != 1 or self.config["is_multi_agent"]:
            raise NotImplementedError("BatchedMetaDriveEnv is the single-agent env; multi-agent envs are separate classes")
        self.num_envs = self.config["num_envs"]
        self.engine = None
        if self.config["random_traffic"] and self.config["auto_reset"] and int(self.config.get("traffic_draws", 1)) <= 1:
            # the reference draws other traffic in EVERY episode (traffic_manager.py:335-337: the stream is not re-seeded at
            # reset); here a new draw happens at an explicit reset() only, and episodes that auto-reset restore the latest one
            import warnings
            warnings.warn("random_traffic=True with auto_reset=True and traffic_draws=1: traffic is re-drawn by env.reset() only; episodes that "
                          "auto-reset in between replay the latest draw (call reset() between episodes, or set auto_reset=False, "
                          "for a new draw per episode)", stacklevel=2)
        lidar = self.config["vehicle_config"]["lidar"]
        n = lidar["num_lasers"] if lidar["distance"] > 0 else 0
        vc = self.config["vehicle_config"]
        n_s = vc["side_detector"]["num_lasers"] if vc["side_detector"]["distance"] > 0 else 0
        n_l = vc["lane_line_detector"]["num_lasers"] if vc["lane_line_detector"]["distance"] > 0 else 0
        n_o = lidar["num_others"] * (8 if lidar["add_others_navi"] else 4) if n > 0 else 0
        self._obs_dim = (2 if self.config["random_agent_model"] else 0) + (n_s or 2) + 6 + (n_l or 1) + 10 + n_o + n
        self.observation_space = Box(-0.0, 1.0, (self._obs_dim, ), np.float32)
        self.action_space = make_action_space(self.config)
        self.start_seed = self.config["start_seed"]
        self.episode_rewards = None

    # -- lifecycle ----------------------------------------------------------------------------
    def lazy_init(self, host=None):
        """`host`: a HostScene already built from this env's config (e.g. before the GPU was touched)."""
        if self.engine is None:
            from metadrive_ped_amd.engine import BatchedEngine
            self.engine = BatchedEngine(self.config, host=host)

    def reset(self, seed=None):
        """seed: None keeps the scenario assignment; an int re-bases it (env e gets scenario
        seed + (env_seed_offset + e) % num_scenarios), regenerating maps/traffic on the host."""
        if seed is not None:
            if not (isinstance(seed, (int, np.integer)) and seed >= 0):
                raise ValueError("seed must be a non-negative int, got {!r}".format(seed))
            if self.engine is not None and int(seed) != self.config["start_seed"]:
                self.config["start_seed"] = int(seed)
                self.engine.host = None
                self.engine.cfg = self.config
                self.engine.build()
            self.config["start_seed"] = int(seed)
        if self.config["random_traffic"] and self.engine is not None:
            # new traffic for the coming episodes (PGTrafficManager with random_traffic: the stream is not re-seeded at
            # reset); episodes that auto-reset in between restart from the latest draw
            self.config["traffic_epoch"] = int(self.config.get("traffic_epoch", 0)) + 1
            self.engine.host = None
            self.engine.cfg = self.config
            self.engine.build()
        self.lazy_init()
        self.engine.reset()
        return self._obs(), self._info()

    def step(self, actions):
        if self.engine is None:
            raise RuntimeError("call reset() before step()")
        torch = self.engine.torch
        a = actions
        if self.config["agent_policy"] == "IDMPolicy":     # the agents drive themselves; `actions` is ignored
            self.engine.step(None)
            terminated, truncated = self._done_flags()
            return self._obs(), self.engine.reward[:, 0], terminated, truncated, self._info()
        if self.config["discrete_action"]:
            a = discrete_to_continuous(torch, self.config, a, (self.num_envs, ), self.engine.device)
        else:
            if not torch.is_tensor(a):
                a = torch.as_tensor(np.asarray(a, dtype=np.float32))
            if a.dim() == 1:
                a = a.unsqueeze(0).expand(self.num_envs, 2)
        if tuple(a.shape) != (self.num_envs, 2):
            raise ValueError("actions must have shape [{}, 2], got {}".format(self.num_envs, tuple(a.shape)))
        self.engine.step(a)
        terminated, truncated = self._done_flags()
        return self._obs(), self.engine.reward[:, 0], terminated, truncated, self._info()

    def _done_flags(self):
        """(terminated, truncated) as two views of ONE [E, 2] bool tensor: every extra device op of an eager loop costs
        about as much as a tenth of the step itself on this GPU, so the two bits are tested in one go."""
        e = self.engine
        if e.done_tt is not None:
            # written by md_step itself (MdState.done_out): no device op at all.  Like obs and reward these are views of
            # the engine's buffers: the next step() overwrites them -- .clone() what has to outlive it
            return e.done_tt[:, 0, 0], e.done_tt[:, 0, 1]
        if getattr(self, "_tt_mask", None) is None or self._tt_mask.device != e.device:
            self._tt_mask = e.torch.tensor([abi.FL_TERMINATED, abi.FL_TRUNCATED], dtype=e.flags.dtype, device=e.device)
        tt = (e.flags[:, 0:1] & self._tt_mask) != 0
        return tt[:, 0], tt[:, 1]

    def close(self):
        self.engine = None

    # -- record / replay of the traffic (RecordManager / ReplayManager / ReplayTrafficParticipantPolicy) ------------
    def start_recording(self, max_steps):
        """Right after reset(): keep the pose of every mover for the next `max_steps` steps (device memory)."""
        self.engine.start_recording(max_steps)

    def stop_recording(self):
        return self.engine.stop_recording()

    # -- traffic participants spawned by the user (engine.spawn_object(Pedestrian, ...) of the reference) -------------
    def spawn_object(self, kind, position, heading_theta=0.0, envs=None):
        """kind "pedestrian" | "cyclist" at `position` (one [x, y] or one per chosen env) -> handle.  It is hit by
        lidar beams, crashing into it sets crash_human, it moves with the velocity given by set_velocity and it is
        gone when its env resets."""
        return self.engine.spawn_object(kind, position, heading_theta, envs)

    def set_velocity(self, handle, direction, value=None, in_local_frame=False, envs=None):
        self.engine.set_velocity(handle, direction, value, in_local_frame, envs)

    def clear_objects(self, handles, envs=None):
        self.engine.clear_objects(list(handles), envs)

    def export_scenarios(self, tracks, envs=None):
        """BaseEnv.export_scenarios (envs/base_env.py:775-836) for a recorded batch: one scenario description (the
        reference's unified dict format, see scenario_export.py) per env of `envs` from stop_recording()'s tracks."""
        from metadrive_ped_amd.scenario_export import tracks_to_scenarios
        return tracks_to_scenarios(tracks, self.engine.host, envs)

    def load_scenarios(self, scenarios):
        """traffic_mode='replay' from scenario descriptions written by export_scenarios() (one per env, same scenario
        seeds): ScenarioEnv-style replay of data recorded here."""
        from metadrive_ped_amd.scenario_export import scenarios_to_tracks
        self.lazy_init()
        tracks = scenarios_to_tracks(scenarios, self.engine.host)
        torch = self.engine.torch
        self.load_tracks(dict(shape=torch.from_numpy(tracks["shape"].view(np.uint8).reshape(tracks["shape"].shape[0], -1)),
                              dyn=torch.from_numpy(tracks["dyn"]), seeds=tracks["seeds"], cap=tracks["cap"]))

    def load_tracks(self, tracks):
        """For an env built with traffic_mode='replay': the traffic follows `tracks` (from stop_recording() of an env
        with the same scenarios) instead of reacting; call before reset()."""
        if self.config["traffic_mode"] != "replay":
            raise ValueError("load_tracks needs config traffic_mode='replay'")
        self.lazy_init()
        self.engine.set_tracks(tracks)

    # -- state checkpoint (the role of BaseEngine/BaseManager get_state / set_state, manager/base_manager.py:116-135,
    #    and of BaseVehicle.get_state / set_state, component/vehicle/base_vehicle.py:808-846: everything that
    #    evolves is already a flat array here, so a checkpoint is a dict of numpy arrays) -----------------------
    def get_state(self):
        """Every evolving array of the batch (poses, dynamics, navigation, PID, flags, obs, RNG ...) as host numpy
        arrays, plus the scenario assignment.  set_state() of the result resumes bit-identically."""
        if self.engine is None:
            raise RuntimeError("call reset() before get_state()")
        st = self.engine.download_state()
        st["__seeds__"] = np.asarray(self.engine.host.seeds, dtype=np.int64)
        if getattr(self.engine, "_staged", None) is not None:       # random_traffic: which staged draw every env is on
            st["__draw_idx__"] = self.engine.draw_idx.cpu().numpy().copy()
        return st

    def set_state(self, state):
        if self.engine is None:
            raise RuntimeError("call reset() before set_state()")
        state = dict(state)
        draw_idx = state.pop("__draw_idx__", None)
        if draw_idx is not None and getattr(self.engine, "_staged", None) is not None:
            self.engine.draw_idx.copy_(self.engine.torch.from_numpy(np.asarray(draw_idx, dtype=np.int32)))
        seeds = np.asarray(state["__seeds__"])
        if seeds.tolist() != list(self.engine.host.seeds):
            raise ValueError("the checkpoint was taken with another scenario assignment (start_seed / num_scenarios / "
                             "env_seed_offset differ): maps and routes would not match")
        arrays = {k: v for k, v in state.items() if k != "__seeds__"}
        ref = self.engine.host.state
        for k, v in arrays.items():
            if k not in ref or np.asarray(v).nbytes != ref[k].nbytes:
                raise ValueError("checkpoint array {!r} does not fit this batch".format(k))
        self.engine.upload_state(arrays)

    # -- helpers --------------------------------------------------------------------------------
    def _obs(self):
        return self.engine.obs[:, 0, :]

    def _info(self):
        e = self.engine
        fl = e.flags[:, 0]
        si = e.step_info[:, 0, :]
        bit = lambda m: (lambda: (fl & m) != 0)
        eager = {
            "velocity": si[:, 1], "steering": e.dyn_f[:, 0, 2], "acceleration": e.dyn_f[:, 0, 3],
            "step_energy": si[:, 2], "episode_energy": si[:, 3], "step_reward": si[:, 0], "episode_reward": si[:, 4],
            "episode_length": e.nav_i[:, 0, 8], "cost": e.cost[:, 0], "total_cost": si[:, 5],
            "raw_action": e.action[:, 0, :], "action": e.action[:, 0, :],
        }
        lazy = {
            "crash_vehicle": bit(abi.FL_CRASH_VEHICLE), "crash_object": bit(abi.FL_CRASH_OBJECT),
            "crash_building": bit(abi.FL_CRASH_BUILDING), "crash_human": bit(abi.FL_CRASH_HUMAN),
            "crash_sidewalk": bit(abi.FL_CRASH_SIDEWALK), "out_of_road": bit(abi.FL_OUT_OF_ROAD),
            "arrive_dest": bit(abi.FL_ARRIVE_DEST), "max_step": bit(abi.FL_MAX_STEP),
            "on_lane": bit(abi.FL_ON_LANE), "on_broken_line": bit(abi.FL_ON_BROKEN),
            "crash": bit(abi.FL_CRASH_VEHICLE | abi.FL_CRASH_OBJECT | abi.FL_CRASH_BUILDING | abi.FL_CRASH_SIDEWALK |
                         abi.FL_CRASH_HUMAN),
            "env_seed": self._env_seed_tensor,
        }
        return LazyInfo(eager, lazy)

    def _env_seed_tensor(self):
        e = self.engine
        key = tuple(e.host.seeds)
        if getattr(self, "_seed_cache", (None, None))[0] != key:
            self._seed_cache = (key, e.torch.as_tensor(np.asarray(e.host.seeds, dtype=np.int64), device=e.device))
        return self._seed_cache[1]

    # -- small parts of BaseEnv's surface that user loops touch (envs/base_env.py:618-700) --------------------------
    def seed(self, seed=None):
        """BaseEnv.seed: scenario seeds are set through reset(seed=...); kept as a no-op like the gymnasium API."""

    def render(self, *args, **kwargs):
        raise NotImplementedError("rendering lies outside this build (DESIGN.md section 1): export_scenarios() gives the "
                                  "episode in the reference's scenario format for its own top-down renderer")

    @property
    def current_seed(self):
        """The scenario seed of env 0 (BaseEnv.current_seed)."""
        return self.current_seeds[0]

    @property
    def episode_step(self):
        """[E] steps taken in the running episode of every env (BaseEnv.episode_step)."""
        return self.engine.nav_i[:, 0, 8]

    @property
    def num_scenarios(self):
        return self.config["num_scenarios"]

    @property
    def current_seeds(self):
        return list(self.engine.host.seeds)


class BatchedSafeMetaDriveEnv(BatchedMetaDriveEnv):
    """SafeMetaDriveEnv (metadrive/envs/safe_metadrive_env.py:7-35): accident scenes on the road (cones,
    broken-down vehicle + warning tripod, barrier), crashes cost but do not terminate, info["total_cost"]
    accumulates the episode cost."""
    SAFE_DEFAULTS = dict(num_scenarios=100, accident_prob=0.8, traffic_density=0.05, crash_vehicle_done=False,
                         crash_object_done=False)

    @classmethod
    def default_config(cls):
        return make_config(dict(cls.SAFE_DEFAULTS))

    def __init__(self, config=None):
        merged = dict(self.SAFE_DEFAULTS)
        merged.update(config or {})
        super().__init__(merged)


class BatchedVaryingDynamicsEnv(BatchedMetaDriveEnv):
    """VaryingDynamicsEnv (metadrive/envs/varying_dynamics_env.py:14-60): the agent's engine force, brake force,
    wheel friction, maximum steering angle and mass are drawn per scenario seed from config["random_dynamics"]
    ({parameter: (min, max) | None}); like there, the same scenario seed always gives the same dynamics, so use
    num_scenarios > 1 for a spread.  `dynamics_parameters()` is the batch form of agent.get_dynamics_parameters()."""
    VARYING_DEFAULTS = dict(
        vehicle_config=dict(vehicle_model="varying_dynamics"),
        random_dynamics=dict(max_engine_force=(100, 3000), max_brake_force=(20, 600), wheel_friction=(0.1, 2.5),
                             max_steering=(10, 80), mass=(300, 3000)))

    @classmethod
    def default_config(cls):
        import copy
        return make_config(copy.deepcopy(cls.VARYING_DEFAULTS))

    def __init__(self, config=None):
        import copy
        merged = copy.deepcopy(self.VARYING_DEFAULTS)
        for k, v in (config or {}).items():
            if k == "vehicle_config":
                merged["vehicle_config"].update(v)
            else:
                merged[k] = v
        super().__init__(merged)

    def dynamics_parameters(self):
        """-> list (one dict per env) of the agent's max_engine_force / max_brake_force / wheel_friction / max_steering
        / mass, as sampled for its scenario."""
        self.lazy_init()
        keys = ("max_engine_force", "max_brake_force", "wheel_friction", "max_steering", "mass")
        h = self.engine.host
        return [{k: h.scenes[s].vehicle_cfgs[0][k] for k in keys if k in h.scenes[s].vehicle_cfgs[0]} for s in h.seeds]

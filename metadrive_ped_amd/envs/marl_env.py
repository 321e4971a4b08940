"""BatchedMultiAgentRoundaboutEnv: MultiAgentRoundaboutEnv (envs/marl_envs/marl_inout_roundabout.py:144-153
over MultiAgentMetaDrive, envs/marl_envs/multi_agent_metadrive.py:65-212) for E lock-stepped environments.

The reference keys every per-agent value by a growing agent name ("agent0", ... "agent57"): a finished
agent leaves the dicts, a respawned vehicle joins under a new name.  The batched form keeps A = num_agents
fixed SLOTS per env; a slot is either active (holds a live agent), dying (its finished vehicle stays
on the road as a static body for `delay_done` steps) or free.  Tensors are [E, A, ...]:

    obs, reward, terminated, truncated, info = env.step(actions)        # actions [E, A, 2]
    info["active"]    [E, A] bool   slot holds a live agent THIS step (only these rows are meaningful); computed when first read, like every derived entry
    info["agent_id"]  [E, A] int    the k of the reference's "agent{k}" currently in the slot
    info["spawned"]   [E, A] bool   slot was (re)filled at the start of this step: obs is its first obs
    terminated/truncated            per slot; `terminated_all` / `truncated_all` [E] play the role of "__all__"

`to_dicts(e, ...)` rebuilds the reference's dict-of-agents view of one env for drop-in code.
"""
import numpy as np

from metadrive_ped_amd import abi
from metadrive_ped_amd.config import make_config
from metadrive_ped_amd.envs.spaces import Box, LazyInfo

MULTI_AGENT_DEFAULTS = dict(
    is_multi_agent=True, num_agents=40, crash_done=True, out_of_road_done=True, delay_done=25, allow_respawn=True,
    horizon=1000, truncate_as_terminate=True, traffic_density=0.0, random_spawn_lane_index=False,
    out_of_road_penalty=10.0, crash_vehicle_penalty=10.0, crash_object_penalty=10.0, crash_vehicle_cost=1.0,
    crash_object_cost=1.0, out_of_road_cost=0.0, marl_map="roundabout",
    map_config=dict(exit_length=60, lane_num=2),
    vehicle_config=dict(vehicle_model="static_default", lidar=dict(num_lasers=72, distance=40, num_others=0)),
)


def _deep_update(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_update(dst[k], v)
        else:
            dst[k] = v
    return dst


class BatchedMultiAgentRoundaboutEnv:
    @classmethod
    def default_config(cls):
        import copy
        return make_config(copy.deepcopy(MULTI_AGENT_DEFAULTS))

    def __init__(self, config=None):
        import copy
        merged = _deep_update(copy.deepcopy(MULTI_AGENT_DEFAULTS), dict(config or {}))
        self.config = make_config(merged)
        self.num_envs = self.config["num_envs"]
        self.num_agents = self.config["num_agents"]
        lidar = self.config["vehicle_config"]["lidar"]
        n = lidar["num_lasers"] if lidar["distance"] > 0 else 0
        vc = self.config["vehicle_config"]
        n_s = vc["side_detector"]["num_lasers"] if vc["side_detector"]["distance"] > 0 else 0
        n_l = vc["lane_line_detector"]["num_lasers"] if vc["lane_line_detector"]["distance"] > 0 else 0
        n_o = lidar["num_others"] * (8 if lidar["add_others_navi"] else 4) if n > 0 else 0
        toll = self.config["marl_map"] == "tollgate"     # no navigation dims, two toll dims after the cloud
        base = 2 if self.config["random_agent_model"] else 0     # [length, width] lead the observation (state_obs.py:70-75)
        self.observation_space = Box(-0.0, 1.0, (base + (n_s or 2) + 6 + (n_l or 1) + (0 if toll else 10) + n_o + n + (2 if toll else 0), ),
                                     np.float32)
        from metadrive_ped_amd.envs.metadrive_env import make_action_space
        self.action_space = make_action_space(self.config)
        self.engine = None

    def reset(self, seed=None):
        if seed is not None:
            self.config["start_seed"] = int(seed)
            if self.engine is not None:
                self.engine.host = None
                self.engine.cfg = self.config
                self.engine.build()
        self.lazy_init()
        self.engine.reset()
        return self.engine.obs, self._info()

    def lazy_init(self, host=None):
        """`host`: a HostScene already built from this env's config (e.g. before the GPU was touched)."""
        if self.engine is None:
            from metadrive_ped_amd.engine import BatchedEngine
            self.engine = BatchedEngine(self.config, host=host)

    def step(self, actions):
        if self.engine is None:
            raise RuntimeError("call reset() before step()")
        torch = self.engine.torch
        if self.config["agent_policy"] == "IDMPolicy":      # every agent is driven by its own IDMPolicy: `actions` is ignored
            a = None
        elif self.config["discrete_action"]:
            from metadrive_ped_amd.envs.metadrive_env import discrete_to_continuous
            a = discrete_to_continuous(torch, self.config, actions, (self.num_envs, self.num_agents), self.engine.device)
        else:
            a = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions, dtype=np.float32))
        if a is not None and tuple(a.shape) != (self.num_envs, self.num_agents, 2):
            raise ValueError("actions must have shape [{}, {}, 2], got {}".format(self.num_envs, self.num_agents, tuple(a.shape)))
        self.engine.step(a)
        A = self.num_agents
        fl = self.engine.flags[:, :A]
        info = self._info()
        if self.engine.done_tt is not None:
            # written by md_step itself (MdState.done_out), zero for slots without a live agent: no device op here -- one md_step
            # launch per step and nothing else (every eager torch op costs ~5 us, 3 % of this step).
            # Views of the engine's buffers, like obs and reward: .clone() what has to outlive the next step()
            terminated, truncated = self.engine.done_tt[:, :A, 0], self.engine.done_tt[:, :A, 1]
        else:
            terminated = ((fl & abi.FL_TERMINATED) != 0) & info["active"]
            truncated = ((fl & abi.FL_TRUNCATED) != 0) & info["active"]
        info._lazy["terminated_all"] = lambda: (terminated | ~info["active"]).all(dim=1)
        info._lazy["truncated_all"] = lambda: (truncated | ~info["active"]).all(dim=1)
        return self.engine.obs, self.engine.reward, terminated, truncated, info

    def _info(self):
        e = self.engine
        A = self.num_agents
        sf = e.shape_f.view(e.torch.int32)[:, :A, 6]
        fl = e.flags[:, :A]
        bit = lambda m: (lambda: (fl & m) != 0)
        eager = {
            "agent_id": e.agent_id[:, :A],
            "velocity": e.step_info[:, :, 1], "step_reward": e.step_info[:, :, 0], "episode_reward": e.step_info[:, :, 4],
            "episode_length": e.nav_i[:, :A, 8], "cost": e.cost,
        }
        lazy = {
            "active": lambda: (sf & (abi.F_ALIVE | abi.F_STATIC)) == abi.F_ALIVE,
            "dying": lambda: (sf & (abi.F_ALIVE | abi.F_STATIC)) == (abi.F_ALIVE | abi.F_STATIC),
            "spawned": lambda: ((sf & (abi.F_ALIVE | abi.F_STATIC)) == abi.F_ALIVE) & (e.nav_i[:, :A, 8] == 0),
            "crash_vehicle": bit(abi.FL_CRASH_VEHICLE), "crash_object": bit(abi.FL_CRASH_OBJECT),
            "crash_sidewalk": bit(abi.FL_CRASH_SIDEWALK), "out_of_road": bit(abi.FL_OUT_OF_ROAD),
            "arrive_dest": bit(abi.FL_ARRIVE_DEST), "max_step": bit(abi.FL_MAX_STEP),
        }
        return LazyInfo(eager, lazy)

    # traffic participants (needs mover_capacity > num_agents: the agents' slots are never handed out)
    def spawn_object(self, kind, position, heading_theta=0.0, envs=None):
        return self.engine.spawn_object(kind, position, heading_theta, envs)

    def set_velocity(self, handle, direction, value=None, in_local_frame=False, envs=None):
        self.engine.set_velocity(handle, direction, value, in_local_frame, envs)

    def clear_objects(self, handles, envs=None):
        self.engine.clear_objects(list(handles), envs)

    def to_dicts(self, e, obs, reward, terminated, truncated, info):
        """The reference's per-agent dict view of env `e` (keys "agent{k}", plus "__all__")."""
        act = info["active"][e].cpu().numpy()
        ids = info["agent_id"][e].cpu().numpy()
        o, r, tm, tc = {}, {}, {}, {}
        for a in np.nonzero(act)[0]:
            k = "agent{}".format(int(ids[a]))
            o[k], r[k] = obs[e, a].cpu().numpy(), float(reward[e, a])
            tm[k], tc[k] = bool(terminated[e, a]), bool(truncated[e, a])
        tm["__all__"] = all(tm.values()) if tm else True
        tc["__all__"] = all(tc.values()) if tc else True
        return o, r, tm, tc

    def actions_from_dicts(self, dicts, info):
        """The way in for the reference's per-env action dicts: `dicts[e]` = {"agent{k}": [steer, throttle], ...} for env e
        (missing agents get [0, 0]) -> the [E, A, 2] tensor step() takes; `info` is the last info (its agent_id / active
        say which slot holds which agent)."""
        torch = self.engine.torch
        ids = info["agent_id"].cpu().numpy()
        act = info["active"].cpu().numpy()
        out = np.zeros((self.num_envs, self.num_agents, 2), np.float32)
        for e, d in enumerate(dicts):
            slot_of = {"agent{}".format(int(ids[e, a])): a for a in np.nonzero(act[e])[0]}
            for name, v in d.items():
                if name not in slot_of:
                    raise KeyError("env {}: no active agent named {!r}".format(e, name))
                out[e, slot_of[name]] = v
        return torch.from_numpy(out).to(self.engine.device)

    def seed(self, seed=None):
        """Scenario seeds are set through reset(seed=...); a no-op like the gymnasium API."""

    def render(self, *args, **kwargs):
        raise NotImplementedError("rendering lies outside this build (DESIGN.md section 1)")

    def close(self):
        self.engine = None


class BatchedMultiAgentIntersectionEnv(BatchedMultiAgentRoundaboutEnv):
    """MultiAgentIntersectionEnv (envs/marl_envs/marl_intersection.py:11-110): 30 agents on a 4-way intersection
    with U-turns, spawn roads = the four arms, destination = a random arm (its own included)."""
    MAP_DEFAULTS = dict(marl_map="intersection", num_agents=30, map_config=dict(exit_length=60, lane_num=2))

    @classmethod
    def default_config(cls):
        import copy
        return make_config(_deep_update(copy.deepcopy(MULTI_AGENT_DEFAULTS), copy.deepcopy(cls.MAP_DEFAULTS)))

    def __init__(self, config=None):
        import copy
        merged = _deep_update(copy.deepcopy(self.MAP_DEFAULTS), dict(config or {}))
        super().__init__(merged)


class BatchedMultiAgentTinyInter(BatchedMultiAgentIntersectionEnv):
    """MultiAgentTinyInter (envs/marl_envs/tinyinter.py:328-420): 8 agents on a ONE-lane 4-way intersection (lane width 4 m, arms
    30 m, no U-turns: a vehicle never gets the arm it came in by as its destination), success reward 10 and penalties 10, and
    finished vehicles leave the road at once (ignore_delay_done=True: MixedIDMAgentManager._finish, :262-268, i.e. no
    delay_done corpse).  `map_config["radius"]` sets the intersection's radius.  The env's two optional extras are not built and
    are refused by name: `num_RL_agents < num_agents` (the rest driven by TinyInterRuleBasedPolicy, :219-247) and
    `use_communication_obs=True` (CommunicationObservation, :14-216)."""
    MAP_DEFAULTS = dict(marl_map="intersection", num_agents=8, success_reward=10.0, out_of_road_penalty=10.0, crash_vehicle_penalty=10.0,
                        crash_object_penalty=10.0, map_config=dict(exit_length=30, lane_num=1, lane_width=4.0, radius=None))
    TINY_KEYS = dict(num_RL_agents=None, ignore_delay_done=True, target_speed=10, use_communication_obs=False)

    def __init__(self, config=None):
        config = dict(config or {})
        tiny = {k: config.pop(k, v) for k, v in self.TINY_KEYS.items()}
        merged = _deep_update(dict(num_agents=self.MAP_DEFAULTS["num_agents"]), config)
        n_rl = merged["num_agents"] if tiny["num_RL_agents"] is None else tiny["num_RL_agents"]
        if n_rl != merged["num_agents"]:
            raise NotImplementedError("MultiAgentTinyInter with num_RL_agents ({}) < num_agents ({}): the rule-based agents "
                                      "(TinyInterRuleBasedPolicy) are not built".format(n_rl, merged["num_agents"]))
        if tiny["use_communication_obs"]:
            raise NotImplementedError("MultiAgentTinyInter: use_communication_obs=True (CommunicationObservation) is not built")
        if tiny["ignore_delay_done"]:
            config["delay_done"] = 0          # the finished vehicle is removed in the step it finishes
        self.tiny_config = dict(tiny, num_RL_agents=n_rl)
        super().__init__(config)


class BatchedMultiAgentRacingEnv(BatchedMultiAgentRoundaboutEnv):
    """MultiAgentRacingEnv (envs/marl_envs/marl_racing_env.py:15-441): up to 12 agents on a hand-built ONE-WAY track (first block
    + 12 straights / curves, 2 lanes, guardrails on both edges), all starting on the first block, no respawn.  Rules of its own
    (MdConfig.ma_kind = 3, on the device): out of road = more than 5 m behind the start of the lane the vehicle is on (the
    guardrails keep it in otherwise); reward = progress + speed, success +20, out of road -5, vehicle crash, sidewalk (guardrail)
    crash -1, idle -1; an agent that moved less than 0.1 m along its lane over its last 100 steps is `idle` and done
    (idle_done); crashes do not end the episode; horizon 3000; side detector (72 beams) and lidar (72 beams, 50 m) in the
    observation.  Like the reference, the default 12 agents need map_config["exit_length"] >= 60 (the spawn manager counts
    (exit_length - 10) // 8 slots per lane; the reference's own tests pass 60, tests/test_env/test_ma_racing.py:83-84).
    Here the guardrail is a sidewalk strip that raises crash_sidewalk, not a wall: the kinematic engine has no contact response."""
    MAP_DEFAULTS = dict(marl_map="racing", num_agents=12, allow_respawn=False, traffic_density=0.0, random_agent_model=False,
                        map_config=dict(lane_num=2, exit_length=20),
                        vehicle_config=dict(lidar=dict(num_lasers=72, distance=50, num_others=0),
                                            side_detector=dict(num_lasers=72, distance=50), enable_reverse=False),
                        out_of_road_penalty=5.0, idle_penalty=1.0, success_reward=20.0, crash_sidewalk_penalty=1.0,
                        cross_yellow_line_done=False, out_of_road_done=True, on_continuous_line_done=False, out_of_route_done=False,
                        crash_done=False, horizon=3000, idle_done=True, crash_sidewalk_done=False, crash_vehicle_done=False)

    @classmethod
    def default_config(cls):
        import copy
        return make_config(_deep_update(copy.deepcopy(MULTI_AGENT_DEFAULTS), copy.deepcopy(cls.MAP_DEFAULTS)))

    def __init__(self, config=None):
        import copy
        merged = _deep_update(copy.deepcopy(self.MAP_DEFAULTS), dict(config or {}))
        super().__init__(merged)

    def _info(self):
        info = super()._info()
        e, A = self.engine, self.num_agents
        info._lazy["idle"] = lambda: (e.flags[:, :A] & abi.FL_IDLE) != 0
        return info


class BatchedMultiAgentBottleneckEnv(BatchedMultiAgentRoundaboutEnv):
    """MultiAgentBottleneckEnv (envs/marl_envs/marl_bottleneck.py:10-140): 20 agents, 4 lanes narrowing to 1 and
    widening again, traffic in both directions, side (4 beams) and lane-line (4 beams) detectors in the observation.
    Its reward / out-of-road rules equal MetaDriveEnv's with on_continuous_line_done (cross_yellow_line_done=True)."""
    MAP_DEFAULTS = dict(marl_map="bottleneck", num_agents=20,
                        map_config=dict(exit_length=60, lane_num=4, neck_lane_num=1, neck_length=20),
                        vehicle_config=dict(side_detector=dict(num_lasers=4, distance=50),
                                            lane_line_detector=dict(num_lasers=4, distance=20)))

    @classmethod
    def default_config(cls):
        import copy
        return make_config(_deep_update(copy.deepcopy(MULTI_AGENT_DEFAULTS), copy.deepcopy(cls.MAP_DEFAULTS)))

    def __init__(self, config=None):
        import copy
        merged = _deep_update(copy.deepcopy(self.MAP_DEFAULTS), dict(config or {}))
        super().__init__(merged)


class BatchedMultiAgentBidirectionEnv(BatchedMultiAgentBottleneckEnv):
    """MultiAgentBidirectionEnv (envs/marl_envs/marl_bidirection.py:11-140): the bottleneck map with a Bidirection block in
    the neck -- ONE lane that both directions share for 40-80 m -- 20 agents from both ends.  Rewards, termination and
    detectors are the bottleneck env's."""
    MAP_DEFAULTS = dict(BatchedMultiAgentBottleneckEnv.MAP_DEFAULTS, marl_map="bidirection")


class BatchedMultiAgentTollgateEnv(BatchedMultiAgentRoundaboutEnv):
    """MultiAgentTollgateEnv (envs/marl_envs/marl_tollgate.py:14-266): 40 agents from both ends of a road that widens from 3
    to 8 lanes, passes a 10 m toll block (a booth on every odd lane, speed limit 3) and narrows again.  Its own rules, all on
    the device (MdConfig.ma_kind = 1): the observation has no navigation dims and ends with [inside the toll block, stayed
    longer than min_pass_steps]; inside the block the speed reward gives way to -overspeed_penalty * v / v_max while the
    vehicle is above the limit; out of road = sidewalk or (cross_yellow_line_done) the yellow line; an agent that leaves the
    block less than `vehicle_config.min_pass_steps` steps after entering it is done, reported as out_of_road."""
    MAP_DEFAULTS = dict(marl_map="tollgate", num_agents=40, cross_yellow_line_done=True, speed_reward=0.0, overspeed_penalty=0.5,
                        map_config=dict(exit_length=70, lane_num=3, toll_lane_num=8, toll_length=10),
                        vehicle_config=dict(min_pass_steps=30, side_detector=dict(num_lasers=72, distance=20),
                                            lane_line_detector=dict(num_lasers=4, distance=20),
                                            lidar=dict(num_lasers=72, distance=20)))

    @classmethod
    def default_config(cls):
        import copy
        return make_config(_deep_update(copy.deepcopy(MULTI_AGENT_DEFAULTS), copy.deepcopy(cls.MAP_DEFAULTS)))

    def __init__(self, config=None):
        import copy
        merged = _deep_update(copy.deepcopy(self.MAP_DEFAULTS), dict(config or {}))
        super().__init__(merged)

    def _info(self):
        info = super()._info()
        e, A = self.engine, self.num_agents
        info._lazy["in_toll_time"] = lambda: e.nav_i[:, :A, 13] & 0xffffff      # MdNav.toll_state, low 24 bits
        return info


class BatchedMultiAgentParkingLotEnv(BatchedMultiAgentRoundaboutEnv):
    """MultiAgentParkingLotEnv (envs/marl_envs/marl_parking_lot.py:22-275): 10 agents on a one-lane road with
    `parking_space_num` spaces at right angles (ParkingLot block) and a T intersection behind it.  An agent entering from
    one of the three entrances is sent to a parking space no other active agent is heading for and holds it until it is
    done; an agent starting in a space leaves through a random entrance; an entrance only lets a new agent in while a space
    is free.  Reversing is on (`enable_reverse`).  Out of road = off the lanes, the yellow line or the sidewalk.
    `info["parking_space"]`: the space an agent holds (-1: none)."""
    MAP_DEFAULTS = dict(marl_map="parking_lot", num_agents=10, parking_space_num=8, map_config=dict(exit_length=20, lane_num=1),
                        vehicle_config=dict(enable_reverse=True))

    @classmethod
    def default_config(cls):
        import copy
        return make_config(_deep_update(copy.deepcopy(MULTI_AGENT_DEFAULTS), copy.deepcopy(cls.MAP_DEFAULTS)))

    def __init__(self, config=None):
        import copy
        merged = _deep_update(copy.deepcopy(self.MAP_DEFAULTS), dict(config or {}))
        super().__init__(merged)

    def _info(self):
        info = super()._info()
        e, A = self.engine, self.num_agents
        info._lazy["parking_space"] = lambda: e.nav_i[:, :A, 14] - 1       # MdNav.toll_entry
        return info


class BatchedMultiAgentMetaDrive(BatchedMultiAgentRoundaboutEnv):
    """MultiAgentMetaDrive itself (envs/marl_envs/multi_agent_metadrive.py:12-128): 15 agents on an ordinary procedurally
    generated map (one per scenario seed, 3 blocks, 3 lanes), all spawning on the first block's exit road (5 slots x 3
    lanes) and driving to the far end of the map."""
    MAP_DEFAULTS = dict(marl_map="pg", num_agents=15, map=3, map_config=dict(exit_length=50, lane_num=3))

    @classmethod
    def default_config(cls):
        import copy
        return make_config(_deep_update(copy.deepcopy(MULTI_AGENT_DEFAULTS), copy.deepcopy(cls.MAP_DEFAULTS)))

    def __init__(self, config=None):
        import copy
        merged = _deep_update(copy.deepcopy(self.MAP_DEFAULTS), dict(config or {}))
        super().__init__(merged)

"""Configuration of the batched envs: the reference's config keys for this path, same defaults.

Mirrors the layering BASE_DEFAULT_CONFIG -> METADRIVE_DEFAULT_CONFIG -> user dict
(metadrive/envs/base_env.py:32-266, envs/metadrive_env.py:16-99) and the Config class's contract
(utils/config.py:23-38,125): an unknown key raises KeyError, a value whose type differs from the
default's raises TypeError.  Keys of the reference that belong to subsystems outside the hot path
(rendering, cameras, recording ...) are accepted only at their default "off" value and rejected
loudly otherwise, so a reference config either behaves the same or fails -- never silently differs.
Batched-engine extras are grouped at the end.
"""
import copy

from metadrive_ped_amd.mapgen.pg import BlockDist

DEFAULT_AGENT = "default_agent"

BASE_DEFAULT_CONFIG = dict(
    # ===== agent =====
    random_agent_model=False,
    num_agents=1,
    is_multi_agent=False,
    allow_respawn=False,
    delay_done=0,
    # ===== action =====
    agent_policy="EnvInputPolicy",   # or "IDMPolicy" (the class of that name is accepted too): envs/base_env.py:53
    discrete_action=False,
    use_multi_discrete=False,
    discrete_steering_dim=5,
    discrete_throttle_dim=5,
    action_check=False,
    # ===== termination =====
    horizon=None,
    truncate_as_terminate=False,
    marl_map=None,          # None | "pg" | "roundabout" | "intersection" | "bottleneck" | "bidirection" | "tollgate" | "parking_lot" | "racing" (set by the multi-agent env classes)
    # ===== vehicle =====
    vehicle_config=dict(
        vehicle_model="default",
        enable_reverse=False,
        spawn_lane_index=None,
        destination=None,
        spawn_longitude=5.0,
        spawn_lateral=0.0,
        width=None, length=None, height=None, mass=None,   # read by vehicle_model="varying_dynamics" only (vehicle_type.py:168-187)
        spawn_velocity=None,             # [vx, vy] m/s at reset (base_vehicle.py:371-372); its component along the heading is kept
        spawn_velocity_car_frame=False,  # True: [forward, left] of the vehicle instead of world axes
        lidar=dict(num_lasers=240, distance=50, num_others=0, gaussian_noise=0.0, dropout_prob=0.0,
                   add_others_navi=False),
        side_detector=dict(num_lasers=0, distance=50, gaussian_noise=0.0, dropout_prob=0.0),
        lane_line_detector=dict(num_lasers=0, distance=20, gaussian_noise=0.0, dropout_prob=0.0),
        min_pass_steps=30,               # MultiAgentTollgateEnv: steps an agent has to spend inside the toll block (marl_tollgate.py:28)
    ),
    # ===== engine =====
    use_render=False,
    image_observation=False,
    physics_world_step_size=2e-2,
    decision_repeat=5,
    map_region_size=1024,
    log_level=20,
)

METADRIVE_DEFAULT_CONFIG = dict(
    start_seed=0,
    num_scenarios=1,
    map=3,
    block_dist_config=None,  # None -> BLOCK_TYPE_DISTRIBUTION_V2
    random_lane_width=False,
    random_lane_num=False,
    map_config=dict(type="block_num", config=None, lane_width=3.5, lane_num=3, exit_length=50,
                    neck_lane_num=1, neck_length=20,    # multi-agent bottleneck map only (marl_bottleneck.py:13)
                    toll_lane_num=8, toll_length=10,    # multi-agent tollgate map only (marl_tollgate.py:19)
                    radius=None),                       # multi-agent intersection maps only (tinyinter.py:349-351)
    store_map=True,
    traffic_density=0.1,
    need_inverse_traffic=False,
    traffic_mode="trigger",
    random_traffic=False,
    accident_prob=0.0,
    static_traffic_object=True,
    random_spawn_lane_index=True,
    agent_configs={DEFAULT_AGENT: dict(spawn_lane_index=(">", ">>", 0))},
    success_reward=10.0,
    out_of_road_penalty=5.0,
    crash_vehicle_penalty=5.0,
    crash_object_penalty=5.0,
    driving_reward=1.0,
    speed_reward=0.1,
    use_lateral_reward=False,
    crash_vehicle_cost=1.0,
    crash_object_cost=1.0,
    out_of_road_cost=1.0,
    out_of_route_done=False,
    on_continuous_line_done=True,
    crash_vehicle_done=True,
    crash_object_done=True,
    crash_human_done=True,
    enable_idm_lane_change=True,
    # multi-agent keys (envs/marl_envs/multi_agent_metadrive.py:12-61); inert for single-agent envs
    crash_done=True,
    out_of_road_done=True,
    force_seed_spawn_manager=False,
    spawn_roads=None,
    cross_yellow_line_done=True,   # bottleneck / bidirection / tollgate envs (marl_bottleneck.py:17,129-135, marl_tollgate.py:22,241-247)
    overspeed_penalty=0.5,         # tollgate env (marl_tollgate.py:25)
    parking_space_num=8,           # parking-lot env (marl_parking_lot.py:31): an even number >= 4
    # racing env (RACING_CONFIG, marl_racing_env.py:46-63)
    crash_sidewalk_penalty=1.0, idle_penalty=1.0, idle_done=True, crash_sidewalk_done=False,
    # VaryingDynamicsEnv (envs/varying_dynamics_env.py:14-25): None = off, else {parameter: (min, max) | None}
    random_dynamics=None,
)

# batched-engine keys (no counterpart in the reference: it steps one world per process)
BATCH_DEFAULT_CONFIG = dict(
    num_envs=1,             # E: environments stepped in lockstep by this process (this GPU's shard)
    env_seed_offset=0,      # global index of this shard's first env (seed = start_seed + (offset + e) % num_scenarios)
    mover_capacity=0,       # slots per env (agents + traffic + props); 0 = smallest multiple of 8 that fits every env
    auto_reset=True,        # restore an env from its reset snapshot on the step after it finished
    device="cuda:0",
    build_workers=0,        # 1 = generate maps in this process; otherwise the persistent workers of hostpool.py do it
    build_cache=False,      # memoise built (map, scene) pairs by scenario seed + config (sub-batches / copies of the same envs)
    traffic_epoch=0,        # random_traffic=True: bumped by every explicit env.reset(); part of the traffic stream's seed
    traffic_draws=4,        # random_traffic=True with auto_reset: traffic draws staged on the device; an env takes the next one
                            # at every reset (an episode meets its n-th predecessor's traffic again); 1 = one draw per env.reset()
    initial_agents=0,       # set by num_agents=-1 (multi-agent): agents present at reset; the other slots start free
    step_kernel="auto",     # single-agent md_step: "wg" = one 4-wave workgroup per env, "wave" = one wave per env, "auto" = by
                            # the number of distinct maps the batch shares (engine.WAVE_KERNEL_MAX_MAPS) (same
                            # results bit for bit; a machine-mapping choice)
)

# Keys of the reference's BASE_DEFAULT_CONFIG (envs/base_env.py:32-266) that only concern rendering, cameras, the GUI,
# debugging displays or asset handling: nothing on this path reads them, so any value is accepted and kept.
COSMETIC_DEFAULT_CONFIG = dict(
    controller="keyboard", norm_pixel=True, stack_size=3, use_chase_camera_follow_lane=False, camera_height=2.2, camera_dist=7.5,
    camera_pitch=None, camera_smooth=True, camera_smooth_buffer_size=20, camera_fov=65, prefer_track_agent=None,
    top_down_camera_initial_x=0, top_down_camera_initial_y=0, top_down_camera_initial_z=200, window_size=(1200, 900),
    image_on_cuda=False, _render_mode="none", force_render_fps=None, force_destroy=False, num_buffering_objects=200,
    render_pipeline=False, daytime="19:00", shadow_range=50, multi_thread_render=True, multi_thread_render_mode="Cull",
    preload_models=True, disable_model_compression=True, cull_lanes_outside_map=False, drivable_area_extension=7,
    height_scale=50, use_mesh_terrain=False, full_size_mesh=True, show_crosswalk=True, show_sidewalk=True, pstats=False,
    debug=False, debug_panda3d=False, debug_physics_world=False, debug_static_world=False, show_coordinates=False,
    show_fps=True, show_logo=True, show_mouse=True, show_skybox=True, show_terrain=True, show_interface=True,
    show_policy_mark=False, show_interface_navi_mark=True, interface_panel=["dashboard"], force_reuse_object_name=False,
    traffic_vehicle_config=dict(show_navi_mark=False, show_dest_mark=False, enable_reverse=False, show_lidar=False,
                                show_lane_line_detector=False, show_side_detector=False),
)
COSMETIC_VEHICLE_CONFIG = dict(show_navi_mark=True, show_dest_mark=False, show_line_to_dest=False, show_line_to_navi_mark=False,
                               use_special_color=False, image_source="rgb_camera", overtake_stat=False, random_color=False,
                               top_down_width=None, top_down_length=None, show_lidar=False, show_side_detector=False,
                               show_lane_line_detector=False)

# Behavioural keys of subsystems that are not built: accepted at the reference's default, rejected loudly otherwise.
_OFF_ONLY = dict(use_render=False, image_observation=False, manual_control=False, agent_observation=None,
                 sensors=None, record_episode=False, replay_episode=None, only_reset_when_replay=False, use_AI_protector=False,
                 save_level=0.5)
_OFF_ONLY_VEHICLE = dict(no_wheel_friction=False, navigation_module=None, spawn_position_heading=None, light=False)
_OFF_HINT = dict(record_episode="use env.start_recording() / stop_recording() / export_scenarios()",
                 replay_episode="use traffic_mode='replay' with env.load_tracks()")


def _merge(dst, src, path=""):
    for k, v in src.items():
        if k not in dst:
            raise KeyError("'{}{}' does not exist in existing config. Please use config.update(..., allow_add_new_key="
                           "True) to allow new keys -- unknown config key".format(path, k))
        if isinstance(dst[k], dict) and isinstance(v, dict) and k != "agent_configs":
            _merge(dst[k], v, path + k + ".")
        else:
            old = dst[k]
            cosmetic = k in COSMETIC_DEFAULT_CONFIG or k in COSMETIC_VEHICLE_CONFIG
            if old is not None and v is not None and not isinstance(old, dict) and not cosmetic:
                # utils/config.py:241-250: int <-> float are interchangeable
                ok = isinstance(v, type(old)) or (isinstance(old, float) and isinstance(v, int)) or \
                    (isinstance(old, int) and not isinstance(old, bool) and isinstance(v, float)) or \
                    (k == "map" and isinstance(v, (int, str))) or (k == "horizon") or (k == "agent_policy")
                if not ok:
                    raise TypeError("Attempting to update '{}{}' with type {}, expected {}".format(
                        path, k, type(v).__name__, type(old).__name__))
            dst[k] = v


def make_config(user=None):
    cfg = copy.deepcopy(BASE_DEFAULT_CONFIG)
    cfg.update(copy.deepcopy(METADRIVE_DEFAULT_CONFIG))
    cfg.update(copy.deepcopy(BATCH_DEFAULT_CONFIG))
    cfg.update(copy.deepcopy(COSMETIC_DEFAULT_CONFIG))
    cfg["vehicle_config"].update(copy.deepcopy(COSMETIC_VEHICLE_CONFIG))
    for k, off in _OFF_ONLY.items():
        cfg.setdefault(k, off)
    for k, off in _OFF_ONLY_VEHICLE.items():
        cfg["vehicle_config"].setdefault(k, off)
    user = dict(user or {})
    if isinstance(user.get("sensors"), dict) and not user["sensors"]:
        user["sensors"] = None                                  # an empty sensor table is the default
    _merge(cfg, user)
    for where, table in ((cfg, _OFF_ONLY), (cfg["vehicle_config"], _OFF_ONLY_VEHICLE)):
        for k, off in table.items():
            if where[k] != off:
                raise NotImplementedError("config['{}']={!r}: this option lies outside the batched step() path built so far "
                                          "(only {!r} is accepted){}".format(k, where[k], off,
                                                                             "; " + _OFF_HINT[k] if k in _OFF_HINT else ""))
    # parse_map_config (component/map/pg_map.py:17-36): `map` shorthand fills map_config
    m = cfg["map"]
    if isinstance(m, int):
        cfg["map_config"]["type"], cfg["map_config"]["config"] = "block_num", m
    elif isinstance(m, str):
        cfg["map_config"]["type"], cfg["map_config"]["config"] = "block_sequence", m
    else:
        raise ValueError("Unknown easy map config: {}".format(m))
    if cfg["block_dist_config"] is None:
        cfg["block_dist_config"] = BlockDist()
    elif isinstance(cfg["block_dist_config"], dict):
        cfg["block_dist_config"] = BlockDist(cfg["block_dist_config"])
    if cfg["is_multi_agent"] and abs(cfg["traffic_density"]) >= 1e-2:
        raise NotImplementedError("traffic_density > 0 in a multi-agent env is not built (the reference's multi-agent "
                                  "envs run without traffic: multi_agent_metadrive.py:58)")
    if cfg["is_multi_agent"] and cfg["random_dynamics"]:
        raise NotImplementedError("random_dynamics: 'Only supporting single-agent now!' (varying_dynamics_env.py:46)")
    if cfg["random_dynamics"]:
        unknown = set(cfg["random_dynamics"]) - {"max_engine_force", "max_brake_force", "wheel_friction", "max_steering", "mass"}
        if unknown:
            raise KeyError("random_dynamics: unknown parameter(s) {}".format(sorted(unknown)))
    if cfg["is_multi_agent"] and abs(cfg["accident_prob"]) >= 1e-2:
        raise NotImplementedError("accident scenes in a multi-agent env are not built")
    # agent_policy: the reference takes a policy CLASS; here its name (or a class of that name)
    pol = cfg["agent_policy"]
    pol = pol if isinstance(pol, str) else getattr(pol, "__name__", repr(pol))
    if pol not in ("EnvInputPolicy", "IDMPolicy"):
        raise NotImplementedError("agent_policy={!r}: built are EnvInputPolicy (actions from step()), IDMPolicy and, in "
                                  "BatchedScenarioEnv only, ReplayEgoCarPolicy".format(pol))
    cfg["agent_policy"] = pol
    if cfg["num_agents"] == -1:
        # "infinite agents" (spawn_manager.py:74-78, agent_manager.py:272-279, multi_agent_metadrive.py:86-92): every spawn
        # point holds an agent at reset and a new agent enters whenever a spawn region is clear, whatever the number on
        # the road.  With fixed slots: capacity = all spawn points, and as many slots again for vehicles still on the
        # road (active or dying) -- a respawn needs a free slot too, which is the one bound the reference does not have.
        if not cfg["is_multi_agent"] or cfg["marl_map"] is None:
            raise ValueError("num_agents=-1 (infinite agents) is a multi-agent env option")
        import math
        from metadrive_ped_amd.marl import PG_SPAWN_ROADS, SPAWN_ROADS
        roads = cfg["spawn_roads"] or (PG_SPAWN_ROADS if cfg["marl_map"] == "pg" else SPAWN_ROADS[cfg["marl_map"]])
        slots = int(math.floor((cfg["map_config"]["exit_length"] - 10) / 8.0))      # max_capacity (spawn_manager.py:108-115)
        if slots <= 0:
            raise ValueError("The exist length {} should greater than minimal longitude interval {}.".format(
                cfg["map_config"]["exit_length"] - 10, 18))
        cfg["initial_agents"] = cfg["map_config"]["lane_num"] * len(roads) * slots
        cfg["num_agents"] = min(128, 2 * cfg["initial_agents"])
    if cfg["spawn_roads"] is not None and not cfg["is_multi_agent"]:
        raise ValueError("spawn_roads is a multi-agent env option")
    if cfg["is_multi_agent"] and cfg["marl_map"] is None:
        raise NotImplementedError("multi-agent configs are built through the multi-agent env classes (marl_map)")
    if cfg["marl_map"] == "parking_lot":
        n = cfg["parking_space_num"]
        assert n % 2 == 0, "number of parking spaces must be multiples of 2"          # marl_parking_lot.py:198-199
        assert n >= 4, "minimal number of parking space is 4"
        if n > 20:
            raise ValueError("parking_space_num > 20: the spawn tables hold 32 places")
    if not cfg["cross_yellow_line_done"] and cfg["marl_map"] not in ("tollgate", "racing"):
        raise NotImplementedError("cross_yellow_line_done=False is built for the tollgate env only")
    if cfg["step_kernel"] not in ("auto", "wg", "wave"):
        raise ValueError("step_kernel must be 'auto', 'wg' or 'wave', got {!r}".format(cfg["step_kernel"]))
    if cfg["mover_capacity"] != 0 and (cfg["mover_capacity"] > 128 or cfg["mover_capacity"] < cfg["num_agents"]):
        raise ValueError("mover_capacity must be 0 (auto) or in [num_agents, 128]")
    return cfg

"""ctypes / numpy mirror of include/mdstep.h.

Every struct of the C-ABI exists here twice: as a numpy structured dtype (for the table arrays the
host builds and uploads) and as a ctypes.Structure (for the three by-pointer argument blocks
MdWorld / MdState / MdConfig).  `check_abi(lib)` compares sizeof() with what the loaded library
reports through md_abi(), so a header/ binding drift fails loudly at import time.
"""
import ctypes as C

import numpy as np

MD_ABI_VERSION = 11
MD_POLY_GROUP = 8     # pieces per MdWorld.poly_ball group
MD_OK, MD_EINVAL, MD_ELAUNCH, MD_ENODEV, MD_EABI = 0, -1, -2, -3, -4
MD_MAX_CAP = 128
MD_MAX_BEAMS = 1024
MD_ROUTE_LEN = 48
MD_IDM_RAND = 8

# mover kinds / flags
KIND_NONE, KIND_VEHICLE, KIND_CONE, KIND_WARNING, KIND_BARRIER, KIND_PEDESTRIAN, KIND_CYCLIST, KIND_BUILDING = range(8)
KIND_MASK = 0xF
F_ALIVE, F_AGENT, F_PENDING, F_STATIC, F_CRASHED_ONCE, F_SPAWNED = 0x10, 0x20, 0x40, 0x80, 0x100, 0x200

# per-step flag word
FL_CRASH_VEHICLE = 0x0001
FL_CRASH_OBJECT = 0x0002
FL_CRASH_HUMAN = 0x0004
FL_CRASH_BUILDING = 0x0008
FL_CRASH_SIDEWALK = 0x0010
FL_ON_WHITE_CONT = 0x0020
FL_ON_YELLOW_CONT = 0x0040
FL_ON_BROKEN = 0x0080
FL_ON_CROSSWALK = 0x0100
FL_ON_LANE = 0x0200
FL_OUT_OF_ROUTE = 0x0400
FL_OUT_OF_ROAD = 0x0800
FL_ARRIVE_DEST = 0x1000
FL_MAX_STEP = 0x2000
FL_TERMINATED = 0x4000
FL_TRUNCATED = 0x8000

Q_LINE_WHITE_CONT, Q_LINE_YELLOW_CONT, Q_LINE_BROKEN, Q_SIDEWALK, Q_CROSSWALK = 1, 2, 3, 4, 5

f4, i4 = np.float32, np.int32

SHAPE_DT = np.dtype([("cx", f4), ("cy", f4), ("c", f4), ("s", f4), ("hl", f4), ("hw", f4), ("flags", i4), ("aux", i4)])
DYN_DT = np.dtype([("heading", f4), ("speed", f4), ("steering", f4), ("throttle", f4), ("last_x", f4),
                   ("last_y", f4), ("last_c", f4), ("last_s", f4)])
PARAM_DT = np.dtype([("max_steer", f4), ("accel_gain", f4), ("brake_gain", f4), ("roll_decel", f4),
                     ("max_speed_kmh", f4), ("lf", f4), ("lr", f4), ("fric_decel", f4)])
NAV_DT = np.dtype([("lane", i4), ("ck0", i4), ("ck1", i4), ("route_len", i4), ("target_lane", i4), ("timer", i4),
                   ("trigger_road", i4), ("trigger_order", i4), ("steps", i4), ("rand_cursor", i4), ("done", i4),
                   ("road0", i4), ("road1", i4), ("toll_state", i4), ("toll_entry", i4), ("toll_exit", i4)])
PID_DT = np.dtype([("hp", f4), ("hi", f4), ("hd", f4), ("lp", f4), ("li", f4), ("ld", f4), ("target_speed", f4),
                   ("energy", f4)])
LANE_DT = np.dtype([("type", i4), ("road", i4), ("idx", i4), ("n_in_road", i4), ("ax", f4), ("ay", f4), ("bx", f4),
                    ("by", f4), ("length", f4), ("width", f4), ("end_phase", f4), ("dirsign", f4), ("angle", f4),
                    ("heading", f4), ("sx", f4), ("sy", f4), ("ex", f4), ("ey", f4), ("x0", f4), ("y0", f4),
                    ("x1", f4), ("y1", f4), ("hull_off", i4), ("hull_n", i4), ("end_phase_w", f4), ("speed_limit", f4),
                    ("elx", f4), ("ely", f4), ("spare", f4, (4, )), ("hull4", f4, (8, ))])
ROAD_DT = np.dtype([("first_lane", i4), ("n_lanes", i4), ("start_node", i4), ("end_node", i4), ("negative", i4),
                    ("block", i4), ("block_kind", i4), ("spare", i4)])
SEG_DT = np.dtype([("sx", f4), ("sy", f4), ("ex", f4), ("ey", f4), ("dx", f4), ("dy", f4), ("len", f4), ("heading", f4),
                   ("cum", f4), ("spare", f4, (3, ))])
MA_DEFAULT, MA_TOLLGATE, MA_PARKING_LOT, MA_RACING = 0, 1, 2, 3
MD_IDLE_WINDOW = 100
FL_IDLE = 0x10000
TM_MOVING, TM_LENGTH_OK, TM_NEVER = 1, 2, 4
SC_ABSENT, SC_REPLAY, SC_IDM, SC_ARRIVED = 0, 1, 2, 3
GRID_DT = np.dtype([("x0", f4), ("y0", f4), ("inv_cell", f4), ("nx", i4), ("ny", i4), ("cell_base", i4),
                    ("spare", i4, (2, ))])

assert SHAPE_DT.itemsize == 32 and DYN_DT.itemsize == 32 and PARAM_DT.itemsize == 32
assert NAV_DT.itemsize == 64 and PID_DT.itemsize == 32 and LANE_DT.itemsize == 160
assert ROAD_DT.itemsize == 32 and GRID_DT.itemsize == 32 and SEG_DT.itemsize == 48

P = C.c_void_p


class MdWorld(C.Structure):
    _fields_ = [
        ("n_maps", C.c_int32), ("n_envs", C.c_int32),
        ("env_map", P), ("lane_off", P), ("lanes", P), ("hull_xy", P), ("road_off", P), ("roads", P),
        ("quad_off", P), ("quads", P), ("quad_kind", P), ("grid", P), ("cell_start", P), ("cell_items", P),
        ("node_adj_off", P), ("node_adj", P), ("node_off", P), ("beam_cs", P),
        ("max_lanes", C.c_int32), ("max_roads", C.c_int32),
        ("spawn_off", P), ("spawn_place", P), ("spawn_lane", P), ("spawn_route", P), ("spawn_route_meta", P),
        ("n_dest", C.c_int32), ("n_vclass", C.c_int32),
        ("poly_off", P), ("segs", P), ("polyv_off", P), ("polyv", P), ("ckpt_off", P), ("ckpt_xy", P), ("track_meta", P), ("vclass", P), ("poly_aux", P), ("side_beam_cs", P), ("ll_beam_cs", P), ("quad_ball", P), ("run_off", P), ("runs", P),
        ("poly_ball", P), ("poly_ball_off", P),
    ]


class MdState(C.Structure):
    _fields_ = [
        ("shape", P), ("dyn", P), ("param", P), ("nav", P), ("pid", P), ("action", P), ("route_nodes", P),
        ("route_roads", P), ("final_lane", P), ("idm_rand", P), ("flags", P), ("obs", P), ("reward", P), ("cost", P),
        ("step_info", P), ("need_reset", P), ("shape0", P), ("dyn0", P), ("nav0", P), ("pid0", P),
        ("route_nodes0", P), ("route_roads0", P), ("final_lane0", P), ("rng", P), ("env_steps", P), ("agent_id", P),
        ("next_agent_id", P),
        ("agent_action", P),
        ("track_shape", P),
        ("track_dyn", P),
        ("detected", P),
        ("scratch", P),
        ("param0", P),
        ("done_out", P),
        ("route_n", P), ("route_segs", P), ("route_verts", P), ("route_aux", P), ("idle_ring", P),
    ]


class MdConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("n_envs", C.c_int32), ("agents_per_env", C.c_int32), ("cap", C.c_int32),
        ("n_beams", C.c_int32), ("obs_dim", C.c_int32), ("substeps", C.c_int32), ("horizon", C.c_int32),
        ("dt", C.c_float), ("lidar_range", C.c_float),
        ("success_reward", C.c_float), ("out_of_road_penalty", C.c_float), ("crash_vehicle_penalty", C.c_float),
        ("crash_object_penalty", C.c_float), ("driving_reward", C.c_float), ("speed_reward", C.c_float),
        ("crash_vehicle_cost", C.c_float), ("crash_object_cost", C.c_float), ("out_of_road_cost", C.c_float),
        ("use_lateral_reward", C.c_int32), ("out_of_route_done", C.c_int32), ("on_continuous_line_done", C.c_int32),
        ("crash_vehicle_done", C.c_int32), ("crash_object_done", C.c_int32), ("crash_human_done", C.c_int32),
        ("truncate_as_terminate", C.c_int32), ("traffic_mode", C.c_int32), ("enable_idm_lane_change", C.c_int32),
        ("auto_reset", C.c_int32),
        ("max_lane_width", C.c_float), ("total_width", C.c_float), ("curve_radius_max", C.c_float),
        ("curve_angle_max", C.c_float),
        ("is_multi_agent", C.c_int32), ("delay_done", C.c_int32), ("allow_respawn", C.c_int32),
        ("crash_done", C.c_int32), ("out_of_road_done", C.c_int32), ("n_side", C.c_int32), ("n_lane_line", C.c_int32),
        ("num_others", C.c_int32),
        ("add_others_navi", C.c_int32),
        ("track_len", C.c_int32),
        ("random_agent_model", C.c_int32),
        ("agent_idm", C.c_int32),
        ("enable_reverse", C.c_int32),
        ("on_lane_line_penalty", C.c_float), ("crash_human_penalty", C.c_float), ("steering_range_penalty", C.c_float),
        ("heading_penalty", C.c_float), ("lateral_penalty", C.c_float), ("max_lateral_dist", C.c_float),
        ("crash_human_cost", C.c_float),
        ("no_negative_reward", C.c_int32), ("relax_out_of_road_done", C.c_int32), ("reactive_traffic", C.c_int32),
        ("filter_overlapping_car", C.c_int32), ("no_static_vehicles", C.c_int32), ("allowed_more_steps", C.c_int32),
        ("scenario_length", C.c_int32),
        ("step_kernel", C.c_int32),
        ("ma_kind", C.c_int32), ("min_pass_steps", C.c_int32), ("overspeed_penalty", C.c_float), ("n_parking", C.c_int32),
        ("side_range", C.c_float), ("ll_range", C.c_float), ("side_mask", C.c_uint32), ("ll_mask", C.c_uint32),
        ("route_seg_cap", C.c_int32), ("route_vert_cap", C.c_int32), ("ego_replay", C.c_int32),
        ("crash_sidewalk_penalty", C.c_float), ("idle_penalty", C.c_float), ("idle_done", C.c_int32), ("crash_sidewalk_done", C.c_int32),
    ]


STRUCT_SIZES = [SHAPE_DT.itemsize, DYN_DT.itemsize, PARAM_DT.itemsize, NAV_DT.itemsize, PID_DT.itemsize,
                LANE_DT.itemsize, ROAD_DT.itemsize, GRID_DT.itemsize, C.sizeof(MdWorld), C.sizeof(MdState),
                C.sizeof(MdConfig)]
STRUCT_NAMES = ["MdShape", "MdDyn", "MdParam", "MdNav", "MdPid", "MdLane", "MdRoad", "MdGrid", "MdWorld", "MdState",
                "MdConfig"]

WORLD_FIELDS = [f for f, t in MdWorld._fields_ if t is P]
STATE_FIELDS = [f for f, t in MdState._fields_ if t is P]

# symbols include/mdstep.h declares; tests check every one is exported
ENTRY_POINTS = ["md_abi", "md_last_error", "md_probe_math", "md_probe_stream_copy", "md_lidar", "md_lidar_detect", "md_line_detector", "md_line_detectors", "md_swap_draw", "md_integrate", "md_localize",
                "md_contacts", "md_observe", "md_idm", "md_traffic_after_step", "md_lifecycle", "md_step"]


def check_abi(abi_fn, what):
    sizes = (C.c_int32 * 11)()
    ver = abi_fn(sizes, 11)
    if ver != MD_ABI_VERSION:
        raise RuntimeError("%s: ABI version %d, binding expects %d" % (what, ver, MD_ABI_VERSION))
    for name, mine, theirs in zip(STRUCT_NAMES, STRUCT_SIZES, list(sizes)):
        if mine != theirs:
            raise RuntimeError("%s: sizeof(%s) = %d in the library but %d in the Python binding" %
                               (what, name, theirs, mine))


def fill_struct(struct, fields, arrays, ptr_of):
    """Set every pointer field of `struct` from `arrays[field]` (None -> NULL) using ptr_of(array)."""
    for f in fields:
        a = arrays.get(f)
        setattr(struct, f, None if a is None else ptr_of(a))
    return struct

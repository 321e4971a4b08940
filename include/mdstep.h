/*
 * mdstep.h -- C-ABI of libmdstep.so: the MI355X (gfx950) batched MetaDrive step() hot path.
 *
 * The reference (zhuhaozh/metadrive_ped, MetaDrive v0.4.2.2) has no FFI: its hot path is Python
 * calling Bullet through the panda3d binding.  Each entry point below names the reference call
 * site(s) it replaces (paths relative to /root/reference/metadrive).  INTEGRATION.md shows the
 * ctypes stub a maintainer would add on the reference side.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only.  Every data pointer is a DEVICE pointer owned by the
 *    caller (PyTorch-ROCm tensors); the library borrows it for the duration of the call, never
 *    allocates, never frees, never synchronises.  Launches are asynchronous on `stream`
 *    (a hipStream_t passed as void*; NULL = the legacy default stream).
 *  - The MdWorld / MdState / MdConfig structs themselves live in HOST memory and are copied by
 *    value into the kernel arguments.
 *  - Return value: MD_OK (0) or a negative MD_E* code.  No exceptions cross the boundary.
 *    md_last_error() returns a static, thread-local description of the last failure.
 *  - Layout: E environments, each with `cap` mover slots.  Mover n = env * cap + slot.  Slots
 *    [0, agents_per_env) are the controlled agents (BaseVehicle driven by EnvInputPolicy), the rest are
 *    traffic vehicles (IDMPolicy) and static props.  All per-mover arrays are indexed by n.
 *  - Coordinates follow the reference's 2-D convention: x forward / y left, heading counter-
 *    clockwise in radians (utils/coordinates_shift.py:8-31); lane lateral is positive to the RIGHT
 *    (component/lane/straight_lane.py:46,69-74).
 */
#ifndef MDSTEP_H
#define MDSTEP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MD_ABI_VERSION 11

/* ---- error codes ------------------------------------------------------------------------- */
#define MD_OK 0
#define MD_EINVAL (-1)   /* bad argument (null pointer, non-positive size, cap too large ...) */
#define MD_ELAUNCH (-2)  /* hipLaunchKernel / hipGetLastError reported a failure               */
#define MD_ENODEV (-3)   /* no usable gfx950 device                                           */
#define MD_EABI (-4)     /* struct size mismatch between caller and library                   */

/* ---- limits ------------------------------------------------------------------------------ */
#define MD_POLY_GROUP 8      /* pieces per MdWorld.poly_ball group                              */
#define MD_MAX_CAP 128        /* mover slots per env the kernels are built for                  */
#define MD_MAX_BEAMS 1024     /* lidar beams per agent                                          */
#define MD_ROUTE_LEN 48       /* checkpoints (road nodes) per route                             */
#define MD_IDM_RAND 8         /* pre-drawn lane-change timer values per traffic vehicle         */

/* ---- mover kinds (MdShape.flags bits 0..3) ------------------------------------------------ */
#define MD_KIND_NONE 0
#define MD_KIND_VEHICLE 1     /* box  W x L   component/vehicle/base_vehicle.py:588              */
#define MD_KIND_CONE 2        /* circle r=0.2 component/static_object/traffic_object.py:43-84    */
#define MD_KIND_WARNING 3     /* circle r=0.5 component/static_object/traffic_object.py:87-127   */
#define MD_KIND_BARRIER 4     /* box 0.3x2.0  component/static_object/traffic_object.py:130-177  */
#define MD_KIND_PEDESTRIAN 5  /* circle r=.35 component/traffic_participants/pedestrian.py:12-30 */
#define MD_KIND_CYCLIST 6     /* box          component/traffic_participants/cyclist.py:28       */
#define MD_KIND_BUILDING 7    /* box: TollGateBuilding 10 x lane width  component/buildings/tollgate_building.py:7-27 */
#define MD_KIND_MASK 0xF
#define MD_F_ALIVE 0x10       /* present in the world this step                                 */
#define MD_F_AGENT 0x20       /* slot is a controlled agent                                     */
#define MD_F_PENDING 0x40     /* traffic vehicle waiting for its block's trigger road            */
#define MD_F_STATIC 0x80      /* never integrated (props, broken-down vehicles)                 */
#define MD_F_CRASHED_ONCE 0x100 /* COST_ONCE object already counted (traffic_object.py)         */
#define MD_F_SPAWNED 0x200    /* (re)spawned at the start of this step: not integrated, observe returns its reset obs */

/* ---- per-step flag word (MdState.flags) --------------------------------------------------- */
#define MD_FL_CRASH_VEHICLE 0x0001
#define MD_FL_CRASH_OBJECT 0x0002
#define MD_FL_CRASH_HUMAN 0x0004
#define MD_FL_CRASH_BUILDING 0x0008
#define MD_FL_CRASH_SIDEWALK 0x0010
#define MD_FL_ON_WHITE_CONT 0x0020
#define MD_FL_ON_YELLOW_CONT 0x0040
#define MD_FL_ON_BROKEN 0x0080
#define MD_FL_ON_CROSSWALK 0x0100
#define MD_FL_ON_LANE 0x0200
#define MD_FL_OUT_OF_ROUTE 0x0400
#define MD_FL_OUT_OF_ROAD 0x0800
#define MD_FL_ARRIVE_DEST 0x1000
#define MD_FL_MAX_STEP 0x2000
#define MD_FL_TERMINATED 0x4000
#define MD_FL_TRUNCATED 0x8000
#define MD_FL_IDLE 0x10000   /* racing env: moved < 0.1 m along its lane over the last 100 steps (beyond the 16 bits MdState.done_out carries) */

/* ---- static quad kinds (MdWorld.quad_kind) ------------------------------------------------ */
#define MD_Q_LINE_WHITE_CONT 1   /* LINE_SOLID_SINGLE_WHITE  block/base_block.py:487-492         */
#define MD_Q_LINE_YELLOW_CONT 2  /* LINE_SOLID_SINGLE_YELLOW                                     */
#define MD_Q_LINE_BROKEN 3       /* LINE_BROKEN_SINGLE_*                                         */
#define MD_Q_SIDEWALK 4          /* BOUNDARY_SIDEWALK strip   pgblock/pg_block.py:294-332        */
#define MD_Q_CROSSWALK 5

/* 32-byte shape record: what lidar and contact tests read.  One per mover. */
typedef struct MdShape {
    float cx, cy;   /* centre                                                     */
    float c, s;     /* cos / sin of heading                                       */
    float hl, hw;   /* half length (along heading) / half width; circle: hl=hw=r  */
    int32_t flags;  /* MD_KIND_* | MD_F_*                                         */
    int32_t aux;    /* lane id (map-local) the object sits on, -1 if unknown      */
} MdShape;

/* 32-byte dynamic record.  For a walking participant (MD_KIND_PEDESTRIAN / MD_KIND_CYCLIST, not MD_F_STATIC):
 * steering / throttle hold its world-frame velocity (vx, vy) in m/s, speed its norm. */
typedef struct MdDyn {
    float heading;  /* psi, radians                                               */
    float speed;    /* signed forward speed, m/s                                  */
    float steering; /* last applied steering action in [-1,1] (vehicle.steering)  */
    float throttle; /* last applied throttle/brake action in [-1,1]               */
    float last_x, last_y;  /* position before this step (vehicle.last_position)   */
    float last_c, last_s;  /* heading dir before this step (last_heading_dir)     */
} MdDyn;

/* 32-byte vehicle parameter record (kinematic stand-in for btRaycastVehicle). */
typedef struct MdParam {
    float max_steer;     /* rad: config max_steering deg -> rad  (pg_space.py:226-272)          */
    float accel_gain;    /* m/s^2 at throttle=1: 4*max_engine_force/mass                        */
    float brake_gain;    /* m/s^2 at brake=1:    4*max_brake_force/(mass*substep_dt), capped    */
    float roll_decel;    /* m/s^2 idle brake 2.0 (base_vehicle.py:473)                          */
    float max_speed_kmh; /* engine cut above this (base_vehicle.py:474)                         */
    float lf, lr;        /* FRONT_WHEELBASE / REAR_WHEELBASE (vehicle_type.py)                  */
    float fric_decel;    /* wheel_friction * g cap on brake decel                               */
} MdParam;

/* 64-byte navigation / policy integer state. */
typedef struct MdNav {
    int32_t lane;        /* current lane (map-local id) or -1 (vehicle.lane)                    */
    int32_t ck0, ck1;    /* _target_checkpoints_index (node_network_navigation.py:99)           */
    int32_t route_len;   /* number of checkpoints (nodes) in the route                          */
    int32_t target_lane; /* IDMPolicy.routing_target_lane or -1  (idm_policy.py:226)            */
    int32_t timer;       /* IDMPolicy.overtake_timer                                            */
    int32_t trigger_road;/* traffic: road id whose entry by an agent activates this vehicle     */
    int32_t trigger_order;/* traffic: block index; blocks are triggered in ascending order      */
    int32_t steps;       /* episode_lengths[agent]                                              */
    int32_t rand_cursor; /* next entry of MdState.idm_rand to consume                           */
    int32_t done;        /* sticky BaseEnv.dones[agent] (envs/base_env.py:600)                  */
    int32_t road0, road1;/* route_roads[ck0] / route_roads[ck1]: the roads of navigation.current_ref_lanes / next_ref_lanes
                          * (node_network_navigation.py:130-168), kept beside the cursors so that the per-step logic never
                          * indexes the 48-entry route arrays; rewritten whenever ck0 / ck1 change (checkpoint advance,
                          * respawn, reset snapshot)                                              */
    /* MultiAgentTollgateEnv only (envs/marl_envs/marl_tollgate.py:38-112; zero elsewhere): */
    int32_t toll_state;  /* TollGateObservation.in_toll_time (low 24 bits) | StayTimeManager.last_block << 24 (0 = none yet) */
    int32_t toll_entry;  /* StayTimeManager.entry_time + 1 in the agent's own steps (0 = not recorded)     */
    int32_t toll_exit;   /* StayTimeManager.exit_time + 1 (0 = not recorded)                               */
} MdNav;

/* 32-byte IDM controller state (PID_controller.py:1-22, idm_policy.py:226-233). */
typedef struct MdPid {
    float hp, hi, hd;    /* heading PID p/i/d errors                                            */
    float lp, li, ld;    /* lateral PID p/i/d errors                                            */
    float target_speed;  /* km/h (NORMAL_SPEED 30 / CREEP_SPEED 5)                              */
    float energy;        /* episode energy consumption                                          */
} MdPid;

/* 160-byte lane record (component/lane/straight_lane.py, circular_lane.py). */
typedef struct MdLane {
    int32_t type;        /* 0 straight, 1 circular                                              */
    int32_t road;        /* map-local road id                                                   */
    int32_t idx;         /* index inside the road (0 = leftmost)                                */
    int32_t n_in_road;   /* lanes in that road                                                  */
    float ax, ay;        /* straight: start ; circular: centre                                  */
    float bx, by;        /* straight: unit direction ; circular: radius, start_phase (wrapped)  */
    float length, width;
    float end_phase;     /* circular: start_phase -/+ angle (not wrapped)                       */
    float dirsign;       /* circular: -1 clockwise, +1 counter-clockwise ; straight: 0          */
    float angle;         /* circular: swept angle (rad)                                         */
    float heading;       /* straight: atan2(dir)                                                */
    float sx, sy, ex, ey;/* centre-line start / end points                                      */
    float x0, y0, x1, y1;/* AABB of the lane's convex hull                                      */
    int32_t hull_off;    /* offset (vertices) into MdWorld.hull_xy                              */
    int32_t hull_n;      /* vertices in the hull (CCW)                                          */
    float end_phase_w;   /* circular: wrap_to_pi(end_phase)                                     */
    float speed_limit;   /* AbstractLane.speed_limit (lane/abs_lane.py:22-29): 3 on the lanes of a TollGate block
                            (pgblock/tollgate.py:18,64-68), 1000 elsewhere -- read by BaseVehicle.overspeed
                            (base_vehicle.py:909-911), which compares it with the speed in km/h                */
    float elx, ely;      /* unit lateral (right-hand) vector at the lane end: position(L, lat) = e + lat*el */
    float spare[4];
    float hull4[8];      /* hull_n == 4 (straight lanes): the four hull vertices inline, so that the
                            containment test needs no second table lookup                         */
} MdLane;

/* 32-byte road record (component/road_network/road.py). */
typedef struct MdRoad {
    int32_t first_lane;  /* map-local id of lane 0                                              */
    int32_t n_lanes;
    int32_t start_node, end_node; /* map-local node ids                                         */
    int32_t negative;    /* Road.is_negative_road()                                             */
    int32_t block;       /* block index that created the road                                   */
    int32_t block_kind;  /* Road.block_ID() (road_network/road.py:42-47): the ASCII code of the block type's letter
                            taken from the end node (start node for a negative road), '>' for the first block */
    int32_t spare;
} MdRoad;

/* 32-byte grid header per map: uniform grid over static geometry (lanes' hulls and quads). */
typedef struct MdGrid {
    float x0, y0;        /* lower-left corner                                                   */
    float inv_cell;      /* 1 / cell size                                                       */
    int32_t nx, ny;
    int32_t cell_base;   /* offset of this map's (nx*ny+1) entries in MdWorld.cell_start        */
    int32_t spare[2];
} MdGrid;

/* 48-byte polyline segment (utils/interpolating_line.py:96-140 `segment_property`): what PointLane --
 * the SDC's reference trajectory and every TrajectoryIDMPolicy route -- is made of.  Scenario mode only. */
typedef struct MdSeg {
    float sx, sy;        /* start_point                                                          */
    float ex, ey;        /* end_point                                                            */
    float dx, dy;        /* direction (unit)                                                     */
    float len;           /* length                                                               */
    float heading;       /* atan2(end - start)                                                   */
    float cum;           /* sum of the lengths of the segments before this one                   */
    float spare[3];
} MdSeg;

/* MdWorld.track_meta[n][4] (scenario mode): first valid run [t0, t1) of the track in slot n and MD_TM_* bits */
#define MD_TM_MOVING 1      /* not a "static car": max std of its valid positions > STATIC_THRESHOLD (scenario_traffic_manager.py:176-181) */
#define MD_TM_LENGTH_OK 2   /* start -> end of the run farther than IDM_CREATE_MIN_LENGTH (:226-228)   */
#define MD_TM_NEVER 4       /* never spawned: noise object (:286-291), unsupported type, empty track    */

/* Scenario-mode slot state lives in MdNav: ck0 = MD_SC_* mode, timer = TrajectoryIDMPolicy.policy_index */
#define MD_SC_ABSENT 0      /* not in the world (not yet valid, removed, filtered): a spawn is tried every step */
#define MD_SC_REPLAY 1      /* ReplayTrafficParticipantPolicy: pose from the track                        */
#define MD_SC_IDM 2         /* TrajectoryIDMPolicy: drives along its own recorded path                    */
#define MD_SC_ARRIVED 3     /* IDM vehicle inside its destination region: removed at the end of this step */
#define MD_MA_DEFAULT 0      /* MdConfig.ma_kind */
#define MD_MA_TOLLGATE 1
#define MD_MA_PARKING_LOT 2
#define MD_MA_RACING 3
#define MD_IDLE_WINDOW 100     /* MultiAgentRacingEnv: steps of movement an agent is judged idle over (marl_racing_env.py:342,392-398) */

/* Static world: everything fixed between resets. */
typedef struct MdWorld {
    int32_t n_maps;
    int32_t n_envs;
    const int32_t* env_map;    /* [n_envs] map id per env                                       */
    const int32_t* lane_off;   /* [n_maps+1] CSR into lanes                                     */
    const MdLane* lanes;
    const float* hull_xy;      /* [2 * n_hull_vertices]                                         */
    const int32_t* road_off;   /* [n_maps+1] CSR into roads                                     */
    const MdRoad* roads;
    const int32_t* quad_off;   /* [n_maps+1] CSR into quads                                     */
    const float* quads;        /* [n_quads][8] four CCW vertices x0,y0..x3,y3                    */
    const int32_t* quad_kind;  /* [n_quads] MD_Q_*                                              */
    const MdGrid* grid;        /* [n_maps]                                                      */
    const int32_t* cell_start; /* CSR per cell into cell_items                                  */
    const int32_t* cell_items; /* item = lane id (>=0, map-local) or ~quad id (<0, map-local)   */
    const int32_t* node_adj_off;   /* [total_nodes+1] CSR: out-neighbours of each node (per map, via node_off) */
    const int32_t* node_adj;       /* [..] pairs flattened: (to_node, road_id)                  */
    const int32_t* node_off;       /* [n_maps+1] offset of each map's nodes in node_adj_off     */
    const float* beam_cs;      /* [n_beams][2] cos/sin of (2*pi*i/n_beams + phase)              */
    int32_t max_lanes;         /* largest lane count of any map (sizes the kernels' LDS lane table) */
    int32_t max_roads;         /* largest road count of any map                                 */
    /* multi-agent respawn tables (SpawnManager.safe_spawn_places, manager/spawn_manager.py:123-161); NULL / 0
     * for single-agent envs */
    const int32_t* spawn_off;  /* [n_maps+1] CSR into the spawn-place arrays                     */
    const float* spawn_place;  /* [n_places][8]: x, y, cos, sin, heading, 0, 0, 0 (slot 0 of each spawn road/lane) */
    const int32_t* spawn_lane; /* [n_places] lane id (map-local)                                 */
    const int32_t* spawn_route;/* [n_places][n_dest][2][MD_ROUTE_LEN]: checkpoint nodes, then roads */
    const int32_t* spawn_route_meta; /* [n_places][n_dest][2]: route_len, final_lane              */
    int32_t n_dest;            /* destinations per spawn place                                   */
    int32_t n_vclass;          /* entries of vclass (0 = respawned agents keep the slot's vehicle) */
    /* scenario mode (traffic_mode 4; NULL otherwise): polylines per mover slot, the SDC route's checkpoints */
    const int32_t* poly_off;   /* [n_envs * cap + 1] CSR into segs: slot 0 = the SDC's reference trajectory
                                  (ScenarioMapManager.current_sdc_route, manager/scenario_map_manager.py:49-63), slot j = the
                                  path of track j over its first valid run (get_idm_route, scenario/parse_object_state.py:19-21) */
    const MdSeg* segs;
    const int32_t* polyv_off;  /* [n_envs * cap + 1] CSR into polyv: outline of that path at width 2 (PointLane.auto_generate_polygon,
                                  component/lane/point_lane.py:60-106): what lane.point_on_lane tests                              */
    const float* polyv;        /* [n_vertices][2]                                                */
    const int32_t* ckpt_off;   /* [n_envs + 1] CSR into ckpt_xy: TrajectoryNavigation.checkpoints (trajectory_navigation.py:96-103) */
    const float* ckpt_xy;      /* [n_ckpt][2]                                                    */
    const int32_t* track_meta; /* [n_envs * cap][4]: t0, t1, MD_TM_* bits, 0                       */
    /* multi-agent + random_agent_model: the vehicle classes a (re)spawned agent is drawn from, uniformly
     * (random_vehicle_type, component/vehicle/vehicle_type.py:269-281): [n_vclass][12] = MdParam (8 floats), half length,
     * half width, 0, 0 */
    const float* vclass;
    /* scenario mode, optional (NULL = derived in the kernel): per mover slot [8] = the end point of its polyline
     * (PointLane.end = position(length, 0)), the bounding box xmin, ymin, xmax, ymax of its outline polygon, 0, 0 --
     * both are functions of the static tables above, kept so that the reactive policy's arrival test and its
     * "is this object on my path" test need no pass over the polyline */
    const float* poly_aux;
    /* optional: beam tables ([n][2] cos, sin; the fans start 90 deg off the heading, SideDetector.__init__
     * distance_detector.py:197) of the side detector (MdConfig.n_side beams) and the lane-line detector (n_lane_line).
     * Where set, md_step fills those observation dims itself (scenario mode: on waves that would otherwise idle) and a
     * separate md_line_detector call is not needed; NULL = the caller runs md_line_detector after md_step. */
    const float* side_beam_cs;
    const float* ll_beam_cs;
    /* optional, derived from `quads` / `quad_kind`: [n_quads][4] = centre x, y, radius of a circle that contains the quad,
     * and its MD_Q_* kind (the integer's bit pattern in the float) -- the detectors cull on these 16 bytes and read the
     * 32-byte quad only for the few (quad, beam) pairs that can meet.  NULL = they derive the circle from the quad. */
    const float* quad_ball;
    /* scenario mode, optional: ALL valid runs of every track -- run_off [n_envs * cap + 1] CSR into runs [n_runs][2] = [t0, t1)
     * (get_max_valid_indicis at any frame of the run, scenario/parse_object_state.py:8-16).  track_meta holds the first run only;
     * with these tables (and MdState.route_*) a vehicle spawned at a frame other than its first run's start gets its reactive
     * policy on a route cut at the spawn frame, as scenario_traffic_manager.py:216-236 does; NULL = such spawns are replayed. */
    const int32_t* run_off;
    const int32_t* runs;
    /* scenario mode, optional (NULL = every projection walks all pieces): per mover slot, for each group of MD_POLY_GROUP consecutive
     * pieces of its static polyline, [4] = centre x, y and radius of a circle that contains the group's pieces (with a margin
     * >= 1e-3 m for the rounding to float), 0 -- poly_ball_off [n_envs * cap + 1] is the CSR into poly_ball (slot n has
     * ceil(n_pieces / MD_POLY_GROUP) groups).  The projections (InterpolatingLine.local_coordinates = arg-min over the pieces)
     * use them as an EXACT cull: only groups whose circle comes within the smallest "farthest point of a circle" are evaluated;
     * the arg-min, its tie-break and every number derived from it are those of the full walk. */
    const float* poly_ball;
    const int32_t* poly_ball_off;
} MdWorld;

/* Dynamic state: one entry per mover unless noted. */
typedef struct MdState {
    MdShape* shape;
    MdDyn* dyn;
    MdParam* param;            /* constant per slot, except multi-agent envs with random_agent_model: a respawn draws a new
                                  vehicle class (VehicleAgentManager._create_agents, manager/agent_manager.py:37-43) */
    MdNav* nav;
    MdPid* pid;
    float* action;             /* [N][2] steering, throttle_brake in [-1,1] (agents: caller-written) */
    int32_t* route_nodes;      /* [N][MD_ROUTE_LEN] checkpoints as map-local node ids (rewritten on respawn) */
    int32_t* route_roads;      /* [N][MD_ROUTE_LEN] road id of (node j, node j+1), -1 past end   */
    int32_t* final_lane;       /* [N] map-local id of navigation.final_lane                      */
    const int32_t* idm_rand;   /* [N][MD_IDM_RAND] pre-drawn np_random.randint(0, 25) values of each IDMPolicy
                                  (policy/idm_policy.py:285), consumed cyclically                 */
    uint32_t* flags;           /* [N] MD_FL_*                                                    */
    float* obs;                /* [n_envs*agents_per_env][obs_dim]                               */
    float* reward;             /* [n_envs*agents_per_env]                                        */
    float* cost;               /* [n_envs*agents_per_env]                                        */
    float* step_info;          /* [n_envs*agents_per_env][8] step_reward, velocity, step_energy, episode_energy, episode_reward, total_cost, long, episode_length */
    int32_t* need_reset;       /* [n_envs] 1 = restore the env from the snapshot before stepping  */
    /* reset snapshot (same layouts) restored by md_step when need_reset[e] != 0                  */
    const MdShape* shape0;
    const MdDyn* dyn0;
    const MdNav* nav0;
    const MdPid* pid0;
    /* multi-agent only */
    const int32_t* route_nodes0; /* reset snapshot of the routes (respawn rewrites them)            */
    const int32_t* route_roads0;
    const int32_t* final_lane0;
    uint32_t* rng;             /* [n_envs] xorshift32 state of the respawn draws (the reference draws these
                                  from an unseeded RandomState: spawn_manager.py:217-220)            */
    int32_t* env_steps;        /* [n_envs] engine.episode_step                                   */
    int32_t* agent_id;         /* [N] running agent number held by the slot ("agent{k}")          */
    int32_t* next_agent_id;    /* [n_envs] VehicleAgentManager.next_agent_count                   */
    /* optional: this step's agent actions straight from the caller's buffer, [n_envs * agents_per_env][2]
     * (steering, throttle).  NULL = the agents' actions are already in `action` (slots 0..A-1 of each env).
     * Saves the scatter into the per-slot array; md_step still writes the sanitised values to `action`. */
    const float* agent_action;
    /* traffic_mode 3 (replay): recorded poses of every non-agent slot, frame-major:
     * track_shape[t * n_envs * cap + n], track_dyn[2 * (t * n_envs * cap + n)] = (heading, speed);
     * frame t is the state at the END of the t-th step of the episode (frame 0 = reset state).  The role of
     * ReplayTrafficParticipantPolicy.act (policy/replay_policy.py:43-67): position / heading / velocity set
     * from the track at the current episode step, no reaction to the agents. */
    const MdShape* track_shape;
    const float* track_dyn;
    /* optional (NULL = not wanted; required when MdConfig.num_others > 0) */
    uint64_t* detected;        /* [n_envs * agents_per_env][2] bit j of the 128-bit set: some beam of the agent's lidar
                                  hit the mover in slot j first -- the `detected_objects` half of Lidar.perceive's
                                  return value (component/sensors/lidar.py:49-73); written by md_step           */
    /* reserved (NULL): was the work space of the phase-per-launch step, removed in ABI v10 (slower at every batch size) */
    uint32_t* scratch;
    const MdParam* param0;     /* reset snapshot of `param` (multi-agent + random_agent_model only; NULL otherwise) */
    /* optional: [n_envs * agents_per_env] 4-byte words, one per agent, written with ONE store (ABI v10; v9 held two bytes):
     *   byte 0 = terminated, byte 1 = truncated -- the MD_FL_TERMINATED / MD_FL_TRUNCATED bits of the agent's flag word as the
     *            two booleans step() returns (envs/base_env.py:586-612), so that the caller needs no kernel to extract them;
     *   bytes 2-3 = the agent's whole step flag word MD_FL_* (all sixteen bits: crash_*, on-line, out_of_road, arrive_dest,
     *            max_step ...), little endian -- what `info` reports.
     * Together with obs and reward this is everything a learner acts on; a caller that lays obs | reward | done_out out in one
     * allocation gathers a rank's whole step output with one collective (metadrive_ped_amd/sharding.py). */
    uint8_t* done_out;
    /* scenario mode, optional (all four or none; needs MdWorld.run_off / runs and MdConfig.route_seg_cap / route_vert_cap):
     * routes built on the device when a track is (re)spawned with a reactive policy at a frame k that is not the start of its
     * first valid run -- PointLane(positions[k:end of the run]) (get_idm_route, scenario/parse_object_state.py:19-21), built by
     * md_step right after the spawn:
     *   route_n     [n_envs * cap][4]  pieces (0 = the slot follows its static polyline MdWorld.segs), outline vertices, k, 0
     *   route_segs  [n_envs * cap][route_seg_cap]      the pieces
     *   route_verts [n_envs * cap][route_vert_cap][2]  the outline at width 2
     *   route_aux   [n_envs * cap][8]                  as MdWorld.poly_aux */
    int32_t* route_n;
    MdSeg* route_segs;
    float* route_verts;
    float* route_aux;
    /* MultiAgentRacingEnv only (NULL elsewhere): [n_envs * agents_per_env][MD_IDLE_WINDOW] the |longitudinal progress| of the
     * agent's last steps, a ring written at index (MdNav.toll_state mod MD_IDLE_WINDOW); MdNav.toll_state counts the entries
     * (movement_between_steps, marl_racing_env.py:342,415) */
    float* idle_ring;
} MdState;

typedef struct MdConfig {
    int32_t struct_size;       /* sizeof(MdConfig) as the caller sees it (ABI check)             */
    int32_t n_envs;
    int32_t agents_per_env;
    int32_t cap;               /* mover slots per env                                            */
    int32_t n_beams;           /* lidar num_lasers (0 = lidar off)                               */
    int32_t obs_dim;           /* [2 if random_agent_model] + (n_side or 2) + 6 + (n_lane_line or 1) + 10 + [others] + n_beams (= 19 + n_beams by default) */
    int32_t substeps;          /* decision_repeat (envs/base_env.py:186)  = 5                     */
    int32_t horizon;           /* 0 = None                                                       */
    float dt;                  /* physics_world_step_size (envs/base_env.py:185) = 0.02          */
    float lidar_range;         /* vehicle_config.lidar.distance = 50                             */
    /* reward scheme (envs/metadrive_env.py:69-77) */
    float success_reward, out_of_road_penalty, crash_vehicle_penalty, crash_object_penalty;
    float driving_reward, speed_reward;
    /* cost scheme (envs/metadrive_env.py:79-82) */
    float crash_vehicle_cost, crash_object_cost, out_of_road_cost;
    /* termination scheme (envs/metadrive_env.py:84-89, base_env.py:80-81) */
    int32_t use_lateral_reward;
    int32_t out_of_route_done, on_continuous_line_done;
    int32_t crash_vehicle_done, crash_object_done, crash_human_done;
    int32_t truncate_as_terminate;
    int32_t traffic_mode;      /* 0 trigger, 1 respawn, 2 hybrid (manager/traffic_manager.py:20-29), 3 replay of recorded tracks,
                                * 4 scenario (ScenarioEnv: envs/scenario_env.py, manager/scenario_traffic_manager.py)              */
    int32_t enable_idm_lane_change;
    int32_t auto_reset;        /* 1: md_step restores envs whose need_reset flag is set          */
    float max_lane_width;      /* BaseMap.MAX_LANE_WIDTH 4.5                                     */
    float total_width;         /* (MAX_LANE_NUM+1)*MAX_LANE_WIDTH = 18 (state_obs.py:92)          */
    float curve_radius_max;    /* BlockParameterSpace.CURVE radius max = 60                      */
    float curve_angle_max;     /* BlockParameterSpace.CURVE angle max = 135 (deg)                */
    /* multi-agent (envs/marl_envs/multi_agent_metadrive.py:12-61) */
    int32_t is_multi_agent;
    int32_t delay_done;        /* steps a finished vehicle stays in place as a static body (25)  */
    int32_t allow_respawn;
    int32_t crash_done, out_of_road_done;
    /* observation layout (obs/state_obs.py:64-151): [side cloud n_side | left,right] + 5 + yaw +
     * [lane-line cloud n_lane_line | lateral] + navi 10 + lidar n_beams */
    int32_t n_side;            /* side_detector.num_lasers (0 = off: two distance dims instead)   */
    int32_t n_lane_line;       /* lane_line_detector.num_lasers (0 = off: one lateral dim instead) */
    int32_t num_others;        /* vehicle_config.lidar.num_others: nearest detected vehicles in the obs (0 = none) */
    int32_t add_others_navi;   /* vehicle_config.lidar.add_others_navi                            */
    int32_t track_len;         /* frames in MdState.track_* (traffic_mode 3); later steps hold the last frame */
    int32_t random_agent_model;/* 1: two extra leading obs dims, length / 10 and width / 2.5 (obs/state_obs.py:70-75) */
    int32_t agent_idm;         /* 1: config agent_policy = IDMPolicy (envs/base_env.py:53, manager/agent_manager.py:37-70): the
                                * agents are driven by the IDM / PID policy of the traffic; MdState.agent_action is not read.
                                * Single-agent envs only. */
    int32_t enable_reverse;    /* vehicle_config.enable_reverse: an agent's negative throttle drives it backwards instead of
                                * braking (base_vehicle.py:476-484) */
    /* scenario mode (traffic_mode 4): ScenarioEnv reward / cost / termination scheme (envs/scenario_env.py:21-95) */
    float on_lane_line_penalty, crash_human_penalty, steering_range_penalty, heading_penalty, lateral_penalty;
    float max_lateral_dist, crash_human_cost;
    int32_t no_negative_reward, relax_out_of_road_done;
    int32_t reactive_traffic, filter_overlapping_car, no_static_vehicles;
    int32_t allowed_more_steps;/* 0 = None                                                       */
    int32_t scenario_length;   /* frames of the scenarios (data_manager.current_scenario_length)  */
    int32_t step_kernel;       /* md_step of single-agent envs: 0 = one 4-wave workgroup per env, 1 = one wave per env (faster when the
                                * batch shares few distinct maps: the host picks, metadrive_ped_amd/engine.py).  Same results bit for
                                * bit; a machine-mapping choice, no reference counterpart. */
    /* which multi-agent env's rules md_observe applies (is_multi_agent only): 0 = MultiAgentMetaDrive and the envs that keep
     * its reward / done / observation, 1 = MultiAgentTollgateEnv (envs/marl_envs/marl_tollgate.py:181-266: no navigation
     * dims, two toll dims after the lidar cloud, overspeed penalty inside the toll block, minimum stay), 2 =
     * MultiAgentParkingLotEnv (envs/marl_envs/marl_parking_lot.py), 3 = MultiAgentRacingEnv (envs/marl_envs/marl_racing_env.py:
     * 338-441: out of road = more than 5 m behind the start of its lane, idle detection, sidewalk / idle penalties) */
    int32_t ma_kind;
    int32_t min_pass_steps;    /* vehicle_config.min_pass_steps (marl_tollgate.py:28)                     */
    float overspeed_penalty;   /* marl_tollgate.py:25                                                     */
    int32_t n_parking;         /* parking lot env: number of parking spaces (destinations 0..n_parking-1 of the spawn tables) */
    float side_range, ll_range;     /* detector ranges in m (vehicle_config.side_detector / lane_line_detector "distance") */
    uint32_t side_mask, ll_mask;    /* which MD_Q_* kinds each detector sees (bit k = kind k), as md_line_detector's kind_mask */
    int32_t route_seg_cap;     /* pieces per slot in MdState.route_segs (>= the longest valid run's frames - 1)          */
    int32_t route_vert_cap;    /* vertices per slot in MdState.route_verts (>= 2 * ceil(longest run's path length + 1) + 4) */
    int32_t ego_replay;        /* scenario mode, agent_policy = ReplayEgoCarPolicy (policy/replay_policy.py:70-82): the agent is put on
                                * frame k of the SDC track (MdState.track_* slot 0) instead of being integrated; actions are ignored */
    /* MultiAgentRacingEnv (ma_kind 3; RACING_CONFIG, marl_racing_env.py:46-63) */
    float crash_sidewalk_penalty, idle_penalty;
    int32_t idle_done, crash_sidewalk_done;
} MdConfig;

/* ---- entry points ------------------------------------------------------------------------- */

/* ABI self-description: fills sizes[0..9] = sizeof(MdShape, MdDyn, MdParam, MdNav, MdPid, MdLane,
 * MdRoad, MdGrid, MdWorld, MdState, MdConfig)[i]; returns MD_ABI_VERSION. */
int md_abi(int32_t* sizes, int n);
const char* md_last_error(void);

/* Arithmetic self-test hook: out[i] = f_op(a[i], b[i]) evaluated ON THE DEVICE with the shared
 * float32 formulas of md_math.h / md_geom.h (op: 0 sin, 1 cos, 2 atan2(a,b), 3 acos, 4 exp, 5 a/b,
 * 6 sqrt, 7 wrap_to_pi, 8 asin, 9 norm(a,b), 10 energy model of base_vehicle.py:255-271 with a = speed
 * km/h, b = distance m).  The parity tests compare it bit for bit with the host build of the same
 * formulas: this is what makes "bit-exact booleans" a checkable claim. */
int md_probe_math(int op, const float* a, const float* b, float* out, int n, void* stream);

/* Measurement hook ("measure the attainable number on the box with a HIP stream-copy
 * kernel"): dst[i] = src[i] over nbytes (a multiple of 16, both 16-byte aligned) with 16-byte
 * accesses, grid-stride; moves 2 * nbytes of HBM traffic.  No reference counterpart. */
int md_probe_stream_copy(void* dst, const void* src, size_t nbytes, void* stream);

/* Lidar: Lidar.perceive / perceive()  component/sensors/lidar.py:49-73,
 * component/sensors/distance_detector.py:27-85, utils/math.py:76-81.
 * For agent a of env e: out[(e*A+a)*out_stride + out_offset + i] = closest hit fraction in [0,1]
 * of beam i against the env's alive movers except its own chassis; 1.0 = no hit. */
int md_lidar(const MdWorld* w, const MdState* s, const MdConfig* c, float* out, int out_stride, int out_offset,
             void* stream);

/* Lidar.perceive complete (component/sensors/lidar.py:49-73): md_lidar's cloud points plus `detected_objects` --
 * detected[(e*A+a)*2 .. +1] = the 128-bit set of the slots some beam of agent a hit first.  Needs only MdState.shape,
 * MdWorld.beam_cs and the sizes in MdConfig: the sensor-level plug (metadrive_ped_amd/sensors.py: BatchedLidar). */
int md_lidar_detect(const MdWorld* w, const MdState* s, const MdConfig* c, float* out, int out_stride, int out_offset,
                    uint64_t* detected, void* stream);

/* Side / lane-line detectors: DistanceDetector.perceive with the static line boxes as targets
 * (component/sensors/distance_detector.py:194-209; obs/state_obs.py:77-86,129-140).
 * kind_mask selects MD_Q_* kinds (bit k set = kind k is a target). beam table = beam_cs. */
int md_line_detector(const MdWorld* w, const MdState* s, const MdConfig* c, const float* beam_cs, int n_beams,
                     float range, uint32_t kind_mask, float* out, int out_stride, int out_offset, void* stream);
/* The same for TWO fans at once -- the side detector and the lane-line detector of one observation (obs/state_obs.py:77-86 and
 * :129-140 call SideDetector.perceive and LaneLineDetector.perceive one after the other): one launch, one pass over the map's
 * line pieces; fan k writes its n_beams_k fractions at out_offset_k of every agent's row.  n_beams0 + n_beams1 <= 255. */
int md_line_detectors(const MdWorld* w, const MdState* s, const MdConfig* c, const float* beam_cs0, int n_beams0, float range0,
                      uint32_t kind_mask0, int out_offset0, const float* beam_cs1, int n_beams1, float range1, uint32_t kind_mask1,
                      int out_offset1, float* out, int out_stride, void* stream);

/* random_traffic (envs/metadrive_env.py:46 "random_traffic": PGTrafficManager does not re-seed its stream at reset,
 * manager/traffic_manager.py:335-337, so every episode sees other traffic) for envs that reset themselves inside md_step: `staged`
 * holds n_draws host-built draws of the traffic, draw-major -- shape0 / dyn0 / nav0 / pid0 and the per-slot constants param,
 * route_nodes, route_roads, final_lane, idm_rand, each [n_draws][n_envs * cap] in the layout of the MdState array of that name
 * (other fields ignored; NULL = that array is not swapped).  For every env with need_reset != 0 (its episode has ended; md_step
 * restores it at the next step) draw_idx[e] advances by one (mod n_draws) and that draw's rows replace the env's snapshot and
 * constants (the *0 twins of the constants too, where the state has them).  Call it after md_step, on the same stream. */
int md_swap_draw(const MdState* s, const MdState* staged, const MdConfig* c, int n_draws, int32_t* draw_idx, void* stream);

/* Dynamics: BaseVehicle.before_step/_set_action/_apply_throttle_brake (component/vehicle/
 * base_vehicle.py:211-232,447-484) + EngineCore.step_physics_world x decision_repeat
 * (engine/core/engine_core.py:350-352, engine/base_engine.py:417-445). Kinematic bicycle. */
int md_integrate(const MdWorld* w, const MdState* s, const MdConfig* c, void* stream);

/* Lane localisation + checkpoint advance: ray_localization (utils/pg/utils.py:151-203),
 * NodeNetworkNavigation._get_current_lane/_update_current_lane/_update_target_checkpoints
 * (component/navigation_module/node_network_navigation.py:181-241,294-304). */
int md_localize(const MdWorld* w, const MdState* s, const MdConfig* c, void* stream);

/* Contacts: collision_callback (engine/core/collision_callback.py:5-42) + BaseVehicle._state_check
 * (component/vehicle/base_vehicle.py:700-767) + rect_region_detection (utils/pg/utils.py:213-256). */
int md_contacts(const MdWorld* w, const MdState* s, const MdConfig* c, void* stream);

/* Observation(19 dims) + reward + cost + done: StateObservation.vehicle_state (obs/state_obs.py:64-151),
 * _get_info_for_checkpoint (node_network_navigation.py:243-292), MetaDriveEnv.reward_function /
 * cost_function / done_function (envs/metadrive_env.py:128-279), _get_step_return
 * (envs/base_env.py:586-623). */
int md_observe(const MdWorld* w, const MdState* s, const MdConfig* c, void* stream);

/* IDM traffic policy + trigger activation: PGTrafficManager.before_step
 * (manager/traffic_manager.py:74-92), IDMPolicy.act (policy/idm_policy.py:235-402). */
int md_idm(const MdWorld* w, const MdState* s, const MdConfig* c, void* stream);

/* Traffic removal after the step: PGTrafficManager.after_step (manager/traffic_manager.py:94-122). */
int md_traffic_after_step(const MdWorld* w, const MdState* s, const MdConfig* c, void* stream);

/* Multi-agent lifecycle at the start of a step: finished agents become static bodies for delay_done
 * steps then vanish, free slots respawn at free spawn places: MultiAgentMetaDrive.step /
 * _after_vehicle_done / _respawn_single_vehicle (envs/marl_envs/multi_agent_metadrive.py:130-212),
 * VehicleAgentManager._finish / before_step (manager/agent_manager.py:115-128,189-202),
 * SpawnManager.get_available_respawn_places (manager/spawn_manager.py:163-209). */
int md_lifecycle(const MdWorld* w, const MdState* s, const MdConfig* c, void* stream);

/* One whole env.step() for all envs: BaseEnv.step (envs/base_env.py:426-463,586-623) =
 * [auto-reset] -> idm -> integrate -> localize -> contacts -> traffic_after_step -> observe -> lidar,
 * fused into ONE launch (one workgroup per env, state staged in LDS). */
int md_step(const MdWorld* w, const MdState* s, const MdConfig* c, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDSTEP_H */

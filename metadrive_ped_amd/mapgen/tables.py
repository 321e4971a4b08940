"""Flatten generated maps into the structure-of-arrays tables the kernels read (include/mdstep.h).

Per map: MdLane records (+ convex hull vertices), MdRoad records, node adjacency (CSR), static
quads (lane-line ghost boxes and side-walk strips) and a uniform grid over lanes' hulls and quads.
`MapTables.concat` stacks many maps into one CSR set indexed by map id.

Geometry definitions restated from the reference:
  lane hull          block/base_block.py:431-468 (convex hull of lane.polygon)
  line ghost boxes   block/base_block.py:470-519 (half width LANE_LINE_WIDTH/4 = 0.0375 m),
                     pgblock/pg_block.py:259-292 (continuous: 4 m pieces; broken: 1.5 m stripe per 3 m),
                     pgblock/pg_block.py:334-362 (which side of which lane gets a line)
  side walk          pgblock/pg_block.py:294-332 (strip from w/2+0.2 to w/2+2.2, 3 m stations)
  map region         constants.py:496-508 (TerrainProperty.point_in_map)
"""
import math

import numpy as np

from metadrive_ped_amd import abi
from metadrive_ped_amd.mapgen.lanes import (COLOR_GREY, LINE_BROKEN, LINE_CONTINUOUS, LINE_GUARDRAIL, LINE_NONE,
                                            LINE_SIDE, CircularLane, convex_hull, wrap_to_pi)
from metadrive_ped_amd.mapgen.pg import DECORATION, is_negative_road

LANE_SEGMENT_LENGTH = 4.0
STRIPE_LENGTH = 1.5
LANE_LINE_WIDTH = 0.15
SIDEWALK_LENGTH = 3.0
SIDEWALK_WIDTH = 2.0
MAP_REGION_SIZE = 1024.0  # BASE_DEFAULT_CONFIG map_region_size (envs/base_env.py)
GRID_CELL = 8.0
GRID_MARGIN = 0.25  # cells get every item whose AABB (+ margin) touches them


def _ccw(q):
    q = np.asarray(q, dtype=np.float64)
    area = 0.0
    for i in range(4):
        j = (i + 1) % 4
        area += q[i, 0] * q[j, 1] - q[j, 0] * q[i, 1]
    return q if area >= 0 else q[::-1].copy()


def _line_box(p0, p1, region=None):
    """Ghost box of one lane-line piece as a CCW quad, or None (zero length / outside the map)."""
    p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
    d = p1 - p0
    length = math.sqrt(d[0] ** 2 + d[1] ** 2)
    mid = (p0 + p1) / 2
    half = (MAP_REGION_SIZE if region is None else region) / 2
    if not (-half <= mid[0] <= half and -half <= mid[1] <= half):
        return None
    if length <= 0:
        return None
    u = d / length
    n = np.array([-u[1], u[0]]) * (LANE_LINE_WIDTH / 4)
    return _ccw([p0 - n, p1 - n, p1 + n, p0 + n])


def lane_line_quads(lane, construct_left_right):
    """Quads + kinds for the lines of one lane; also the side-walk quads when a line is SIDE."""
    quads, kinds = [], []
    has_sidewalk = False       # PGBlock.sidewalks is keyed by the lane: the first SIDE / GUARDRAIL line of a lane makes its strip
    for side, ltype, color, need in zip((-1, 1), lane.line_types, lane.line_colors, construct_left_right):
        if not need or ltype == LINE_NONE:
            continue
        lateral = side * lane.width / 2
        if ltype in (LINE_CONTINUOUS, LINE_SIDE, LINE_GUARDRAIL):
            kind = abi.Q_LINE_WHITE_CONT if color == COLOR_GREY else abi.Q_LINE_YELLOW_CONT
            n = int(lane.length / LANE_SEGMENT_LENGTH)
            pieces = []
            if n == 0:
                pieces.append((lane.position(0, lateral), lane.position(lane.length, lateral)))
            for k in range(n):
                a = lane.position(LANE_SEGMENT_LENGTH * k, lateral)
                b = lane.position(lane.length, lateral) if k == n - 1 else lane.position((k + 1) * LANE_SEGMENT_LENGTH, lateral)
                pieces.append((a, b))
            for a, b in pieces:
                q = _line_box(a, b)
                if q is not None:
                    quads.append(q)
                    kinds.append(kind)
            if ltype in (LINE_SIDE, LINE_GUARDRAIL) and not has_sidewalk:
                # SIDE: the sidewalk on the lane's right; GUARDRAIL: the same strip on whichever side the line is
                # (_generate_sidewalk_from_line(lane, GUARDRAIL_HEIGHT, lateral_direction=idx), pgblock/pg_block.py:341-353) --
                # a contact raises crash_sidewalk either way (base_vehicle.py:757-767)
                has_sidewalk = True
                for q in sidewalk_quads(lane, side if ltype == LINE_GUARDRAIL else 1):
                    quads.append(q)
                    kinds.append(abi.Q_SIDEWALK)
        elif ltype == LINE_BROKEN:
            n = int(lane.length / (2 * STRIPE_LENGTH))
            for k in range(n):
                a = lane.position(k * STRIPE_LENGTH * 2, lateral)
                b = lane.position(k * STRIPE_LENGTH * 2 + STRIPE_LENGTH, lateral)
                if k == n - 1:
                    b = lane.position(lane.length - STRIPE_LENGTH, lateral)
                q = _line_box(a, b)
                if q is not None:
                    quads.append(q)
                    kinds.append(abi.Q_LINE_BROKEN)
        else:
            raise ValueError("line type {}".format(ltype))
    return quads, kinds


def sidewalk_quads(lane, lateral_direction=1):
    start_lat = lane.width / 2 + 0.2
    side_lat = start_lat + SIDEWALK_WIDTH
    start_lat *= lateral_direction
    side_lat *= lateral_direction
    if lane.radius != 0 and side_lat > lane.radius:
        return []
    longs = np.arange(0, lane.length + SIDEWALK_LENGTH, SIDEWALK_LENGTH)
    longs = [min(lane.length + 0.1, s) for s in longs]
    out = []
    for s0, s1 in zip(longs[:-1], longs[1:]):
        if s1 <= s0:
            continue
        q = _ccw([lane.position(s0, start_lat), lane.position(s1, start_lat), lane.position(s1, side_lat),
                  lane.position(s0, side_lat)])
        out.append(q)
    return out


def road_block_id(start_node, end_node):
    """Road.block_ID() (component/road_network/road.py:42-47) as an ASCII code: the block type's letter, read from the end
    node -- from the start node for a negative road --, '>' for the first block's nodes; 0 where the name holds none
    (decoration lanes)."""
    import re
    node = start_node if is_negative_road(end_node) else end_node
    if ">" in node:
        return ord(">")
    m = re.search("[a-zA-Z$]", node)
    return ord(m.group(0)) if m else 0


class StaticTables:
    """The static tables of a map that consists of line bodies only (scenario scenes: their lanes are polylines of
    their own, nothing is localised on the grid): quads + grid, and placeholders where WorldTables expects a map's
    lanes / roads / nodes."""
    _build_grid = None   # bound below

    def __init__(self, quads, kinds):
        self.respawn = None
        self.lanes = np.zeros(1, dtype=abi.LANE_DT)
        self.lanes["x0"], self.lanes["x1"] = 1.0, -1.0          # an empty box: the grid skips it, nothing ever hits it
        self.hull_xy = np.zeros((1, 2), np.float32)
        self.roads = np.zeros(1, dtype=abi.ROAD_DT)
        self.quads = np.asarray(quads, dtype=np.float32).reshape(-1, 8) if len(quads) else np.zeros((0, 8), np.float32)
        self.quad_kind = np.asarray(kinds, dtype=np.int32)
        self.node_adj_off = np.zeros(2, dtype=np.int32)
        self.node_adj = np.zeros((0, 2), np.int32)
        self._build_grid()


class MapTables:
    """numpy tables of ONE map (all offsets map-local)."""
    def __init__(self, pg_map):
        self.pg_map = pg_map
        self.respawn = None    # respawn-lane tables (traffic modes 'respawn' / 'hybrid'), see respawn_tables
        net = pg_map.net
        nodes = {}
        roads, lanes_flat = [], []
        self.lane_id = {}      # (a, b, i) -> lane id
        self.road_id = {}      # (a, b) -> road id
        self.lane_objs = []

        def node_id(name):
            if name not in nodes:
                nodes[name] = len(nodes)
            return nodes[name]

        block_of_road = {}
        for bi, blk in enumerate(pg_map.blocks):
            for a, b, _ in blk.net.roads():
                block_of_road[(a, b)] = bi
        for a, b, lanes in net.roads():
            # decoration lanes (ramp merge tapers) are real lane bodies in the reference too
            # (pgblock/pg_block.py:248-257 walks the whole graph), they just never appear on a route
            rid = len(roads)
            self.road_id[(a, b)] = rid
            roads.append((len(lanes_flat), len(lanes), node_id(a), node_id(b), int(is_negative_road(b)),
                          block_of_road.get((a, b), -1), road_block_id(a, b)))
            for i, l in enumerate(lanes):
                self.lane_id[(a, b, i)] = len(lanes_flat)
                lanes_flat.append((l, rid, i, len(lanes)))
                self.lane_objs.append(l)
        self.node_names = list(nodes.keys())
        self.node_index = nodes
        n_nodes = len(nodes)

        # --- lanes + hulls ---
        self.lanes = np.zeros(len(lanes_flat), dtype=abi.LANE_DT)
        hull_pts = []
        hull_off = 0
        for k, (l, rid, i, n) in enumerate(lanes_flat):
            r = self.lanes[k]
            r["road"], r["idx"], r["n_in_road"] = rid, i, n
            r["length"], r["width"] = l.length, l.width
            r["speed_limit"] = getattr(l, "speed_limit", 1000.0)
            r["sx"], r["sy"] = l.start
            r["ex"], r["ey"] = l.end
            el = l.end_lateral()
            r["elx"], r["ely"] = el
            if isinstance(l, CircularLane):
                r["type"] = 1
                r["ax"], r["ay"] = l.center
                r["bx"], r["by"] = l.radius, l.start_phase
                r["end_phase"] = l.end_phase
                r["end_phase_w"] = wrap_to_pi(l.end_phase)
                r["dirsign"] = l.direction
                r["angle"] = l.angle
            else:
                r["type"] = 0
                r["ax"], r["ay"] = l.start
                r["bx"], r["by"] = l.direction
                r["heading"] = l.heading
            hull = convex_hull(l.polygon()).astype(np.float32)
            r["hull_off"], r["hull_n"] = hull_off, len(hull)
            if len(hull) == 4:
                r["hull4"] = hull.reshape(-1)
            r["x0"], r["y0"] = hull[:, 0].min(), hull[:, 1].min()
            r["x1"], r["y1"] = hull[:, 0].max(), hull[:, 1].max()
            hull_pts.append(hull)
            hull_off += len(hull)
        self.hull_xy = np.concatenate(hull_pts, axis=0).astype(np.float32) if hull_pts else np.zeros((0, 2), np.float32)

        # --- roads / node adjacency ---
        self.roads = np.zeros(len(roads), dtype=abi.ROAD_DT)
        for k, (fl, n, sa, sb, neg, blk, kind) in enumerate(roads):
            self.roads[k] = (fl, n, sa, sb, neg, blk, kind, 0)
        adj = [[] for _ in range(n_nodes)]
        for k, (fl, n, sa, sb, neg, blk, kind) in enumerate(roads):
            adj[sa].append((sb, k))
        self.node_adj_off = np.zeros(n_nodes + 1, dtype=np.int32)
        flat = []
        for i in range(n_nodes):
            self.node_adj_off[i + 1] = self.node_adj_off[i] + len(adj[i])
            flat += adj[i]
        self.node_adj = np.asarray(flat, dtype=np.int32).reshape(-1, 2) if flat else np.zeros((0, 2), np.int32)

        # --- static quads (pgblock/pg_block.py:248-257: left+right on lane 0 of positive roads, else right only) ---
        quads, kinds = [], []
        for a, b, lanes in net.roads():
            pos = not is_negative_road(b)
            for i, l in enumerate(lanes):
                q, kd = lane_line_quads(l, (True, True) if (i == 0 and pos) else (False, True))
                quads += q
                kinds += kd
        self.quads = np.asarray(quads, dtype=np.float32).reshape(-1, 8) if quads else np.zeros((0, 8), np.float32)
        self.quad_kind = np.asarray(kinds, dtype=np.int32)
        self._build_grid()

    def _build_grid(self):
        boxes = []  # (x0, y0, x1, y1, item)
        for k in range(len(self.lanes)):
            r = self.lanes[k]
            if r["x1"] < r["x0"]:
                continue      # a placeholder lane (scenario scenes: nothing is localised on the grid there)
            boxes.append((r["x0"], r["y0"], r["x1"], r["y1"], k))
        for k in range(len(self.quads)):
            q = self.quads[k].reshape(4, 2)
            boxes.append((q[:, 0].min(), q[:, 1].min(), q[:, 0].max(), q[:, 1].max(), ~k))
        if not boxes:         # nothing static at all: one empty cell that covers everything
            self.grid = np.zeros(1, dtype=abi.GRID_DT)
            self.grid[0]["x0"], self.grid[0]["y0"], self.grid[0]["inv_cell"] = -1.0e7, -1.0e7, 5.0e-8
            self.grid[0]["nx"], self.grid[0]["ny"] = 1, 1
            self.cell_start = np.zeros(2, dtype=np.int32)
            self.cell_items = np.zeros(0, dtype=np.int32)
            return
        b = np.asarray([bb[:4] for bb in boxes], dtype=np.float64)
        x0 = math.floor(b[:, 0].min() - 1.0)
        y0 = math.floor(b[:, 1].min() - 1.0)
        x1, y1 = b[:, 2].max() + 1.0, b[:, 3].max() + 1.0
        nx = int(math.ceil((x1 - x0) / GRID_CELL))
        ny = int(math.ceil((y1 - y0) / GRID_CELL))
        cells = [[] for _ in range(nx * ny)]
        for (bx0, by0, bx1, by1, item) in boxes:
            gx0 = max(0, int(math.floor((bx0 - GRID_MARGIN - x0) / GRID_CELL)))
            gx1 = min(nx - 1, int(math.floor((bx1 + GRID_MARGIN - x0) / GRID_CELL)))
            gy0 = max(0, int(math.floor((by0 - GRID_MARGIN - y0) / GRID_CELL)))
            gy1 = min(ny - 1, int(math.floor((by1 + GRID_MARGIN - y0) / GRID_CELL)))
            for gy in range(gy0, gy1 + 1):
                for gx in range(gx0, gx1 + 1):
                    cells[gy * nx + gx].append(item)
        self.grid = np.zeros(1, dtype=abi.GRID_DT)
        self.grid[0]["x0"], self.grid[0]["y0"] = x0, y0
        self.grid[0]["inv_cell"] = 1.0 / GRID_CELL
        self.grid[0]["nx"], self.grid[0]["ny"] = nx, ny
        self.cell_start = np.zeros(nx * ny + 1, dtype=np.int32)
        items = []
        for i, c in enumerate(cells):
            # lane items ascending first (the localisation tie-break is "lowest lane id"), then quads
            c_sorted = sorted([x for x in c if x >= 0]) + sorted([x for x in c if x < 0], reverse=True)
            items += c_sorted
            self.cell_start[i + 1] = len(items)
        self.cell_items = np.asarray(items, dtype=np.int32)


StaticTables._build_grid = MapTables._build_grid

RESPAWN_REGION_LONGITUDE = 8.0  # manager/spawn_manager.py:28


def spawn_tables(mt, spawn_roads, lane_num, fixed_destination=False, dests=None, exclude_own_road=False):
    """Respawn places (slot 0 of every spawn road x lane) and, for each, the route to every destination
    (end node of the reversed spawn roads): SpawnManager._auto_fill_spawn_roads_randomly /
    get_available_respawn_places (manager/spawn_manager.py:123-209), RoundaboutSpawnManager.
    update_destination_for (envs/marl_envs/marl_inout_roundabout.py:136-141)."""
    from metadrive_ped_amd.mapgen.pg import negate_road
    pg = mt.pg_map
    if dests is None:
        dests = [negate_road(*r)[1] for r in spawn_roads]
    if fixed_destination:   # one destination per place: the default of NodeNetworkNavigation.reset
        dests = [None]
    all_dests = list(dests)
    places, lanes, routes, meta = [], [], [], []
    for road in spawn_roads:
        if exclude_own_road:
            # disable_u_turn (marl_intersection.py:80-82): the end roads are the spawn roads but the vehicle's own, in order; every
            # place keeps len(spawn_roads) - 1 destinations and the device draws an index among them
            dests = [d for r, d in zip(spawn_roads, all_dests) if tuple(r) != tuple(road)]
        for li in range(lane_num):
            lane = pg.net.lanes(*road)[li]
            long = RESPAWN_REGION_LONGITUDE / 2
            pos = lane.position(long, 0.0)
            h = wrap_to_pi(lane.heading_theta_at(long))
            places.append([pos[0], pos[1], math.cos(h), math.sin(h), h, 0.0, 0.0, 0.0])
            lanes.append(mt.lane_id[(road[0], road[1], li)])
            for d in dests:
                if d is None:
                    d = destination_for(pg, pg.seed, (road[0], road[1], li))
                nodes, roads_, n, fin = route_arrays(mt, (road[0], road[1], li), d)
                routes.append([nodes, roads_])
                meta.append([n, fin])
    return dict(spawn_place=np.asarray(places, np.float32), spawn_lane=np.asarray(lanes, np.int32),
                spawn_route=np.asarray(routes, np.int32), spawn_route_meta=np.asarray(meta, np.int32), n_dest=len(dests),
                dests=all_dests)


def destination_for(pg_map, seed, lane_index):
    """NodeNetworkNavigation.reset (navigation_module/node_network_navigation.py:43-71): a vehicle on a
    negative road drives to the first block's socket, any other to a socket of the last block."""
    from metadrive_ped_amd.rng import get_np_random
    negative = lane_index[1].find("-") != -1
    block = pg_map.blocks[0] if negative else pg_map.blocks[-1]
    sockets = list(block.sockets.values())
    socket = sockets[0] if len(sockets) == 1 else sockets[int(get_np_random(seed).choice(len(sockets)))]
    return socket.negative[1] if negative else socket.positive[1]


def respawn_lanes(pg_map):
    """PGTrafficManager._get_available_respawn_lanes (manager/traffic_manager.py:279-296): a road named by
    two blocks is an inner road and drops out."""
    roads = []
    for block in pg_map.blocks:
        for road in block.respawn_roads:
            if road in roads:
                roads.remove(road)
            else:
                roads.append(road)
    out = []
    for road in roads:
        out += pg_map.net.lanes(*road)
    return out


def respawn_tables(mt, seed):
    """MdWorld.spawn_* for the traffic modes 'respawn' / 'hybrid': one entry per respawn lane with the
    route a vehicle starting there follows (n_dest = 1; spawn_place is not used on this path)."""
    pg = mt.pg_map
    lanes, routes, meta = [], [], []
    for lane in respawn_lanes(pg):
        idx = tuple(lane.index)
        nodes, roads_, n, fin = route_arrays(mt, idx, destination_for(pg, seed, idx))
        lanes.append(mt.lane_id[idx])
        routes.append([nodes, roads_])
        meta.append([n, fin])
    return dict(spawn_place=np.zeros((len(lanes), 8), np.float32), spawn_lane=np.asarray(lanes, np.int32),
                spawn_route=np.asarray(routes, np.int32).reshape(-1, 2, abi.MD_ROUTE_LEN),
                spawn_route_meta=np.asarray(meta, np.int32).reshape(-1, 2), n_dest=1)


def route_arrays(mt, lane_index, dest):
    """(route_nodes[24], route_roads[24], n_checkpoints, final_lane) of set_route (node_network_navigation.py:94-128)."""
    pg = mt.pg_map
    ckpts = pg.bfs_route(lane_index[0], dest)
    if len(ckpts) <= 2:
        ckpts = [lane_index[0], lane_index[1]]
    if len(ckpts) > abi.MD_ROUTE_LEN:
        raise ValueError("route with {} checkpoints exceeds MD_ROUTE_LEN".format(len(ckpts)))
    nodes = np.full(abi.MD_ROUTE_LEN, -1, np.int32)
    roads = np.full(abi.MD_ROUTE_LEN, -1, np.int32)
    for j, name in enumerate(ckpts):
        nodes[j] = mt.node_index[name]
    for j in range(len(ckpts) - 1):
        roads[j] = mt.road_id[(ckpts[j], ckpts[j + 1])]
    fr = mt.roads[mt.road_id[(ckpts[-2], ckpts[-1])]]
    return nodes, roads, len(ckpts), int(fr["first_lane"] + fr["n_lanes"] - 1)


class WorldTables:
    """Many maps stacked into the CSR layout of MdWorld."""
    def __init__(self, maps, env_map, beam_cs):
        self.maps = maps
        n = len(maps)
        self.arrays = {}
        lane_off, road_off, quad_off, node_off = [0], [0], [0], [0]
        hull_base, cell_base, item_base, adj_base = 0, 0, 0, 0
        lanes, hulls, roads, quads, qk, grids, cstart, citems, adj_off, adj = [], [], [], [], [], [], [], [], [], []
        for m in maps:
            l = m.lanes.copy()
            l["hull_off"] += hull_base
            lanes.append(l)
            hulls.append(m.hull_xy)
            hull_base += len(m.hull_xy)
            roads.append(m.roads)
            quads.append(m.quads)
            qk.append(m.quad_kind)
            g = m.grid.copy()
            g["cell_base"] = cell_base
            grids.append(g)
            cstart.append(m.cell_start + item_base)
            cell_base += len(m.cell_start)
            citems.append(m.cell_items)
            item_base += len(m.cell_items)
            adj_off.append(m.node_adj_off[:-1] + adj_base)
            adj.append(m.node_adj)
            adj_base += len(m.node_adj)
            lane_off.append(lane_off[-1] + len(l))
            road_off.append(road_off[-1] + len(m.roads))
            quad_off.append(quad_off[-1] + len(m.quads))
            node_off.append(node_off[-1] + len(m.node_adj_off) - 1)
        a = self.arrays
        a["env_map"] = np.asarray(env_map, dtype=np.int32)
        a["lane_off"] = np.asarray(lane_off, dtype=np.int32)
        a["lanes"] = np.concatenate(lanes)
        a["hull_xy"] = np.concatenate(hulls).astype(np.float32).reshape(-1, 2)
        a["road_off"] = np.asarray(road_off, dtype=np.int32)
        a["roads"] = np.concatenate(roads)
        a["quad_off"] = np.asarray(quad_off, dtype=np.int32)
        a["quads"] = np.concatenate(quads).astype(np.float32).reshape(-1, 8)
        a["quad_kind"] = np.concatenate(qk).astype(np.int32)
        # bounding circles of the quads (MdWorld.quad_ball): centre = mean of the vertices, radius rounded up
        q4 = a["quads"].astype(np.float64).reshape(-1, 4, 2)
        ctr = q4.mean(axis=1)
        rad = np.sqrt(((q4 - ctr[:, None, :]) ** 2).sum(axis=2)).max(axis=1) * 1.0001 + 1.0e-3
        ball = np.zeros((len(q4), 4), np.float32)
        ball[:, 0:2] = ctr
        ball[:, 2] = rad
        ball[:, 3] = a["quad_kind"].view(np.float32) if len(q4) == len(a["quad_kind"]) else 0.0
        a["quad_ball"] = ball
        a["grid"] = np.concatenate(grids)
        a["cell_start"] = np.concatenate(cstart).astype(np.int32)
        a["cell_items"] = np.concatenate(citems).astype(np.int32)
        a["node_adj_off"] = np.concatenate(adj_off + [np.asarray([adj_base], dtype=np.int32)]).astype(np.int32)
        a["node_adj"] = np.concatenate(adj).astype(np.int32).reshape(-1, 2)
        a["node_off"] = np.asarray(node_off, dtype=np.int32)
        a["beam_cs"] = np.asarray(beam_cs, dtype=np.float32).reshape(-1, 2)
        self.n_maps = n
        self.n_envs = len(env_map)
        if n and all(getattr(m, "respawn", None) is not None for m in maps):
            off = [0]
            for m in maps:
                off.append(off[-1] + len(m.respawn["spawn_lane"]))
            a["spawn_off"] = np.asarray(off, np.int32)
            for k in ("spawn_place", "spawn_lane", "spawn_route", "spawn_route_meta"):
                a[k] = np.ascontiguousarray(np.concatenate([m.respawn[k] for m in maps]))
            self.n_dest = 1
        # guard: padding arrays that may legitimately be empty
        for k in ("quads", "quad_kind", "quad_ball", "cell_items", "node_adj", "hull_xy"):
            if a[k].size == 0:
                a[k] = np.zeros((1, ) + a[k].shape[1:], dtype=a[k].dtype)


def beam_table(n_beams, phase=0.0):
    """cos/sin of DistanceDetector._get_lidar_range (component/sensors/distance_detector.py:177-180)."""
    if n_beams <= 0:
        return np.zeros((1, 2), np.float32)
    ang = np.arange(0, n_beams) * (2 * np.pi / n_beams) + phase
    return np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)

"""Host-side (float64) lane geometry used while a map is being generated at reset().

Two lane primitives exist in PG maps -- a straight segment and a circular arc -- and a road is a list
of parallel lanes between two graph nodes.  This module keeps them as small value objects; the
per-step Frenet math lives on the GPU (include/md_geom.h) and works from the flattened MdLane table
produced by mapgen/tables.py.

Behavioural source (restated, not copied): metadrive/component/lane/straight_lane.py:14-95,
circular_lane.py:11-177, abs_lane.py:76-89, pg_lane.py:8 (POLYGON_SAMPLE_RATE = 1).
Lateral coordinate is positive to the RIGHT of the driving direction.
"""
import math

import numpy as np

LINE_NONE, LINE_BROKEN, LINE_CONTINUOUS, LINE_SIDE, LINE_GUARDRAIL = "none", "broken", "continuous", "side", "guardrail"
COLOR_GREY, COLOR_YELLOW = "grey", "yellow"
POLYGON_SAMPLE_RATE = 1.0


def wrap_to_pi(x):
    """(-pi, pi]  (metadrive/utils/math.py:29-41)"""
    a = x % (2 * np.pi)
    if a > np.pi:
        a -= 2 * np.pi
    return a


class Lane:
    """Common part: width, line decoration, graph index."""
    kind = None
    # AbstractLane.speed_limit (lane/abs_lane.py:22): 1000 unless a builder sets it -- the bends of create_bend_straight
    # get 20 (create_pg_block_utils.py:28), the TollGate block's lanes 3 (pgblock/tollgate.py:64-68); side lanes made from a
    # reference lane inherit it (CreateRoadFrom), an extension starts again at 1000 (ExtendStraightLane re-initialises the lane).
    # Only MultiAgentTollgateEnv reads it (BaseVehicle.overspeed); the ramps' own SPEED_LIMIT constants are not carried.
    speed_limit = 1000.0

    def __init__(self, width, line_types):
        self.width = float(width)
        self.line_types = list(line_types)
        self.line_colors = [COLOR_GREY, COLOR_GREY]
        self.index = None  # (start_node, end_node, i) once added to a road network
        self.radius = 0.0

    def clone(self):
        other = object.__new__(type(self))
        other.__dict__.update(self.__dict__)
        other.line_types = list(self.line_types)
        other.line_colors = list(self.line_colors)
        return other

    def distance(self, p):
        s, r = self.local_coordinates(p)
        return abs(r) + max(s - self.length, 0.0) + max(-s, 0.0)

    def is_previous_lane_of(self, other, tol=1e-1):
        e, s = self.end, other.start
        return math.hypot(e[0] - s[0], e[1] - s[1]) < tol

    def width_at(self, s):
        return self.width


class StraightLane(Lane):
    kind = 0

    def __init__(self, start, end, width=3.5, line_types=(LINE_BROKEN, LINE_BROKEN)):
        super().__init__(width, line_types)
        self.start = np.asarray(start, dtype=np.float64)
        self.end = np.asarray(end, dtype=np.float64)
        self.refresh()

    def refresh(self):
        d = self.end - self.start
        self.length = math.sqrt(d[0] ** 2 + d[1] ** 2)
        self.heading = math.atan2(d[1], d[0])
        self.direction = d / self.length
        self.direction_lateral = np.array([self.direction[1], -self.direction[0]])

    def position(self, s, lat):
        return self.start + s * self.direction + lat * self.direction_lateral

    def heading_theta_at(self, s):
        return self.heading

    def local_coordinates(self, p):
        dx, dy = p[0] - self.start[0], p[1] - self.start[1]
        return (float(dx * self.direction[0] + dy * self.direction[1]),
                float(dx * self.direction_lateral[0] + dy * self.direction_lateral[1]))

    def end_lateral(self):
        return self.direction_lateral

    def shifted(self, lat):
        """Parallel copy at lateral offset `lat`.  Mirrors how side lanes are made in
        create_pg_block_utils.py:104-111: start/end are moved, length/heading/direction are kept."""
        o = self.clone()
        o.start = self.position(0.0, lat)
        o.end = self.position(self.length, lat)
        return o

    def extended(self, extra, line_types):
        """ExtendStraightLane (create_pg_block_utils.py:178-195)."""
        o = self.clone()
        o.speed_limit = 1000.0
        o.start = self.end
        o.end = self.position(self.length + extra, 0.0)
        o.line_types = list(line_types)
        o.refresh()
        return o

    def polygon(self):
        longs = np.arange(0, self.length + POLYGON_SAMPLE_RATE, POLYGON_SAMPLE_RATE)
        pts = [self.position(s, +self.width / 2) for s in longs]
        pts += [self.position(s, -self.width / 2) for s in longs[::-1]]
        return np.asarray(pts)


class CircularLane(Lane):
    kind = 1

    def __init__(self, center, radius, start_phase, angle, clockwise=True, width=3.5,
                 line_types=(LINE_BROKEN, LINE_BROKEN)):
        assert angle > 0
        super().__init__(width, line_types)
        self.center = np.asarray(center, dtype=np.float64)
        self.radius = float(radius)
        self.clockwise = bool(clockwise)
        self.start_phase = wrap_to_pi(start_phase)
        self.angle = float(angle)
        self.end_phase = self.start_phase + (-self.angle if self.clockwise else self.angle)
        self.direction = -1 if self.clockwise else 1
        self.refresh()

    def refresh(self):
        self.length = abs(self.radius * (self.end_phase - self.start_phase))
        assert self.length > 0
        self.start = self.position(0.0, 0.0)
        self.end = self.position(self.length, 0.0)

    def position(self, s, lat):
        phi = self.direction * s / self.radius + self.start_phase
        return self.center + (self.radius + lat * self.direction) * np.array([math.cos(phi), math.sin(phi)])

    def heading_theta_at(self, s):
        phi = self.direction * s / self.radius + self.start_phase
        return phi + math.pi / 2 * self.direction

    def local_coordinates(self, p):
        dx, dy = p[0] - self.center[0], p[1] - self.center[1]
        abs_phase = wrap_to_pi(math.atan2(dy, dx))
        sp, ep = wrap_to_pi(self.start_phase), wrap_to_pi(self.end_phase)
        d_start = abs(wrap_to_pi(abs_phase - sp))
        d_end = abs(wrap_to_pi(abs_phase - ep))
        if d_start > d_end:
            diff = self.end_phase - abs_phase if self.clockwise else abs_phase - self.end_phase
            s = wrap_to_pi(diff) * self.radius + self.length
        else:
            diff = self.start_phase - abs_phase if self.clockwise else abs_phase - self.start_phase
            s = wrap_to_pi(diff) * self.radius
        return s, self.direction * (math.hypot(dx, dy) - self.radius)

    def end_lateral(self):
        phi = self.direction * self.length / self.radius + self.start_phase
        return self.direction * np.array([math.cos(phi), math.sin(phi)])

    def with_radius(self, radius):
        o = self.clone()
        o.radius = float(radius)
        o.refresh()
        return o

    def polygon(self):
        """Outline with the +-1 m tangent extensions at both ends (circular_lane.py:123-174)."""
        h0 = self.heading_theta_at(0.0)
        d0 = np.array([math.cos(h0), math.sin(h0)])
        h1 = self.heading_theta_at(self.length)
        d1 = np.array([math.cos(h1), math.sin(h1)])
        longs = np.arange(0, self.length + POLYGON_SAMPLE_RATE, POLYGON_SAMPLE_RATE)
        pts = []
        for k, lat in enumerate([+self.width / 2, -self.width / 2]):
            ss = longs if k == 0 else longs[::-1]
            last = len(ss) - 1
            for t, s in enumerate(ss):
                p = self.position(s, lat)
                at_lane_start = (t == 0 and k == 0) or (t == last and k == 1)
                at_lane_end = (t == last and k == 0) or (t == 0 and k == 1)
                if at_lane_start:
                    pts.append(p)
                    pts.append(p - d0 * POLYGON_SAMPLE_RATE)
                elif at_lane_end:
                    pts.append(p)
                    pts.append(p + d1 * POLYGON_SAMPLE_RATE)
                else:
                    pts.append(p)
        return np.asarray(pts)


def convex_hull(points):
    """Andrew monotone chain; CCW, no collinear points.  The reference feeds the lane outline into a
    Bullet convex-hull shape (block/base_block.py:456-466); containment in that hull is `on_lane`."""
    pts = sorted(set((float(p[0]), float(p[1])) for p in points))
    if len(pts) < 3:
        return np.asarray(pts)

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower = []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    upper = []
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return np.asarray(lower[:-1] + upper[:-1])

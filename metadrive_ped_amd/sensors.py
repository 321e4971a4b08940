"""Sensor-level plug: the reference's Lidar.perceive (component/sensors/lidar.py:49-73, distance_detector.py:27-85,
118-127) answered by the HIP library, for a maintainer who keeps the reference's engine and swaps only the sensor.

    lidar = BatchedLidar(device="cuda:0")
    cloud_points, detected_objects = lidar.perceive(ego, world, num_lasers=240, distance=50)

mirrors `Lidar.perceive(base_vehicle, physics_world, num_lasers, distance, height=None, detector_mask=None, show=False)`:
`physics_world` is anything iterable over the bodies in the world (the reference hands Bullet's dynamic world; here the
objects themselves: every one with `.position`, `.heading_theta` and `.LENGTH` / `.WIDTH`, or `.RADIUS` for cones /
warnings / pedestrians); the result is `(cloud_points: list[float] of length num_lasers in [0, 1], detected_objects: set)`,
beam 0 along the heading, counter-clockwise, 1.0 = nothing within `distance`, the vehicle's own chassis excluded.
`perceive_batch` does many (vehicle, world) pairs in one md_lidar_detect launch.  There is no CPU path: without the
library or a ROCm device this raises.
"""
import ctypes as C
import math

import numpy as np

from metadrive_ped_amd import abi
from metadrive_ped_amd.mapgen.tables import beam_table


def _shape_of(obj):
    """MdShape fields of a reference-style object"""
    x, y = float(obj.position[0]), float(obj.position[1])
    h = float(getattr(obj, "heading_theta", 0.0))
    if hasattr(obj, "RADIUS") and not hasattr(obj, "LENGTH"):
        r = float(obj.RADIUS)
        kind = abi.KIND_PEDESTRIAN if getattr(obj, "TYPE_NAME", "") == "pedestrian" else (abi.KIND_WARNING if r >= 0.45 else abi.KIND_CONE)
        return x, y, math.cos(h), math.sin(h), r, r, kind
    kind = getattr(obj, "md_kind", abi.KIND_VEHICLE)
    return x, y, math.cos(h), math.sin(h), float(obj.LENGTH) / 2.0, float(obj.WIDTH) / 2.0, kind


class BatchedLidar:
    DEFAULT_HEIGHT = 1.2          # Lidar.DEFAULT_HEIGHT (lidar.py:19): beams are horizontal at this height; 2-D here

    def __init__(self, *args, device="cuda:0"):
        """`BatchedLidar("cuda:0")`, or -- the way the reference's engine builds its sensors, `cls(*args, engine)`
        (engine/core/engine_core.py:523-534) -- `BatchedLidar(engine)` / `BatchedLidar("cuda:1", engine)`"""
        import torch
        for a in args:
            if isinstance(a, (str, torch.device)):
                device = a
        from metadrive_ped_amd import _lib
        self.torch = torch
        self.lib = _lib.load()
        self._check = _lib.check
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.MdStepError("BatchedLidar needs a ROCm device; there is no CPU fallback")
        self._beams = {}

    def _beam_tensor(self, n):
        if n not in self._beams:
            self._beams[n] = self.torch.from_numpy(beam_table(n)).to(self.device)
        return self._beams[n]

    def perceive(self, base_vehicle, physics_world, num_lasers, distance, height=None, detector_mask=None, show=False):
        cloud, det = self.perceive_batch([base_vehicle], [physics_world], num_lasers, distance, detector_mask=None
                                         if detector_mask is None else [detector_mask])
        return cloud[0], det[0]

    def perceive_batch(self, vehicles, worlds, num_lasers, distance, detector_mask=None):
        """vehicles[e] sees worlds[e] (an iterable of objects; the vehicle itself may be among them and is skipped).
        -> (list of cloud-point lists, list of detected-object sets).  `detector_mask[e]` (bool per beam): beams that are
        off report 1.0, as in distance_detector.py:46-49."""
        torch = self.torch
        E = len(vehicles)
        if E == 0 or len(worlds) != E:
            raise ValueError("need one world per vehicle")
        if not (0 < int(num_lasers) <= abi.MD_MAX_BEAMS):
            raise ValueError("num_lasers must be in 1..{}".format(abi.MD_MAX_BEAMS))
        objs = [[o for o in wld if o is not vehicles[e]] for e, wld in enumerate(worlds)]
        cap = max(2, 1 + max(len(o) for o in objs))
        if cap > abi.MD_MAX_CAP:
            raise ValueError("at most {} bodies per world".format(abi.MD_MAX_CAP - 1))
        shape = np.zeros((E, cap), dtype=abi.SHAPE_DT)
        shape["aux"] = -1
        for e in range(E):
            for j, o in enumerate([vehicles[e]] + objs[e]):
                x, y, c, s, hl, hw, kind = _shape_of(o)
                shape[e, j] = (x, y, c, s, hl, hw, kind | abi.F_ALIVE | (abi.F_AGENT if j == 0 else 0), -1)
        d_shape = torch.from_numpy(shape.view(np.uint8).reshape(-1)).to(self.device)
        out = torch.empty(E, int(num_lasers), dtype=torch.float32, device=self.device)
        det = torch.zeros(E, 2, dtype=torch.int64, device=self.device)
        w, s, k = abi.MdWorld(), abi.MdState(), abi.MdConfig()
        w.n_maps, w.n_envs = 1, E
        w.beam_cs = self._beam_tensor(int(num_lasers)).data_ptr()
        s.shape = d_shape.data_ptr()
        k.struct_size = C.sizeof(abi.MdConfig)
        k.n_envs, k.agents_per_env, k.cap, k.n_beams, k.obs_dim = E, 1, cap, int(num_lasers), int(num_lasers)
        k.lidar_range = float(distance)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self._check(self.lib.md_lidar_detect(C.byref(w), C.byref(s), C.byref(k), C.c_void_p(out.data_ptr()), int(num_lasers), 0,
                                                 C.c_void_p(det.data_ptr()), stream), "md_lidar_detect")
        cloud = out.cpu().numpy()
        bits = det.cpu().numpy().view(np.uint64)
        clouds, sets = [], []
        for e in range(E):
            row = cloud[e]
            if detector_mask is not None and detector_mask[e] is not None:
                row = np.where(np.asarray(detector_mask[e], bool), row, np.float32(1.0))
            clouds.append([float(x) for x in row])
            found = set()
            for j, o in enumerate(objs[e], start=1):
                if (int(bits[e, j >> 6]) >> (j & 63)) & 1:
                    found.add(o)
            sets.append(found)
        return clouds, sets

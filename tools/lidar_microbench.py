"""Kernel micro-bench for the roofline row of SURVEY 8(d): md_lidar alone on synthetic shape tables.

E in {4096, 65536} envs, one agent each at the origin of its env, M in {4, 16, 64} OBBs uniformly in a
100 m square around it, headings U(-pi, pi), sizes of the reference's traffic vehicle classes
(component/vehicle/vehicle_type.py), 240 beams, range 50 m, seed 0.  Prints one JSON line per case:
algorithmic GB/s (16 + 24 M + 4 B bytes per agent) against the HBM peak and the measured stream copy,
and FP32 op/s (30 B M per agent) against the vector peak -- the honest ceiling is the lower of the two.

Usage: python tools/lidar_microbench.py [--reps 50]
(the same synthetic cases are compared bit for bit with the CPU oracle by tests/test_gpu_parity.py)
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
FP32_PEAK_TFLOPS = 157.3
# (length, width) of S / M / L / XL / default vehicles (vehicle_type.py LENGTH / WIDTH)
SIZES = np.array([[4.25, 1.7], [4.6, 1.85], [4.5, 1.86], [5.74, 2.3], [4.515, 1.852]], np.float32)


def make_case(E, M, B, seed=0):
    from metadrive_ped_amd import abi
    from metadrive_ped_amd.mapgen.tables import beam_table
    rng = np.random.RandomState(seed)
    cap = M + 1
    shape = np.zeros((E, cap), dtype=abi.SHAPE_DT)
    shape["aux"] = -1
    ego_h = rng.uniform(-np.pi, np.pi, E).astype(np.float32)
    shape["cx"][:, 0], shape["cy"][:, 0] = 0.0, 0.0
    shape["c"][:, 0], shape["s"][:, 0] = np.cos(ego_h), np.sin(ego_h)
    shape["hl"][:, 0], shape["hw"][:, 0] = SIZES[4, 0] / 2, SIZES[4, 1] / 2
    shape["flags"][:, 0] = abi.KIND_VEHICLE | abi.F_ALIVE | abi.F_AGENT
    h = rng.uniform(-np.pi, np.pi, (E, M)).astype(np.float32)
    k = rng.randint(0, 4, (E, M))
    shape["cx"][:, 1:] = rng.uniform(-50, 50, (E, M))
    shape["cy"][:, 1:] = rng.uniform(-50, 50, (E, M))
    shape["c"][:, 1:], shape["s"][:, 1:] = np.cos(h), np.sin(h)
    shape["hl"][:, 1:], shape["hw"][:, 1:] = SIZES[k, 0] / 2, SIZES[k, 1] / 2
    shape["flags"][:, 1:] = abi.KIND_VEHICLE | abi.F_ALIVE
    return shape.reshape(-1), beam_table(B), cap


def structs(abi, E, cap, B, shape_ptr, beam_ptr):
    w = abi.MdWorld()
    w.n_maps, w.n_envs, w.max_lanes, w.max_roads = 1, E, 1, 1
    w.beam_cs = beam_ptr
    s = abi.MdState()
    s.shape = shape_ptr
    k = abi.MdConfig()
    k.struct_size = C.sizeof(abi.MdConfig)
    k.n_envs, k.agents_per_env, k.cap, k.n_beams, k.obs_dim = E, 1, cap, B, B
    k.lidar_range = 50.0
    return w, s, k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--beams", type=int, default=240)
    args = ap.parse_args()
    import torch
    from metadrive_ped_amd import _lib, abi
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B = args.beams
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    # attainable HBM: stream copy
    n = 1 << 30
    a, b = torch.ones(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    for _ in range(3):
        _lib.check(lib.md_probe_stream_copy(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), n, st), "copy")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _lib.check(lib.md_probe_stream_copy(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), n, st), "copy")
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 2.0 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    print(json.dumps(dict(kernel="stream_copy_kernel", bytes=2 * n, GBps=round(copy_gbs, 1), frac_of_spec=round(copy_gbs / HBM_PEAK_GBS, 3))))
    for E in (4096, 65536):
        for M in (4, 16, 64):
            shape, beams, cap = make_case(E, M, B)
            t_shape = torch.from_numpy(shape.view(np.uint8)).to(dev)
            t_beams = torch.from_numpy(beams).to(dev)
            out = torch.empty(E, B, device=dev)
            w, s, k = structs(abi, E, cap, B, t_shape.data_ptr(), t_beams.data_ptr())
            call = lambda: _lib.check(lib.md_lidar(C.byref(w), C.byref(s), C.byref(k), C.c_void_p(out.data_ptr()), B, 0, st), "md_lidar")
            for _ in range(5):
                call()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
            for x, y in evs:
                x.record()
                call()
                y.record()
            torch.cuda.synchronize()
            ms = sorted(x.elapsed_time(y) for x, y in evs)
            avg = sum(ms) / len(ms)
            byts = (16.0 + 24.0 * M + 4.0 * B) * E
            flops = 30.0 * B * M * E
            res = out.cpu().numpy()
            line = dict(kernel="md_lidar", envs=E, shapes_per_env=M, beams=B, avg_us=round(avg * 1e3, 2), min_us=round(ms[0] * 1e3, 2),
                        bytes_per_launch=int(byts), GBps=round(byts / (avg * 1e-3) / 1e9, 1),
                        hbm_frac_spec=round(byts / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        hbm_frac_attainable=round(byts / (avg * 1e-3) / 1e9 / copy_gbs, 4),
                        fp32_tflops=round(flops / (avg * 1e-3) / 1e12, 2),
                        fp32_frac=round(flops / (avg * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                        agent_lidars_per_s=round(E / (avg * 1e-3), 0), hit_fraction=round(float((res < 1.0).mean()), 3))
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

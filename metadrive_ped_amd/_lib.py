"""Loader of the HIP product library (metadrive_ped_amd/lib/libmdstep.so) through its C-ABI.

There is NO fallback: if the library is missing or its ABI does not match the binding, importing
the engine raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from
here.)  torch is imported first on purpose: its bundled libamdhip64.so.7 then satisfies the
library's DT_NEEDED entry by soname, so the tensors' device pointers, the torch stream handle and
our kernel launches all live in ONE HIP runtime instance.
"""
import ctypes as C
import os

from metadrive_ped_amd import abi

_LIB = None
# MD_LIB_PATH: tuning experiments only (A/B of differently compiled builds of the same source)
LIB_PATH = os.environ.get("MD_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmdstep.so")


class MdStepError(RuntimeError):
    pass


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    import torch  # noqa: F401  (see module docstring)
    if not os.path.exists(LIB_PATH):
        raise MdStepError(
            "HIP library not found: {}\nBuild it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.".format(LIB_PATH))
    lib = C.CDLL(LIB_PATH)
    lib.md_abi.restype = C.c_int
    lib.md_abi.argtypes = [C.POINTER(C.c_int32), C.c_int]
    lib.md_last_error.restype = C.c_char_p
    W, S, K = C.POINTER(abi.MdWorld), C.POINTER(abi.MdState), C.POINTER(abi.MdConfig)
    for name in ("md_integrate", "md_localize", "md_contacts", "md_observe", "md_idm", "md_traffic_after_step", "md_lifecycle", "md_step"):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [W, S, K, C.c_void_p]
    lib.md_lidar.restype = C.c_int
    lib.md_lidar.argtypes = [W, S, K, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.md_lidar_detect.restype = C.c_int
    lib.md_lidar_detect.argtypes = [W, S, K, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.md_swap_draw.restype = C.c_int
    lib.md_swap_draw.argtypes = [S, S, K, C.c_int, C.c_void_p, C.c_void_p]
    lib.md_line_detectors.restype = C.c_int
    lib.md_line_detectors.argtypes = [W, S, K, C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_uint32,
                                      C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.md_line_detector.restype = C.c_int
    lib.md_line_detector.argtypes = [W, S, K, C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_void_p, C.c_int, C.c_int,
                                     C.c_void_p]
    lib.md_probe_math.restype = C.c_int
    lib.md_probe_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.md_probe_stream_copy.restype = C.c_int
    lib.md_probe_stream_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    abi.check_abi(lib.md_abi, LIB_PATH)
    _LIB = lib
    return lib


def check(rc, what):
    if rc != abi.MD_OK:
        msg = load().md_last_error().decode("utf8", "replace")
        raise MdStepError("{} failed with code {}: {}".format(what, rc, msg))

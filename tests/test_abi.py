"""The C-ABI library builds for gfx950, loads without a GPU, exports every symbol include/mdstep.h
declares, and agrees with the Python binding on every struct size.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build_hip()
    from metadrive_ped_amd import _lib
    return _lib.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mdstep.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(md_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    from metadrive_ped_amd import abi
    syms = declared_symbols()
    assert set(syms) == set(abi.ENTRY_POINTS), (syms, abi.ENTRY_POINTS)
    for s in syms:
        assert hasattr(lib, s), "libmdstep.so does not export %s" % s


def test_struct_sizes_match(lib):
    from metadrive_ped_amd import abi
    sizes = (C.c_int32 * 11)()
    assert lib.md_abi(sizes, 11) == abi.MD_ABI_VERSION
    assert list(sizes) == abi.STRUCT_SIZES


def test_bad_arguments_are_rejected_without_touching_a_device(lib):
    from metadrive_ped_amd import abi
    w, s, k = abi.MdWorld(), abi.MdState(), abi.MdConfig()
    assert lib.md_step(None, None, None, None) == abi.MD_EINVAL
    k.struct_size = 4
    assert lib.md_step(C.byref(w), C.byref(s), C.byref(k), None) == abi.MD_EABI
    k.struct_size = C.sizeof(abi.MdConfig)
    k.n_envs, k.cap, k.agents_per_env = 4, 1000, 1
    assert lib.md_step(C.byref(w), C.byref(s), C.byref(k), None) == abi.MD_EINVAL
    assert b"cap" in lib.md_last_error()
    k.cap = 32
    w.n_envs = 4
    assert lib.md_lidar(C.byref(w), C.byref(s), C.byref(k), None, 240, 0, None) == abi.MD_EINVAL  # null shape
    assert lib.md_probe_math(0, None, None, None, 4, None) == abi.MD_EINVAL


def test_engine_refuses_cpu_device():
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import BatchedEngine
    from metadrive_ped_amd._lib import MdStepError
    with pytest.raises(MdStepError):
        BatchedEngine(make_config(dict(device="cpu", map="S")))


def test_oracle_abi_matches():
    import oracle_binding as ob
    ob.load()  # raises on any struct-size mismatch

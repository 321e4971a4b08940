"""Device arithmetic vs host arithmetic of the shared float32 formulas, bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OPS = {0: "sin", 1: "cos", 2: "atan2", 3: "acos", 4: "exp", 5: "div", 6: "sqrt", 7: "wrap_to_pi", 8: "asin", 9: "norm",
       10: "energy"}


@pytest.mark.parametrize("op", sorted(OPS))
def test_probe_math_bit_exact(op):
    import torch
    from metadrive_ped_amd import _lib
    import oracle_binding as ob
    lib, ref = _lib.load(), ob.load()
    rng = np.random.RandomState(op)
    n = 1 << 16
    if op in (3, 8):
        a = rng.uniform(-1.2, 1.2, n)
    elif op == 4:
        a = rng.uniform(-20, 20, n)
    elif op == 6:
        a = rng.uniform(0, 1e4, n)
    elif op == 10:
        a = rng.uniform(0, 120, n)
    else:
        a = rng.uniform(-400, 400, n)
    b = rng.uniform(-400, 400, n) if op != 10 else rng.uniform(0, 5, n)
    a[:8] = [0.0, -0.0, 1.0, -1.0, 0.5, 1e-30, 3.1415927, -3.1415927]
    a, b = a.astype(np.float32), b.astype(np.float32)
    want = np.zeros(n, np.float32)
    ref.ref_probe_math(op, a.ctypes.data, b.ctypes.data, want.ctypes.data, n)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    out = torch.zeros(n, device="cuda")
    rc = lib.md_probe_math(op, ta.data_ptr(), tb.data_ptr(), out.data_ptr(), n,
                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    got = out.cpu().numpy()
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    assert len(bad) == 0, "{}: {} of {} differ, e.g. a={} b={} gpu={} cpu={}".format(
        OPS[op], len(bad), n, a[bad[:3]], b[bad[:3]], got[bad[:3]], want[bad[:3]])


def test_stream_copy_probe_copies():
    """md_probe_stream_copy (the attainable-bandwidth probe of bench.py) is a faithful copy, tail included."""
    import ctypes as C
    import torch
    from metadrive_ped_amd import _lib
    lib = _lib.load()
    n = 16 * (1024 * 37 + 123)
    src = torch.randint(0, 255, (n, ), dtype=torch.uint8, device="cuda")
    dst = torch.zeros_like(src)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.md_probe_stream_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n, st), "copy")
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    assert lib.md_probe_stream_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), 24, st) != 0

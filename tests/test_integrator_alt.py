"""Row a-1 (vehicle dynamics) against a SECOND, independent spelling: oracle/md_integrator_alt.c.

`md_integrate_mover` (include/md_entity.h + md_geom.h) is compiled into the HIP kernel AND into the oracle, so for this
row "HIP == oracle bit for bit" says the two compilers agree, not that the formulas are right.  md_integrator_alt.c is
written from the model (reference kinematic bicycle, component/vehicle_model/kinematics.py:148-158, plus the documented
engine / brake / grip / yaw-slew terms) in double precision with libm and shares no code with the headers.  Here both
the oracle's and the HIP kernel's integrate phase are compared with it, step by step from identical start states.

Tolerance (float32 product path vs double): 2e-4 m on positions (coordinates of a few hundred metres carry 3e-5 m of
float32 rounding per operation), 2e-5 rad on the heading, 2e-5 m/s on the speed, 2e-6 on (cos, sin) of the heading.
"""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import make_cfg, scripted_actions
from metadrive_ped_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POS_TOL, ANG_TOL, V_TOL, CS_TOL = 2e-4, 2e-5, 2e-5, 2e-6


def _alt():
    path = os.path.join(ROOT, "oracle", "_build", "libmdalt.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(path)
    lib.alt_integrate.restype = None
    lib.alt_integrate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def _alt_step(lib, pre, md_config):
    n = len(pre["shape"])
    out = np.full((n, 6), np.nan, np.float64)
    act = np.ascontiguousarray(pre["action"], np.float32)
    lib.alt_integrate(pre["shape"].ctypes.data, pre["dyn"].ctypes.data, pre["param"].ctypes.data, act.ctypes.data, n,
                      C.byref(md_config), out.ctypes.data)
    return out


def _compare(pre, post, alt, where):
    """post: state arrays after the product's integrate phase; alt: alt_integrate's [n, 6] (nan = did not drive)"""
    drove = ~np.isnan(alt[:, 0])
    assert drove.sum() > 0
    sh, dy = post["shape"], post["dyn"]
    still = ~drove
    assert np.array_equal(sh["cx"][still], pre["shape"]["cx"][still]) and np.array_equal(sh["cy"][still], pre["shape"]["cy"][still]), where
    err = dict(x=np.abs(sh["cx"][drove] - alt[drove, 0]).max(), y=np.abs(sh["cy"][drove] - alt[drove, 1]).max(),
               v=np.abs(dy["speed"][drove] - alt[drove, 3]).max(),
               c=np.abs(sh["c"][drove] - alt[drove, 4]).max(), s=np.abs(sh["s"][drove] - alt[drove, 5]).max())
    dpsi = dy["heading"][drove].astype(np.float64) - alt[drove, 2]
    err["psi"] = np.abs((dpsi + np.pi) % (2 * np.pi) - np.pi).max()
    assert err["x"] < POS_TOL and err["y"] < POS_TOL, (where, err)
    assert err["psi"] < ANG_TOL and err["v"] < V_TOL and err["c"] < CS_TOL and err["s"] < CS_TOL, (where, err)
    return int(drove.sum()), err


def _snapshot(state):
    return {k: state[k].copy() for k in ("shape", "dyn", "param", "action")}


@pytest.mark.parametrize("mode,extra", [("respawn", {}), ("trigger", dict(vehicle_config=dict(enable_reverse=True)))])
def test_oracle_integrate_against_independent_spelling(cs_dist, mode, extra):
    import oracle_binding as ob
    lib = _alt()
    E = 12
    cfg = make_cfg(cs_dist, num_envs=E, num_scenarios=E, traffic_density=0.3, traffic_mode=mode, auto_reset=True, horizon=120,
                   start_seed=40, **extra)
    from metadrive_ped_amd.engine import HostScene
    host = HostScene(cfg)
    o = ob.OracleWorld(host)
    o.reset()
    n_checked, moved, worst = 0, 0.0, {}
    for t in range(150):
        a = scripted_actions(E, 1, t, seed=5)
        if t % 40 > 25:
            a[:, :, 1] = -0.8                      # brake to a stop and (enable_reverse) back up
        if t % 3 == 0:
            # the integrate phase alone, on a copy, from the state the rollout has reached
            side = ob.OracleWorld(host, state={k: v.copy() for k, v in o.state.items()})
            side.state["action"].reshape(E, -1, 2)[:, :1, :] = a
            pre = _snapshot(side.state)
            side.call("ref_integrate")
            alt = _alt_step(lib, pre, host.md_config)
            n, err = _compare(pre, side.state, alt, where="%s t=%d" % (mode, t))
            n_checked += n
            moved = max(moved, float(np.abs(side.state["shape"]["cx"] - pre["shape"]["cx"]).max()))
            worst = {k: max(worst.get(k, 0.0), float(v)) for k, v in err.items()}
        o.step(a)
    assert n_checked > 300 and moved > 0.5, (n_checked, moved)
    # the check has teeth: a formula error of one part in a thousand would sit far above the tolerance
    assert moved * 1e-3 > POS_TOL


@pytest.mark.gpu
def test_hip_integrate_against_independent_spelling(cs_dist):
    """md_integrate (HIP, through the C-ABI) vs md_integrator_alt.c: the kernel against code it shares nothing with."""
    import torch
    from metadrive_ped_amd.engine import BatchedEngine
    lib = _alt()
    E = 64
    cfg = make_cfg(cs_dist, num_envs=E, num_scenarios=E, traffic_density=0.3, traffic_mode="respawn", auto_reset=True, horizon=150,
                   start_seed=60, vehicle_config=dict(enable_reverse=True))
    eng = BatchedEngine(cfg)
    eng.reset()
    n_checked = 0
    for t in range(120):
        a = scripted_actions(E, 1, t, seed=9)
        if t % 40 > 28:
            a[:, :, 1] = -0.9
        if t % 4 == 0:
            saved = eng.download_state()
            eng.action[:, :1, :] = torch.from_numpy(a).to(eng.device)
            pre = _snapshot(eng.download_state())
            eng.call("md_integrate")
            post = eng.download_state()
            alt = _alt_step(lib, pre, eng.host.md_config)
            n, _ = _compare(pre, post, alt, where="hip t=%d" % t)
            n_checked += n
            eng.upload_state(saved)                 # the rollout goes on from where it was
        eng.step(torch.from_numpy(a).to(eng.device))
    assert n_checked > 2000

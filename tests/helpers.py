"""Shared helpers of the parity tests."""
import numpy as np

from metadrive_ped_amd import abi

STATE_KEYS_EXACT = ["shape", "dyn", "nav", "pid", "action", "flags", "obs", "reward", "cost", "step_info", "need_reset"]


def make_cfg(cs_dist, **kw):
    from metadrive_ped_amd.config import make_config
    base = dict(num_envs=16, num_scenarios=16, block_dist_config=cs_dist, traffic_density=0.1)
    base.update(kw)
    return make_config(base)


def scripted_actions(E, A, step, seed=0):
    """Deterministic pseudo-random actions in [-1,1]^2, biased to drive forward so that episodes reach
    traffic, curves, lines and crashes."""
    rng = np.random.RandomState(seed * 100003 + step)
    a = rng.uniform(-1, 1, size=(E, A, 2)).astype(np.float32)
    a[..., 0] *= 0.3
    a[..., 1] = np.abs(a[..., 1]) * 0.8 + 0.2
    # a few envs brake / steer hard
    a[::7, :, 1] = -0.5
    a[3::11, :, 0] = 0.9
    return a


def assert_state_equal(gpu, ref, keys=STATE_KEYS_EXACT, where=""):
    """Bit-exact comparison of two state dicts (numpy).  Floats are compared through their bit patterns
    so that NaN == NaN and -0.0 != 0.0 are both visible."""
    for k in keys:
        g, r = gpu[k], ref[k]
        gb = np.ascontiguousarray(g).view(np.uint8)
        rb = np.ascontiguousarray(r).view(np.uint8)
        if not np.array_equal(gb, rb):
            bad = np.nonzero(gb.reshape(len(g), -1) != rb.reshape(len(r), -1))[0]
            rows = np.unique(bad)[:5]
            msg = ["{} state['{}'] differs in {} rows (of {}); first rows: {}".format(where, k, len(np.unique(bad)), len(g), rows)]
            for row in rows[:3]:
                msg.append("  row {} gpu={} ref={}".format(row, g[row], r[row]))
            raise AssertionError("\n".join(msg))


def flag_names(v):
    names = [n for n in dir(abi) if n.startswith("FL_")]
    return [n for n in names if v & getattr(abi, n)]

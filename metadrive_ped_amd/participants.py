"""User-spawned traffic participants: pedestrians and cyclists.

The reference's way (tests/test_functionality/test_pedestrian.py:38-55):

    ped = env.engine.spawn_object(Pedestrian, position=[30, 0], heading_theta=0)
    ped.set_velocity([1, 0], 1, in_local_frame=True)

A participant is a body of its own kind -- pedestrian: cylinder r = 0.35 m (traffic_participants/pedestrian.py:15-30),
cyclist: box 1.75 x 0.4 m (cyclist.py:28,42-47) -- that lidar beams hit, that an agent crashes into (`crash_human`,
terminal with crash_human_done) and that moves with the linear velocity it was given until told otherwise; it lives
until the environment resets (BaseEngine.reset clears spawned objects).  Here it takes a free mover slot of the
chosen environments; the step kernel moves it (md_integrate_mover, include/md_entity.h).

These functions work on the numpy state arrays (HostScene.state layout): the engine pulls the three arrays involved,
applies the edit and pushes them back -- spawning is a rare, host-driven event -- and the tests apply the same edit
to the oracle's state.
"""
import numpy as np

from metadrive_ped_amd import abi

KINDS = {"pedestrian": abi.KIND_PEDESTRIAN, "cyclist": abi.KIND_CYCLIST,
         abi.KIND_PEDESTRIAN: abi.KIND_PEDESTRIAN, abi.KIND_CYCLIST: abi.KIND_CYCLIST}
HALF_EXTENT = {abi.KIND_PEDESTRIAN: (0.35, 0.35), abi.KIND_CYCLIST: (1.75 / 2, 0.4 / 2)}


def _env_index(envs, E):
    idx = np.arange(E) if envs is None else np.atleast_1d(np.asarray(envs, dtype=np.int64))
    if idx.size == 0 or idx.min() < 0 or idx.max() >= E:
        raise IndexError("envs out of range [0, {})".format(E))
    return idx


def spawn(state, E, cap, A, kind, position, heading_theta=0.0, envs=None):
    """Put a participant of `kind` ("pedestrian" | "cyclist") at `position` ([x, y], or one per env [n, 2]) into the
    highest mover slot that is free -- never used by the scene (reset snapshot) and empty now -- in EVERY chosen env;
    returns that slot index: the handle for set_velocity / clear.  Raises if no slot is free everywhere."""
    if kind not in KINDS:
        raise ValueError("spawn_object: kind must be 'pedestrian' or 'cyclist', got {!r}".format(kind))
    k = KINDS[kind]
    idx = _env_index(envs, E)
    sh = state["shape"].reshape(E, cap)
    sh0 = state["shape0"].reshape(E, cap)
    free = ((sh["flags"][idx] & abi.KIND_MASK) == 0) & ((sh0["flags"][idx] & abi.KIND_MASK) == 0)
    free[:, :A] = False
    ok = np.nonzero(free.all(axis=0))[0]
    if ok.size == 0:
        raise RuntimeError("spawn_object: no mover slot is free in all {} chosen envs; build the env with a larger "
                           "`mover_capacity` (now {})".format(len(idx), cap))
    slot = int(ok[-1])
    pos = np.broadcast_to(np.asarray(position, dtype=np.float32).reshape(-1, 2), (len(idx), 2))
    hd = np.broadcast_to(np.asarray(heading_theta, dtype=np.float32).reshape(-1), (len(idx), ))
    rec = sh[idx, slot]
    rec["cx"], rec["cy"] = pos[:, 0], pos[:, 1]
    rec["c"], rec["s"] = np.cos(hd), np.sin(hd)
    rec["hl"], rec["hw"] = HALF_EXTENT[k]
    rec["flags"] = k | abi.F_ALIVE
    rec["aux"] = -1
    sh[idx, slot] = rec
    d = state["dyn"].reshape(E, cap)
    drec = np.zeros(len(idx), dtype=abi.DYN_DT)
    drec["heading"] = hd
    drec["last_x"], drec["last_y"] = pos[:, 0], pos[:, 1]
    drec["last_c"], drec["last_s"] = rec["c"], rec["s"]
    d[idx, slot] = drec
    state["flags"].reshape(E, cap)[idx, slot] = 0
    return slot


def set_velocity(state, E, cap, slot, direction, value=None, in_local_frame=False, envs=None):
    """BaseObject.set_velocity (base_object.py:310-328): the direction is normalised to `value` m/s when a value is
    given, else taken as the velocity itself; in_local_frame: direction is (forward, left) of the participant."""
    idx = _env_index(envs, E)
    sh = state["shape"].reshape(E, cap)
    d = state["dyn"].reshape(E, cap)
    kinds = sh["flags"][idx, slot] & abi.KIND_MASK
    if not np.isin(kinds, (abi.KIND_PEDESTRIAN, abi.KIND_CYCLIST)).all() or not ((sh["flags"][idx, slot] & abi.F_ALIVE) != 0).all():
        raise ValueError("set_velocity: slot {} does not hold a participant in every chosen env".format(slot))
    dx, dy = float(direction[0]), float(direction[1])
    c, s = sh["c"][idx, slot], sh["s"][idx, slot]
    if in_local_frame:
        vx, vy = dx * c - dy * s, dx * s + dy * c
    else:
        vx, vy = np.full(len(idx), dx, np.float32), np.full(len(idx), dy, np.float32)
    if value is not None:
        ratio = np.float32(value) / (np.hypot(vx, vy) + np.float32(1e-6))
        vx, vy = vx * ratio, vy * ratio
    d["steering"][idx, slot] = vx
    d["throttle"][idx, slot] = vy
    d["speed"][idx, slot] = np.hypot(vx, vy)


def clear(state, E, cap, slot, envs=None):
    """engine.clear_objects for a participant slot."""
    idx = _env_index(envs, E)
    sh = state["shape"].reshape(E, cap)
    kinds = sh["flags"][idx, slot] & abi.KIND_MASK
    if not np.isin(kinds, (abi.KIND_NONE, abi.KIND_PEDESTRIAN, abi.KIND_CYCLIST)).all():
        raise ValueError("clear_objects: slot {} holds something that was not spawned as a participant".format(slot))
    rec = np.zeros(len(idx), dtype=abi.SHAPE_DT)
    rec["aux"] = -1
    sh[idx, slot] = rec
    state["dyn"].reshape(E, cap)[idx, slot] = np.zeros(len(idx), dtype=abi.DYN_DT)


__all__ = ["spawn", "set_velocity", "clear", "KINDS", "HALF_EXTENT"]

"""Host build workers (metadrive_ped_amd/hostpool.py): started before the GPU is touched, kept, never re-forked."""
import os

import numpy as np
import pytest

from metadrive_ped_amd import hostpool
from metadrive_ped_amd.config import make_config
from metadrive_ped_amd.engine import HostScene


def _same(h1, h2):
    assert h1.cap == h2.cap
    for k in h1.state:
        assert np.array_equal(h1.state[k].view(np.uint8), h2.state[k].view(np.uint8)), k
    for k in h1.world.arrays:
        assert np.array_equal(np.ascontiguousarray(h1.world.arrays[k]).view(np.uint8),
                              np.ascontiguousarray(h2.world.arrays[k]).view(np.uint8)), k


def test_workers_build_what_the_process_builds():
    user = dict(num_envs=40, num_scenarios=40, map=3, traffic_density=0.1, start_seed=300)
    pool = hostpool.get()
    assert pool is not None and pool.alive()            # conftest started it
    h_pool = HostScene(make_config(user))
    h_ser = HostScene(make_config(dict(user, build_workers=1)))
    _same(h_pool, h_ser)


def test_memo_gives_fresh_equal_objects():
    hostpool.clear_memo()
    user = dict(num_envs=20, num_scenarios=20, map=2, traffic_density=0.1, start_seed=77, build_cache=True)
    h1 = HostScene(make_config(user))
    n = len(hostpool._MEMO)
    assert n == 20
    h2 = HostScene(make_config(dict(user, mover_capacity=h1.cap + 8)))      # another capacity: the same built scenes, cut differently
    assert len(hostpool._MEMO) == n and h2.cap == h1.cap + 8
    h3 = HostScene(make_config(user))
    _same(h1, h3)
    assert h1.state["shape0"] is not h3.state["shape0"]
    hostpool.clear_memo()


def test_worker_error_is_reported():
    with pytest.raises(RuntimeError, match="host build failed in a worker"):
        hostpool.get().map(os.path.getsize, ["/nonexistent/%d" % i for i in range(4)])
    assert hostpool.get().alive()                       # the workers survive a failing job
    assert hostpool.get().map(abs, [-1, -2, -3]) == [1, 2, 3]


def test_start_refuses_after_gpu_init(monkeypatch):
    pool = hostpool._POOL
    monkeypatch.setattr(hostpool, "_POOL", None)
    monkeypatch.setattr(hostpool, "gpu_initialised", lambda: True)
    with pytest.raises(RuntimeError, match="already initialised the GPU"):
        hostpool.start()
    assert hostpool.get() is None                       # and builds fall back to this process
    h = HostScene(make_config(dict(num_envs=17, num_scenarios=17, map=2, traffic_density=0.1)))
    assert h.E == 17
    monkeypatch.setattr(hostpool, "_POOL", pool)


@pytest.mark.gpu
def test_reset_with_new_seeds_never_forks_the_gpu_process(monkeypatch):
    """A process with a GPU context builds new scenes through the workers it already has: no fork, no exec, no new child."""
    import subprocess
    import time
    import torch
    from metadrive_ped_amd.envs.metadrive_env import BatchedMetaDriveEnv
    env = BatchedMetaDriveEnv(dict(num_envs=256, num_scenarios=256, map=3, traffic_density=0.1, start_seed=0))
    env.reset()
    assert torch.cuda.is_initialized() and hostpool.get() is not None

    def boom(*a, **k):
        raise AssertionError("the GPU process tried to start a child process")
    monkeypatch.setattr(os, "fork", boom)
    monkeypatch.setattr(subprocess, "Popen", boom)
    t0 = time.time()
    env.reset(seed=5000)
    dt = time.time() - t0
    a = torch.zeros(256, 2, device=env.engine.device)
    a[:, 1] = 0.5
    for _ in range(5):
        env.step(a)
    assert env.engine.host.seeds[:3] == [5000, 5001, 5002]
    assert dt < 30.0, dt
    env.close()


def test_sticky_jobs_meet_the_same_worker_again():
    pool = hostpool.get()
    a = pool.map(hostpool.worker_pid, list(range(64)), sticky=True)
    b = pool.map(hostpool.worker_pid, list(range(64)), sticky=True)
    assert a == b and len(set(a)) > 1


def test_random_traffic_reset_reuses_the_maps():
    """random_traffic=True: env.reset() draws new traffic on the SAME maps -- the maps are not generated again (engine._MAP_CACHE),
    the traffic differs, and what comes out equals a build from scratch with the same traffic epoch."""
    from metadrive_ped_amd import engine
    user = dict(num_envs=6, num_scenarios=6, map=3, traffic_density=0.2, start_seed=900, random_traffic=True, build_workers=1)
    engine._MAP_CACHE.clear()
    h0 = HostScene(make_config(dict(user, traffic_epoch=0)))
    made = engine.MAPS_GENERATED[0]
    h1 = HostScene(make_config(dict(user, traffic_epoch=1)))
    assert engine.MAPS_GENERATED[0] == made                      # no map generated for the second draw
    assert not np.array_equal(h0.state["shape0"].view(np.uint8), h1.state["shape0"].view(np.uint8))
    for k in h0.world.arrays:                                     # same maps
        assert np.array_equal(np.ascontiguousarray(h0.world.arrays[k]).view(np.uint8), np.ascontiguousarray(h1.world.arrays[k]).view(np.uint8)), k
    engine._MAP_CACHE.clear()
    _same(h1, HostScene(make_config(dict(user, traffic_epoch=1))))
    # respawn tables are added to a COPY: the cached tables stay as generated
    HostScene(make_config(dict(user, traffic_mode="respawn", traffic_epoch=0)))
    assert all(mt.respawn is None for mt in engine._MAP_CACHE.values())


def test_config_errors_keep_their_type_through_the_workers():
    """A scene the builder refuses (more movers than the capacity) raises ValueError whether it is built here or on a worker."""
    user = dict(num_envs=20, num_scenarios=20, map=7, traffic_density=0.9, start_seed=450, mover_capacity=12)
    with pytest.raises(ValueError, match="mover"):
        HostScene(make_config(dict(user, build_workers=1)))
    with pytest.raises(ValueError, match="mover"):
        HostScene(make_config(user))

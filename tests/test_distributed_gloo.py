"""N>1 path on CPU: two gloo ranks each own half of the env range (shard_config), step their shard
with the oracle, gather their packed step-output slab (obs | reward | terminated, truncated, flags) with the
ONE collective bench.py uses (sharding.gather_step_slab = one all_gather_into_tensor), and the gathered batch must
be bit-identical to the unsharded single-process batch -- i.e. results
do not depend on how many GPUs the batch is split over."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, E_per, steps, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from collections import OrderedDict
    from metadrive_ped_amd.mapgen.pg import BLOCK_TYPE_DISTRIBUTION_V2
    from metadrive_ped_amd.config import make_config
    from metadrive_ped_amd.engine import HostScene
    from metadrive_ped_amd.sharding import gather_step_slab, shard_config, slab_layout, split_step_slab
    from helpers import scripted_actions
    import oracle_binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = OrderedDict((k, 0.0) for k in BLOCK_TYPE_DISTRIBUTION_V2)
    d["Curve"], d["Straight"] = 0.6, 0.4
    base = make_config(dict(num_envs=E_per, num_scenarios=E_per * world, block_dist_config=d, start_seed=20))
    cfg = shard_config(base, rank, world)
    host = HostScene(cfg)
    o = ob.OracleWorld(host)
    o.reset()
    for t in range(steps):
        a = scripted_actions(E_per * world, 1, t)[rank * E_per:(rank + 1) * E_per]
        o.step(a)
    # the rank's step-output slab exactly as BatchedEngine lays it out (obs | reward | done_out in one allocation), then the ONE
    # collective bench.py's with_gather leg issues
    layout, total = slab_layout(E_per, 1, o.obs.shape[-1])
    slab = torch.zeros(total, dtype=torch.uint8)
    for name, arr in (("obs", o.obs), ("reward", o.state["reward"]), ("done_out", o.state["done_out"])):
        off, n = layout[name]
        slab[off:off + n] = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1))
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    gathered = gather_step_slab(slab)
    dist.all_gather_into_tensor = real
    assert len(calls) == 1
    g = split_step_slab(gathered, E_per, 1, o.obs.shape[-1])
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), **{k: v.numpy() for k, v in g.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process(tmp_path, cs_dist):
    import torch.multiprocessing as mp
    from helpers import make_cfg, scripted_actions
    from metadrive_ped_amd.engine import HostScene
    import oracle_binding as ob
    ob.load()
    E_per, world, steps = 3, 2, 40
    port = _free_port()
    mp.spawn(_worker, args=(world, port, E_per, steps, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    host = HostScene(make_cfg(cs_dist, num_envs=E_per * world, num_scenarios=E_per * world, start_seed=20))
    o = ob.OracleWorld(host)
    o.reset()
    for t in range(steps):
        o.step(scripted_actions(E_per * world, 1, t))
    from metadrive_ped_amd import abi
    n = E_per * world
    assert got["obs"].shape == (n, 1, 259)
    assert got["obs"].tobytes() == o.obs.tobytes()
    assert got["reward"].tobytes() == o.state["reward"].tobytes()
    fl = o.state["flags"].reshape(n, -1)[:, 0]
    assert (got["flags"][:, 0].astype(np.int64) & 0xFFFF == fl & 0xFFFF).all()          # the agents' whole step flag words
    assert (got["terminated"][:, 0] == ((fl & abi.FL_TERMINATED) != 0)).all()
    assert (got["truncated"][:, 0] == ((fl & abi.FL_TRUNCATED) != 0)).all()
    assert (fl & abi.FL_ON_LANE).any()                                                  # the flag words are not trivially zero

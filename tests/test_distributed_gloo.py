"""N>1 path on CPU: two gloo ranks each own half of the env range (shard_config), step their shard
with the oracle, gather obs/reward with the same collective bench.py uses (all_gather_into_tensor),
and the gathered batch must be bit-identical to the unsharded single-process batch -- i.e. results
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
    from metadrive_ped_amd.sharding import gather_step_outputs, shard_config
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
    g = gather_step_outputs(dict(obs=torch.from_numpy(o.obs.copy()), reward=torch.from_numpy(o.state["reward"].copy()),
                                 flags=torch.from_numpy(o.state["flags"].reshape(E_per, -1)[:, 0].astype(np.int64))))
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
    assert got["obs"].shape == (E_per * world, 259)
    assert got["obs"].tobytes() == o.obs.tobytes()
    assert got["reward"].tobytes() == o.state["reward"].tobytes()
    assert (got["flags"] == o.state["flags"].reshape(E_per * world, -1)[:, 0]).all()

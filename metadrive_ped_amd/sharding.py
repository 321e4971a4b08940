"""Multi-GPU layout: environments are independent worlds, so the global batch is cut into contiguous
env ranges, one per rank (= one process per GPU), with NO collective inside step().  Scenario seeds
depend on the GLOBAL env index, so the union of the shards is bit-identical to the unsharded batch
(SURVEY 8e).  The only exchange a trainer may want is presenting the per-rank slabs as one batch:
gather_step_outputs() is that single collective -- torch.distributed all_gather_into_tensor, which is
RCCL on ROCm ("nccl" backend, over xGMI) and gloo in the CPU tests.
"""
import copy


def shard_config(cfg, rank, world_size):
    """Config of rank `rank`'s shard: same scenario table, env range [rank*E, (rank+1)*E)."""
    c = copy.copy(cfg)
    c["env_seed_offset"] = cfg.get("env_seed_offset", 0) + rank * cfg["num_envs"]
    return c


def gather_step_outputs(tensors, group=None):
    """All-gather a dict of per-rank tensors (leading dim = this rank's envs) into global tensors.
    One collective per tensor; obs is ~1 KB/agent so at 8192 envs/GPU this is ~8.6 MB per rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = {}
    for k, t in tensors.items():
        t = t.contiguous()
        g = torch.empty((world * t.shape[0], ) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(g, t, group=group)
        out[k] = g
    return out

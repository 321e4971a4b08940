"""Multi-GPU layout: environments are independent worlds, so the global batch is cut into contiguous
env ranges, one per rank (= one process per GPU), with NO collective inside step().  Scenario seeds
depend on the GLOBAL env index, so the union of the shards is bit-identical to the unsharded batch
(SURVEY 8e).  The only exchange a trainer may want is presenting the per-rank step outputs as one batch:
gather_step_slab() is that single collective -- ONE torch.distributed all_gather_into_tensor (RCCL on ROCm:
"nccl" backend, over xGMI; gloo in the CPU tests) of the rank's output slab obs | reward | done_out, which the
engine keeps in one allocation (BatchedEngine.out_slab; done_out carries terminated, truncated and the step's
flag word, include/mdstep.h).  split_step_slab() turns the gathered bytes back into typed per-rank views.
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


def slab_layout(E, A, obs_dim):
    """Byte layout of a rank's step-output slab (BatchedEngine._pack_step_outputs): name -> (offset, nbytes), total."""
    layout, at = {}, 0
    for k, n in (("obs", E * A * obs_dim * 4), ("reward", E * A * 4), ("done_out", E * A * 4)):
        layout[k] = (at, n)
        at = (at + n + 15) // 16 * 16
    return layout, at


def gather_step_slab(slab, out=None, group=None):
    """ONE collective: every rank's output slab (uint8, same size on every rank) -> [world, slab_bytes] on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    slab = slab.contiguous().view(torch.uint8).reshape(-1)
    if out is None:
        out = torch.empty((world, slab.numel()), dtype=torch.uint8, device=slab.device)
    dist.all_gather_into_tensor(out.view(-1), slab, group=group)
    return out


def split_step_slab(gathered, E, A, obs_dim):
    """Typed views of a gathered [world, slab_bytes] tensor: obs [world*E, A, obs_dim] f32, reward [world*E, A] f32,
    terminated / truncated [world*E, A] bool, flags [world*E, A] int16 (MD_FL_* of the step).  No copy for world == 1;
    otherwise one reshaping copy per field (the parts of different ranks are not adjacent in memory)."""
    import torch
    world = gathered.shape[0]
    layout, total = slab_layout(E, A, obs_dim)
    assert gathered.shape[1] == total, (gathered.shape, total)

    def part(name):
        o, n = layout[name]
        return gathered[:, o:o + n].contiguous()
    obs = part("obs").view(torch.float32).view(world * E, A, obs_dim)
    reward = part("reward").view(torch.float32).view(world * E, A)
    done = part("done_out")
    tt = done.view(torch.bool).view(world * E, A, 4)
    flags = done.view(torch.int16).view(world * E, A, 2)[:, :, 1]
    return dict(obs=obs, reward=reward, terminated=tt[:, :, 0], truncated=tt[:, :, 1], flags=flags)

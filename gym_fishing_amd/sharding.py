"""Multi-GPU sharding of the env batch.

Envs never interact, so the batch splits into contiguous blocks, one per rank
(one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).  The noise
stream is keyed by the GLOBAL env index (env_offset + i), so every env's trajectory
is the same for any world size.  The only collective on the path is one all-reduce of
the 4-double episodic-return record per rollout; it is latency-bound (32 bytes), so
no bucketing or ring tuning applies.
"""
import math


def shard_range(total_envs, rank, world_size, multiple=4):
    """Contiguous [offset, offset + count) block of `rank`.  Offsets are multiples of
    `multiple` (4: noise quads and 16-byte rows must not straddle shards); the blocks
    tile [0, total_envs) exactly."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank %r / world_size %r" % (rank, world_size))
    if total_envs < 0:
        raise ValueError("total_envs must be >= 0")
    units = -(-total_envs // multiple)            # ceil: number of `multiple`-sized groups
    base, extra = divmod(units, world_size)
    start_u = rank * base + min(rank, extra)
    count_u = base + (1 if rank < extra else 0)
    start = min(start_u * multiple, total_envs)
    end = min((start_u + count_u) * multiple, total_envs)
    return start, end - start


def dist_info():
    """(rank, world_size, local_rank) from torch.distributed / the torchrun env."""
    import os
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", dist.get_rank()))
    except Exception:  # noqa: BLE001
        pass
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def summarize_record(rec):
    """{sum R, sum R^2, n_episodes, sum length} -> dict with mean/std (host floats)."""
    s, s2, n, L = (float(x) for x in rec.tolist())
    out = {"sum_return": s, "sum_sq_return": s2, "n_episodes": n, "sum_length": L}
    if n > 0:
        mean = s / n
        out["mean_return"] = mean
        out["std_return"] = math.sqrt(max(s2 / n - mean * mean, 0.0))
        out["mean_length"] = L / n
    return out


def all_reduce_record(rec, group=None):
    """Sum the 4-double record across ranks (returns the reduced tensor).  RCCL ("nccl")
    reduces the device tensor in place; any other backend (gloo in CPU tests, or gloo ranks
    sharing one GPU) goes through a host copy -- 32 bytes."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rec
    if rec.is_cuda and dist.get_backend(group) != "nccl":
        host = rec.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        rec.copy_(host)
        return rec
    dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=group)
    return rec


def make_sharded(env_id, total_envs, rank=None, world_size=None, **kwargs):
    """This rank's shard of a `total_envs`-wide vec-env: make(id, num_envs=count,
    env_offset=offset, ...).  Every rank passes the same seed."""
    from . import ensure_dmabuf_ipc, make
    r, w, _ = dist_info()
    if (w if world_size is None else world_size) > 1:
        ensure_dmabuf_ipc()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    offset, count = shard_range(total_envs, rank, world_size)
    if count == 0:
        raise ValueError("rank %d has no envs (total %d over %d ranks)" % (rank, total_envs, world_size))
    return make(env_id, num_envs=count, env_offset=offset, **kwargs)

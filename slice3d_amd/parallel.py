"""One-process-per-GPU sharding of the hot path.

Inference shards with NO data-path collective except one final gather:
  * object-parallel: rank r handles objects r, r+N, ... (what bench.py measures; zero communication)
  * query-parallel (one object, e.g. a dense 256^3 grid): every rank runs the cheap encoder itself
    (242 GFLOP, ~4 ms — cheaper than broadcasting the 183 MB latent over one 153 GB/s xGMI link) and
    decodes a contiguous 1/N slab of the queries; one all_gather of the fp32 logits at the end.
The functions below are backend-agnostic (`nccl` = RCCL on the GPUs, `gloo` in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) slab of `n_items` for `rank`; slabs differ by at most one item and tile
    [0, n_items) exactly (empty slabs allowed when n_items < world)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def object_indices(n_objects, rank, world):
    return list(range(rank, n_objects, world))


def gather_slabs(local, n_total, group=None):
    """all_gather variable-length 1-D slabs (as produced by shard_range) into the full vector."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    assert local.numel() == sizes[rank], (local.numel(), sizes[rank])
    pad = max(sizes)
    buf = torch.zeros(pad, dtype=local.dtype, device=local.device)
    buf[:local.numel()] = local.reshape(-1)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)])


def decode_points_sharded(decode_fn, qry, group=None):
    """Query-parallel decode of one object: `decode_fn(qry_slab (1,q,3)) -> (1,q)` runs on this rank's
    slab; returns the full (1,Q) result on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = qry.shape[1]
    lo, hi = shard_range(n, rank, world)
    local = decode_fn(qry[:, lo:hi].contiguous()) if hi > lo else qry.new_zeros((1, 0))
    return gather_slabs(local.reshape(-1), n, group).view(1, n)


def all_reduce_mean_(flat, group=None):
    """Data-parallel gradient exchange: in-place mean over ranks of ONE flat bucket.
    With RCCL this is a single all-reduce of the 83 MB gradient buffer; no-op for a single process."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(dist.get_world_size(group))
    return flat

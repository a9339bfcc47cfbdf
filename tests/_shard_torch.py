"""TEST INFRASTRUCTURE (round 2's torch.distributed twin of csrc/shard.hip; the product path is amk_shard_*).  Multi-GPU sharding of the batched-scenes sweep: scenes are independent, so
they are block-partitioned over ranks (one process per GPU) with no data-path collective; the one
exchange step is the gather of the per-scene results (4 doubles of control + flags) so that every
rank -- the reference's single FSM would be rank 0 -- holds all controls.  RCCL over xGMI on the
GPU box (`backend="nccl"`), gloo in the CPU tests."""
import torch
import torch.distributed as dist


def scene_range(rank, world, total):
    """Contiguous block of global scene ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_controls(u_local, counts=None, out=None):
    """all_gather of per-scene controls [S_local, 4] -> [S_total, 4] in global scene order.
    Uneven shards are padded to the largest one (RCCL/gloo all_gather wants equal shapes).
    out: optional preallocated [world * S_local, 4] for the equal-shard case (one collective, no temporaries)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and out is None):
        return u_local
    world = dist.get_world_size()   # (world 1 with `out`: the collective still runs -- the single-GPU RCCL exercise)
    if counts is None:
        if out is not None:
            dist.all_gather_into_tensor(out, u_local.contiguous())
            return out
        counts = [u_local.shape[0]] * world
    m = max(counts)
    pad = u_local
    if u_local.shape[0] < m:
        pad = torch.cat([u_local, u_local.new_zeros((m - u_local.shape[0],) + tuple(u_local.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous())
    return torch.cat([b[:c] for b, c in zip(bufs, counts)])


def max_over_ranks(seconds, device):
    """Wall time of the slowest rank (bench.py's timing contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

"""Multi-GPU layout: the instance batch is split into contiguous, independent sub-batches, one per rank
(one process per GPU).  Instances share no state, so there is no collective on the data path; the
only cross-rank operations are the barrier around the timed region and the max-reduction of its duration."""
import os


def shard_range(n_total, rank, world):
    """Contiguous [first, last) slice of a global instance range owned by `rank`; sizes differ by at most 1."""
    base, extra = divmod(n_total, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (the timed region's duration); identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())

"""Source-point data parallelism: the Abbe sum over source points is a plain
sum, so each rank takes a contiguous shard of the (dy,dx) list and the partial intensities
are combined by a single all-reduce.  Pure index arithmetic -- usable with gloo on CPU tensors
for tests and with RCCL ("nccl" backend on ROCm) on the GPU box."""
import os


def shard_bounds(S: int, rank: int, world: int):
    """Contiguous, balanced split of range(S): the first S % world ranks get one extra point."""
    base, extra = divmod(S, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def resolve_group(group):
    """None unless the caller passed a group, or a default group exists and
    LITHO_SHARD_SOURCES=1 asks for sharding."""
    if group is not None:
        return group
    if os.environ.get("LITHO_SHARD_SOURCES", "0") != "1":
        return None
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.group.WORLD
    return None

"""Sharding of independent IQ channels over the GPUs of one node.

Channels never interact (one reference `Demod` per stream, simple_fm.rs:137), so the path
shards with NO collective: rank r owns a contiguous block of channels and runs the same
kernel on its own GPU.  An optional result gather (off the hot path) is provided for callers
that want every rank's audio on rank 0.
"""


def channel_range(total_channels, world_size, rank):
    """Contiguous block partition; the first (total % world) ranks get one extra channel."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(total_channels, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def owner_of(channel, total_channels, world_size):
    base, extra = divmod(total_channels, world_size)
    edge = extra * (base + 1)
    if channel < edge:
        return channel // (base + 1)
    return extra + (channel - edge) // base if base else world_size - 1


def gather_audio(local_audio, total_channels, group=None, dst=0):
    """Gather per-channel int16 arrays from every rank to `dst` (torch.distributed; gloo or RCCL).

    local_audio: list of 1-D numpy int16 arrays for this rank's channel_range, in order.
    Returns the full list on dst, None elsewhere.  Off the timed path by construction.
    """
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    gathered = [None] * world if rank == dst else None
    dist.gather_object([a.tobytes() for a in local_audio], gathered, dst=dst, group=group)
    if rank != dst:
        return None
    import numpy as np
    full = []
    for r in range(world):
        lo, hi = channel_range(total_channels, world, r)
        assert len(gathered[r]) == hi - lo
        full.extend(np.frombuffer(b, dtype=np.int16) for b in gathered[r])
    return full

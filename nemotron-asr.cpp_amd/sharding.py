"""Stream sharding across the GPUs of one node.

Streams are independent units (reference: all state is per stream, src/nemo-stream.h:177-262),
so rank r simply owns streams {r*B .. r*B+B-1} on GPU r with replicated weights.  There is no
data-path collective; torch.distributed is used only to line the ranks up and to take the
max-over-ranks of the elapsed time.  Backend "nccl" (= RCCL) on GPUs, "gloo" in CPU tests."""
from __future__ import annotations


def stream_ids(rank: int, world: int, streams_per_rank: int) -> list:
    assert 0 <= rank < world
    return list(range(rank * streams_per_rank, (rank + 1) * streams_per_rank))


def owner_of(stream_id: int, streams_per_rank: int) -> int:
    return stream_id // streams_per_rank


def barrier(dist, sync_device=None):
    if sync_device is not None:
        sync_device()
    if dist is not None:
        dist.barrier()
    if sync_device is not None:
        sync_device()


def max_over_ranks(dist, value: float, device="cpu") -> float:
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_rtfx(world: int, audio_s_per_rank: float, elapsed_max: float) -> float:
    """Whole-job audio seconds per wall second (weak scaling: every rank processes its own audio)."""
    return world * audio_s_per_rank / elapsed_max

"""Stream sharding across the GPUs of one node.

Streams are independent units (reference: all state is per stream, src/nemo-stream.h:177-262), weights are
replicated, and there is no data-path collective; torch.distributed is used only to line the ranks up and to
take the max-over-ranks of the elapsed time.  Backend "nccl" (= RCCL) on GPUs, "gloo" in CPU tests.

ONE placement rule for the whole repo: stream s lives on GPU  s mod G  (the socket server's
`--devices` rule, host/nemo_server.cpp: "stream s is served by entry s mod count").  Rank r of a G-rank
bench.py job therefore owns streams r, r + G, r + 2G, ... -- the streams the server would give GPU r."""
from __future__ import annotations

PLACEMENT = "stream s on GPU s mod G"


def stream_ids(rank: int, world: int, streams_per_rank: int) -> list:
    assert 0 <= rank < world
    return list(range(rank, world * streams_per_rank, world))


def owner_of(stream_id: int, world: int) -> int:
    return stream_id % world


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

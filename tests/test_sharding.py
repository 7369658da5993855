"""N > 1 path on CPU: world_size-2 gloo job.  Each rank runs ITS shard of the streams (through the
CPU oracle, standing in for the per-GPU engine) and the ranks agree on the max elapsed time; the
union of the shards equals the single-process result stream by stream."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as ge  # noqa: E402  (spawned workers import this module without conftest)

ge.load_package()
from nemotron_asr_amd import sharding, synth  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, out_dir):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    import __graft_entry__ as ge
    ge.load_package()
    from oracle import binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["NASR_ORACLE_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W = synth.make_weights(n_layers=1)
    model = ob.OracleModel(W, 1)
    ids = sharding.stream_ids(rank, world, B)
    sharding.barrier(dist)
    import time
    t0 = time.perf_counter()
    toks = {}
    for sid in ids:
        st = ob.OracleStream(model, 0)
        pcm = synth.make_pcm(sid, 0.8)
        toks[sid] = st.process(pcm) + st.finalize()
    sharding.barrier(dist)
    el = time.perf_counter() - t0
    mx = sharding.max_over_ranks(dist, el)
    assert mx >= el
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, el, toks))
    if rank == 0:
        assert max(g[1] for g in gathered) == pytest.approx(mx)
        merged = {}
        for g in gathered:
            merged.update(g[2])
        np.save(os.path.join(out_dir, "merged.npy"), np.array([merged[k] for k in sorted(merged)], dtype=object), allow_pickle=True)
        with open(os.path.join(out_dir, "rtfx.txt"), "w") as f:
            f.write(str(sharding.aggregate_rtfx(world, B * 0.8, mx)))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding(tmp_path):
    from oracle import binding as ob
    world, B = 2, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    merged = np.load(tmp_path / "merged.npy", allow_pickle=True)
    assert len(merged) == world * B
    W = synth.make_weights(n_layers=1)
    model = ob.OracleModel(W, 1)
    for sid in range(world * B):
        st = ob.OracleStream(model, 0)
        ref = st.process(synth.make_pcm(sid, 0.8)) + st.finalize()
        assert list(merged[sid]) == ref
        assert sharding.owner_of(sid, world) == sid % world       # the server's rule: stream s on device s mod count
    assert float((tmp_path / "rtfx.txt").read_text()) > 0


def test_stream_partition_is_exact():
    for world in (1, 2, 4, 8):
        per_rank = [sharding.stream_ids(r, world, 64) for r in range(world)]
        assert sorted(sum(per_rank, [])) == list(range(64 * world))                     # every stream exactly once
        assert all(len(ids) == 64 for ids in per_rank)
        assert all(sharding.owner_of(s, world) == r for r, ids in enumerate(per_rank) for s in ids)

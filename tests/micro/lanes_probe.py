"""micro-benchmark: N independent engines ("lanes": own HIP streams, own host thread, own copy of the weights) on ONE GPU,
each stepping total_batch / N streams, against one engine stepping them all.  Independent launch chains share the chip
(tests/micro/overlap_probe.py); this measures what that is worth at a given batch / lookahead.
Measured (round 1): 64 streams x R=13: 1 lane 17 300 RTFx, 4 lanes x 16 streams 9 400; 8 streams x R=0: 1 lane 420, 2 lanes 380,
8 lanes x 1 stream 146 -- batching rows into one chain beats concurrent chains everywhere; lanes are for several GPUs.
usage: lanes_probe.py <total_batch> <right_context> <lanes> [steps] [pipeline]"""
import ctypes as C
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from nemotron_asr_amd import capi, synth  # noqa: E402


NL = int(os.environ.get("LAYERS", "24"))


def make(W, B, R, steps, lane, pipeline):
    eng = capi.Engine(W, n_layers=NL, dtype=capi.DTYPE_BF16, max_streams=B)
    eng.set_option("pipeline", pipeline)
    sts = [eng.stream(R) for _ in range(B)]
    n = synth.shift_samples(R)
    devs = [eng.upload(synth.make_pcm(100 * lane + b, steps * n / 16000 + 0.1)[:steps * n]) for b in range(B)]
    L = capi.lib()
    h = (C.c_void_p * B)(*[s.h for s in sts])
    toks = [np.zeros(16 * (1 + R), np.int32) for _ in range(B)]
    tp = (C.c_void_p * B)(*[t.ctypes.data for t in toks])
    cap = (C.c_int32 * B)(*([16 * (1 + R)] * B))
    nt = (C.c_int32 * B)()
    ns = (C.c_int32 * B)(*([n] * B))
    ptrs = [(C.c_void_p * B)(*[d + 2 * k * n for d in devs]) for k in range(steps)]

    def step(k):
        if L.nasr_engine_step(eng.h, h, B, ptrs[k], ns, tp, cap, nt, capi.FLAG_PCM_DEVICE) < 0:
            raise RuntimeError(L.nasr_last_error().decode())
    step.keep = (toks, devs)                   # the engine writes tokens into these buffers: they must outlive make()
    return eng, step, sts


def main():
    total, R, lanes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
    pipeline = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    W = synth.make_weights(n_layers=NL)
    B = total // lanes
    engs = [make(W, B, R, steps + 5, lane, pipeline) for lane in range(lanes)]
    for _, st, _s in engs:
        for k in range(5):
            st(k)
    for e, _, _s in engs:
        e.synchronize()
    go = threading.Barrier(lanes + 1)

    def run(st, e):
        go.wait()
        for k in range(5, 5 + steps):
            st(k)
        e.synchronize()

    if os.environ.get("INMAIN"):                # diagnostic: the single lane steps on the main thread
        go = threading.Barrier(1)
        t0 = time.perf_counter()
        run(engs[0][1], engs[0][0])
        dt = time.perf_counter() - t0
    else:
        ths = [threading.Thread(target=run, args=(st, e)) for e, st, _s in engs]
        for t in ths:
            t.start()
        go.wait()
        t0 = time.perf_counter()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
    audio = lanes * B * steps * synth.shift_samples(R) / 16000
    print(f"batch {total} R={R}: {lanes} lane(s) x {B} streams, pipeline={pipeline}: {1e3 * dt / steps:.3f} ms per round of steps, RTFx {audio / dt:.0f}")
    for e, _, sts in engs:                     # streams before their engine
        for st in sts:
            st.destroy()
        e.close()


if __name__ == "__main__":
    main()

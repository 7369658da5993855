"""micro-benchmark: how much does a second, independent launch chain on its own HIP stream slow the batch-1 step chain?
Engine 1 = the 24-layer batch-1 bench step; engine 2 (own stream, own host thread) = a 1-layer model stepping continuously
(front end + 1 layer + decode, ~0.14 ms per step): a much denser interferer than a decode graph overlapped with the next
encoder would be.  Prints ms per step of engine 1 alone and beside engine 2."""
import ctypes as C
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from nemotron_asr_amd import capi, synth  # noqa: E402


def make(n_layers, steps):
    W = synth.make_weights(n_layers=n_layers)
    eng = capi.Engine(W, n_layers=n_layers, dtype=capi.DTYPE_BF16, max_streams=1)
    st = eng.stream(0)
    pcm = synth.make_pcm(1, steps * 0.08 + 0.1)[:steps * 1280]
    dev = eng.upload(pcm)
    L = capi.lib()
    h = (C.c_void_p * 1)(st.h)
    tok = np.zeros(64, np.int32)
    tp = (C.c_void_p * 1)(tok.ctypes.data)
    cap = (C.c_int32 * 1)(64)
    nt = (C.c_int32 * 1)()
    ns = (C.c_int32 * 1)(1280)

    def step(k):
        ptr = (C.c_void_p * 1)(dev + 2 * k * 1280)
        if L.nasr_engine_step(eng.h, h, 1, ptr, ns, tp, cap, nt, capi.FLAG_PCM_DEVICE) < 0:
            raise RuntimeError(L.nasr_last_error().decode())
    step.keep = (tok, st)                      # buffers the engine writes into must outlive make()
    return eng, step


def main():
    n = 400
    e1, step1 = make(24, n + 20)
    e2, step2 = make(1, 20000)
    for k in range(20):
        step1(k)
    t = time.perf_counter()
    for k in range(20, 20 + n // 2):
        step1(k)
    alone = (time.perf_counter() - t) / (n // 2)
    stop = False
    count = [0]

    def side():
        k = 0
        while not stop and k < 19999:
            step2(k)
            k += 1
        count[0] = k

    th = threading.Thread(target=side)
    th.start()
    time.sleep(0.05)
    t = time.perf_counter()
    for k in range(20 + n // 2, 20 + n):
        step1(k)
    beside = (time.perf_counter() - t) / (n // 2)
    stop = True
    th.join()
    print(f"engine 1 alone {1e3 * alone:.4f} ms/step, beside a continuously stepping 1-layer engine {1e3 * beside:.4f} ms/step "
          f"({count[0]} side steps)")


if __name__ == "__main__":
    main()

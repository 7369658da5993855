"""Throughput of the diarization side-car on the GPU (BASELINE config 5 shape: 64 streams):
VAD windows per second and speaker embeddings per second, random-init MarbleNet / TitaNet-L."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import __graft_entry__ as ge

ge.load_package()
from nemotron_asr_amd import capi, synth

B = 64
W = synth.make_diar_weights()
for dtype, name in ((capi.DTYPE_BF16 | capi.DIAR_VAD_BF16, "bf16, VAD on the bf16 MFMA"), (capi.DTYPE_BF16, "bf16, VAD f32"), (capi.DTYPE_F32, "f32")):
    eng = capi.Diar(W, dtype=dtype, max_windows=8192, max_segments=96)
    # one ASR step at R = 13 brings 1.12 s of new audio per stream: 112 new VAD windows (plus 0.63 s of history)
    audio = [synth.make_pcm(s, 1.75 + 0.02)[:10080 + 111 * 160].astype(np.float32) / 32768.0 for s in range(B)]
    eng.vad(audio)
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        out = eng.vad(audio)
    dt = (time.perf_counter() - t0) / n
    nw = sum(o.size for o in out)
    print(f"[{name}] VAD: {nw} windows of {B} streams in {dt * 1e3:.2f} ms = {nw / dt / 1e3:.0f} k windows/s "
          f"= {nw / 100.0 / dt:.0f} x real time (100 windows per audio-second)", flush=True)
    segs = [synth.make_pcm(100 + s, 1.5 + 0.01)[:24000].astype(np.float32) / 32768.0 for s in range(96)]
    eng.embed(segs)
    t0 = time.perf_counter()
    for _ in range(n):
        e = eng.embed(segs)
    dt = (time.perf_counter() - t0) / n
    print(f"[{name}] TitaNet-L: {len(segs)} sub-segments in {dt * 1e3:.2f} ms = {len(segs) / dt:.0f} embeddings/s "
          f"({4.3 * len(segs) / dt / 1e3:.1f} TFLOP/s at 4.3 GFLOP each); 64 streams need ~85/s", flush=True)
    if name.startswith("bf16, VAD on"):
        # configs[4]'s own call: sub-segments as device pointers into s16 PCM already resident in HBM (no host hand-over inside the call)
        asr = capi.Engine(synth.make_weights(n_layers=1), n_layers=1, dtype=capi.DTYPE_BF16, max_streams=1)      # only for its device allocator (torch's own
        # CUDA initialisation fails in a process whose GPU was first initialised through the library, unless a profiler did it even earlier)
        ptrs = [asr.upload(synth.make_pcm(100 + s, 1.5 + 0.01)[:24000]) for s in range(96)]
        e2 = eng.embed_device_s16(ptrs)
        t0 = time.perf_counter()
        for _ in range(n):
            e2 = eng.embed_device_s16(ptrs)
        dt = (time.perf_counter() - t0) / n
        assert np.abs(e2 - e).max() < 1e-2 * np.abs(e).max(), "device s16 path disagrees with the float host path"
        print(f"[{name}] TitaNet-L, device-resident s16: {len(ptrs)} sub-segments in {dt * 1e3:.3f} ms, {eng.last_gpu_ms('embed'):.3f} ms on the device ({0.511 / dt:.0f} TFLOP/s at 511 GFLOP of GEMMs per call)", flush=True)
        asr.close()
    eng.close()

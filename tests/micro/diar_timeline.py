"""GPU timeline of one nasr_diar_embed / nasr_diar_vad call on device-resident s16 PCM (run under rocprofv3 --kernel-trace):
prints wall time per call; the trace tells kernel time and gaps."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import __graft_entry__ as ge

ge.load_package()
from nemotron_asr_amd import capi, synth

B, S = 64, 96
dW = synth.make_diar_weights()
deng = capi.Diar(dW, dtype=capi.DTYPE_BF16, max_segments=128)
asr = capi.Engine(synth.make_weights(n_layers=1), n_layers=1, dtype=capi.DTYPE_BF16, max_streams=1)       # for its device allocator
pcm = [synth.make_pcm(b, 3.0) for b in range(B)]
dev = [asr.upload(p) for p in pcm]
vad_n = [10080 - 160 + 17920] * B
seg_ptrs = [dev[i % B] + 2 * 12000 * (i // B) for i in range(S)]
for _ in range(2):
    deng.vad_device_s16(dev, vad_n); deng.embed_device_s16(seg_ptrs)
for name, fn in (("vad", lambda: deng.vad_device_s16(dev, vad_n)), ("embed", lambda: deng.embed_device_s16(seg_ptrs))):
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    print(f"{name}: wall per call {1e3 * min(t):.3f} .. {1e3 * max(t):.3f} ms", flush=True)
deng.close(); asr.close()

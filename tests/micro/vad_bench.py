import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import __graft_entry__ as ge
ge.load_package()
from nemotron_asr_amd import capi, synth
B = 64
W = synth.make_diar_weights(spk=False)
audio = [synth.make_pcm(s, 1.75 + 0.02)[:10080 + 111 * 160].astype(np.float32) / 32768.0 for s in range(B)]
for dtype, name in ((capi.DTYPE_BF16 | capi.DIAR_VAD_BF16, "VAD bf16"), (capi.DTYPE_BF16 | capi.DIAR_VAD_F16, "VAD f16"), (capi.DTYPE_BF16, "VAD f32")):
    eng = capi.Diar(W, dtype=dtype, max_windows=8192)
    eng.vad(audio)
    t0 = time.perf_counter()
    for _ in range(20):
        out = eng.vad(audio)
    dt = (time.perf_counter() - t0) / 20
    print(name, f"{dt*1e3:.3f} ms per call, {sum(o.size for o in out)} windows", flush=True)
    eng.close()

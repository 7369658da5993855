"""TitaNet-L embeddings of the bf16 engine against the F32 oracle: max |err| / scale and cosine per segment (tolerance of
tests/test_gpu_diar.py::test_speaker_embeddings_match_oracle: 6e-2, cosine > 0.999); batch == alone."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import __graft_entry__ as ge

ge.load_package()
from nemotron_asr_amd import capi, synth
from oracle import diar_binding as db

W = synth.make_diar_weights(vad=False)
om = db.DiarModel(W)
segs = [synth.make_pcm(10 + i, 1.5 + 0.01)[:24000].astype(np.float32) / 32768.0 for i in range(7)]
lens = [24000, 24000, 12000, 4321, 100, 23999, 160]
ref = np.stack([om.spk_embed(a, l) for a, l in zip(segs, lens)])
scale = np.abs(ref).max()
for dtype, name in ((capi.DTYPE_BF16, "bf16"), (capi.DTYPE_F32, "f32")):
    eng = capi.Diar(W, dtype=dtype, max_segments=4)
    got = eng.embed(segs, lens)
    assert np.isfinite(got).all(), "non-finite embedding"
    for i, (g, r) in enumerate(zip(got, ref)):
        print(f"[{name}] seg {i} len {lens[i]:6d}: max err / scale {np.abs(g - r).max() / scale:.3e}  cosine {float(g @ r / (np.linalg.norm(g) * np.linalg.norm(r))):.6f}")
    alone = eng.embed(segs[:1], lens[:1])[0]
    print(f"[{name}] batch == alone: {np.abs(alone - got[0]).max() / scale:.3e}")
    eng.close()

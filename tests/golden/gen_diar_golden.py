"""Generates tests/golden/diar_golden_v1.npz from the reference's own src/diarize_audio.cpp compiled into oracle/_ref
(run in the build container, where /root/reference exists).  Outputs only: no reference source is stored."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from nemotron_asr_amd import synth  # noqa: E402
from oracle import diar_binding as db  # noqa: E402

W = synth.make_diar_weights(spk=False)
out = {}
for n, norm in [(10080, False), (24000, True), (16000 + 77, False), (4000, True)]:
    a = synth.make_pcm(9, n / 16000.0 + 0.01)[:n].astype(np.float32) / 32768.0
    mel, tv = db.ref_logmel(a, W["vad.preprocessor.featurizer.fb"], W["vad.preprocessor.featurizer.window"], norm)
    out[f"mel_{n}_{int(norm)}"] = mel
    print(n, norm, mel.shape, tv)
np.savez_compressed(Path(__file__).parent / "diar_golden_v1.npz", **out)

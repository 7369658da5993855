"""SURVEY.md section 8 f-4 -- CPU side: the diarization oracle.
The 80-mel front end is pinned to the reference's own src/diarize_audio.cpp (compiled unmodified into oracle/_ref,
and through the committed golden vectors where the reference tree is absent).  MarbleNet / TitaNet-L cannot be pinned
(ggml graphs, no diarize.gguf): their tests are structural properties of the restatement."""
from pathlib import Path

import numpy as np
import pytest

from nemotron_asr_amd import synth
from oracle import binding as ob
from oracle import diar_binding as db

GOLD = Path(__file__).parent / "golden" / "diar_golden_v1.npz"
NEMO = Path(__file__).parent / "golden" / "nemo_diar_v1.npz"     # NeMo-generated fixtures the reference's tests hold


def _audio(seed, n):
    return (synth.make_pcm(seed, n / 16000.0 + 0.01)[:n].astype(np.float32) / 32768.0)


@pytest.fixture(scope="module")
def W():
    return synth.make_diar_weights()


@pytest.fixture(scope="module")
def model(W):
    return db.DiarModel(W)


@pytest.mark.parametrize("n,norm", [(10080, False), (24000, True), (16000 + 77, False), (4000, True)])
def test_logmel_matches_committed_golden(W, n, norm):
    g = np.load(GOLD)
    mel, tv = db.logmel(_audio(9, n), W["vad.preprocessor.featurizer.fb"], W["vad.preprocessor.featurizer.window"], norm)
    ref = g[f"mel_{n}_{int(norm)}"]
    assert mel.shape == ref.shape and tv == n // 160 and mel.shape[1] % 16 == 0
    # same operation order as the reference build; logf / FMA contraction differences only
    assert np.abs(mel - ref).max() < (2e-3 if norm else 2e-5)
    assert (mel[:, tv:] == 0).all()                      # the +1 STFT frame and the pad-to-16 tail are zeros


@pytest.mark.skipif(not ob.have_ref(), reason="compiled reference not available on this box")
@pytest.mark.parametrize("n,norm", [(10080, False), (24000, True), (1234, False)])
def test_logmel_matches_compiled_reference(W, n, norm):
    a = _audio(21, n)
    fb, win = W["spk.preprocessor.featurizer.fb"], W["spk.preprocessor.featurizer.window"]
    mel, tv = db.logmel(a, fb, win, norm)
    ref, tvr = db.ref_logmel(a, fb, win, norm)
    assert mel.shape == ref.shape and tv == tvr
    assert np.abs(mel - ref).max() < (2e-3 if norm else 2e-5)


@pytest.mark.parametrize("which,norm", [("vad_ref", False), ("spk_ref", True)])
def test_logmel_matches_nemo_fixture(which, norm):
    """The reference's own check of this stage (tests/test_diarize_preproc.cpp: max_abs < 1e-3 against the PyTorch
    fixture), with NeMo's filterbank rebuilt from its definition (librosa Slaney) instead of read from diarize.gguf:
    83 200 samples without normalisation (MarbleNet) and 24 000 with per-feature normalisation (TitaNet)."""
    g = np.load(NEMO)
    audio, ref = g[f"{which}_audio"], g[f"{which}_mel"]
    mel, tv = db.logmel(audio, synth.slaney_filterbank(80), synth.hann_window(), norm)
    assert mel.shape == ref.shape and tv == audio.size // 160
    assert np.abs(mel[:, :tv] - ref[:, :tv]).max() < 2e-4          # measured 7e-5; the reference accepts 1e-3
    assert (mel[:, tv:] == 0).all() and (ref[:, tv:] == 0).all()


def test_vad_window_properties(model):
    a = _audio(3, 10080 + 160 * 5)
    p = model.vad_batch(a)
    assert p.shape == (6,) and ((p > 0) & (p < 1)).all()
    assert abs(p[2] - model.vad_window(a[320:320 + 10080])) < 1e-7        # batch == window by window
    # masked frames do not leak: samples beyond lens_samples change nothing once their frames are masked ...
    b = a[:10080].copy()
    b[8000:] = 0.25
    full = model.vad_window(a[:10080], 10080)
    assert model.vad_window(a[:10080], 7000) != full
    # ... but the mel frames that straddle the boundary still see them (the reference masks frames, not samples)
    assert np.isfinite(model.vad_window(b, 7000))


def test_spk_embedding_properties(model):
    a = _audio(5, 24000)
    e = model.spk_embed(a)
    assert e.shape == (192,) and np.isfinite(e).all() and np.abs(e).max() > 1e-3
    assert np.abs(model.spk_embed(a) - e).max() == 0.0                    # deterministic
    e2 = model.spk_embed(_audio(6, 24000))
    cos = float(e @ e2 / (np.linalg.norm(e) * np.linalg.norm(e2)))
    assert cos < 0.9999                                                   # the input matters
    short = model.spk_embed(a, 12000)                                     # 75 valid frames
    assert np.isfinite(short).all() and np.abs(short - e).max() > 1e-4

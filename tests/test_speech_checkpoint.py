"""CPU: the 'speech' synthetic checkpoint (synth.make_weights(margins="speech")) under the F32 oracle -- the audio generator is
deterministic, the decoder has the refractory behaviour it was built for, and on HELD-OUT audio (streams the fit never saw)
the oracle's greedy transcript is the phone sequence with >= 99 % of its decisions at a top-2 margin >= 0.5 logits."""
import hashlib

import numpy as np

from nemotron_asr_amd import synth
from oracle import binding as ob


def test_speech_pcm_is_deterministic_and_well_formed():
    pcm, ev = synth.make_speech_pcm(3, 12.0)
    pcm2, ev2 = synth.make_speech_pcm(3, 12.0)
    assert np.array_equal(pcm, pcm2) and ev == ev2
    assert hashlib.sha256(pcm.tobytes()).hexdigest()[:16] == hashlib.sha256(synth.make_speech_pcm(3, 12.0)[0].tobytes()).hexdigest()[:16]
    assert 15 <= len(ev) <= 30 and np.abs(pcm).max() < 30000
    for (k0, a0, b0), (k1, a1, b1) in zip(ev, ev[1:]):
        assert k0 != k1 and a1 - b0 >= 0.08 * 16000 - 1 and 0.24 * 16000 - 1 <= b0 - a0 <= 0.48 * 16000
    assert len({synth.phone_token(k) for k in range(synth.N_PHONES)}) == synth.N_PHONES
    w, b = synth.load_speech_readout()
    assert w.shape == (synth.N_PHONES + 1, 1024) and b.shape == (synth.N_PHONES + 1,)


def test_speech_decoder_emits_each_phone_once():
    """detector k high -> token k; after emitting k its logit sits SPEECH_SUPPRESS target units lower and blank wins; the
    state is the last token only"""
    w = np.zeros((17, 1024), np.float32)
    w[np.arange(17), np.arange(17)] = 1.0
    W = synth.make_weights(1, margins="speech", readout=(w, np.zeros(17, np.float32)))
    M = ob.OracleModel(W, 1)
    toks = [synth.phone_token(k) for k in range(16)]
    A = synth.SPEECH_LOGIT_SCALE
    z = np.zeros(1280, np.float32)
    e = np.zeros(1024, np.float32)
    e[3] = 1.0
    lg, h1, c1 = M.decoder_joint(1024, z, z, e)
    assert int(np.argmax(lg)) == toks[3] and lg[toks[3]] - lg[1024] > 0.4 * A
    lg, h1, c1 = M.decoder_joint(toks[3], z, z, e)                 # same frame, token 3 just emitted
    assert int(np.argmax(lg)) == 1024 and lg[1024] - np.delete(lg, 1024).max() > 0.3 * A
    e5 = np.zeros(1024, np.float32)
    e5[5] = 1.0
    lg, h2, c2 = M.decoder_joint(toks[3], z, z, e5)                # next phone: not held back by the previous token
    assert int(np.argmax(lg)) == toks[5]
    lg, _, _ = M.decoder_joint(toks[5], h1, c1, e)                 # phone 3 again after one other phone: the state has let go of it
    assert int(np.argmax(lg)) == toks[3] and lg[toks[3]] - lg[1024] > 0.3 * A


def test_speech_checkpoint_transcribes_heldout_audio_with_wide_margins():
    W = synth.make_weights(24, margins="speech")
    om = ob.OracleModel(W, 24)
    R, n = 13, synth.shift_samples(13)
    pcm, ev = synth.make_speech_pcm(5, 22.4)                       # stream 5: held out (the fit used streams 100..147)
    ost = ob.OracleStream(om, R)
    ost.enable_decision_log()
    toks = []
    for o in range(0, pcm.size, n):
        toks += ost.process(pcm[o:o + n])
    toks += ost.finalize()
    assert toks == [synth.phone_token(k) for k, _, _ in ev] and len(toks) > 30
    m = ost.decision_log()["margin"]
    assert m.size > 300 and (m >= 0.5).mean() >= 0.99 and m.min() > 0.05, (m.size, (m >= 0.5).mean(), m.min())

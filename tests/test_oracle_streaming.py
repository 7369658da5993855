"""Pin the oracle's cache / ring / mask / chunk logic (which the scalar reference does not
have) by an independent float64 numpy restatement that processes the WHOLE sequence at once
with an explicit "70 left / chunk-block" mask, and check the driver
arithmetic of reference src/nemo-stream.h:65-100 and src/nemo-stream.cpp:1145-1293."""
import numpy as np
import pytest

from nemotron_asr_amd import synth
from oracle import binding as ob
from tests.golden import inputs as gi


def _ln(x, w, b):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + 1e-5) * w + b


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def full_sequence_layer(W, l, x, T, ks=9):
    """float64, whole sequence, explicit mask: frame n (chunk c = n // T) sees keys
    [c*T - 70, c*T + T) clipped at 0; relative position n_q - n_k."""
    g = lambda k: W[f"encoder.layers.{l}.{k}"].astype(np.float64)
    N = x.shape[0]
    x = x.astype(np.float64).copy()
    a = _ln(x, g("norm_feed_forward1.weight"), g("norm_feed_forward1.bias"))
    h = a @ g("feed_forward1.linear1.weight").T
    x += 0.5 * ((h * _sig(h)) @ g("feed_forward1.linear2.weight").T)
    a = _ln(x, g("norm_self_att.weight"), g("norm_self_att.bias"))
    q = a @ g("self_attn.linear_q.weight").T
    k = a @ g("self_attn.linear_k.weight").T
    v = a @ g("self_attn.linear_v.weight").T
    u, vb = g("self_attn.pos_bias_u"), g("self_attn.pos_bias_v")
    rel_max = 70 + T - 1
    rels = np.arange(-(T - 1), rel_max + 1)
    emb = np.stack([ob.pos_emb(int(r)) for r in rels]).astype(np.float64)
    P = emb @ g("self_attn.linear_pos.weight").T            # [n_rel][1024]
    ctx = np.zeros_like(q)
    for n in range(N):
        c = n // T
        lo, hi = max(0, c * T - 70), c * T + T
        for hd in range(8):
            sl = slice(hd * 128, hd * 128 + 128)
            keys = np.arange(lo, hi)
            s1 = (q[n, sl] + u[hd]) @ k[keys, sl].T
            pr = P[(n - keys) + (T - 1)][:, sl]
            s2 = np.einsum("d,jd->j", q[n, sl] + vb[hd], pr)
            s = (s1 + s2) / np.sqrt(128.0)
            w = np.exp(s - s.max())
            w /= w.sum()
            ctx[n, sl] = w @ v[keys, sl]
    x += ctx @ g("self_attn.linear_out.weight").T
    a = _ln(x, g("norm_conv.weight"), g("norm_conv.bias"))
    y = a @ g("conv.pointwise_conv1.weight").T
    glu = y[:, :1024] * _sig(y[:, 1024:])
    z = np.concatenate([np.zeros((ks - 1, 1024)), glu])
    dw = g("conv.depthwise_conv.weight")
    cv = sum(z[kk:kk + N] * dw[kk] for kk in range(ks))
    cv = _ln(cv, g("conv.batch_norm.weight"), g("conv.batch_norm.bias"))
    x += (cv * _sig(cv)) @ g("conv.pointwise_conv2.weight").T
    a = _ln(x, g("norm_feed_forward2.weight"), g("norm_feed_forward2.bias"))
    h = a @ g("feed_forward2.linear1.weight").T
    x += 0.5 * ((h * _sig(h)) @ g("feed_forward2.linear2.weight").T)
    return _ln(x, g("norm_out.weight"), g("norm_out.bias"))


@pytest.mark.parametrize("R,n_chunks", [(0, 75), (1, 40), (6, 12)])
def test_cached_streaming_equals_full_sequence_mask(weights2, R, n_chunks):
    """Runs past 70 cached frames so the window slides and cache_valid_len saturates."""
    T = 1 + R
    model = ob.OracleModel(weights2, 2)
    st = ob.OracleStream(model, R)
    sub_tap, lay_tap = st.enable_taps()
    rng = np.random.default_rng(5)
    n_mel = 9 + 8 * T * n_chunks
    mel = (rng.standard_normal((n_mel, 128)) * 2 - 4).astype(np.float32)
    subs, outs = [], []
    for c in range(n_chunks):
        chunk = mel[c * 8 * T: c * 8 * T + st.chunk_mel]
        out = st.encode_chunk(chunk)
        subs.append(sub_tap.copy())
        outs.append(out.copy())
        assert st.cache_valid_len == min(70, (c + 1) * T)
    x = np.concatenate(subs)                                   # post-subsampling frames
    y = x
    for l in range(2):
        y = full_sequence_layer(weights2, l, y, T)
    got = np.concatenate(outs)
    assert np.abs(got - y).max() < 2e-3                        # reference layer threshold 2e-3


def test_chunk_arithmetic():
    # reference docs/GGML_NEMO_MAPPING.md:133-140: 17/25/65/121
    assert [synth.chunk_mel_frames(r) for r in (0, 1, 6, 13)] == [17, 25, 65, 121]
    assert [synth.shift_samples(r) for r in (0, 1, 6, 13)] == [1280, 2560, 8960, 17920]


def test_driver_counts_10s(weights1):
    """10 s at R=0 -> 124 steps, no tail; R=13 -> 8 steps + tail of 12."""
    model = ob.OracleModel(weights1, 1)
    pcm = synth.make_pcm(3, 10.0)
    for R, steps, tail in ((0, 124, 0), (13, 8, 1)):
        st = ob.OracleStream(model, R)
        toks = []
        piece = synth.shift_samples(R)
        for o in range(0, pcm.size, piece):
            toks += st.process(pcm[o:o + piece])
        assert st.total_chunks == steps
        toks += st.finalize()
        assert st.total_chunks == steps + tail
        assert all(0 <= t < 1024 for t in toks)


def test_push_size_independence(weights1):
    """Tokens do not depend on how PCM is split (reference CLI reads chunk_samples, :144)."""
    model = ob.OracleModel(weights1, 1)
    pcm = synth.make_pcm(1, 3.0)
    res = []
    for piece in (1280, 2720, 5000, 48000):
        st = ob.OracleStream(model, 0)
        toks = []
        for o in range(0, pcm.size, piece):
            toks += st.process(pcm[o:o + piece])
        toks += st.finalize()
        res.append(toks)
    assert res[0] == res[1] == res[2] == res[3]


def test_decode_state_commit_only_on_emit(weights1):
    model = ob.OracleModel(weights1, 1)
    st = ob.OracleStream(model, 0)
    h0, c0, p0 = st.decoder_state()
    assert p0 == 1024 and not h0.any() and not c0.any()
    enc = gi.enc_frames(64)
    n = 0
    for t in range(64):
        hb, cb, pb = st.decoder_state()
        toks = st.decode(enc[t:t + 1])
        ha, ca, pa = st.decoder_state()
        assert len(toks) <= 10
        if not toks:
            assert np.array_equal(hb, ha) and np.array_equal(cb, ca) and pb == pa
        else:
            assert pa == toks[-1]
        n += len(toks)
    assert n > 0


def test_bf16_emulation_close_to_f32(weights1):
    m32 = ob.OracleModel(weights1, 1)
    m16 = ob.OracleModel(weights1, 1, emulate_bf16=True)
    x = gi.layer_input(14)
    a, b = m32.layer_chunk0(0, x), m16.layer_chunk0(0, x)
    d = np.abs(a - b).max()
    assert 1e-5 < d < 5e-2

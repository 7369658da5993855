"""Round-5 GPU parity tests, all through the C ABI against the CPU oracle:

* the greedy loop's 10-symbol cap (src/nemo-stream.cpp:849, :865-927) forced on every frame and on some frames, f32 and bf16, batch 1
  and 64: tokens, emission frames and decode-iteration counts;
* 512 streams x R = 13 in ONE step (3 584 ... 7 168 rows: the 224 / 256-row GEMM tiles, the banded tile order, three-piece pipelined
  steps) against the oracle instead of against the engine's other variants (two layers synchronous, six layers pipelined in three pieces);
* BASELINE configs[3]'s own flavour: F32 tensors rounded to bf16 at upload, 64 streams x R = 13, 24 layers;
* a stream that starts on a used slot: the K/V rings are no longer cleared (one reset launch instead of 53 fills) -- stale rows, and
  rows deliberately filled with large values, must never reach a result.
"""
import numpy as np
import pytest

from nemotron_asr_amd import capi, synth
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _acc(worst, diff):
    """running maximum of |diff| that does NOT swallow NaN (Python's max(0.0, nan) is 0.0: round 5 found three parity tests blind to an all-NaN engine)"""
    m = float(np.abs(diff).max())
    assert np.isfinite(m), "non-finite values in the engine's output"
    return max(worst, m)

EPS_MARGIN = 0.05          # tests/test_gpu_configs.py: the logit noise a bf16 first divergence may hide in
BLANK = 1024


@pytest.fixture(scope="module")
def W2():
    return synth.make_weights(n_layers=2)


def _with_blank_bias(W, delta):
    w = dict(W)
    b = np.array(W["joint.joint_net.2.bias"], np.float32, copy=True)
    b[BLANK] += delta
    w["joint.joint_net.2.bias"] = b
    return w


def _oracle_run(om, R, pcm, n):
    ost = ob.OracleStream(om, R)
    ost.enable_decision_log()
    ref = []
    for o in range(0, pcm.size, n):
        ref += ost.process(pcm[o:o + n])
    ref += ost.finalize()
    return ost, ref


def _frames_hist(frames):
    """symbols emitted per encoder frame (only frames that emitted)"""
    return np.bincount(np.asarray(frames, np.int64)) if len(frames) else np.zeros(0, np.int64)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("R,B", [(0, 1), (13, 1), (13, 64)])
def test_ten_symbol_cap_forced_on_every_frame(W2, dtype, R, B):
    """blank bias -1e9: blank never wins, so EVERY frame runs into max_symbols_per_step = 10 (src/nemo-stream.cpp:849: the loop ends
    by its bound, not by a blank) and the decoder state advances ten times per frame.  Per stream: 10 tokens at every decoded
    frame, iterations = 10 x frames, tokens / frames / iterations == the oracle (f32: exact; bf16: frames and iterations exact -- the
    cap does not depend on which non-blank token wins -- and any token difference is a near-tie of the oracle).  At 64 x R = 13 a step
    needs 141 decode iterations: far beyond the 12 a step graph carries, so the eager continuation rounds run as well."""
    L = 2
    W = _with_blank_bias(W2, -1e9)
    n = synth.shift_samples(R)
    n_push = 4 if R else 12
    pcms = [synth.make_pcm(900 + b, n_push * n / 16000 + 0.3) for b in range(B)]          # + 0.3 s: a tail for finalize
    eng = capi.Engine(W, n_layers=L, dtype=capi.DTYPE_F32 if dtype == "f32" else capi.DTYPE_BF16, max_streams=B)
    sts = [eng.stream(R) for _ in range(B)]
    toks = [[] for _ in range(B)]
    for o in range(0, pcms[0].size, n):
        for b, t in enumerate(eng.step(sts, [p[o:o + n] for p in pcms])):
            toks[b] += t
    for b, t in enumerate(eng.finalize(sts)):
        toks[b] += t
    om = ob.OracleModel(W, L, emulate_bf16=dtype == "bf16")
    for b in sorted({0, B // 2, B - 1}):
        ost, ref = _oracle_run(om, R, pcms[b], n)
        rf, gf = ost.token_frames(), sts[b].token_frames()
        assert len(ref) >= 10 * 8 and len(ref) % 10 == 0
        assert set(_frames_hist(rf)[_frames_hist(rf) > 0].tolist()) == {10}          # the oracle itself: ten symbols at every frame
        assert gf == rf, b
        assert sts[b].stats().decode_iterations == ost.decode_iterations == len(ref)
        assert BLANK not in toks[b]
        if dtype == "f32":
            assert toks[b] == ref, b
        else:
            div = ob.first_divergence(ost.decision_log(), ref, rf, toks[b], gf)
            assert div is None or (div["decision"] >= 0 and div["margin"] < EPS_MARGIN), (b, div)
    eng.close()


@pytest.mark.parametrize("R,B", [(13, 8), (0, 3)])
def test_ten_symbol_cap_on_some_frames_f32(W2, R, B):
    """A milder blank bias: the oracle's own decisions show frames that stop at a blank after 0 .. 9 symbols AND frames that are cut
    off by the cap (10 symbols, no blank evaluated: the iteration count tells the two apart).  The f32 engine reproduces tokens,
    frames, iteration counts and the committed decoder state."""
    L = 2
    n = synth.shift_samples(R)
    pcms = [synth.make_pcm(950 + b, (5 if R else 30) * n / 16000 + 0.2) for b in range(B)]
    chosen = None
    # the near-tie checkpoint's logits are ~1e-3 apart: -0.2 on the blank is already "some frames" (-0.5 caps every frame); pick the
    # first bias under which stream 0's ORACLE decisions show both kinds of frame
    for delta in (-0.2, -0.15, -0.3, -0.1, -0.4):
        W = _with_blank_bias(W2, delta)
        om = ob.OracleModel(W, L)
        ost, ref = _oracle_run(om, R, pcms[0], n)
        n_frames = len(set(ost.decision_log()["frame"].tolist()))          # frames the loop visited
        h = np.bincount(np.asarray(ost.token_frames(), np.int64), minlength=n_frames)
        if (h == 10).sum() >= 2 and (h < 10).sum() >= 2:
            chosen = (delta, W, om)
            break
    assert chosen, "no blank bias gives both capped and uncapped frames"
    delta, W, om = chosen
    eng = capi.Engine(W, n_layers=L, dtype=capi.DTYPE_F32, max_streams=B)
    sts = [eng.stream(R) for _ in range(B)]
    toks = [[] for _ in range(B)]
    for o in range(0, pcms[0].size, n):
        for b, t in enumerate(eng.step(sts, [p[o:o + n] for p in pcms])):
            toks[b] += t
    for b, t in enumerate(eng.finalize(sts)):
        toks[b] += t
    capped = 0
    for b in range(B):
        ost, ref = _oracle_run(om, R, pcms[b], n)
        assert toks[b] == ref and sts[b].token_frames() == ost.token_frames(), (b, delta)
        assert sts[b].stats().decode_iterations == ost.decode_iterations
        hh, cc, pp = ost.decoder_state()
        ds = sts[b].tap(capi.TAP_DEC_STATE)
        assert int(ds[-1]) == pp and np.abs(ds[:1280] - hh).max() < 1e-4 and np.abs(ds[1280:2560] - cc).max() < 1e-4
        capped += int((_frames_hist(ost.token_frames()) == 10).sum())
    assert capped >= 2
    eng.close()


# ---- 512 streams in one step against the oracle ----------------------------------------------------------------------------------
def _spot_state_vs_oracle(st, ost, L, T):
    worst = 0.0
    for l in range(L):
        for which, tap in ((0, capi.TAP_K_CACHE), (1, capi.TAP_V_CACHE)):
            worst = _acc(worst, st.tap(tap, l, cap=70 * 1024).reshape(70, 1024) - ost.get_cache(which, l))
        worst = _acc(worst, st.tap(capi.TAP_CONV_CACHE, l, cap=8 * 1024).reshape(8, 1024) - ost.get_cache(2, l))
    return worst


@pytest.mark.parametrize("pipeline,L", [(0, 2), (4, 6)])
def test_512_streams_R13_one_step_vs_oracle(W2, pipeline, L):
    """north_star's third batch size on one GPU: 512 streams x R = 13 = 7 168 rows per GEMM -- k_gemm_wide<256, 7 / 8> for W1 / pw1 / QKV (and,
    pipelined, for every GEMM of the layer), the banded tile order, and with pipeline = 4 the three-piece step.  Round 4 checked these
    against each other only (one digest for every variant); here four spot streams' encoder output (= last layer + norm_out) of every step
    and their K / V / conv caches are compared with the bf16-emulating oracle, on the SHIPPED path (graph replay, no debug mode)."""
    B, R, T = 512, 13, 14
    W = W2 if L == 2 else synth.make_weights(n_layers=L)      # six layers: three real pieces (1 + 3 + 2 layers)
    eng = capi.Engine(W, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
    eng.set_option("pipeline", pipeline)
    om = ob.OracleModel(W, L, emulate_bf16=True)
    om32 = ob.OracleModel(W, L)                # the PINNED oracle: F32, no emulation
    n = synth.shift_samples(R)
    n_push = 4
    pcms = [synth.make_pcm(1200 + (b % 97) * 5 + b // 97, n_push * n / 16000 + 0.01)[:n_push * n] for b in range(B)]
    spots = (0, 171, 340, 511)
    sts = [eng.stream(R) for _ in range(B)]
    osts = {b: ob.OracleStream(om, R) for b in spots}
    taps = {b: osts[b].enable_taps() for b in spots}
    osts32 = {b: ob.OracleStream(om32, R) for b in spots[::3]}
    taps32 = {b: osts32[b].enable_taps() for b in osts32}
    worst = worst32 = mean32 = 0.0
    for k in range(n_push):
        eng.step(sts, [p[k * n:(k + 1) * n] for p in pcms])
        if pipeline:
            eng.synchronize()                  # completes the step in flight; its workspace set holds the encoder output
        for b in spots:
            c0 = osts[b].total_chunks
            osts[b].process(pcms[b][k * n:(k + 1) * n])
            if b in osts32:
                osts32[b].process(pcms[b][k * n:(k + 1) * n])
            if osts[b].total_chunks > c0:
                got = sts[b].tap(capi.TAP_ENCODER_OUT).reshape(-1, 1024)[:T]
                worst = _acc(worst, got - taps[b][1][L - 1])      # last layer incl. norm_out = the encoder output
                if b in osts32:
                    worst32 = _acc(worst32, got - taps32[b][1][L - 1])
                    mean32 = max(mean32, float(np.abs(got - taps32[b][1][L - 1]).mean()))
    assert all(osts[b].total_chunks == n_push - 1 for b in spots)
    assert worst < (3e-2 if L == 2 else 5e-2), worst      # the 2-layer bar of tests/test_gpu_parity.py; six layers: between it and the 24-layer bar (0.1)
    # THE stated tolerance of the bf16 path against the pinned F32 oracle: 2 layers max < 2.5e-2 / mean < 5e-3
    # (measured 2.0e-2 / 3.8e-3); six layers: sqrt(3) x that, max < 4.5e-2 / mean < 9e-3 (the error is a random walk over the layers' roundings)
    assert worst32 < (2.5e-2 if L == 2 else 4.5e-2) and mean32 < (5e-3 if L == 2 else 9e-3), (worst32, mean32)
    kv = max(_spot_state_vs_oracle(sts[b], osts[b], L, T) for b in spots)
    assert kv < 1.2e-1, kv                     # K rows are not LayerNorm-scaled (one bf16 ulp at |k| ~ 4 = 0.03)
    if pipeline:
        assert eng.counter("pipelined_steps") >= n_push - 1
    eng.close()


def test_512_streams_f32_engine_tokens_equal_oracle(W2):
    """... and the tokens: the f32 engine (f32 MFMA GEMMs at 7 168 rows) on 512 streams x R = 13, spot streams token-, frame- and
    iteration-exact against the oracle."""
    L, B, R = 2, 512, 13
    eng = capi.Engine(W2, n_layers=L, dtype=capi.DTYPE_F32, max_streams=B)
    n = synth.shift_samples(R)
    n_push = 4
    pcms = [synth.make_pcm(1300 + (b % 89) * 3 + b // 89, n_push * n / 16000 + 0.2) for b in range(B)]
    sts = [eng.stream(R) for _ in range(B)]
    toks = [[] for _ in range(B)]
    for o in range(0, pcms[0].size, n):
        for b, t in enumerate(eng.step(sts, [p[o:o + n] for p in pcms])):
            toks[b] += t
    for b, t in enumerate(eng.finalize(sts)):
        toks[b] += t
    om = ob.OracleModel(W2, L)
    total = 0
    for b in (0, 100, 255, 256, 400, 511):
        ost, ref = _oracle_run(om, R, pcms[b], n)
        assert toks[b] == ref and sts[b].token_frames() == ost.token_frames(), b
        assert sts[b].stats().decode_iterations == ost.decode_iterations
        total += len(ref)
    assert total > 10
    eng.close()


# ---- configs[3]'s own flavour at full size ------------------------------------------------------------------------------------------
def test_config4_f32_tensors_to_bf16_engine_64_streams_R13_24_layers():
    """BASELINE configs[3] names bf16 from the F32 checkpoint (64 streams x R = 13 per GPU, x 8 GPUs): round 4 ran that flavour at 24 layers
    only fed from Q8_0-dequantised tensors and on the f32 engine.  Here: the f32 tensors rounded to bf16 at upload, 24 layers, one
    64-stream step on the shipped (graph) path, four spot streams' encoder output against the 24-layer bf16-emulating oracle (bar of
    test_config3_full_size_one_step_and_q8_semantics: max < 0.1, mean < 1.5e-2) and against the pinned F32 oracle (asserted: max < 8e-2, mean < 1.6e-2)."""
    L, B, R, T = 24, 64, 13, 14
    W = synth.make_weights(n_layers=L)
    eng = capi.Engine(W, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
    n = synth.shift_samples(R)
    pcms = [synth.make_pcm(1500 + b, 2 * n / 16000 + 0.01)[:2 * n] for b in range(B)]
    sts = [eng.stream(R) for _ in range(B)]
    for k in range(2):
        eng.step(sts, [p[k * n:(k + 1) * n] for p in pcms])
    assert all(s.progress().chunks == 1 for s in sts)
    spots = (0, 21, 42, 63)
    got = {b: sts[b].tap(capi.TAP_ENCODER_OUT).reshape(-1, 1024)[:T].copy() for b in spots}
    eng.close()
    om = ob.OracleModel(W, L, emulate_bf16=True)
    d_max = d_mean = 0.0
    for b in spots:
        ost = ob.OracleStream(om, R)
        tap = ost.enable_taps()
        ost.process(pcms[b])
        assert ost.total_chunks == 1
        d_max = _acc(d_max, got[b] - tap[1][L - 1])
        d_mean = max(d_mean, float(np.abs(got[b] - tap[1][L - 1]).mean()))
    assert np.isfinite(d_max) and d_max < 1e-1 and d_mean < 1.5e-2, (d_max, d_mean)
    del om
    # ... and against the PINNED oracle (F32, no emulation): THE stated tolerance of the bf16 path at 24 layers on the near-tie checkpoint,
    # max < 8e-2 / mean < 1.6e-2 (measured 5.7e-2 / 1.2e-2; INTEGRATION.md "Tolerances")
    om = ob.OracleModel(W, L)
    f_max = f_mean = 0.0
    for b in spots[:2]:
        ost = ob.OracleStream(om, R)
        tap = ost.enable_taps()
        ost.process(pcms[b])
        f_max = _acc(f_max, got[b] - tap[1][L - 1])
        f_mean = max(f_mean, float(np.abs(got[b] - tap[1][L - 1]).mean()))
    assert f_max < 8e-2 and f_mean < 1.6e-2, (f_max, f_mean)


# ---- a stream that starts on a used slot --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("R", [0, 13])
def test_stale_kv_rows_never_reach_a_result(W2, dtype, R):
    """Stream start / reset is ONE launch and leaves the K/V rings alone (rounds 1-4: 53 fills, 32 MB per stream): rows behind
    cache_valid_len get -1e9 and weigh exactly 0 (src/nemo-stream.cpp:1037-1043; the reference's own reset relies on it, :95-115).
    Engine A: fresh.  Engine B: slot 0 is first used by another stream (other audio, other lookahead), then its rings are overwritten
    with +-1000 (nasr_stream_debug_fill_kv), then the stream under test starts on it; and once more through reset().  Tokens, frames,
    the encoder output of every step and the logical K / V / conv caches are BIT-identical to engine A's."""
    L = 2
    dt = capi.DTYPE_F32 if dtype == "f32" else capi.DTYPE_BF16
    n = synth.shift_samples(R)
    pcm = synth.make_pcm(77, 8 * n / 16000 + 0.2)

    def run(eng, st):
        toks, encs = [], []
        for o in range(0, pcm.size, n):
            c0 = st.progress().chunks
            toks += eng.step([st], [pcm[o:o + n]])[0]
            if st.progress().chunks > c0:
                encs.append(st.tap(capi.TAP_ENCODER_OUT).reshape(-1, 1024)[:1 + R].copy())
        state = [st.tap(tap, l, cap=70 * 1024).copy() for l in range(L) for tap in (capi.TAP_K_CACHE, capi.TAP_V_CACHE)]
        state += [st.tap(capi.TAP_CONV_CACHE, l, cap=8 * 1024).copy() for l in range(L)]
        frames = st.token_frames()
        toks += eng.finalize([st])[0]
        return toks, frames, np.stack(encs), state

    engA = capi.Engine(W2, n_layers=L, dtype=dt, max_streams=2)
    sa = engA.stream(R)
    ref = run(engA, sa)
    engA.close()
    assert len(ref[0]) > 3 and np.isfinite(ref[2]).all()

    engB = capi.Engine(W2, n_layers=L, dtype=dt, max_streams=2)
    other = engB.stream(13 - R if R in (0, 13) else 0)
    engB.step([other], [synth.make_pcm(5, 4.0)])           # fills slot 0's rings and conv caches with another stream's rows
    other.destroy()
    sb = engB.stream(R)                                     # same slot
    got = run(engB, sb)
    sb.reset()
    sb.debug_fill_kv(1000.0)
    got2 = run(engB, sb)
    sb.destroy()
    sc = engB.stream(R)
    sc.debug_fill_kv(-3.0e4)
    got3 = run(engB, sc)
    engB.close()
    for g in (got, got2, got3):
        assert g[0] == ref[0] and g[1] == ref[1]
        assert np.array_equal(g[2], ref[2]), float(np.abs(g[2] - ref[2]).max())
        for a, b in zip(g[3], ref[3]):
            assert np.array_equal(a, b)


def test_256_streams_R6_vs_oracle(W2):
    """256 streams x R = 6 (T = 7, 1 792 rows): the shapes round 5's small kernels take besides 512 x R = 13 -- k_dwconv_stream<7> with ONE workgroup per
    stream (it also writes the cache), k_sub_dw_row at another image height, split-K partial slabs + k_post at a size where the residual fold does not apply
    (two K slices of 128 x 128 tiles).  Spot streams' encoder output of every step and their K / V / conv caches against the bf16-emulating oracle, pipelined."""
    L, B, R, T = 2, 256, 6, 7
    eng = capi.Engine(W2, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
    eng.set_option("pipeline", 4)
    om = ob.OracleModel(W2, L, emulate_bf16=True)
    n = synth.shift_samples(R)
    n_push = 5
    pcms = [synth.make_pcm(1700 + (b % 61) * 7 + b // 61, n_push * n / 16000 + 0.01)[:n_push * n] for b in range(B)]
    spots = (0, 85, 170, 255)
    sts = [eng.stream(R) for _ in range(B)]
    osts = {b: ob.OracleStream(om, R) for b in spots}
    taps = {b: osts[b].enable_taps() for b in spots}
    worst = 0.0
    for k in range(n_push):
        eng.step(sts, [p[k * n:(k + 1) * n] for p in pcms])
        for b in spots:
            c0 = osts[b].total_chunks
            osts[b].process(pcms[b][k * n:(k + 1) * n])
            if osts[b].total_chunks > c0:
                got = sts[b].tap(capi.TAP_ENCODER_OUT).reshape(-1, 1024)[:T]
                worst = _acc(worst, got - taps[b][1][L - 1])
    assert all(osts[b].total_chunks == n_push - 1 for b in spots)
    assert worst < 3e-2, worst
    kv = max(_spot_state_vs_oracle(sts[b], osts[b], L, T) for b in spots)
    assert kv < 1.2e-1, kv
    eng.close()


def test_256_streams_R13_vs_oracle(W2):
    """256 streams x R = 13 (3 584 rows = 224 tiles of 128 x 128): round 5's split rule hands the residual GEMMs ONE K slice from 200 tiles, so their
    products are added to the residual stream in the GEMM epilogue (no partial slabs) on the 128 x 128 kernels -- a shape between the welded 128 x 64 form
    (64 streams) and the 224 x 256 tiles (512 streams).  Spot streams against the bf16-emulating oracle, pipelined; tokens of the f32 engine are covered at 512."""
    L, B, R, T = 2, 256, 13, 14
    eng = capi.Engine(W2, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
    eng.set_option("pipeline", 4)
    om = ob.OracleModel(W2, L, emulate_bf16=True)
    n = synth.shift_samples(R)
    n_push = 4
    pcms = [synth.make_pcm(1900 + (b % 53) * 11 + b // 53, n_push * n / 16000 + 0.01)[:n_push * n] for b in range(B)]
    spots = (0, 99, 200, 255)
    sts = [eng.stream(R) for _ in range(B)]
    osts = {b: ob.OracleStream(om, R) for b in spots}
    taps = {b: osts[b].enable_taps() for b in spots}
    worst = 0.0
    for k in range(n_push):
        eng.step(sts, [p[k * n:(k + 1) * n] for p in pcms])
        for b in spots:
            c0 = osts[b].total_chunks
            osts[b].process(pcms[b][k * n:(k + 1) * n])
            if osts[b].total_chunks > c0:
                got = sts[b].tap(capi.TAP_ENCODER_OUT).reshape(-1, 1024)[:T]
                worst = _acc(worst, got - taps[b][1][L - 1])
    assert worst < 3e-2, worst
    kv = max(_spot_state_vs_oracle(sts[b], osts[b], L, T) for b in spots)
    assert kv < 1.2e-1, kv
    eng.close()


def test_ragged_rows_on_the_wide_tiles_of_pipelined_steps():
    """Round 5 lets pipelined steps take the 224 x 256 GEMM tiles from 32 tiles (was 96): at 100 / 130 streams x R = 13 (1 400 / 1 820 rows: ragged last row tile) W1, QKV (K / V ring
    rows out of the epilogue) and pointwise_conv1 (GLU pairs) now run on k_gemm_wide2.  Same digest (tokens, encoder output, K and conv caches of every stream) as with 128 x 128 tiles
    everywhere ("wide_tiles" = 0) and as rounds 1-4's loops ("gemm_prio" = 20): tests/micro/ragged_wide_check.py."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tests" / "micro" / "ragged_wide_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stdout.count("==") == 3 and "!=" not in r.stdout

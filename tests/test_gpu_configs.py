"""GPU parity at the workloads BASELINE.json names (configs[2] / configs[3]: 64 streams x 1.12 s lookahead, Q8_0 tensors ->
bf16 engine, M = 896 rows per GEMM) and the token-agreement analysis of the reduced-precision engine against the f32
oracle (= the reference's arithmetic), plus the reference-exact reset and the never-drop token hand-over.

Tolerances: bf16 engine vs the bf16-emulating oracle (same rounding points, other summation order) 3e-2 on LayerNorm-scale
activations for 2 layers; 24 layers: stated in the test.  Token agreement: greedy RNN-T diverges at the first near-tie a
rounding flips, so the check is on WHERE it diverges: the oracle's top-2 logit margin at the first differing decision of
every stream must be below EPS_MARGIN (the logit noise the bf16 path is allowed), not on an agreement rate."""
import ctypes as C
import difflib
import json
import os
from pathlib import Path

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

# logit noise budget of the bf16 path at 24 layers: encoder rows differ from f32 by ~1e-2 (max 5-7e-2) on |x| <= 4 and the
# (f32) joint maps that to ~1e-2 on the logits.  Measured first-divergence margins (round 2, 12 streams): 0.001-0.016.
# A first divergence at an oracle margin above EPS_MARGIN = 3x the largest of those would be a bug, not a rounding flip.
EPS_MARGIN = 0.05


def _report(name, payload):
    d = os.environ.get("NASR_REPORT_DIR")
    if d:
        Path(d).mkdir(parents=True, exist_ok=True)
        (Path(d) / f"{name}.json").write_text(json.dumps(payload, indent=1))


@pytest.fixture(scope="module")
def W2():
    return synth.make_weights(n_layers=2)


@pytest.fixture(scope="module")
def W24():
    return synth.make_weights(n_layers=24)


@pytest.fixture(scope="module")
def Q24(W24):
    """(engine tensors with Q8_0 blocks, dequantised f32 values, raw blocks by name)"""
    engW, deq = synth.quantize_weights(W24, "q8_0")
    return engW, deq, {k: v[1] for k, v in engW.items() if isinstance(v, tuple)}


def _pieces(R, n_push, seed0, B):
    n = synth.shift_samples(R)
    return n, [synth.make_pcm(seed0 + b, n_push * n / 16000 + 0.01)[:n_push * n] for b in range(B)]


def test_config3_q8_0_64_streams_R13_two_layers(W2):
    """BASELINE configs[2] at its exact launch shape: 64 streams x R = 13 (M = 896 rows: 7 m-chunks of the 128 x 128 tile
    kernel, XCD remap, split-K + reduction), Q8_0 tensors dequantised and packed to bf16 at upload, PCM in.  EVERY stream's
    last-layer output of every step and its K / V / conv caches against the bf16-emulating oracle on the dequantised values."""
    L, B, R, T = 2, 64, 13, 14
    engW, deqW = synth.quantize_weights(W2, "q8_0")
    om = ob.OracleModel(deqW, L, emulate_bf16=True)
    eng = capi.Engine(engW, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
    eng.set_debug(True)
    n, pcms = _pieces(R, 6, 400, B)
    sts = [eng.stream(R) for _ in range(B)]
    osts = [ob.OracleStream(om, R) for _ in range(B)]
    taps = [o.enable_taps() for o in osts]
    for o in osts:
        o.enable_decision_log()
    worst, steps, toks_g, toks_o = 0.0, 0, [[] for _ in range(B)], [[] for _ in range(B)]
    for k in range(6):
        out = eng.step(sts, [p[k * n:(k + 1) * n] for p in pcms])
        stepped = False
        for b in range(B):
            toks_g[b] += out[b]
            c0 = osts[b].total_chunks
            toks_o[b] += osts[b].process(pcms[b][k * n:(k + 1) * n])
            assert sts[b].progress().chunks == osts[b].total_chunks
            if osts[b].total_chunks > c0:
                stepped = True
                got = sts[b].tap(capi.TAP_LAYER_OUT, L - 1).reshape(T, 1024)
                worst = _acc(worst, got - taps[b][1][L - 1])
        steps += stepped
    assert steps >= 4 and worst < 3e-2, (steps, worst)
    kv_worst = 0.0
    for b in range(B):
        for l in range(L):
            for which, tap in ((0, capi.TAP_K_CACHE), (1, capi.TAP_V_CACHE)):
                kv_worst = _acc(kv_worst, sts[b].tap(tap, l, cap=70 * 1024).reshape(70, 1024) - osts[b].get_cache(which, l))
            kv_worst = _acc(kv_worst, sts[b].tap(capi.TAP_CONV_CACHE, l, cap=8 * 1024).reshape(8, 1024) - osts[b].get_cache(2, l))
    assert kv_worst < 1.2e-1, kv_worst          # K rows are not LayerNorm-scaled (|k| up to ~4: one bf16 ulp = 0.03)
    # tokens: where a stream leaves the oracle's greedy path, the oracle's top-2 margin there is within the rounding noise
    # (token-for-token exactness is asserted on the f32 engine; 64 streams x 5 steps always contain a few near-ties)
    n_tok, n_same = 0, 0
    for b in range(B):
        div = ob.first_divergence(osts[b].decision_log(), toks_o[b], osts[b].token_frames(), toks_g[b], sts[b].token_frames())
        assert div is None or (div["decision"] >= 0 and div["margin"] < EPS_MARGIN), (b, div)
        n_same += div is None
        n_tok += len(toks_o[b])
    assert n_tok > 20 and n_same >= B // 2, (n_tok, n_same)
    eng.close()


def test_config3_full_size_one_step_and_q8_semantics(Q24):
    """The same launch shape at BASELINE's model size (24 layers): one 64-stream x R = 13 step from Q8_0 tensors, four spot
    streams against the 24-layer bf16-emulating oracle.  bf16 re-rounding over 24 layers is a random walk (~100 roundings
    of 0.004-0.008 on |x| <= 4): typical difference 0.01-0.02, worst element below 0.1 (stated, measured 0.05-0.07).
    Then the question config 3 asks -- how close is this to the ggml CPU path ON Q8_0? -- as numbers: the oracle in
    ORC_EMU_Q8_ACT mode (activation rows quantised per 32, int8 dot products: ggml's published Q8_0 mul_mat) against the
    f32 product of the same dequantised weights, and the engine against both."""
    engW, deq, blocks = Q24
    L, B, R, T = 24, 64, 13, 14
    eng = capi.Engine(engW, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
    eng.set_debug(True)
    n, pcms = _pieces(R, 2, 500, B)
    sts = [eng.stream(R) for _ in range(B)]
    for k in range(2):
        eng.step(sts, [p[k * n:(k + 1) * n] for p in pcms])
    assert all(s.progress().chunks == 1 for s in sts)
    spots = (0, 21, 42, 63)
    got = {b: sts[b].tap(capi.TAP_LAYER_OUT, L - 1).reshape(T, 1024).copy() for b in spots}
    eng.close()
    res = {}
    for mode, kw in (("bf16", dict(emulate_bf16=True)), ("f32", {}), ("q8_act", dict(emulate_q8_act=True, q8_blocks=blocks))):
        om = ob.OracleModel(deq, L, **kw)
        outs = {}
        for b in spots if mode == "bf16" else spots[:2]:
            ost = ob.OracleStream(om, R)
            tap = ost.enable_taps()
            ost.process(pcms[b])
            assert ost.total_chunks == 1
            outs[b] = tap[1][L - 1].copy()
        res[mode] = outs
        del om
    assert all(np.isfinite(got[b]).all() for b in spots)
    d_bf16 = max(float(np.abs(got[b] - res["bf16"][b]).max()) for b in spots)
    m_bf16 = max(float(np.abs(got[b] - res["bf16"][b]).mean()) for b in spots)
    assert d_bf16 < 1e-1 and m_bf16 < 1.5e-2, (d_bf16, m_bf16)
    def dist(a, b):
        return dict(max=max(float(np.abs(a[k] - b[k]).max()) for k in spots[:2]), mean=max(float(np.abs(a[k] - b[k]).mean()) for k in spots[:2]))

    rep = dict(engine_vs_bf16_oracle=dict(max=d_bf16, mean=m_bf16), q8_act_oracle_vs_f32_oracle=dist(res["f32"], res["q8_act"]),
               engine_vs_q8_act_oracle=dist(got, res["q8_act"]), engine_vs_f32_oracle=dist(got, res["f32"]))
    _report("config3_full_size", rep)
    # the engine (bf16 operands) must be no further from ggml's Q8_0 semantics than 3x what those semantics are from f32
    assert rep["engine_vs_q8_act_oracle"]["mean"] < 3 * max(rep["q8_act_oracle_vs_f32_oracle"]["mean"], 5e-3), rep
    # THE tolerance of the bf16 path, against the PINNED F32 oracle -- not only against the oracle's own bf16 emulation:
    # 24 layers, near-tie (random) checkpoint: max < 8e-2, mean < 1.6e-2 of |x| <= 4 (measured 5.7e-2 / 1.2e-2; the reference's f32 ladder is 5e-3, tests/test_compute.cpp:2351)
    assert rep["engine_vs_f32_oracle"]["max"] < 8e-2 and rep["engine_vs_f32_oracle"]["mean"] < 1.6e-2, rep["engine_vs_f32_oracle"]


def _run_engine_tokens(eng, R, pcms, n):
    B = len(pcms)
    sts = [eng.stream(R) for _ in range(B)]
    toks = [[] for _ in range(B)]
    for o in range(0, pcms[0].size, n):
        for b, t in enumerate(eng.step(sts, [p[o:o + n] for p in pcms])):
            toks[b] += t
    for b, t in enumerate(eng.finalize(sts)):
        toks[b] += t
    frames = [s.token_frames() for s in sts]
    for s in sts:
        s.destroy()
    return toks, frames


def _agreement(om, R, pcms, n, toks, frames):
    rows = []
    for b, pcm in enumerate(pcms):
        ost = ob.OracleStream(om, R)
        ost.enable_decision_log()
        ref = []
        for o in range(0, pcm.size, n):
            ref += ost.process(pcm[o:o + n])
        ref += ost.finalize()
        rf = ost.token_frames()
        div = ob.first_divergence(ost.decision_log(), ref, rf, toks[b], frames[b])
        prefix = div["index"] if div else len(ref)
        rows.append(dict(stream=b, ref_tokens=len(ref), engine_tokens=len(toks[b]), common_prefix=prefix,
                         aligned_ratio=round(difflib.SequenceMatcher(None, ref, toks[b], autojunk=False).ratio(), 4),
                         first_divergence=div))
    return rows


@pytest.mark.parametrize("R,B,n_push,pipeline", [(13, 8, 27, 0), (0, 1, 375, 4)])
def test_bf16_token_agreement_vs_f32_oracle(W24, R, B, n_push, pipeline):
    """24 layers on the near-tie (random) checkpoint against the F32 oracle: 8 streams x 30 s at R = 13 (graph path, M = 112 rows) and
    -- round 4 -- configs[1]'s own shape, ONE stream x R = 0 x 30 s on four lanes (fused one-row path).
    Reported: tokens, common prefix, aligned agreement; asserted: every stream's first divergence happens at a decision
    whose top-2 margin in the oracle is below EPS_MARGIN, and (R = 13) the aligned agreement over all streams is above 0.5."""
    L = 24
    n, pcms = _pieces(R, n_push, 700, B)                 # 27 x 1.12 s = 30.2 s / 375 x 80 ms = 30 s per stream
    eng = capi.Engine(W24, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
    eng.set_option("pipeline", pipeline)
    toks, frames = _run_engine_tokens(eng, R, pcms, n)
    eng.close()
    om = ob.OracleModel(W24, L)
    rows = _agreement(om, R, pcms, n, toks, frames)
    del om
    tot_ref = sum(r["ref_tokens"] for r in rows)
    rate = sum(r["aligned_ratio"] * r["ref_tokens"] for r in rows) / max(tot_ref, 1)
    _report(f"bf16_token_agreement_b{B}_R{R}", dict(rows=rows, aligned_rate=rate, eps_margin=EPS_MARGIN))
    assert tot_ref > (50 if B > 1 else 10), tot_ref
    for r in rows:
        d = r["first_divergence"]
        assert d is None or (d["decision"] >= 0 and d["margin"] < EPS_MARGIN), r
    assert B == 1 or rate > 0.5, rows


def test_f32_engine_64_streams_is_token_exact_on_the_near_tie_checkpoint(W24):
    """Round 4 (missing #2): the configuration whose parity does not depend on a friendly checkpoint, at configs[2]'s
    batch.  64 streams x R = 13, 24 layers, NASR_DTYPE_F32 -- every GEMM of the step (M = 896) on v_mfma_f32_32x32x2_f32
    (k_gemm_f32_mfma, bit-identical to the FMA tile kernel: tests/test_gpu_parity.py) -- on the RANDOM checkpoint, where 1-3 % of
    the greedy decisions are near-ties: tokens AND emission frames of all 64 streams equal the F32 oracle's, pipelined as
    benchmarked (reference bar: exact tokens, tests/test_compute.cpp:2805-2817).  bench.py times this configuration
    (f32_engine.b64_R13_ms_per_step)."""
    L, R, B = 24, 13, 64
    n, pcms = _pieces(R, 3, 1200, B)                     # 3 pushes of 1.12 s + the tail flush: 2 chunks + a partial one per stream
    eng = capi.Engine(W24, n_layers=L, dtype=capi.DTYPE_F32, max_streams=B)
    eng.set_option("pipeline", 4)
    toks, frames = _run_engine_tokens(eng, R, pcms, n)
    eng.close()
    om = ob.OracleModel(W24, L)
    total, bad = 0, []
    for b, pcm in enumerate(pcms):
        ost = ob.OracleStream(om, R)
        ref = []
        for o in range(0, pcm.size, n):
            ref += ost.process(pcm[o:o + n])
        ref += ost.finalize()
        total += len(ref)
        if toks[b] != ref or frames[b] != ost.token_frames():
            bad.append(b)
    del om
    _report("f32_engine_b64_R13_tokens", dict(streams=B, oracle_tokens=total, streams_that_differ=bad))
    assert total > 100 and not bad, (total, bad)


def test_q8_0_token_agreement_vs_ggml_q8_semantics(Q24):
    """BASELINE config 3's parity question: the engine fed Q8_0 tensors (dequantised to bf16) against ggml-CPU Q8_0
    semantics (oracle ORC_EMU_Q8_ACT, parity unpinned: ggml is absent from the reference tree).  4 streams x 11 s, R = 13.
    Same bar as above: first divergences only at oracle margins below EPS_MARGIN."""
    engW, deq, blocks = Q24
    L, R, B = 24, 13, 4
    n, pcms = _pieces(R, 10, 900, B)
    eng = capi.Engine(engW, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
    toks, frames = _run_engine_tokens(eng, R, pcms, n)
    eng.close()
    om = ob.OracleModel(deq, L, emulate_q8_act=True, q8_blocks=blocks)
    rows = _agreement(om, R, pcms, n, toks, frames)
    del om
    tot_ref = sum(r["ref_tokens"] for r in rows)
    rate = sum(r["aligned_ratio"] * r["ref_tokens"] for r in rows) / max(tot_ref, 1)
    _report("q8_0_token_agreement", dict(rows=rows, aligned_rate=rate, eps_margin=EPS_MARGIN))
    assert tot_ref > 10
    for r in rows:
        d = r["first_divergence"]
        assert d is None or (d["decision"] >= 0 and d["margin"] < EPS_MARGIN), r


def _shipped_pipeline_identity(eng, R, pcms, n_steps, ragged_at, L, spot_streams, modes=(0, 1, 2, 3, 4)):
    """steps `pcms` through `eng` with pipeline = 0..4 (8 = the grouped pipeline) and returns per mode: tokens, token frames, encoder-out taps at three
    points of the run, decoder state, K / V / conv caches of every layer for the spot streams, per-stream counters"""
    B, n = len(pcms), synth.shift_samples(R)
    res = {}
    for mode in modes:
        eng.set_option("pipeline", mode)
        sts = [eng.stream(R) for _ in range(B)]
        toks, encs = [[] for _ in range(B)], []
        for k in range(n_steps):
            if k == ragged_at:            # a ragged push (two uneven parts): not graph-eligible, lands mid-pipeline
                cuts = (slice(k * n, k * n + 777), slice(k * n + 777, (k + 1) * n))
            else:
                cuts = (slice(k * n, (k + 1) * n),)
            for c in cuts:
                for b, t in enumerate(eng.step(sts, [p[c] for p in pcms])):
                    toks[b] += t
            if k in (n_steps // 3, 2 * n_steps // 3):     # a tap in the middle of the run (completes what is in flight)
                encs.append(np.stack([sts[b].tap(capi.TAP_ENCODER_OUT) for b in spot_streams]))
        for b, t in enumerate(eng.finalize(sts)):
            toks[b] += t
        encs.append(np.stack([sts[b].tap(capi.TAP_ENCODER_OUT) for b in spot_streams]))
        dec = np.stack([s.tap(capi.TAP_DEC_STATE) for s in sts])
        caches = np.stack([np.concatenate([sts[b].tap(capi.TAP_K_CACHE, l, cap=70 * 1024), sts[b].tap(capi.TAP_V_CACHE, l, cap=70 * 1024),
                                           sts[b].tap(capi.TAP_CONV_CACHE, l, cap=8 * 1024)]) for b in spot_streams for l in range(L)])
        counters = [(s.stats().chunks, s.stats().decode_iterations, s.stats().tokens, s.stats().cache_valid_len) for s in sts]
        frames = [s.token_frames() for s in sts]
        res[mode] = (toks, frames, encs, dec, caches, counters)
        for s in sts:
            s.destroy()
    return res


def _assert_modes_identical(res):
    toks0, frames0, encs0, dec0, caches0, counters0 = res[0]
    assert sum(len(t) for t in toks0) > 0
    for mode in [m for m in res if m != 0]:
        toks, frames, encs, dec, caches, counters = res[mode]
        assert toks == toks0 and frames == frames0, mode
        assert counters == counters0, mode
        assert all(np.array_equal(a, b) for a, b in zip(encs, encs0)), mode
        assert np.array_equal(dec, dec0), mode
        assert np.array_equal(caches, caches0), mode


def test_shipped_pipeline_bit_identity_batch1_R0_24_layers(W24):
    """What bench.py times at configs[1]: 24 layers, bf16, ONE stream x 80 ms, the fused 8-launch layer, pieces cut at
    launch granularity (E = 3: launches 57 | 126; E = 4: 48 | 104 | 164 of 192).  160 steps (+ a ragged push in the middle)
    with pipeline = 1, 2, 3, 4 == synchronous stepping: tokens, token frames, encoder output at three points of the run,
    decoder state, the K / V / conv cache of every layer, the counters -- bit for bit."""
    L, R = 24, 0
    n_steps = 160
    pcm = synth.make_pcm(31, n_steps * 0.08 + 0.01)[:n_steps * 1280]
    eng = capi.Engine(W24, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=2)
    res = _shipped_pipeline_identity(eng, R, [pcm], n_steps, 81, L, (0,), modes=(0, 1, 2, 3, 4, 8))
    assert eng.counter("grouped_steps") > n_steps - 10          # mode 8 really ran the grouped pipeline
    eng.close()
    assert res[0][5][0][0] >= n_steps - 1
    _assert_modes_identical(res)


def test_shipped_pipeline_bit_identity_b64_R13_q8_0_24_layers(Q24):
    """What bench.py times at configs[2]: 24 layers, Q8_0 tensors -> bf16 engine, 64 streams x 1.12 s (M = 896: tiled GEMMs,
    split-K, XCD remap), pieces of 6 + 7 + 7 + 4 layers.  150 steps (+ a ragged push) with pipeline = 1, 2, 3, 4 ==
    synchronous stepping, bit for bit (caches of 4 spot streams x 24 layers)."""
    engW, _, _ = Q24
    L, R, B = 24, 13, 64
    n_steps, n = 150, synth.shift_samples(13)
    base = [synth.make_pcm(800 + b, 25 * n / 16000 + 0.01)[:25 * n] for b in range(B)]
    pcms = [np.tile(p, n_steps // 25) for p in base]        # bit-identity does not care that the audio repeats every 28 s
    eng = capi.Engine(engW, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
    res = _shipped_pipeline_identity(eng, R, pcms, n_steps, 75, L, (0, 21, 42, 63))
    eng.close()
    assert res[0][5][0][0] >= n_steps - 1
    _assert_modes_identical(res)


def test_graph_cache_is_bounded_over_10000_calls_with_random_batches(W2):
    """A server's batch size changes from call to call (tests/server_load.py: 1 .. 63 streams per engine call with 64 live
    streams).  With option graph_cache = 4 the engine keeps at most 4 step shapes per slot: over 10 000 pipelined calls with a
    random subset of 12 streams each, the number of live hipGraphExec objects stays under its bound (and is the same at call
    2 000 as at call 10 000), shapes are evicted and re-captured, and every stream's tokens equal those of an engine that never
    evicts; three streams are checked against the oracle."""
    L, R, S, N = 2, 0, 12, 10000
    rng = np.random.default_rng(1234)
    plan = [rng.permutation(S)[:rng.integers(1, S + 1)] for _ in range(N)]
    pushes = np.zeros(S, np.int64)
    for sel in plan:
        pushes[sel] += 1
    base = [synth.make_pcm(900 + b, 30.0)[:375 * 1280] for b in range(S)]
    pcms = [np.tile(base[b], int(pushes[b]) // 375 + 1) for b in range(S)]
    res, counters = {}, {}
    for cap in (4, 64):
        eng = capi.Engine(W2, n_layers=L, dtype=capi.DTYPE_F32, max_streams=S)
        eng.set_option("pipeline", 4)
        eng.set_option("graph_cache", cap)
        sts = [eng.stream(R) for _ in range(S)]
        toks, pos, mid = [[] for _ in range(S)], np.zeros(S, np.int64), None
        for i, sel in enumerate(plan):
            out = eng.step([sts[b] for b in sel], [pcms[b][pos[b] * 1280:(pos[b] + 1) * 1280] for b in sel])
            for b, t in zip(sel, out):
                toks[b] += t
            pos[sel] += 1
            if i == 2000:
                mid = eng.counter("graph_execs")
        for b, t in enumerate(eng.finalize(sts)):
            toks[b] += t
        counters[cap] = dict(execs_mid=mid, execs_end=eng.counter("graph_execs"), shapes=eng.counter("graph_shapes"),
                             evictions=eng.counter("graph_evictions"), frames=[s.token_frames() for s in sts[:3]])
        res[cap] = toks
        eng.close()
    _report("graph_cache", {str(k): {kk: vv for kk, vv in v.items() if kk != "frames"} for k, v in counters.items()})
    c4, c64 = counters[4], counters[64]
    assert c4["evictions"] > 100 and c64["evictions"] == 0
    per_shape = 5 * (min(4, L) + 1)                     # NSLOT slots x (encoder pieces: one per layer at most + the decode graph)
    assert c4["execs_end"] <= 4 * per_shape and c4["execs_mid"] == c4["execs_end"]      # bounded, and flat from call 2 000 to call 10 000
    assert c64["shapes"] == S and c64["execs_end"] == S * per_shape
    assert res[4] == res[64] and sum(len(t) for t in res[4]) > 100
    om = ob.OracleModel(W2, L)
    for b in range(3):                                    # the first 40 s of three streams against the oracle
        ost = ob.OracleStream(om, R)
        ref = ost.process(pcms[b][:500 * 1280])
        rf = ost.token_frames()
        n = sum(1 for f in rf if f < 480)
        got = [t for t, f in zip(res[4][b], c4["frames"][b]) if f < 480]
        assert got == ref[:n] and n > 0, b


def test_grouped_pipeline_two_streams_ragged_and_mixed_calls():
    """pipeline = 8 (two chains of four-problem launches, 8 steps in flight) on an 8-layer model, two streams: together (M = 2
    rows per problem), alone, a ragged push (drains, runs eagerly), a stats call in the middle (drains: bubbles run through the
    stages), another lookahead in between (not eligible: lanes mode), finalize -- tokens, frames, decoder state and encoder
    output equal synchronous stepping bit for bit."""
    L = 8
    W = synth.make_weights(n_layers=L)
    pcms = [synth.make_pcm(70 + b, 6.0) for b in range(3)]
    res = {}
    for mode in (0, 8):
        eng = capi.Engine(W, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=3)
        eng.set_option("pipeline", mode)
        a, b = eng.stream(0), eng.stream(0)
        c = eng.stream(1)
        toks = [[], [], []]
        n = 1280
        for k in range(60):
            if k % 7 == 3:                          # stream a alone
                toks[0] += eng.step([a], [pcms[0][k * n:(k + 1) * n]])[0]
                toks[1] += eng.step([b], [pcms[1][k * n:(k + 1) * n]])[0]
            elif k == 31:                           # ragged: two uneven parts
                for sl in (slice(k * n, k * n + 500), slice(k * n + 500, (k + 1) * n)):
                    out = eng.step([a, b], [pcms[0][sl], pcms[1][sl]])
                    toks[0] += out[0]; toks[1] += out[1]
            else:
                out = eng.step([a, b], [pcms[0][k * n:(k + 1) * n], pcms[1][k * n:(k + 1) * n]])
                toks[0] += out[0]; toks[1] += out[1]
            if k % 9 == 4:                          # a third stream with another lookahead (T = 2: M = 2 rows, eligible too)
                toks[2] += eng.step([c], [pcms[2][(k // 9) * 2560:(k // 9 + 1) * 2560]])[0]
            if k == 20:
                assert a.stats().chunks > 0         # drains
            if k == 57:                             # right after a joint step (valid until the engine's next chunk step; drains)
                enc = np.stack([s.tap(capi.TAP_ENCODER_OUT) for s in (a, b)])
        out = eng.finalize([a, b]) + eng.finalize([c])
        for i in range(3):
            toks[i] += out[i]
        res[mode] = (toks, [s.token_frames() for s in (a, b, c)], np.stack([s.tap(capi.TAP_DEC_STATE) for s in (a, b, c)]), enc, eng.counter("grouped_steps"))
        eng.close()
    assert res[8][4] > 40 and res[0][4] == 0
    assert res[8][0] == res[0][0] and res[8][1] == res[0][1] and sum(len(t) for t in res[0][0]) > 0
    assert np.array_equal(res[8][2], res[0][2]) and np.array_equal(res[8][3], res[0][3])


def test_reset_reference_mode_keeps_what_the_reference_keeps(W2):
    """nemo_stream_reset as coded (src/nemo-stream.cpp:95-115): conv cache and preprocessor carry survive, K/V contents
    survive but are masked.  Engine (f32) == oracle twin token for token and on the encoder output; and the quirk is
    visible: the first chunk after a reference reset differs from the first chunk after a fresh reset, by the stale conv
    cache (R = 0: exactly the first kernel_size - 1 = 8 frames read stale rows in layer 0)."""
    L = 2
    om = ob.OracleModel(W2, L)
    eng = capi.Engine(W2, n_layers=L, dtype=capi.DTYPE_F32, max_streams=2)
    eng.set_debug(True)
    first, second = synth.make_pcm(7, 1.53), synth.make_pcm(8, 2.0)      # 1.53 s: leaves un-framed samples in the preprocessor
    for R in (0, 13):
        st, ost = eng.stream(R), ob.OracleStream(om, R)
        tap = ost.enable_taps()
        assert eng.step([st], [first])[0] + eng.finalize([st])[0] == ost.process(first) + ost.finalize()
        st.reset(reference=True)
        ost.reset(reference=True)
        assert st.progress().chunks == 0 and st.progress().cache_valid_len == 0 and st.progress().mel_frames_buffered == 9
        n = synth.shift_samples(R)
        tg, to, enc_ref, sub_ref = [], [], [], []
        for o in range(0, second.size, n):
            c0 = ost.total_chunks
            tg += eng.step([st], [second[o:o + n]])[0]
            to += ost.process(second[o:o + n])
            assert st.stats().chunks == ost.total_chunks
            if ost.total_chunks > c0:
                e = st.tap(capi.TAP_LAYER_OUT, L - 1).reshape(-1, 1024)
                assert np.abs(e - tap[1][L - 1]).max() < 2e-3
                enc_ref.append(e.copy())
                sub_ref.append(st.tap(capi.TAP_SUBSAMPLED).reshape(-1, 1024).copy())
        tg += eng.finalize([st])[0]
        to += ost.finalize()
        assert tg == to and st.token_frames() == ost.token_frames()
        # the same audio after a FRESH reset: other mel frames (no preprocessor carry) and no stale conv rows
        st.reset()
        enc_fresh, sub_fresh = [], []
        for o in range(0, second.size, n):
            c0 = st.progress().chunks
            eng.step([st], [second[o:o + n]])
            if st.progress().chunks > c0:
                enc_fresh.append(st.tap(capi.TAP_LAYER_OUT, L - 1).reshape(-1, 1024).copy())
                sub_fresh.append(st.tap(capi.TAP_SUBSAMPLED).reshape(-1, 1024).copy())
        assert len(enc_fresh) >= 1 and np.abs(enc_fresh[0] - enc_ref[0][:enc_fresh[0].shape[0]]).max() > 1e-3
        st.destroy()
    # isolate the conv-cache effect (mel pushed directly: no preprocessor involved), layer 0, R = 0
    rng = np.random.default_rng(5)
    mel_a = (rng.standard_normal((9 * 8, 128)) * 2 - 4).astype(np.float32)
    mel_b = (rng.standard_normal((12 * 8, 128)) * 2 - 4).astype(np.float32)
    outs = {}
    for mode in (False, True):
        st = eng.stream(0)
        for c in range(9):
            eng.step_mel([st], [mel_a[c * 8:(c + 1) * 8]])
        st.reset(reference=mode)
        rows = []
        for c in range(12):
            eng.step_mel([st], [mel_b[c * 8:(c + 1) * 8]])
            rows.append(st.tap(capi.TAP_LAYER_OUT, 0).reshape(-1, 1024)[0].copy())
        outs[mode] = np.stack(rows)
        st.destroy()
    diff = np.abs(outs[True] - outs[False]).max(axis=1)
    assert (diff[:8] > 1e-4).all(), diff            # frames 0..7 read at least one stale conv-cache row
    # from frame 8 on the conv window holds new rows only, and layer 0's K/V rows never depended on the conv module:
    # layer 0 is bit-identical again (deeper layers keep the difference in their K/V rows for 70 frames)
    assert diff[8:].max() == 0.0, diff
    eng.close()


def test_tokens_that_do_not_fit_stay_queued(W2):
    """The hand-over never drops a token: with a 3-entry buffer the step returns 3, the rest comes out of collect /
    finalize in order; the concatenation equals the oracle's tokens."""
    L = 2
    eng = capi.Engine(W2, n_layers=L, dtype=capi.DTYPE_F32, max_streams=1)
    st = eng.stream(0)
    pcm = synth.make_pcm(11, 6.0)
    ost = ob.OracleStream(ob.OracleModel(W2, L), 0)
    ref = ost.process(pcm) + ost.finalize()
    assert len(ref) > 12
    lib = capi.lib()
    h = (C.c_void_p * 1)(st.h)
    buf = np.zeros(3, np.int32)
    tp = (C.c_void_p * 1)(buf.ctypes.data)
    cap = (C.c_int32 * 1)(3)
    nt = (C.c_int32 * 1)()
    pp = (C.c_void_p * 1)(pcm.ctypes.data)
    ns = (C.c_int32 * 1)(pcm.size)
    assert lib.nasr_engine_step(eng.h, h, 1, pp, ns, tp, cap, nt, 0) == 0
    got = buf[:nt[0]].tolist()
    assert nt[0] == 3 and st.progress().reserved > 0          # queued on the stream, visible without a device sync
    while True:
        assert lib.nasr_engine_collect(eng.h, h, 1, tp, cap, nt) == 0
        got += buf[:nt[0]].tolist()
        if nt[0] < 3:
            break
    got += eng.finalize([st])[0]
    assert got == ref
    eng.close()


def test_progress_does_not_drain_the_pipeline(W2):
    """nasr_stream_get_progress is host state only: with pipelined steps the decode of the last step stays in flight (its
    tokens arrive with the next call), where nasr_stream_get_stats completes it."""
    eng = capi.Engine(W2, n_layers=2, dtype=capi.DTYPE_F32, max_streams=1)
    eng.set_option("pipeline", 1)
    st = eng.stream(0)
    pcm = synth.make_pcm(3, 4.0)
    sync = []
    for o in range(0, pcm.size, 1280):
        sync += eng.step([st], [pcm[o:o + 1280]])[0]
    n_before = len(sync)
    chunks = st.progress().chunks
    assert chunks > 40 and st.progress().decode_iterations == -1
    assert st.stats().chunks == chunks                           # drains: the last step's tokens are now queued
    tail = eng.collect([st])[0] + eng.finalize([st])[0]
    ost = ob.OracleStream(ob.OracleModel(W2, 2), 0)
    ref = []
    for o in range(0, pcm.size, 1280):
        ref += ost.process(pcm[o:o + 1280])
    ref += ost.finalize()
    assert sync + tail == ref and n_before <= len(ref)
    eng.close()


@pytest.mark.parametrize("args", [("800", "11"), ("1200", "7", "4", "L8"), ("1000", "3", "4", "L4", "SOAK_STREAMS=13x72,0x4"),      # ("1200", "7", "8") moved out (round 6: suite time); run it with tests/micro/soak_pipeline.py
                                  ("300", "11", "4", "L4", "SOAK_STREAMS=13x100", "SOAK_OPTS=large_step_pieces=2")])
def test_pipelined_engine_soak(args):
    """tests/micro/soak_pipeline.py, short form: random push sizes (partial chunks, several chunks, ragged groups), random subsets of
    five streams of three lookaheads, resets, finalize / collect in between -- a pipelined engine emits exactly the tokens of a
    synchronous one.  (800 calls, seed 11): four lanes on 4 layers (round 2).  (1200 calls, seed 7, 8 layers): the sequence that
    exposed the lane inconsistency of the round-2 cuts at calls 475-499 (a stream alternating between one-to-four-row steps and larger
    ones while both are in flight; enqueue_encoder: snap8) -- with four lanes, and with the grouped pipeline (mode 8).  Last case: 72
    streams of R = 13 (+ 4 of R = 0), ~58 of them per call: the rows of a step cross 768 both ways, so steps on the deep-ring GEMM
    kernels and steps on the co-resident ones are in flight together.  Round 4: 100 streams of R = 13, ~80 per call, pushes of 1-4 chunks: the
    rows of a step cross 3 584 both ways, so steps cut into two pieces (engine option "large_step_pieces" = 2) and into four, on 224-row tiles
    and on 128-row ones, follow each other."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ)
    env.update(a.split("=", 1) for a in args if "=" in a)
    args = [a for a in args if "=" not in a]
    r = subprocess.run([sys.executable, str(root / "tests" / "micro" / "soak_pipeline.py"), *args], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "pipelined == synchronous" in r.stdout


def test_gemm_ring_variants_are_bit_identical():
    """The large-M GEMMs come in two ring depths: deep rings (one 96-128 KiB workgroup per CU: synchronous steps below 1 792 rows) and
    shallow ones of which two share a CU (k_gemm_tiled2_k32<4>, k_gemm_t64<3>: pipelined steps above 768 rows, every step from 1 792 rows).
    Same MFMAs in the same order: tokens, encoder output, K/V and conv caches of 16 / 64 / 128 streams x R = 13 (M = 224, 896, 1 792;
    synchronous and four lanes) have ONE digest whether the choice is left to the engine or forced either way (engine option "gemm_cores").
    Round 4: 256 and 512 streams (M = 3 584, 7 168) are in the digest too, where the PERSISTENT tile loop (k_gemm_persist: loader, consumer
    and storer waves, a tile parked beside the ring while the next one is multiplied) serves every GEMM with >= 1.75 tiles per CU -- the
    same digest with engine option "persistent_gemm" = 1 (off by default: it does not pay inside the engine, profiles/r4_persistent_gemm.md).
    And the same digest with the kernels and the tile order of round 3 ("wide_tiles" = 0, "tile_bands" = 0, "t64_tiles" = 127: no 256- / 224-row tiles, the row
    chunk fastest, half-width tiles up to 127 tiles) and with the 256-row form only and column-group bands at every size ("wide_tiles" = 256, "tile_bands" = 1).
    Round 5: and with "resid_epilogue" = 0 -- split-K partial slabs + k_post everywhere, against the default where the residual GEMMs add to the residual
    stream in their own epilogue (k_gemm_t64w: both K slices in one workgroup; launches without split-K) -- and with that fold in synchronous steps too
    ("resid_epilogue" = 2) and the per-frame depthwise conv ("dwconv_stream" = 0), and with "chain" = 2: k_post as the head phase of the GEMM that reads its rows,
    handed over through per-row-chunk flags inside one launch (measured slower and off by default, but a correct hand-off: same digest)."""
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tests" / "micro" / "gemm_variant_identity.py"), "opt:gemm_cores=1", "opt:gemm_cores=0", "opt:persistent_gemm=1",
                        "opt:persistent_gemm=1 opt:gemm_cores=0", "opt:wide_tiles=0 opt:tile_bands=0 opt:t64_tiles=127", "opt:wide_tiles=256 opt:tile_bands=1", "opt:resid_epilogue=0", "opt:resid_epilogue=2 opt:dwconv_stream=0", "opt:chain=2",
                        "opt:gemm_prio=20", "opt:gemm_prio=16 opt:epilogue16=0"],          # rounds 1-4's GEMM loops (k_gemm_wide, k_gemm_tiled2_k32); round 5's with 8-byte epilogue stores
                       capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stdout.count("==") == 12 and "!=" not in r.stdout


def test_lanes_option_gives_queues_back_and_keeps_results(W2):
    """Engine option "lanes": the engine keeps fewer encoder lanes (before or after it has picked them) -- fewer pieces in flight,
    identical tokens."""
    R, piece = 0, synth.shift_samples(0)
    pcm = synth.make_pcm(77, 6.0)
    res = []
    for lanes_before, lanes_after in ((None, None), (2, None), (None, 1)):
        eng = capi.Engine(W2, n_layers=2, dtype=capi.DTYPE_BF16, max_streams=1)
        if lanes_before:
            eng.set_option("lanes", lanes_before)
        eng.set_option("pipeline", 3 if lanes_before or lanes_after else 0)
        st = eng.stream(R)
        toks = []
        for k in range(pcm.size // piece):
            toks += eng.step([st], [pcm[k * piece:(k + 1) * piece]])[0]
            if lanes_after and k == 20:
                eng.set_option("lanes", lanes_after)          # drains the pipeline, destroys the lanes beyond the first
        toks += eng.finalize([st])[0]
        res.append(toks)
        st.destroy()
        eng.close()
    assert len(res[0]) > 0 and res[0] == res[1] == res[2]
    with pytest.raises(capi.NasrError):
        e = capi.Engine(W2, n_layers=2, dtype=capi.DTYPE_BF16, max_streams=1)
        try:
            e.set_option("lanes", 0)
        finally:
            e.close()


def test_steps_cut_into_different_piece_counts_keep_their_order():
    """Round 4: from 3 584 rows per step the engine cuts a pipelined step into at most three pieces instead of four (every further lane is
    another GEMM's working set in the same L2s; engine option "large_step_pieces", here 2 so that the two cuts differ as much as they can).  Steps in flight order their layers through the lanes, so a step
    that is cut differently from the ones before it has to wait for them: calls of 256 streams x R = 13 (3 584 rows, two pieces) alternate
    with calls of the first 64 of those streams (896 rows, four pieces), then the other 192 catch up -- tokens, encoder output and caches
    of every stream must be those of synchronous stepping, bit for bit.  (With the wait taken out the test still passes: the overtaking
    step would have to get through more layers than the overtaken one has left before it is launched, which these sizes do not produce.
    It covers the mixed path; the wait is there because stream order alone does not guarantee it.)"""
    L, R, B = 4, 13, 256
    W = synth.make_weights(n_layers=L)
    n = synth.shift_samples(R)
    n_chunks = 6
    pcms = [synth.make_pcm(500 + (b % 64), n_chunks * n / 16000 + 0.01)[:n_chunks * n] for b in range(B)]
    pcms = [np.roll(p, 131 * (b // 64)) for b, p in enumerate(pcms)]
    runs = []
    for pipeline in (0, 4):
        eng = capi.Engine(W, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
        eng.set_option("pipeline", pipeline)
        eng.set_option("large_step_pieces", 2)
        sts = [eng.stream(R) for _ in range(B)]
        pos = [0] * B
        toks = [[] for _ in range(B)]

        def call(idx):
            out = eng.step([sts[b] for b in idx], [pcms[b][pos[b] * n:(pos[b] + 1) * n] for b in idx])
            for b, t in zip(idx, out):
                toks[b] += t
                pos[b] += 1

        everyone, first = list(range(B)), list(range(64))
        call(everyone); call(first); call(everyone); call(first); call(first); call(everyone)        # streams 0-63: 6 chunks, the rest: 3
        rest = list(range(64, B))
        call(rest); call(rest); call(rest)
        assert pos == [n_chunks] * B
        for b, t in enumerate(eng.finalize(sts)):
            toks[b] += t
        spot = (0, 63, 64, 255)
        state = [np.concatenate([sts[b].tap(capi.TAP_ENCODER_OUT).ravel()] + [sts[b].tap(tap, l, cap=70 * 1024).ravel() for l in range(L) for tap in (capi.TAP_K_CACHE, capi.TAP_CONV_CACHE)])
                 for b in spot]
        runs.append((toks, state))
        if pipeline:
            assert eng.counter("pipelined_steps") >= 6
        eng.close()
    (t0, s0), (t1, s1) = runs
    assert sum(len(t) for t in t0) > 0
    assert t0 == t1
    for a, b in zip(s0, s1):
        assert np.array_equal(a, b)

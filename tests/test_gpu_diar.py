"""SURVEY.md section 8 f-4 on the GPU, through the C ABI: the batched sliding-window VAD and the speaker embeddings
against the CPU restatement (oracle/diar_oracle.c)."""
import numpy as np
import pytest

from nemotron_asr_amd import capi, synth
from oracle import diar_binding as db

pytestmark = pytest.mark.gpu


def _acc(worst, diff):
    """running maximum of |diff| that does NOT swallow NaN (Python's max(0.0, nan) is 0.0: round 5 found three parity tests blind to an all-NaN engine)"""
    m = float(np.abs(diff).max())
    assert np.isfinite(m), "non-finite values in the engine's output"
    return max(worst, m)


def _audio(seed, n):
    return synth.make_pcm(seed, n / 16000.0 + 0.01)[:n].astype(np.float32) / 32768.0


@pytest.fixture(scope="module")
def Wv():
    return synth.make_diar_weights(spk=False)


def test_vad_batched_windows_match_oracle(Wv):
    """3 buffers of different lengths (one shorter than a window, one exactly one window) in ONE call: every 0.63 s window
    at a 10 ms shift == the oracle's window-by-window run; also through tiles smaller than the window count."""
    om = db.DiarModel(Wv)
    audios = [_audio(1, 10080 + 160 * 37 + 55), _audio(2, 9000), _audio(3, 10080)]
    ref = [om.vad_batch(a) for a in audios]
    for max_windows in (4096, 16):
        eng = capi.Diar(Wv, max_windows=max_windows)
        got = eng.vad(audios)
        assert [g.size for g in got] == [38, 0, 1] == [r.size for r in ref]
        for g, r in zip(got, ref):
            if r.size:
                assert np.abs(g - r).max() < 2e-5, np.abs(g - r).max()
        assert 0.0 < got[0].min() and got[0].max() < 1.0 and np.ptp(got[0]) > 1e-4     # not a constant
        eng.close()


def test_vad_on_the_bf16_mfma_stays_close_to_the_f32_network(Wv):
    """NASR_DIAR_VAD_BF16 (what bench.py's configs[4] entry runs): MarbleNet's pointwise convolutions on the bf16 MFMA, bf16
    activation planes.  P(speech) of every window against the oracle's f32 network: stated tolerance 2e-2 absolute (measured
    below 1e-2 on the synthetic network), same window counts, same handling of short buffers, tiles smaller than the window count."""
    om = db.DiarModel(Wv)
    audios = [_audio(1, 10080 + 160 * 137 + 55), _audio(2, 9000), _audio(3, 10080), _audio(4, 10080 + 160 * 300)]
    ref = [om.vad_batch(a) for a in audios]
    worst = 0.0
    for max_windows in (4096, 64):
        eng = capi.Diar(Wv, dtype=capi.DTYPE_BF16 | capi.DIAR_VAD_BF16, max_windows=max_windows)
        got = eng.vad(audios)
        assert [g.size for g in got] == [r.size for r in ref] == [138, 0, 1, 301]
        for g, r in zip(got, ref):
            if r.size:
                worst = _acc(worst, g - r)
        eng.close()
    assert worst < 2e-2, worst
    f32 = capi.Diar(Wv, dtype=capi.DTYPE_BF16)            # without the flag MarbleNet stays f32: the 2e-5 parity above
    assert np.abs(f32.vad(audios)[0] - ref[0]).max() < 2e-5
    f32.close()


def _poison_lds(pattern):
    import ctypes as C
    from pathlib import Path
    so = Path(__file__).parent / "helpers" / "liblds_poison.so"
    if not so.exists():
        pytest.skip("tests/helpers/liblds_poison.so not built (python __graft_entry__.py warns when the helper fails to compile)")
    L = C.CDLL(str(so))
    L.lds_poison.argtypes = [C.c_int, C.c_uint]
    got = L.lds_poison(0, pattern)
    assert got >= 64 * 1024, got


@pytest.mark.parametrize("flags", [capi.DIAR_VAD_BF16, capi.DIAR_VAD_F16, 0])
def test_vad_does_not_depend_on_what_lds_held_before(Wv, flags):
    """Round-3 advisor: block 0 of k_vad_marblenet_bf16 multiplied LDS columns 80..95 of a plane nobody had written by zero weights
    -- 0 x NaN = NaN in the MFMA.  Every CU's LDS is filled with bf16 NaN pairs / Inf pairs / f32 NaN (tests/helpers/lds_poison.hip,
    a whole-LDS workgroup per CU, 8 rounds) right before the call: same bits as a call after a zero fill."""
    audios = [_audio(1, 10080 + 160 * 137 + 55), _audio(4, 10080 + 160 * 300)]
    eng = capi.Diar(Wv, dtype=capi.DTYPE_BF16 | flags)
    _poison_lds(0)
    ref = eng.vad(audios)
    for pattern in (0x7fc07fc0, 0x7f807f80, 0xffc0ff80, 0x7fc00000):
        _poison_lds(pattern)
        got = eng.vad(audios)
        for g, r in zip(got, ref):
            assert np.isfinite(g).all(), hex(pattern)
            assert np.array_equal(g, r), hex(pattern)
    eng.close()


def _calibrated_vad(W, d, spread=4.0):
    """The synthetic MarbleNet's logit difference d = lg1 - lg0 moves by +-0.6 around -6.7 (P(speech) ~ 0.001-0.006: never a
    segment).  An affine map of the decoder layer -- d' = a (d - median), a = spread / std -- makes P(speech) cross the reference's
    onset 0.9 / offset 0.5 along the audio, so that segments exist and the bf16 network's error is seen at the decision level
    (the map also multiplies that error by a: a stress, not a favour)."""
    W = dict(W)
    w, b = W["vad.decoder.decoder_layers.0.weight"], W["vad.decoder.decoder_layers.0.bias"]
    a, m = spread / float(np.std(d)), float(np.median(d))
    W["vad.decoder.decoder_layers.0.weight"] = np.stack([np.zeros(128, np.float32), (a * (w[1] - w[0])).astype(np.float32)])
    W["vad.decoder.decoder_layers.0.bias"] = np.asarray([0.0, a * (float(b[1]) - float(b[0]) - m)], np.float32)
    return W, a


# asserted: max |dP| vs the oracle at batch 64 (synthetic network, P ~ 0.003), max logit error, and on the calibrated stress track: max
# |dP|, share of frames whose onset / offset threshold decision differs, streams (of 64) whose segment COUNT may differ, worst boundary
# shift in 10 ms frames where the count agrees.  Measured on MI355X (profiles/r4_vad_16bit_segments.md): bf16 1.1e-4 / 0.041 / 0.081 /
# 0.6 % / 9 / 82; f16 2.1e-5 / 0.0061 / 0.012 / 0.1 % / 5 / 2.
_VAD16 = {"bf16": (capi.DIAR_VAD_BF16, 1e-3, 0.08, 0.15, 0.02, 16, 200), "f16": (capi.DIAR_VAD_F16, 1e-4, 0.012, 0.025, 0.004, 8, 3)}


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_config5_vad_on_the_16_bit_mfma_at_batch_64_and_at_the_segment_level(Wv, kind):
    """What bench.py's configs[4] entry runs (MarbleNet on the 16-bit MFMA), at its batch and at the level the pipeline consumes it.
    (i) ONE call over 64 buffers x 112 windows (7 168 windows): every window of 8 streams against the oracle's f32 network, every
    window of all 64 against the f32 kernel (itself within 2e-5 of the oracle).  (ii) Segment level: a decoder calibrated so that
    P(speech) crosses onset 0.9 / offset 0.5 (above), 64 streams x 20 s of the phone / silence audio: nasr_diar_plan on the 16-bit
    probabilities against the f32 ones -- segments per stream, boundary shifts in VAD frames (10 ms), exact-equality rate.
    The calibrated track is a stress by construction: its median sits ON the offset threshold and it crosses onset 0.9 about 85
    times per 20 s stream, so every rounding error that exists meets a decision (a trained VAD is bimodal and rarely near a threshold).
    What is asserted is listed at _VAD16; IEEE-half planes (NASR_DIAR_VAD_F16, what bench.py runs) keep boundaries within 3 frames.
    gpurun_out/r4_vad_<kind>_segments.json."""
    import json
    from pathlib import Path
    from tests.test_diar_pipeline_plan import plan
    flag, tol_p, tol_logit, tol_cal, tol_flip, max_count_differs, tol_shift = _VAD16[kind]
    om = db.DiarModel(Wv)
    B, n = 64, 10080 - 160 + 17920
    pcms = [synth.make_speech_pcm(600 + b, 21.0)[0] for b in range(B)]
    f32 = capi.Diar(Wv, dtype=capi.DTYPE_BF16, max_windows=8192)
    b16 = capi.Diar(Wv, dtype=capi.DTYPE_BF16 | flag, max_windows=8192)
    pf, pb = f32.vad([p[:n] for p in pcms]), b16.vad([p[:n] for p in pcms])
    assert [x.size for x in pb] == [112] * B == [x.size for x in pf]
    worst_oracle = 0.0
    for b in range(0, B, 8):
        ref = om.vad_batch(pcms[b][:n].astype(np.float32) / 32768.0)
        worst_oracle = _acc(worst_oracle, pb[b] - ref)
        assert np.abs(pf[b] - ref).max() < 2e-5
    assert all(np.isfinite(x).all() and np.isfinite(y).all() for x, y in zip(pb, pf))
    worst_f32 = max(float(np.abs(x - y).max()) for x, y in zip(pb, pf))
    lg = lambda p: np.log(p / (1.0 - p))
    d = np.concatenate([lg(x.astype(np.float64)) for x in pf])
    dl = np.concatenate([lg(x.astype(np.float64)) for x in pb]) - d
    f32.close(); b16.close()
    # (ii) calibrated decoder, long buffers
    Wc, gain = _calibrated_vad(Wv, d)
    nl = 20 * 16000
    f32 = capi.Diar(Wc, dtype=capi.DTYPE_BF16, max_windows=8192)
    b16 = capi.Diar(Wc, dtype=capi.DTYPE_BF16 | flag, max_windows=8192)
    qf, qb = f32.vad([p[:nl] for p in pcms]), b16.vad([p[:nl] for p in pcms])
    ref0 = db.DiarModel(Wc).vad_batch(pcms[0][:nl // 4].astype(np.float32) / 32768.0)
    assert np.abs(qf[0][:ref0.size] - ref0).max() < 2e-4            # the calibrated f32 kernel is still the oracle's network
    f32.close(); b16.close()
    n_seg, same, worst_shift, n_sub_same, crossing, count_differs = 0, 0, 0, 0, 0, 0
    frames = sum(x.size for x in qf)
    flips = sum(int(np.sum((x >= 0.9) != (y >= 0.9)) + np.sum((x < 0.5) != (y < 0.5))) for x, y in zip(qf, qb)) / (2.0 * frames)
    for b in range(B):
        sf, uf = plan(qf[b], nl, 0.9, 0.5)
        sb, ub = plan(qb[b], nl, 0.9, 0.5)
        n_seg += len(sf)
        crossing += int(np.sum((qf[b][1:] >= 0.9) != (qf[b][:-1] >= 0.9)))
        same += int(sf == sb)
        n_sub_same += int(uf == ub)
        if len(sf) != len(sb):
            count_differs += 1
            continue
        for x, y in zip(sf, sb):
            worst_shift = max(worst_shift, abs(x[0] - y[0]), abs(x[1] - y[1]))
    assert all(np.isfinite(x).all() and np.isfinite(y).all() for x, y in zip(qb, qf))
    worst_cal = max(float(np.abs(x - y).max()) for x, y in zip(qb, qf))
    report = dict(planes=kind, windows_per_call=B * 112, max_abs_dP_vs_oracle=worst_oracle, max_abs_dP_vs_f32_kernel=worst_f32,
                  logit_error_max=float(np.abs(dl).max()), logit_error_rms=float(np.sqrt(np.mean(dl * dl))), logit_signal_std=float(np.std(d)),
                  calibration_gain=gain, calibrated_max_abs_dP=worst_cal, streams=B, seconds=20, segments_f32=n_seg,
                  onset_crossings_f32=crossing, threshold_decisions_that_differ=flips, streams_with_identical_segments=same, streams_with_identical_sub_segments=n_sub_same,
                  streams_with_another_segment_count=count_differs, worst_boundary_shift_frames=worst_shift)
    out = Path(__file__).resolve().parent.parent / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / f"r4_vad_{kind}_segments.json").write_text(json.dumps(report, indent=1))
    print(report)
    assert worst_oracle < tol_p and worst_f32 < tol_p and report["logit_error_max"] < tol_logit, report
    assert n_seg >= B, report                      # the calibrated track does produce segments
    assert worst_cal < tol_cal and flips < tol_flip, report
    assert count_differs <= max_count_differs and worst_shift <= tol_shift, report


@pytest.mark.parametrize("dtype,tol", [(capi.DTYPE_F32, 2e-3), (capi.DTYPE_BF16, 6e-2)])
def test_speaker_embeddings_match_oracle(dtype, tol):
    """5 sub-segments (full, short, minimal lens) in ONE launch sequence, tiled by max_segments = 2: TitaNet-L embeddings
    == the oracle's one-by-one run.  f32 engine: 2e-3 of the embedding scale; bf16 pointwise convs: 6e-2, cosine > 0.999."""
    W = synth.make_diar_weights(vad=False)
    om = db.DiarModel(W)
    segs = [_audio(10 + i, 24000) for i in range(5)]
    lens = [24000, 24000, 12000, 4321, 100]
    ref = np.stack([om.spk_embed(a, l) for a, l in zip(segs, lens)])
    eng = capi.Diar(W, dtype=dtype, max_segments=2)
    got = eng.embed(segs, lens)
    scale = np.abs(ref).max()
    assert np.isfinite(got).all() and np.abs(got - ref).max() < tol * scale, (np.abs(got - ref).max(), scale)
    for g, r in zip(got, ref):
        assert float(g @ r / (np.linalg.norm(g) * np.linalg.norm(r))) > 0.999
    assert np.abs(eng.embed(segs[:1], lens[:1])[0] - got[0]).max() < 1e-5 * scale      # batch == alone
    eng.close()


def test_config5_batch_64_streams():
    """BASELINE config 5 at its batch: ONE nasr_diar_vad call over 64 buffers of 1.12 s of new audio (+ 0.62 s of history:
    112 windows each, 7 168 windows) and ONE nasr_diar_embed call over 96 sub-segments (64 streams x 1.12 s at a 0.75 s
    shift), s16 PCM, the dtype the benchmark uses (TitaNet pointwise convs on the bf16 MFMA) -- a sample of windows and
    segments against the oracle."""
    W = synth.make_diar_weights()
    om = db.DiarModel(W)
    B, n = 64, 10080 - 160 + 17920
    pcms = [synth.make_pcm(600 + b, 2.6)[:max(n, 36000)] for b in range(B)]
    eng = capi.Diar(W, dtype=capi.DTYPE_BF16, max_windows=8192, max_segments=96)
    probs = eng.vad([p[:n] for p in pcms])
    assert [p.size for p in probs] == [112] * B
    allp = np.concatenate(probs)
    assert np.isfinite(allp).all() and 0.0 < allp.min() and allp.max() < 1.0
    worst = 0.0
    for b in (0, 17, 40, 63):
        a = pcms[b][:n].astype(np.float32) / 32768.0
        for w in (0, 1, 55, 110, 111):
            worst = max(worst, abs(float(probs[b][w]) - om.vad_window(a[w * 160:w * 160 + 10080])))
    assert worst < 2e-5, worst
    segs = [pcms[i % B][12000 * (i // B):12000 * (i // B) + 24000].astype(np.float32) / 32768.0 for i in range(96)]
    emb = eng.embed(segs)
    assert emb.shape == (96, 192) and np.isfinite(emb).all()
    for i in (0, 31, 64, 95):
        r = om.spk_embed(segs[i])
        assert np.abs(emb[i] - r).max() < 6e-2 * np.abs(r).max(), i
        assert float(emb[i] @ r / (np.linalg.norm(emb[i]) * np.linalg.norm(r))) > 0.999, i
    assert np.abs(eng.embed(segs[40:41])[0] - emb[40]).max() < 1e-2 * np.abs(emb[40]).max()      # batch of 96 == alone (bf16 tiles differ)
    eng.close()


def test_s16_device_resident_audio_equals_float_host_audio(Wv):
    """NASR_FLAG_AUDIO_S16 | NASR_FLAG_PCM_DEVICE: the side-car reads s16 PCM that is already in HBM (the ASR streams' own
    buffers) -- same probabilities / embeddings as float host audio (x / 32768 is exact in f32)."""
    W = dict(Wv)
    W.update(synth.make_diar_weights(vad=False))
    eng = capi.Diar(W, dtype=capi.DTYPE_F32)
    asr = capi.Engine(synth.make_weights(n_layers=1), n_layers=1, dtype=capi.DTYPE_BF16, max_streams=1)   # only for its device allocator
    pcm = [synth.make_pcm(30 + b, 1.6)[:24000 + 800 * b] for b in range(2)]
    dev = [asr.upload(p) for p in pcm]
    host = eng.vad([p.astype(np.float32) / 32768.0 for p in pcm])
    for got in (eng.vad(pcm), eng.vad_device_s16(dev, [p.size for p in pcm])):
        for g, h in zip(got, host):
            assert g.size == h.size > 0 and np.abs(g - h).max() < 1e-6
    e_host = eng.embed([p[:24000].astype(np.float32) / 32768.0 for p in pcm])
    e_dev = eng.embed_device_s16(dev)
    assert np.abs(e_dev - e_host).max() < 1e-5 * np.abs(e_host).max()
    eng.close()
    asr.close()


@pytest.mark.parametrize("which,norm", [("vad_ref", False), ("spk_ref", True)])
def test_front_end_matches_nemo_fixture(which, norm):
    """The device front end against the NeMo-generated fixtures of the reference's tests/diarize/ (the check of
    tests/test_diarize_preproc.cpp, threshold 1e-3), with NeMo's Slaney filterbank in place of the synthetic one."""
    from pathlib import Path
    g = np.load(Path(__file__).parent / "golden" / "nemo_diar_v1.npz")
    ns = which[:3]
    W = synth.make_diar_weights(vad=ns == "vad", spk=ns == "spk")
    W[f"{ns}.preprocessor.featurizer.fb"] = synth.slaney_filterbank(80)
    eng = capi.Diar(W, dtype=capi.DTYPE_F32, max_segments=1)
    audio, ref = g[f"{which}_audio"], g[f"{which}_mel"]
    mel, tv = eng.logmel(audio, ns, norm)
    assert mel.shape == ref.shape and tv == audio.size // 160
    assert np.abs(mel[:, :tv] - ref[:, :tv]).max() < 2e-4
    assert (mel[:, tv:] == 0).all()
    eng.close()


def test_vad_errors(Wv):
    eng = capi.Diar(Wv)
    with pytest.raises(capi.NasrError):
        eng.embed([_audio(1, 24000)])            # no 'spk.*' tensors in this engine
    with pytest.raises(capi.NasrError):
        capi.Diar({"encoder.x": np.zeros(4, np.float32)})
    bad = dict(Wv)
    del bad["vad.encoder.encoder.3.res.0.1.running_var"]
    with pytest.raises(capi.NasrError, match="running_var"):
        capi.Diar(bad)
    eng.close()


def test_side_car_on_a_stream_lent_by_the_asr_engine(Wv):
    """nasr_engine_lend_stream + nasr_diar_set_stream: the side-car runs on the ASR engine's fourth stream (a hardware queue no
    encoder lane uses) -- same probabilities as on its own stream; the engine keeps stepping (one piece fewer), same tokens."""
    audios = [_audio(1, 10080 + 160 * 37 + 55), _audio(3, 10080)]
    d0 = capi.Diar(Wv)
    ref = d0.vad(audios)
    d0.close()
    W = synth.make_weights(n_layers=4)
    pcm = synth.make_pcm(9, 4.0)
    piece = synth.shift_samples(0)
    toks = []
    for lend in (False, True):
        eng = capi.Engine(W, n_layers=4, dtype=capi.DTYPE_BF16, max_streams=1)
        eng.set_option("pipeline", 4)
        st = eng.stream(0)
        out = []
        diar = capi.Diar(Wv)
        for k in range(pcm.size // piece):
            out += eng.step([st], [pcm[k * piece:(k + 1) * piece]])[0]
            if lend and k == 10:
                diar.set_stream(eng.lend_stream())
            if k % 8 == 3:
                got = diar.vad(audios)
                for g, r in zip(got, ref):
                    assert g.size == r.size and (r.size == 0 or np.array_equal(g, r))
        out += eng.finalize([st])[0]
        toks.append(out)
        diar.close()                   # the borrower first
        st.destroy()
        eng.close()
    assert len(toks[0]) > 0 and toks[0] == toks[1]


def test_engine_destroyed_before_its_borrower_leaves_nothing_dangling(Wv, capfd):
    """Round-2 advisor / round-3 verdict: the side-car held a raw hipStream_t that nasr_engine_destroy destroyed under it.  Borrowers
    are counted now: the engine going first is reported (stderr + nasr_last_error) and the stream lives on until the side-car lets
    go -- the side-car keeps working on it, with the same results."""
    audios = [_audio(1, 10080 + 160 * 37 + 55)]
    d0 = capi.Diar(Wv)
    ref = d0.vad(audios)
    d0.close()
    eng = capi.Engine(synth.make_weights(n_layers=1), n_layers=1, dtype=capi.DTYPE_BF16, max_streams=1)
    eng.set_option("pipeline", 4)
    diar = capi.Diar(Wv)
    diar.set_stream(eng.lend_stream())
    assert np.array_equal(diar.vad(audios)[0], ref[0])
    eng.close()                                    # the lender first
    err = capfd.readouterr().err
    assert "still held by 1 client" in err, err
    assert np.array_equal(diar.vad(audios)[0], ref[0])        # the stream is still there
    diar.close()                                   # the last borrower destroys it
    # the usual order stays silent
    eng = capi.Engine(synth.make_weights(n_layers=1), n_layers=1, dtype=capi.DTYPE_BF16, max_streams=1)
    eng.set_option("pipeline", 4)
    diar = capi.Diar(Wv)
    diar.set_stream(eng.lend_stream())
    diar.close()
    eng.close()
    assert "still held" not in capfd.readouterr().err

"""Fits the output layer of the 'speech' synthetic checkpoint (nemotron-asr.cpp_amd/synth.py: speech_decoder_tensors) and
writes nemotron-asr.cpp_amd/data/speech_readout_v1.npz.

Runs on a GPU box (the 24-layer F32 engine extracts the encoder features of the calibration audio in seconds; the CPU
oracle needs ~3 minutes per minute of audio):

    gpurun -- python tests/golden/gen_speech_joint.py gpurun_out/speech

What it does
  1. calibration audio: synth.make_speech_pcm(stream = 100 .. 100 + N_CAL), 40 s each, through the F32 engine with the
     speech checkpoint's encoder (default synthetic weights, residual branches scaled by synth.SPEECH_RESIDUAL_SCALE) at R = 0
     and R = 13; the encoder output of every frame is read back (ENCODER_OUT tap).
  2. frame labels from the audio's own event list: phone k where the phone covers the whole 80 ms frame, silence where no
     phone touches it; partially covered frames are left out of the fit.
  3. ridge regression from the 1024-d encoder output to one-hot targets (16 phones + silence), both R pooled
     ->  w [17][1024], b [17]: the 17 detector rows of joint.enc.
  4. held-out check (streams 0 .. N_VAL - 1, 60 s, never used in the fit): score margins, the F32 engine's transcript
     against the phone sequence of the audio, and the bf16 engine's tokens against the F32 engine's.
The npz holds the fit only (70 KB); everything else of the checkpoint is generated from the seed.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from nemotron_asr_amd import capi, synth  # noqa: E402

N_CAL, CAL_SECONDS, N_VAL, VAL_SECONDS, LAMBDA = 48, 40.0, 16, 60.0, 3.0
K = synth.N_PHONES


def frame_labels(ev, n_frames):
    """>= 0 phone (frame fully inside the phone), -1 silence (no phone touches the frame), <= -2 partly covered by phone -2 - lab"""
    lab = np.full(n_frames, -1, np.int32)
    for k, a, b in ev:
        for f in range(a // 1280, min((b - 1) // 1280 + 1, n_frames)):
            c = (min(b, (f + 1) * 1280) - max(a, f * 1280)) / 1280.0
            lab[f] = k if c >= 0.999 else -2 - k
    return lab


def run_features(eng, R, streams, seconds):
    """encoder output of every frame of every stream: [B][frames][1024]"""
    n = synth.shift_samples(R)
    T = 1 + R
    pcms, evs = zip(*[synth.make_speech_pcm(s, seconds) for s in streams])
    sts = [eng.stream(R) for _ in streams]
    out = [[] for _ in streams]
    n_push = pcms[0].size // n
    for k in range(n_push):
        before = [s.progress().chunks for s in sts]
        eng.step(sts, [p[k * n:(k + 1) * n] for p in pcms])
        for b, s in enumerate(sts):
            if s.progress().chunks > before[b]:
                out[b].append(s.tap(capi.TAP_ENCODER_OUT).reshape(T, 1024))
    for s in sts:
        s.destroy()
    return [np.concatenate(o) for o in out], evs


def run_tokens(eng, R, streams, seconds):
    n = synth.shift_samples(R)
    pcms, evs = zip(*[synth.make_speech_pcm(s, seconds) for s in streams])
    sts = [eng.stream(R) for _ in streams]
    toks = [[] for _ in streams]
    for k in range(pcms[0].size // n):
        for b, t in enumerate(eng.step(sts, [p[k * n:(k + 1) * n] for p in pcms])):
            toks[b] += t
    for b, t in enumerate(eng.finalize(sts)):
        toks[b] += t
    frames = [s.token_frames() for s in sts]
    for s in sts:
        s.destroy()
    return toks, frames, evs


def main():
    out_dir = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/speech")
    out_dir.mkdir(parents=True, exist_ok=True)
    t0 = time.time()
    W = synth.make_weights(24)
    rep = {}
    import os
    if os.environ.get("SPEECH_ALPHAS"):
        # exploration: how much of the frame's phone is linearly readable off the encoder for a given residual-branch scale
        for alpha in [float(x) for x in os.environ["SPEECH_ALPHAS"].split(",")]:
            Wa = synth.scale_residual_branches(dict(W), alpha)
            eng = capi.Engine(Wa, n_layers=24, dtype=capi.DTYPE_F32, max_streams=32)
            b16 = capi.Engine(Wa, n_layers=24, dtype=capi.DTYPE_BF16, max_streams=8)
            row = {}
            for R in (0, 13):
                feats, evs = run_features(eng, R, range(100, 124), 30.0)
                vf, vev = run_features(eng, R, range(8), 30.0)
                vf16, _ = run_features(b16, R, range(8), 30.0)
                H_, Y_ = [], []
                for e, ev in zip(feats, evs):
                    lab = frame_labels(ev, e.shape[0])
                    keep = lab >= -1
                    y = np.zeros((e.shape[0], K + 1))
                    y[np.arange(e.shape[0]), np.where(lab >= 0, lab, K)] = 1.0
                    H_.append(e[keep].astype(np.float64)); Y_.append(y[keep])
                H_, Y_ = np.concatenate(H_), np.concatenate(Y_)
                H1 = np.concatenate([H_, np.ones((H_.shape[0], 1))], axis=1)
                sol = np.linalg.solve(H1.T @ H1 + LAMBDA * np.eye(H1.shape[1]), H1.T @ Y_)
                mg, noise = [], []
                for e, e16, ev in zip(vf, vf16, vev):
                    lab = frame_labels(ev, e.shape[0])
                    f32s = np.concatenate([e, np.ones((e.shape[0], 1), np.float32)], axis=1) @ sol
                    f16s = np.concatenate([e16, np.ones((e.shape[0], 1), np.float32)], axis=1) @ sol
                    f32s[:, K] += synth.SPEECH_BLANK_BIAS
                    keep = lab >= -1
                    cls = np.where(lab >= 0, lab, K)
                    own = f32s[np.arange(e.shape[0]), cls]
                    rest = f32s.copy(); rest[np.arange(e.shape[0]), cls] = -1e9
                    mg.append((own - rest.max(axis=1))[keep]); noise.append((f16s - f32s)[keep][:, :K].ravel())
                mg, noise = np.concatenate(mg), np.concatenate(noise)
                row[f"R{R}"] = dict(train_rms=float(np.sqrt(((H1 @ sol - Y_) ** 2).mean())), heldout_acc=float((mg > 0).mean()),
                                    margin_pct_0p1_1_5_50=[round(float(x), 3) for x in np.percentile(mg, [0.1, 1, 5, 50])],
                                    bf16_noise_rms=float(np.sqrt((noise ** 2).mean())), bf16_noise_max=float(np.abs(noise).max()),
                                    enc_err_mean=float(np.mean([np.abs(a - b_).mean() for a, b_ in zip(vf, vf16)])))
            eng.close(); b16.close()
            rep[f"alpha_{alpha}"] = row
            print(alpha, json.dumps(row), f"{time.time() - t0:.0f}s", flush=True)
        (out_dir / "speech_alpha_scan.json").write_text(json.dumps(rep, indent=1))
        return
    synth.scale_residual_branches(W, synth.SPEECH_RESIDUAL_SCALE)
    eng = capi.Engine(W, n_layers=24, dtype=capi.DTYPE_F32, max_streams=max(N_CAL, N_VAL))
    H, Y = [], []
    val = {}
    for R in (0, 13):
        feats, evs = run_features(eng, R, range(100, 100 + N_CAL), CAL_SECONDS)
        for e, ev in zip(feats, evs):
            lab = frame_labels(ev, e.shape[0])
            keep = lab >= -1
            y = np.zeros((e.shape[0], K + 1), np.float64)
            y[np.arange(e.shape[0]), np.where(lab >= 0, lab, K)] = 1.0
            H.append(e[keep].astype(np.float64))
            Y.append(y[keep])
        val[R] = run_features(eng, R, range(N_VAL), VAL_SECONDS)
        print(f"R={R}: features done, {time.time() - t0:.0f}s", flush=True)
    eng.close()
    H, Y = np.concatenate(H), np.concatenate(Y)
    H1 = np.concatenate([H, np.ones((H.shape[0], 1))], axis=1)
    G = H1.T @ H1 + LAMBDA * np.eye(H1.shape[1])
    G[-1, -1] -= LAMBDA                      # the intercept is not regularised
    sol = np.linalg.solve(G, H1.T @ Y)
    w, b = sol[:-1].T.astype(np.float32), sol[-1].astype(np.float32)
    np.savez(out_dir / synth.SPEECH_READOUT_FILE, w=w, b=b)
    rep["fit"] = dict(frames=int(H.shape[0]), n_cal=N_CAL, cal_seconds=CAL_SECONDS, ridge=LAMBDA,
                      train_residual_rms=float(np.sqrt(((H1 @ sol - Y) ** 2).mean())))

    # held-out: margins of the frame scores (target units) -- phone frames: own score vs the best of the rest incl. blank + bias
    bf = capi.Engine(W, n_layers=24, dtype=capi.DTYPE_BF16, max_streams=N_VAL)
    for R in (0, 13):
        feats, evs = val[R]
        feats16, _ = run_features(bf, R, range(N_VAL), VAL_SECONDS)
        mg, noise = [], []
        for e, e16, ev in zip(feats, feats16, evs):
            lab = frame_labels(ev, e.shape[0])
            s32 = e @ w.T + b
            s16 = e16 @ w.T + b
            s32[:, K] += synth.SPEECH_BLANK_BIAS
            s16[:, K] += synth.SPEECH_BLANK_BIAS
            keep = lab >= -1
            cls = np.where(lab >= 0, lab, K)
            own = s32[np.arange(e.shape[0]), cls]
            rest = s32.copy()
            rest[np.arange(e.shape[0]), cls] = -1e9
            mg.append((own - rest.max(axis=1))[keep])
            noise.append((s16 - s32)[keep].ravel())
        mg, noise = np.concatenate(mg), np.concatenate(noise)
        rep[f"heldout_R{R}"] = dict(frames=int(mg.size), frame_accuracy=float((mg > 0).mean()),
                                    margin_percentiles_0p1_1_5_50=[float(x) for x in np.percentile(mg, [0.1, 1, 5, 50])],
                                    bf16_score_noise_rms=float(np.sqrt((noise ** 2).mean())), bf16_score_noise_max=float(np.abs(noise).max()))
    bf.close()

    # end to end with the speech checkpoint: F32 engine's transcript vs the audio's phone sequence, bf16 engine vs F32 engine
    Ws = synth.apply_speech_decoder(W, readout=(w, b))
    res = {}
    for name, dt in (("f32", capi.DTYPE_F32), ("bf16", capi.DTYPE_BF16)):
        e = capi.Engine(Ws, n_layers=24, dtype=dt, max_streams=N_VAL)
        res[name] = {R: run_tokens(e, R, range(N_VAL), VAL_SECONDS) for R in (0, 13)}
        e.close()
    for R in (0, 13):
        t32, f32_, evs = res["f32"][R]
        t16, f16, _ = res["bf16"][R]
        want = [[synth.phone_token(k) for k, _, _ in ev] for ev in evs]
        rep[f"tokens_R{R}"] = dict(
            streams=N_VAL, phones=sum(len(x) for x in want), f32_tokens=sum(len(x) for x in t32),
            f32_transcript_equals_phone_sequence=sum(a == b_ for a, b_ in zip(t32, want)),
            bf16_tokens_equal_f32=sum(a == b_ for a, b_ in zip(t16, t32)),
            bf16_frames_equal_f32=sum(a == b_ for a, b_ in zip(f16, f32_)),
            frame_shifts=int(sum(sum(x != y for x, y in zip(a, b_)) for a, b_ in zip(f16, f32_) if len(a) == len(b_))))
    (out_dir / "speech_fit_report.json").write_text(json.dumps(rep, indent=1))
    print(json.dumps(rep, indent=1))
    print(f"done in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()

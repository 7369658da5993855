"""Where does the bf16 engine stop being token-exact?  

The speech checkpoint (synth.py) shows token-for-token agreement of the bf16 engine with the F32 arithmetic where it is easy:
residual branches scaled by 0.1 and 8 logits per target unit.  This sweep makes it harder in both directions:

    SPEECH_RESIDUAL_SCALE alpha in {0.1, 0.2, 0.3, 0.5}   (the read-out is refitted per alpha: the encoder changes)
    SPEECH_LOGIT_SCALE    A     in {8, 4, 2}               (logits per target unit: margins shrink, the bf16 noise in logits too --
                                                            what changes is the share of decisions inside the noise)

and reports per cell, at configs[1]'s shape (1 stream x R = 0, 16 streams one after the other) and configs[2]'s (64 streams x
R = 13): bf16 engine tokens == F32 engine tokens (the F32 engine's tokens are the oracle's: tests/test_gpu_parity.py), emission-frame
shifts, the measured bf16 noise of the detector logits, and the oracle's decision margins (1st percentile, share >= 0.5).

Two phases, because the oracle is CPU work and GPU-box minutes are not:
    gpurun -- python tests/micro/margin_sweep.py gpu gpurun_out/margin      # fits, engine runs; writes readout_a<alpha>.npz + gpu.json
    python tests/micro/margin_sweep.py cpu gpurun_out/margin                 # oracle margins per alpha (A = 8; margins scale with A) + one cell at A = 4
    python tests/micro/margin_sweep.py merge gpurun_out/margin profiles/r4_margin_sweep.json
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
from nemotron_asr_amd import synth  # noqa: E402

import os
ALPHAS = tuple(float(x) for x in os.environ["MARGIN_ALPHAS"].split(",")) if os.environ.get("MARGIN_ALPHAS") else (0.1, 0.2, 0.3, 0.5)      # MARGIN_ALPHAS=0.4: one more column, merged with the rest
SCALES = (8.0, 4.0, 2.0)
N_CAL, CAL_SECONDS, LAMBDA = 24, 30.0, 3.0
N_TOK, TOK_SECONDS = 16, 30.0          # streams compared per cell at R = 0; R = 13 runs 64 streams
ORACLE_STREAMS = 4


def tag(alpha):
    return f"a{alpha:g}".replace(".", "p")


def fit_readout(capi, gj, W):
    """ridge read-out of the phone off the frozen encoder (tests/golden/gen_speech_joint.py, fewer calibration streams)"""
    K = synth.N_PHONES
    eng = capi.Engine(W, n_layers=24, dtype=capi.DTYPE_F32, max_streams=N_CAL)
    H, Y = [], []
    for R in (0, 13):
        feats, evs = gj.run_features(eng, R, range(100, 100 + N_CAL), CAL_SECONDS)
        for e, ev in zip(feats, evs):
            lab = gj.frame_labels(ev, e.shape[0])
            keep = lab >= -1
            y = np.zeros((e.shape[0], K + 1), np.float64)
            y[np.arange(e.shape[0]), np.where(lab >= 0, lab, K)] = 1.0
            H.append(e[keep].astype(np.float64))
            Y.append(y[keep])
    eng.close()
    H, Y = np.concatenate(H), np.concatenate(Y)
    H1 = np.concatenate([H, np.ones((H.shape[0], 1))], axis=1)
    G = H1.T @ H1 + LAMBDA * np.eye(H1.shape[1])
    G[-1, -1] -= LAMBDA
    sol = np.linalg.solve(G, H1.T @ Y)
    acc = float(((H1 @ sol).argmax(axis=1) == Y.argmax(axis=1)).mean())
    return sol[:-1].T.astype(np.float32), sol[-1].astype(np.float32), acc


def phase_gpu(out: Path):
    from nemotron_asr_amd import capi
    sys.path.insert(0, str(ROOT / "tests" / "golden"))
    import gen_speech_joint as gj
    out.mkdir(parents=True, exist_ok=True)
    t0 = time.time()
    rep = {}
    for alpha in ALPHAS:
        W = synth.scale_residual_branches(synth.make_weights(24), alpha)
        w, b, train_acc = fit_readout(capi, gj, W)
        np.savez(out / f"readout_{tag(alpha)}.npz", w=w, b=b)
        # bf16 noise of the detector scores (target units), held-out streams
        noise = {}
        f32e = capi.Engine(W, n_layers=24, dtype=capi.DTYPE_F32, max_streams=8)
        b16e = capi.Engine(W, n_layers=24, dtype=capi.DTYPE_BF16, max_streams=8)
        for R in (0, 13):
            f32f, evs = gj.run_features(f32e, R, range(8), 20.0)
            b16f, _ = gj.run_features(b16e, R, range(8), 20.0)
            d = np.concatenate([(x16 - x32) @ w.T for x16, x32 in zip(b16f, f32f)])
            enc = np.concatenate([np.abs(x16 - x32).ravel() for x16, x32 in zip(b16f, f32f)])
            noise[f"R{R}"] = dict(score_noise_rms=float(np.sqrt((d ** 2).mean())), score_noise_max=float(np.abs(d).max()), encoder_err_mean=float(enc.mean()), encoder_err_max=float(enc.max()))
        f32e.close(); b16e.close()
        print(f"alpha {alpha}: fit (train accuracy {train_acc:.4f}) + noise done, {time.time() - t0:.0f} s", flush=True)
        for A in SCALES:
            synth.SPEECH_LOGIT_SCALE = A
            Ws = synth.apply_speech_decoder(W, readout=(w, b))
            cell = dict(alpha=alpha, logit_scale=A, readout_train_accuracy=train_acc, bf16_noise=noise)
            for R, B, shape in ((0, 1, "b1_R0"), (13, 64, "b64_R13")):
                streams = list(range(N_TOK)) if B == 1 else list(range(64))
                res = {}
                for name, dt in (("f32", capi.DTYPE_F32), ("bf16", capi.DTYPE_BF16)):
                    eng = capi.Engine(Ws, n_layers=24, dtype=dt, max_streams=B)
                    eng.set_option("pipeline", 4)
                    toks, frames, evs = [], [], []
                    for s0 in range(0, len(streams), B):
                        t, f, ev = gj.run_tokens(eng, R, streams[s0:s0 + B], TOK_SECONDS)
                        toks += t; frames += f; evs += list(ev)
                    eng.close()
                    res[name] = (toks, frames, evs)
                t32, f32_, evs = res["f32"]
                t16, f16, _ = res["bf16"]
                want = [[synth.phone_token(k) for k, _, _ in ev] for ev in evs]
                n_tok = sum(len(x) for x in t32)
                cell[shape] = dict(streams=len(streams), f32_tokens=n_tok, phones=sum(len(x) for x in want),
                                   f32_transcript_is_the_phone_sequence=int(sum(a == b_ for a, b_ in zip(t32, want))),
                                   streams_tokens_equal=int(sum(a == b_ for a, b_ in zip(t16, t32))),
                                   tokens_equal=bool(all(a == b_ for a, b_ in zip(t16, t32))),
                                   tokens_differing_streams=[i for i, (a, b_) in enumerate(zip(t16, t32)) if a != b_][:8],
                                   frame_shifts=int(sum(sum(x != y for x, y in zip(a, b_)) for a, b_ in zip(f16, f32_) if len(a) == len(b_))),
                                   bf16_logit_noise_rms=A * noise[f"R{R}"]["score_noise_rms"], bf16_logit_noise_max=A * noise[f"R{R}"]["score_noise_max"])
            rep[f"{tag(alpha)}_A{A:g}"] = cell
            print(json.dumps(cell), f"{time.time() - t0:.0f} s", flush=True)
            (out / ("gpu.json" if not os.environ.get("MARGIN_ALPHAS") else f"gpu_{tag(ALPHAS[0])}.json")).write_text(json.dumps(rep, indent=1))
    synth.SPEECH_LOGIT_SCALE = 8.0


def phase_cpu(out: Path):
    from oracle import binding as ob
    rep = {}
    t0 = time.time()
    cells = [(a, 8.0) for a in ALPHAS] + ([(0.3, 4.0)] if 0.3 in ALPHAS else [])
    for alpha, A in cells:
        z = np.load(out / f"readout_{tag(alpha)}.npz")
        synth.SPEECH_LOGIT_SCALE = A
        W = synth.apply_speech_decoder(synth.scale_residual_branches(synth.make_weights(24), alpha), readout=(z["w"], z["b"]))
        om = ob.OracleModel(W, 24)
        row = {}
        for R in (0, 13):
            margins, n_tok = [], 0
            for s in range(ORACLE_STREAMS):
                pcm, _ = synth.make_speech_pcm(s, TOK_SECONDS)
                ost = ob.OracleStream(om, R)
                ost.enable_decision_log()
                n_tok += len(ost.process(pcm) + ost.finalize())
                margins.append(np.asarray(ost.decision_log()["margin"], np.float64))
            m = np.concatenate(margins)
            row[f"R{R}"] = dict(decisions=int(m.size), tokens=n_tok, margin_pct_0p1_1_5_50=[round(float(x), 4) for x in np.percentile(m, [0.1, 1, 5, 50])],
                                share_ge_0p5=round(float((m >= 0.5).mean()), 4), share_lt_0p05=round(float((m < 0.05).mean()), 5))
        del om
        rep[f"{tag(alpha)}_A{A:g}"] = row
        print(alpha, A, json.dumps(row), f"{time.time() - t0:.0f} s", flush=True)
        (out / ("cpu.json" if not os.environ.get("MARGIN_ALPHAS") else f"cpu_{tag(ALPHAS[0])}.json")).write_text(json.dumps(rep, indent=1))
    synth.SPEECH_LOGIT_SCALE = 8.0


def phase_merge(out: Path, dest: Path):
    gpu, cpu = {}, {}
    for f in sorted(out.glob("gpu*.json")):
        gpu.update(json.loads(f.read_text()))
    for f in sorted(out.glob("cpu*.json")):
        cpu.update(json.loads(f.read_text()))
    gpu = dict(sorted(gpu.items(), key=lambda kv: (kv[1]["alpha"], -kv[1]["logit_scale"])))
    cells = {}
    for key, cell in gpu.items():
        a_tag, A = key.split("_A")
        base = cpu.get(f"{a_tag}_A8")
        if base:
            for R, shape in ((0, "b1_R0"), (13, "b64_R13")):
                o = base[f"R{R}"]
                k = float(A) / 8.0          # every logit that takes part in a decision is proportional to the scale
                cell[shape]["oracle_margin_1st_percentile"] = round(o["margin_pct_0p1_1_5_50"][1] * k, 4)
                cell[shape]["oracle_margin_0p1_percentile"] = round(o["margin_pct_0p1_1_5_50"][0] * k, 4)
                cell[shape]["oracle_margins_from"] = "A = 8 oracle run, scaled by A / 8" if float(A) != 8.0 else "oracle run"
                direct = cpu.get(key)
                if direct and float(A) != 8.0:
                    cell[shape]["oracle_margin_1st_percentile_direct"] = direct[f"R{R}"]["margin_pct_0p1_1_5_50"][1]
        cells[key] = cell
    dest.write_text(json.dumps(dict(
        what="bf16 engine tokens vs F32 engine tokens on the speech checkpoint with harder residual scales / smaller logit scales (tests/micro/margin_sweep.py)",
        streams_per_cell={"b1_R0": N_TOK, "b64_R13": 64}, seconds_per_stream=TOK_SECONDS, oracle_streams_for_margins=ORACLE_STREAMS, cells=cells, oracle=cpu), indent=1))
    for key, cell in cells.items():
        print(key, {s: (cell[s]["tokens_equal"], cell[s]["streams_tokens_equal"], cell[s]["frame_shifts"], cell[s].get("oracle_margin_1st_percentile"), round(cell[s]["bf16_logit_noise_max"], 4))
                    for s in ("b1_R0", "b64_R13")})


if __name__ == "__main__":
    phase, out = sys.argv[1], Path(sys.argv[2])
    if phase == "gpu":
        phase_gpu(out)
    elif phase == "cpu":
        phase_cpu(out)
    else:
        phase_merge(out, Path(sys.argv[3]))

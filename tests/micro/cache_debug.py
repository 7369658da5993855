import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge
ge.load_package()
import numpy as np
from nemotron_asr_amd import capi, synth
L, R, T = 2, 13, 14
W2 = synth.make_weights(n_layers=L)
n = synth.shift_samples(R)
for B, pipe, debug in ((64, 0, True), (64, 4, False), (40, 0, True), (128, 0, True), (65, 0, True)):
    pcms = [synth.make_pcm(400 + b, 3 * n / 16000 + 0.01)[:3 * n] for b in range(B)]
    eng = capi.Engine(W2, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
    eng.set_option("pipeline", pipe)
    if debug:
        eng.set_debug(True)
    sts = [eng.stream(R) for _ in range(B)]
    for k in range(3):
        eng.step(sts, [p[k * n:(k + 1) * n] for p in pcms])
        if k >= 1:
            bad = []
            for b in range(B):
                if debug:
                    s_ = sts[b].tap(capi.TAP_SUBSAMPLED)
                    l0 = sts[b].tap(capi.TAP_LAYER_OUT, 0)
                    l1 = sts[b].tap(capi.TAP_LAYER_OUT, 1)
                    if not (np.isfinite(s_).all() and np.isfinite(l0).all() and np.isfinite(l1).all()):
                        bad.append((b, bool(np.isfinite(s_).all()), bool(np.isfinite(l0).all()), bool(np.isfinite(l1).all()), int((~np.isfinite(l0.reshape(T, 1024))).any(1).sum())))
                else:
                    e_ = sts[b].tap(capi.TAP_ENCODER_OUT)
                    if not np.isfinite(e_).all():
                        bad.append((b, int((~np.isfinite(e_.reshape(-1, 1024)[:T])).any(1).sum())))
            print(f"B={B} pipe={pipe} step {k}: streams with non-finite taps {len(bad)} {bad[:6]}", flush=True)
    eng.close()

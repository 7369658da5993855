"""Encoder output of the large-M path under the engine options that must not change a bit of it (64 streams x R = 13 by default).
    python tests/micro/path_identity_debug.py [B]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge
ge.load_package()
import numpy as np
from nemotron_asr_amd import capi, synth

L, R, T = 2, 13, 14
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = synth.make_weights(n_layers=L)
n = synth.shift_samples(R)
pcms = [synth.make_pcm(400 + b, 4 * n / 16000 + 0.01)[:4 * n] for b in range(B)]
ref = None
for name, opts, debug, pipe in (("old", dict(resid_epilogue=0), False, 0), ("old-debug", dict(resid_epilogue=0), True, 0),
                                ("resid-always", dict(resid_epilogue=2), False, 0), ("default", dict(), False, 0), ("default-debug", dict(), True, 0),
                                ("default-pipe", dict(), False, 4), ("old-pipe", dict(resid_epilogue=0), False, 4), ("chain-pipe", dict(chain=1), False, 4)):
    eng = capi.Engine(W, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.set_option("pipeline", pipe)
    if debug:
        eng.set_debug(True)
    sts = [eng.stream(R) for _ in range(B)]
    outs = []
    for k in range(4):
        eng.step(sts, [p[k * n:(k + 1) * n] for p in pcms])
        if k >= 1:
            outs.append(np.stack([sts[b].tap(capi.TAP_ENCODER_OUT).reshape(-1, 1024)[:T] for b in (0, B // 2, B - 1)]))
    outs = np.stack(outs)
    if ref is None:
        ref = outs
    d = np.abs(outs - ref)
    print(f"{name:12s} finite={np.isfinite(outs).all()} max|x|={np.abs(outs).max():.3f} diff vs old: max {d.max():.4f} mean {d.mean():.5f} per step {[round(float(x), 4) for x in d.reshape(3, -1).max(1)]}", flush=True)
    eng.close()

"""Emission statistics of the synthetic decoder (GPU): tokens per stream per step at batch B, right context R,
for a sweep of (EMBED_GAIN, SUPPRESS, blank gain) -- how synth.py's decoder constants were chosen."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import __graft_entry__ as ge
ge.load_package()
from nemotron_asr_amd import capi, synth

B, R, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
configs = [tuple(float(v) for v in a.split(",")) for a in sys.argv[4:]] or [(synth.EMBED_GAIN, synth.SUPPRESS, synth.BLANK_GAIN)]
W = synth.make_weights(n_layers=24)
n = synth.shift_samples(R)
pcm = [synth.make_pcm(s, steps * n / 16000.0)[:steps * n] for s in range(B)]
for lg, pg, bg in configs:
    synth.EMBED_GAIN, synth.SUPPRESS = lg, pg
    W.update(synth.make_weights(n_layers=24, layers=[], blank_bias=bg))
    eng = capi.Engine(W, n_layers=24, dtype=capi.DTYPE_BF16, max_streams=B)
    sts = [eng.stream(R) for _ in range(B)]
    per = np.zeros((steps, B), int)
    for k in range(steps):
        out = eng.step(sts, [p[k * n:(k + 1) * n] for p in pcm])
        per[k] = [len(t) for t in out]
    it = sum(s.stats().decode_iterations for s in sts)
    frames = steps * B * (1 + R)
    tot = per.sum(0)
    print(f"embed {lg} suppress {pg} blank {bg}: tokens/frame {per.sum() / frames:.3f}  evaluations/frame {it / frames:.3f}  "
          f"per-stream tokens min/median/max {tot.min()}/{int(np.median(tot))}/{tot.max()}  "
          f"max tokens of one stream in one step {per.max()} (frames/step {1 + R})  silent streams {(tot == 0).sum()}  "
          f"mean over steps of (busiest stream's tokens) {per.max(1).mean():.1f}  stream-steps with >= 20 tokens {(per >= 20).mean() * 100:.2f}%", flush=True)
    eng.close()

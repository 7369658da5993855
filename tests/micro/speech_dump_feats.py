import sys, json, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]; sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge; ge.load_package()
from nemotron_asr_amd import capi, synth
sys.path.insert(0, str(ROOT / "tests" / "golden"))
import gen_speech_joint as g
alpha = float(sys.argv[1]); out = Path(sys.argv[2]); out.mkdir(parents=True, exist_ok=True)
W = synth.scale_residual_branches(synth.make_weights(24), alpha)
eng = capi.Engine(W, n_layers=24, dtype=capi.DTYPE_F32, max_streams=40)
b16 = capi.Engine(W, n_layers=24, dtype=capi.DTYPE_BF16, max_streams=8)
d = {}
for R in (0,):
    f, ev = g.run_features(eng, R, range(100, 132), 30.0)
    d[f"cal_R{R}"] = np.stack(f).astype(np.float16)
    f, ev = g.run_features(eng, R, range(8), 30.0)
    d[f"val_R{R}"] = np.stack(f).astype(np.float32)
    f, ev = g.run_features(b16, R, range(8), 30.0)
    d[f"val16_R{R}"] = np.stack(f).astype(np.float32)
np.savez_compressed(out / f"feats_a{alpha}.npz", **d)
print({k: v.shape for k, v in d.items()})

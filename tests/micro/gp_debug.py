import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import __graft_entry__ as ge
ge.load_package()
from nemotron_asr_amd import capi, synth
L = 8
W = synth.make_weights(n_layers=L)
pcms = [synth.make_pcm(70 + b, 6.0) for b in range(3)]
n = 1280
def run(mode, use_alt, use_ragged, use_c, use_stats):
    eng = capi.Engine(W, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=3)
    eng.set_option("pipeline", mode)
    a, b = eng.stream(0), eng.stream(0)
    c = eng.stream(1)
    toks = [[], [], []]
    for k in range(60):
        if use_alt and k % 7 == 3:
            toks[0] += eng.step([a], [pcms[0][k * n:(k + 1) * n]])[0]
            toks[1] += eng.step([b], [pcms[1][k * n:(k + 1) * n]])[0]
        elif use_ragged and k == 31:
            for sl in (slice(k * n, k * n + 500), slice(k * n + 500, (k + 1) * n)):
                out = eng.step([a, b], [pcms[0][sl], pcms[1][sl]])
                toks[0] += out[0]; toks[1] += out[1]
        else:
            out = eng.step([a, b], [pcms[0][k * n:(k + 1) * n], pcms[1][k * n:(k + 1) * n]])
            toks[0] += out[0]; toks[1] += out[1]
        if use_c and k % 9 == 4:
            toks[2] += eng.step([c], [pcms[2][(k // 9) * 2560:(k // 9 + 1) * 2560]])[0]
        if use_stats and k == 20:
            a.stats()
    out = eng.finalize([a, b]) + eng.finalize([c])
    for i in range(3): toks[i] += out[i]
    fr = [s.token_frames() for s in (a, b, c)]
    eng.close()
    return toks, fr
for flags in ((1,1,1,1), (0,1,1,1), (1,0,1,1), (1,1,0,1), (1,1,1,0), (0,0,1,0)):
    t0, f0 = run(0, *flags); t8, f8 = run(8, *flags)
    print(flags, "tokens equal", t0 == t8, "frames equal", f0 == f8, [len(x) for x in t0], [len(x) for x in t8], flush=True)
    if t0 != t8:
        for i in range(3):
            if t0[i] != t8[i]: print("  stream", i, t0[i], t8[i], f0[i], f8[i])

"""One small invocation of the hot path on cuda:0, checked against the CPU oracle.

The oracle is the checker only (see oracle/nasr_oracle.h); the thing exercised is the HIP
engine through the C ABI."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

from . import capi, synth

ROOT = Path(__file__).resolve().parent.parent


def run(verbose=True):
    if str(ROOT) not in sys.path:
        sys.path.insert(0, str(ROOT))
    from oracle import binding as ob

    n_layers = 2
    W = synth.make_weights(n_layers=n_layers)
    pcm = synth.make_pcm(2, 4.0)
    model = ob.OracleModel(W, n_layers, emulate_bf16=True)
    ost = ob.OracleStream(model, 0)
    ref_tokens = ost.process(pcm) + ost.finalize()

    eng = capi.Engine(W, n_layers=n_layers, dtype=capi.DTYPE_BF16, max_streams=2)
    eng.set_debug(True)
    st = eng.stream(0)
    toks = []
    for o in range(0, pcm.size, 1280):
        toks += eng.step([st], [pcm[o:o + 1280]])[0]
    toks += eng.finalize([st])[0]
    stats = st.stats()
    assert stats.chunks == ost.total_chunks, (stats.chunks, ost.total_chunks)
    enc = st.tap(capi.TAP_ENCODER_OUT).reshape(-1, 1024)
    assert np.isfinite(enc).all()
    n = max(len(toks), len(ref_tokens))
    agree = 1.0 if n == 0 else sum(a == b for a, b in zip(toks, ref_tokens)) / n
    if verbose:
        print(f"smoke: {stats.chunks} chunks, {len(toks)} tokens (oracle {len(ref_tokens)}), agreement {agree:.3f}")
    assert len(ref_tokens) > 0, "smoke workload should emit tokens"
    assert agree >= 0.8, (toks, ref_tokens)
    st.destroy()
    eng.close()
    return True

"""last-layer activation error of the f32 / bf16 / Q8_0->bf16 engines against the F32 oracle (and the
bf16-emulating one), 2 and 24 layers, both synthetic checkpoints.  Run on a GPU box: python tests/micro/tolerance_table.py out.json"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from nemotron_asr_amd import capi, synth  # noqa: E402
from oracle import binding as ob  # noqa: E402


def run(L, ckpt, R=13, n_chunks=6):
    W = synth.make_weights(L, margins=ckpt) if L == 24 else synth.make_weights(L)
    if L != 24 and ckpt == "speech":
        synth.scale_residual_branches(W, synth.SPEECH_RESIDUAL_SCALE)
    engQ, deq = synth.quantize_weights(W, "q8_0")
    T, n = 1 + R, synth.shift_samples(R)
    pcm = (synth.make_speech_pcm(3, n_chunks * n / 16000 + 1.0)[0] if ckpt == "speech" else synth.make_pcm(3, n_chunks * n / 16000 + 1.0))[:(n_chunks + 1) * n]
    refs = {}
    for name, weights, kw in (("f32", W, {}), ("bf16emu", W, dict(emulate_bf16=True)), ("f32_q8w", deq, {}), ("bf16emu_q8w", deq, dict(emulate_bf16=True))):
        om = ob.OracleModel(weights, L, **kw)
        ost = ob.OracleStream(om, R)
        tap = ost.enable_taps()
        outs = []
        for k in range(n_chunks + 1):
            c0 = ost.total_chunks
            ost.process(pcm[k * n:(k + 1) * n])
            if ost.total_chunks > c0:
                outs.append(tap[1][L - 1].copy())
        refs[name] = np.stack(outs)
        del om
    res = {}
    for name, weights, dt, ref32, refemu in (("f32_engine", W, capi.DTYPE_F32, "f32", None), ("bf16_engine", W, capi.DTYPE_BF16, "f32", "bf16emu"),
                                             ("q8_0_engine", engQ, capi.DTYPE_BF16, "f32_q8w", "bf16emu_q8w")):
        eng = capi.Engine(weights, n_layers=L, dtype=dt, max_streams=1)
        eng.set_debug(True)
        st = eng.stream(R)
        outs = []
        for k in range(n_chunks + 1):
            c0 = st.progress().chunks
            eng.step([st], [pcm[k * n:(k + 1) * n]])
            if st.progress().chunks > c0:
                outs.append(st.tap(capi.TAP_LAYER_OUT, L - 1).reshape(T, 1024).copy())
        got = np.stack(outs)
        d = np.abs(got - refs[ref32])
        res[name] = dict(vs_f32_oracle=dict(max=float(d.max()), mean=float(d.mean())))
        if refemu:
            d2 = np.abs(got - refs[refemu])
            res[name]["vs_bf16_emulating_oracle"] = dict(max=float(d2.max()), mean=float(d2.mean()))
        eng.close()
    return res


out = {}
for L in (2, 24):
    for ckpt in ("random", "speech"):
        out[f"{L}_layers_{ckpt}"] = run(L, ckpt)
        print(L, ckpt, json.dumps(out[f"{L}_layers_{ckpt}"]), flush=True)
Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tolerance_table.json").write_text(json.dumps(out, indent=1))

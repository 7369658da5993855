"""Generate tests/golden/golden_v1.npz from the UNMODIFIED reference sources.

Run in the build container only (needs /root/reference):
    python tests/golden/gen_golden.py
It compiles oracle/_ref/libnemo_ref.so (reference src/preprocessor.cpp +
src/reference/*.cpp, see oracle/Makefile), feeds it the seeded synthetic weights and
inputs defined in nemotron-asr.cpp_amd/synth.py and tests/golden/inputs.py, and stores
only the OUTPUTS.  Inputs are regenerated from the seed by every consumer.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from nemotron_asr_amd import synth  # noqa: E402
from oracle import binding as ob  # noqa: E402
from tests.golden import inputs as gi  # noqa: E402


def main():
    assert ob.have_ref(), "needs /root/reference to build oracle/_ref"
    W = synth.make_weights(n_layers=1)
    out = {}
    # a-1: mel of 1 s of PCM fed in irregular pieces (pins carry-over state)
    pcm = gi.pcm()
    mel = ob.ref_preproc(W["preprocessor.featurizer.fb"], W["preprocessor.featurizer.window"], gi.split_pcm(pcm))
    out["mel"] = mel
    # a-2: subsampling of a 17- and a 121-frame chunk (the R=0 / R=13 graph widths)
    out["sub_17"] = ob.ref_subsampling(W, gi.mel_chunk(mel, 17))
    out["sub_121"] = ob.ref_subsampling(W, gi.mel_chunk(mel, 121))
    # a-3..a-9: one ConformerLayer (offline == streaming chunk 0), T = 1, 2, 14, 16
    for T in (1, 2, 14, 16):
        out[f"layer_T{T}"] = ob.ref_conformer_layer(W, 0, gi.layer_input(T))
    # a-7: positional table
    out["pos_emb_5"] = ob.ref_pos_emb(5)
    # a-6: rel_shift on the reference's own closed-form test input, in[h][i][p] = 100 h + 10 i + p, heads 2, qlen 4
    # (tests/test_compute.cpp:1028-1052), plus qlen 14 (the R = 13 chunk length)
    for q in (4, 14):
        out[f"rel_shift_q{q}"] = ob.ref_rel_shift(gi.rel_shift_input(2, q))
    # a-7 again at the chunk widths the streaming slice is cut from (src/nemo-stream.cpp:168-177): KV = 71 and 84
    out["pos_emb_84"] = ob.ref_pos_emb(84)
    # a-12/13: decoder+joint logits for the reference's own test sequence
    # (tests/test_compute.cpp:2407: {1024, 0, 100, 500})
    enc = gi.enc_frames(64)
    lg, h, c = ob.ref_decoder_joint_seq(W, gi.DEC_TOKENS, enc[:4])
    out["dec_logits"], out["dec_h"], out["dec_c"] = lg, h, c
    # a-14: greedy tokens
    out["greedy_tokens"] = np.asarray(ob.ref_greedy(W, enc), np.int32)
    np.savez_compressed(ROOT / "tests" / "golden" / "golden_v1.npz", **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)
    print("greedy tokens:", out["greedy_tokens"].size, "over 64 frames")


if __name__ == "__main__":
    main()

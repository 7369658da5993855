"""Collect the NeMo-generated diarization fixtures the reference's own tests hold into one small npz.

Run in the build container only (needs /root/reference):
    python tests/golden/gen_nemo_diar_fixtures.py
These are DATA files (inputs and the outputs NeMo produced for them), committed by the reference under
tests/diarize/ for tests/test_diarize_preproc.cpp and tests/test_diarize_cluster.cpp; no source is copied.
  vad_ref/input_audio.f32 -> vad_ref/mel.f32   80-mel log-mel of an 83 200-sample buffer, no normalisation  [80][528]
  spk_ref/input_audio.f32 -> spk_ref/mel.f32   the same front end with per-feature normalisation (1.5 s)   [80][160]
  cluster_ref/{embeddings,affinity,labels}     NME-SC clustering: 60 embeddings x 192 -> cosine affinity -> labels
Everything else under tests/diarize/ (block outputs, logits, embeddings of the real networks) needs the
diarize.gguf weights, which are not in the tree.
"""
import json
from pathlib import Path

import numpy as np

REF = Path("/root/reference/tests/diarize")
OUT = Path(__file__).resolve().parent / "nemo_diar_v1.npz"


def main():
    out = {}
    for d in ("vad_ref", "spk_ref"):
        audio = np.fromfile(REF / d / "input_audio.f32", dtype=np.float32)
        mel = np.fromfile(REF / d / "mel.f32", dtype=np.float32)
        assert mel.size % 80 == 0
        out[f"{d}_audio"] = audio
        out[f"{d}_mel"] = mel.reshape(80, mel.size // 80)
    meta = json.loads((REF / "cluster_ref" / "cluster_meta.json").read_text())
    N, D = meta["N"], meta["D"]
    out["cluster_embeddings"] = np.fromfile(REF / "cluster_ref" / "embeddings.f32", dtype=np.float32).reshape(N, D)
    out["cluster_affinity"] = np.fromfile(REF / "cluster_ref" / "affinity.f32", dtype=np.float32).reshape(N, N)
    out["cluster_labels"] = np.fromfile(REF / "cluster_ref" / "labels.i32", dtype=np.int32)
    out["cluster_meta"] = np.array([meta["est_num_spk"], meta["p_hat"], meta["max_num_speakers"], meta["sparse_search_volume"],
                                    meta["nme_mat_size"]], np.int32)
    out["cluster_max_rp_threshold"] = np.float32(meta["max_rp_threshold"])
    np.savez_compressed(OUT, **out)
    for k, v in out.items():
        print(k, np.asarray(v).shape, np.asarray(v).dtype)
    print(OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()

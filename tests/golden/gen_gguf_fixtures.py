#!/usr/bin/env python3
"""Golden fixtures for the GGUF side of the path, produced by the REFERENCE's own converter.

Runs only in the build container (needs /root/reference): it imports
`/root/reference/scripts/convert_to_gguf.py` as a module and stores OUTPUTS only:

  * `quantize_q8_0` / `quantize_q4_0` bytes (reference scripts/convert_to_gguf.py:118-204) for seeded arrays that
    cover the edge cases: all-zero blocks, values on rounding ties (np.round = half to even), a single non-zero
    element, a tail that is not a multiple of 32 (zero padded by the converter), tiny blocks whose fp16 scale is
    subnormal, large values;
  * complete GGUF v3 files written by `convert_to_gguf()` itself (:309-540: header, KV section incl. the string-array
    vocabulary, the legacy 8-byte vocabulary blob and the prompt dictionary, tensor infos, 32-byte alignment, data)
    from a tiny synthetic `.nemo` archive (model_config.yaml + model_weights.ckpt) built here: F32, F16, Q8_0 and
    Q4_0 flavours of an English-style checkpoint (short tokens -> legacy blob present) and a multilingual-style one
    (long tokens -> blob omitted, prompt dictionary present).

Nothing of the converter's text is copied; the fixture is data (inputs + the bytes it produced).

    python tests/golden/gen_gguf_fixtures.py        # writes tests/golden/gguf_ref_v1.npz
"""
from __future__ import annotations

import importlib.util
import io
import json
import sys
import tarfile
import tempfile
from contextlib import redirect_stdout
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent / "gguf_ref_v1.npz"


def load_converter():
    spec = importlib.util.spec_from_file_location("ref_convert_to_gguf", REF / "scripts" / "convert_to_gguf.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def quant_inputs():
    """name -> float32 array (1-D); every case a few blocks long"""
    rng = np.random.default_rng(0x6A6F)
    cases = {}
    cases["normal"] = (rng.standard_normal(32 * 8) * 0.03).astype(np.float32)               # weight-like
    z = (rng.standard_normal(32 * 4)).astype(np.float32)
    z[32:64] = 0.0                                                                           # an all-zero block
    cases["zero_block"] = z
    # amax = 127 -> scale exactly 1.0 (Q8_0): x.5 values are rounding ties (half to even), -127.5 would wrap
    t = np.zeros(64, np.float32)
    t[:32] = np.array([127.0, 0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 63.5, 64.5, -126.5, 126.5, 3.49999, 3.50001] + [0.25] * 19, np.float32)
    # amax = 7 -> scale exactly 1.0 (Q4_0): ties and the asymmetric [-8, 7] clip
    t[32:] = np.array([7.0, -7.0, 0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 6.5, -6.5, 5.5, -5.5, 6.49, -6.51] + [0.75] * 18, np.float32)
    cases["ties"] = t
    one = np.zeros(96, np.float32)
    one[5], one[40], one[95] = 3.0, -1e-3, 1e4                                               # one non-zero element per block
    cases["single_nonzero"] = one
    cases["tail_70"] = (rng.standard_normal(70) * 0.5).astype(np.float32)                    # not a multiple of 32: padded
    tiny = (rng.standard_normal(64) * 1e-6).astype(np.float32)                               # fp16 scale is subnormal / zero
    tiny[32:] *= 1e-3
    cases["tiny"] = tiny
    cases["large"] = (rng.standard_normal(64) * 3e3).astype(np.float32)
    neg = -np.abs(rng.standard_normal(32)).astype(np.float32)                                # maximum magnitude on the negative side
    cases["negative_max"] = neg
    return cases


def vocab_en(n=64):
    # SentencePiece-like, every token <= 7 bytes in UTF-8 so that the legacy blob is written
    return [("▁" if i % 3 == 0 else "") + np.base_repr(i, 36).lower() for i in range(n)]


def vocab_ml(n=48):
    v = [("▁" if i % 4 == 0 else "") + "tok" + np.base_repr(i, 36).lower() for i in range(n)]
    v[7] = "▁verylongmultilingualtoken"      # > 7 bytes: the converter drops the legacy blob
    v[9] = "éè中文"              # multi-byte UTF-8
    return v


def model_config(vocab, num_prompts, prompt_dict):
    cfg = {
        "encoder": {"d_model": 1024, "n_heads": 8, "feat_in": 128, "ff_expansion_factor": 4, "n_layers": 2,
                    "conv_kernel_size": 9, "subsampling_factor": 8,
                    "att_context_size": [[70, 13], [70, 6], [70, 1], [70, 0]]},
        "decoder": {"prednet": {"pred_hidden": 640}},
        "joint": {"num_classes": len(vocab), "vocabulary": vocab, "jointnet": {"joint_hidden": 640}},
    }
    if num_prompts:
        cfg["num_prompts"] = num_prompts
        cfg["model_defaults"] = {"prompt_dictionary": prompt_dict}
    return cfg


def tiny_tensors():
    """PyTorch-side names and layouts (what a .nemo checkpoint holds), small shapes.  Which of them the converter
    quantises follows from its own rules (pattern :246-263, >= 256 elements, >= 2 dims, never the depthwise conv)."""
    rng = np.random.default_rng(0x6775)
    f = lambda *s: (rng.standard_normal(s) * 0.05).astype(np.float32)
    return {
        "encoder.pre_encode.out.weight": f(6, 40),                               # not an encoder layer: F32
        "encoder.layers.0.feed_forward1.linear1.weight": f(16, 64),              # quantised
        "encoder.layers.0.feed_forward1.linear2.weight": f(8, 32),               # 256 elements: quantised (>= 256)
        "encoder.layers.0.self_attn.linear_q.weight": f(4, 32),                  # 128 elements: stays F32
        "encoder.layers.0.self_attn.linear_pos.weight": f(8, 96),                # quantised
        "encoder.layers.0.self_attn.pos_bias_u": f(8, 128),                      # not *.weight: F32
        "encoder.layers.0.conv.pointwise_conv1.weight": f(16, 32, 1),            # squeezed to (16, 32), quantised
        "encoder.layers.0.conv.depthwise_conv.weight": f(32, 1, 9),              # -> (9, 32), never quantised
        "encoder.layers.0.conv.batch_norm.weight": f(300),                       # matches the pattern but 1-D: F32
        "encoder.layers.0.norm_out.weight": f(16),
        "encoder.layers.1.feed_forward2.linear1.weight": f(12, 64),              # quantised; 3072 B of f32 before it? alignment padding
        "decoder.prediction.embed.weight": f(5, 8),
        "joint.joint_net.2.bias": f(7),                                          # 28 bytes: the next tensor needs padding
        "joint.joint_net.2.weight": f(7, 8),
    }


def make_nemo(path: Path, cfg: dict, tensors: dict):
    import torch
    import yaml
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        (td / "model_config.yaml").write_text(yaml.safe_dump(cfg, allow_unicode=True))
        torch.save({k: torch.from_numpy(v) for k, v in tensors.items()}, td / "model_weights.ckpt")
        with tarfile.open(path, "w") as tar:
            tar.add(td / "model_config.yaml", arcname="./model_config.yaml")
            tar.add(td / "model_weights.ckpt", arcname="./model_weights.ckpt")


def main():
    if not (REF / "scripts" / "convert_to_gguf.py").exists():
        sys.exit("needs /root/reference (build container only)")
    conv = load_converter()
    out = {}
    meta = {"quant_cases": [], "files": {}}
    for name, x in quant_inputs().items():
        out[f"qin.{name}"] = x
        out[f"q8.{name}"] = np.frombuffer(conv.quantize_q8_0(x), np.uint8)
        out[f"q4.{name}"] = np.frombuffer(conv.quantize_q4_0(x), np.uint8)
        meta["quant_cases"].append(name)
    tensors = tiny_tensors()
    for k, v in tensors.items():
        out[f"src.{k}"] = v
    prompt_dict = {"en-US": 0, "de-DE": 2, "auto": 3, "fr-FR": 1}
    variants = {
        "en_f32": (vocab_en(), 0, None, None),
        "en_f16": (vocab_en(), 0, None, "f16"),
        "en_q8_0": (vocab_en(), 0, None, "q8_0"),
        "en_q4_0": (vocab_en(), 0, None, "q4_0"),
        "ml_q8_0": (vocab_ml(), 4, prompt_dict, "q8_0"),
    }
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        for tag, (vocab, n_prompts, pd, quant) in variants.items():
            nemo = td / f"tiny-{tag.split('_')[0]}.nemo"
            make_nemo(nemo, model_config(vocab, n_prompts, pd), tensors)
            gg = td / f"{tag}.gguf"
            with redirect_stdout(io.StringIO()):
                conv.convert_to_gguf(str(nemo), str(gg), quant_type=quant)
            out[f"file.{tag}"] = np.frombuffer(gg.read_bytes(), np.uint8)
            meta["files"][tag] = {"vocab": vocab, "num_prompts": n_prompts, "prompt_dict": pd, "quant": quant,
                                  "model_name": nemo.stem}
    out["meta"] = np.frombuffer(json.dumps(meta, ensure_ascii=False).encode(), np.uint8)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes, {len(out)} arrays)")


if __name__ == "__main__":
    main()

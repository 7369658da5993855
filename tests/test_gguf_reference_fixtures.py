"""Row f-2 pinned to the reference: tests/golden/gguf_ref_v1.npz holds bytes produced by the reference's own converter
(`scripts/convert_to_gguf.py`, imported by tests/golden/gen_gguf_fixtures.py in the build container): Q8_0 / Q4_0 blocks
for edge-case arrays and complete GGUF v3 files.  Checked here, on the CPU:

  * `synth.pack_q8_0` / `pack_q4_0` (what the synthetic GGUFs of the GPU tests and of `bench.py --weights q8_0` are made
    of) are byte-identical to the converter's quantisers, ties / zero blocks / ragged tails / subnormal scales included;
  * `gguf_io.write_gguf` reproduces the converter's files byte for byte, `synth.quantize_weights` picks the tensors the
    converter picks;
  * the C++ reader (`host/gguf_reader.cpp`) parses the converter's files: KV values, vocabulary (string array and legacy
    8-byte blob), prompt dictionary, tensor table, data bytes;
  * the engine's upload-time dequantisation (`nasr_tensor_to_f32`, host code of nasr_engine.hip) of the converter's blocks
    equals d * q computed here from the raw bytes.
"""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

from nemotron_asr_amd import capi, gguf_io, synth

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "nemotron-asr.cpp_amd" / "bin"


@pytest.fixture(scope="module")
def fx():
    z = np.load(ROOT / "tests" / "golden" / "gguf_ref_v1.npz")
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def _fnv(data: bytes) -> str:
    h = 0
    for b in data:
        h = (h * 1099511628211 + b) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def test_packers_are_byte_identical_to_the_reference_quantisers(fx):
    z, meta = fx
    assert set(meta["quant_cases"]) >= {"normal", "zero_block", "ties", "single_nonzero", "tail_70", "tiny", "large", "negative_max"}
    for name in meta["quant_cases"]:
        x = z[f"qin.{name}"]
        assert synth.pack_q8_0(x).tobytes() == z[f"q8.{name}"].tobytes(), name
        assert synth.pack_q4_0(x).tobytes() == z[f"q4.{name}"].tobytes(), name
    # the cases do what they are named for
    q8 = z["q8.ties"].view(np.dtype([("d", np.float16), ("q", np.int8, 32)]))
    assert float(q8["d"][0]) == 1.0 and q8["q"][0][:11].tolist() == [127, 0, 2, 2, 0, -2, -2, 64, 64, -126, 126]   # half to even
    q4 = z["q4.ties"].view(np.dtype([("d", np.float16), ("q", np.uint8, 16)]))
    assert float(q4["d"][1]) == 1.0
    assert z["q8.zero_block"].view(np.dtype([("d", np.float16), ("q", np.int8, 32)]))[1].tobytes() == bytes(34)
    assert z["q8.tail_70"].size == 3 * 34 and z["q4.tail_70"].size == 3 * 18


def _dequant_expected(ty, raw, n):
    if ty == synth.TYPE_F16:
        return raw.view(np.float16).astype(np.float32)[:n]
    if ty == synth.TYPE_Q8_0:
        blk = raw.view(np.dtype([("d", np.float16), ("q", np.int8, 32)]))
        return (blk["d"].astype(np.float32)[:, None] * blk["q"].astype(np.float32)).reshape(-1)[:n]
    blk = raw.view(np.dtype([("d", np.float16), ("q", np.uint8, 16)]))
    q = np.concatenate([(blk["q"] & 0xF).astype(np.int32) - 8, (blk["q"] >> 4).astype(np.int32) - 8], axis=1)
    return (blk["d"].astype(np.float32)[:, None] * q.astype(np.float32)).reshape(-1)[:n]


def test_engine_dequantisation_of_reference_blocks(fx):
    """nasr_tensor_to_f32 = the conversion nasr_engine_create applies at upload (no GPU involved)."""
    z, meta = fx
    for name in meta["quant_cases"]:
        x = z[f"qin.{name}"]
        n = x.size // 32 * 32            # whole blocks (the converter never hands the engine a ragged tensor: >= 2-D matrices)
        if n == 0:
            continue
        for ty, key in ((synth.TYPE_Q8_0, "q8"), (synth.TYPE_Q4_0, "q4")):
            raw = z[f"{key}.{name}"][: n // 32 * (34 if key == "q8" else 18)]
            got = capi.tensor_to_f32(ty, raw, (n,))
            want = _dequant_expected(ty, raw, n)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, key)     # bit-exact, subnormals and -0 included
            amax = np.abs(x[:n].reshape(-1, 32)).max(axis=1, keepdims=True)
            step = np.maximum(amax, 6.2e-5 * (127 if key == "q8" else 7)) / (127.0 if key == "q8" else 7.0)     # fp16 scales below 2^-14 are subnormal
            assert np.all(np.abs(got.reshape(-1, 32) - x[:n].reshape(-1, 32)) <= step * (0.52 if key == "q8" else 1.02) + 1e-12), (name, key)
    h = (np.arange(64, dtype=np.float32) / 7 - 3).astype(np.float16)
    assert np.array_equal(capi.tensor_to_f32(synth.TYPE_F16, h, (2, 32)).reshape(-1), h.astype(np.float32))
    with pytest.raises(capi.NasrError):
        capi.tensor_to_f32(synth.TYPE_Q8_0, np.zeros(34, np.uint8), (33,))       # not a multiple of 32
    with pytest.raises(capi.NasrError):
        capi.tensor_to_f32(5, np.zeros(64, np.uint8), (32,))                     # a ggml type the converter never writes


def _as_converter_sees(name, a):
    if name.endswith(("conv.pointwise_conv1.weight", "conv.pointwise_conv2.weight")) and a.ndim == 3:
        return a[:, :, 0]                                  # scripts/convert_to_gguf.py:399-405
    if name.endswith("conv.depthwise_conv.weight") and a.ndim == 3:
        return np.ascontiguousarray(a[:, 0, :].T)          # (C, 1, k) -> (k, C), :406-411
    return a


@pytest.mark.parametrize("tag", ["en_f32", "en_f16", "en_q8_0", "en_q4_0", "ml_q8_0"])
def test_writer_reproduces_the_reference_file_byte_for_byte(fx, tag, tmp_path):
    z, meta = fx
    info = meta["files"][tag]
    src = {k[4:]: _as_converter_sees(k[4:], z[k]) for k in z.files if k.startswith("src.")}
    if info["quant"]:
        eng, deq = synth.quantize_weights(src, info["quant"])
    else:
        eng = src
    hp = gguf_io.default_hparams(n_layers=2, num_prompts=info["num_prompts"], kernel_size=9, vocab_size=len(info["vocab"]) + 1)
    out = tmp_path / f"{tag}.gguf"
    gguf_io.write_gguf(out, eng, hp, info["vocab"], prompt_dict=info["prompt_dict"], legacy_vocab_blob=True, name=info["model_name"])
    ref = z[f"file.{tag}"].tobytes()
    got = out.read_bytes()
    assert len(got) == len(ref)
    assert got == ref
    if info["quant"]:      # which tensors the converter quantised: its rules, restated in synth.quantize_weights
        kv, tensors, start = gguf_io.read_gguf(out)
        tid = {"f16": 1, "q8_0": 8, "q4_0": 2}[info["quant"]]
        quantised = sorted(n for n, t in tensors.items() if t[0] == tid)
        assert quantised == sorted(["encoder.layers.0.feed_forward1.linear1.weight", "encoder.layers.0.feed_forward1.linear2.weight",
                                    "encoder.layers.0.self_attn.linear_pos.weight", "encoder.layers.0.conv.pointwise_conv1.weight",
                                    "encoder.layers.1.feed_forward2.linear1.weight"])
        assert all(t[0] == 0 for n, t in tensors.items() if n not in quantised)


@pytest.mark.parametrize("tag", ["en_f32", "en_q8_0", "en_q4_0", "ml_q8_0"])
def test_cpp_reader_on_reference_files(fx, tag, tmp_path):
    z, meta = fx
    info = meta["files"][tag]
    if not (BIN / "gguf_dump").exists():
        subprocess.check_call(["make", "-C", str(ROOT / "nemotron-asr.cpp_amd" / "host"), "../bin/gguf_dump"])
    path = tmp_path / f"{tag}.gguf"
    raw = z[f"file.{tag}"].tobytes()
    path.write_bytes(raw)
    out = json.loads(subprocess.check_output([str(BIN / "gguf_dump"), str(path), "--full"]))
    assert out["version"] == 3 and out["data_start"] % 32 == 0
    kv = out["kv"]
    assert (kv["nemo.n_mels"], kv["nemo.d_model"], kv["nemo.n_heads"], kv["nemo.d_head"], kv["nemo.d_ff"]) == (128, 1024, 8, 128, 4096)
    assert (kv["nemo.n_layers"], kv["nemo.kernel_size"], kv["nemo.vocab_size"]) == (2, 9, len(info["vocab"]) + 1)
    assert (kv["nemo.decoder_dim"], kv["nemo.joint_dim"], kv["nemo.subsampling_factor"], kv["nemo.att_left_context"]) == (640, 640, 8, 70)
    assert kv["nemo.num_prompts"] == info["num_prompts"]
    assert out["name"] == info["model_name"]
    assert out["vocab"] == info["vocab"]
    if tag.startswith("en"):
        assert out["legacy_vocab"] == info["vocab"]            # the 8-byte records decode to the same pieces
        assert out["prompt_langs"] == [] and out["n_kv"] == 2 + 1 + 1 + 13
    else:
        assert out["legacy_vocab"] is None                     # a token longer than 7 bytes: the converter omits the blob
        pd = info["prompt_dict"]
        assert out["prompt_langs"] == sorted(pd) and out["prompt_ids"] == [pd[k] for k in sorted(pd)]
        assert out["n_kv"] == 2 + 1 + 2 + 13
    # tensor table and data: independently parsed by the Python reader, values against the source arrays
    kvp, tensors, start = gguf_io.read_gguf(path)
    assert out["data_start"] == start
    src = {k[4:]: _as_converter_sees(k[4:], z[k]) for k in z.files if k.startswith("src.")}
    assert [t["name"] for t in out["tensors"]] == list(src)
    for t in out["tensors"]:
        ty, dims, off, nb = tensors[t["name"]]
        a = src[t["name"]]
        assert t["ne"][:t["n_dims"]] == list(reversed(a.shape)) and t["n_dims"] == a.ndim
        assert (t["type"], t["offset"], t["nbytes"]) == (ty, off, nb) and off % 32 == 0
        data = raw[start + off:start + off + nb]
        assert t["hash"] == _fnv(data)
        if ty == 0:
            assert data == a.astype(np.float32).tobytes()
        else:
            vals = capi.tensor_to_f32(ty, np.frombuffer(data, np.uint8), a.shape)
            tol = {1: 1e-3, 8: 1.0 / 127 * 0.52, 2: 1.0 / 7 * 1.02}[ty]
            assert np.abs(vals - a).max() <= tol * np.abs(a).max() + 1e-7
    dw = next(t for t in out["tensors"] if "depthwise" in t["name"])
    assert dw["ne"][:2] == [32, 9] and dw["type"] == 0          # (k, C): ne[1] = kernel size (src/nemo-ggml.cpp:357-360)


def test_reader_rejects_corrupt_headers(fx, tmp_path):
    """ADVICE (round 1): alignment 0 / not a power of two, offsets that wrap, non-positive extents."""
    import struct
    z, meta = fx
    raw = bytearray(z["file.en_f32"].tobytes())
    path = tmp_path / "good.gguf"
    path.write_bytes(bytes(raw))
    kv, tensors, start = gguf_io.read_gguf(path)
    name0 = next(iter(tensors)).encode()
    pos = bytes(raw).index(struct.pack("<Q", len(name0)) + name0)       # tensor-info record of the first tensor
    p_ndims = pos + 8 + len(name0)
    ndims = struct.unpack_from("<I", raw, p_ndims)[0]
    p_dims = p_ndims + 4
    p_off = p_dims + 8 * ndims + 4

    def run(mut, name):
        b = bytearray(raw)
        mut(b)
        f = tmp_path / name
        f.write_bytes(bytes(b))
        r = subprocess.run([str(BIN / "gguf_dump"), str(f)], capture_output=True)
        return r.returncode, r.stderr

    rc, err = run(lambda b: struct.pack_into("<Q", b, p_off, 0xFFFFFFFFFFFFFFF0), "wrap.gguf")
    assert rc == 2 and b"out of file bounds" in err
    rc, err = run(lambda b: struct.pack_into("<q", b, p_dims, 0), "zero_dim.gguf")
    assert rc == 2 and b"bad tensor shape" in err
    rc, err = run(lambda b: struct.pack_into("<q", b, p_dims, -4), "neg_dim.gguf")
    assert rc == 2 and b"bad tensor shape" in err
    rc, err = run(lambda b: struct.pack_into("<q", b, p_dims, 1 << 61), "huge_dim.gguf")
    assert rc == 2
    # general.alignment = 0 / 24 (the converter never writes the key: insert the KV in front of the first one)
    good = bytes(raw)
    hdr_end = good.index(struct.pack("<Q", len(b"general.architecture")) + b"general.architecture")
    for val in (0, 24):
        extra = struct.pack("<Q", 17) + b"general.alignment" + struct.pack("<iI", 4, val)
        b = bytearray(good[:hdr_end] + extra + good[hdr_end:])
        struct.pack_into("<q", b, 16, struct.unpack_from("<q", good, 16)[0] + 1)      # n_kv + 1
        f = tmp_path / f"align{val}.gguf"
        f.write_bytes(bytes(b))
        r = subprocess.run([str(BIN / "gguf_dump"), str(f)], capture_output=True)
        assert r.returncode == 2 and b"alignment" in r.stderr, (val, r.stderr)

"""Minimal GGUF v3 writer/reader for the `nemo.*` schema the reference uses
(layout: reference scripts/convert_to_gguf.py:60-116, :491-540; reader side
src/nemo-ggml.cpp:99-182).  Used to produce synthetic model files for the C++ host
(host/gguf_reader.cpp, host/nemo_amd.cpp) and to cross-check that reader."""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"GGUF"
VERSION = 3
ALIGN = 32
T_UINT32, T_INT32, T_FLOAT32, T_STRING, T_ARRAY = 4, 5, 6, 8, 9
GGML_F32, GGML_F16, GGML_Q4_0, GGML_Q8_0 = 0, 1, 2, 8
_BLOCK = {GGML_F32: (1, 4), GGML_F16: (1, 2), GGML_Q4_0: (32, 18), GGML_Q8_0: (32, 34)}


def _wstr(f, s):
    b = s.encode() if isinstance(s, str) else s
    f.write(struct.pack("<Q", len(b)))
    f.write(b)


def default_hparams(n_layers=24, num_prompts=0, kernel_size=9, vocab_size=1025):
    """the `nemo.*` keys in the order the reference converter writes them (scripts/convert_to_gguf.py:356-371)"""
    return {"nemo.n_mels": 128, "nemo.d_model": 1024, "nemo.n_heads": 8, "nemo.d_head": 128, "nemo.d_ff": 4096,
            "nemo.n_layers": n_layers, "nemo.kernel_size": kernel_size, "nemo.vocab_size": vocab_size, "nemo.decoder_dim": 640,
            "nemo.joint_dim": 640, "nemo.subsampling_factor": 8, "nemo.att_left_context": 70, "nemo.num_prompts": num_prompts}


def synthetic_vocab(n=1024):
    """SentencePiece-like pieces: a third start a word (U+2581 prefix); every piece fits a legacy 8-byte record."""
    return [("▁" if i % 3 == 0 else "") + "t" + np.base_repr(i, 36).lower() for i in range(n)]


LEGACY_WORD = 8


def pack_vocab_legacy(vocab):
    """fixed 8-byte NUL-terminated records, or None when a token does not fit -- the converter then omits the KV
    (scripts/convert_to_gguf.py:288-306; reader: src/nemo-ggml.cpp:156-165)"""
    enc = [t.encode() for t in vocab]
    if any(len(e) + 1 > LEGACY_WORD for e in enc):
        return None
    return b"".join(e.ljust(LEGACY_WORD, b"\0") for e in enc)


def write_gguf(path, weights: dict, hparams: dict, vocab: list, prompt_dict: dict | None = None,
               legacy_vocab_blob=False, name="synthetic-nemotron"):
    """weights: name -> float32 ndarray, or (ggml_type, raw uint8/float16 ndarray, shape).
    Byte-for-byte the file the reference converter writes for the same content (tests/test_gguf_reference_fixtures.py).
    legacy_vocab_blob: also write `tokenizer.vocab` when every token fits (the converter always does)."""
    blob = pack_vocab_legacy(vocab) if legacy_vocab_blob else None
    legacy_vocab_blob = blob is not None
    if prompt_dict:
        prompt_dict = {k: prompt_dict[k] for k in sorted(prompt_dict)}     # :381-383
    infos, off = [], 0
    for nm, v in weights.items():
        if isinstance(v, tuple):
            t, raw, shape = v
            data = np.ascontiguousarray(raw).tobytes()
        else:
            t, shape, data = GGML_F32, v.shape, np.ascontiguousarray(v, np.float32).tobytes()
        off = (off + ALIGN - 1) // ALIGN * ALIGN
        infos.append((nm, list(reversed(shape)), t, off, data))
        off += len(data)
    kv = [("general.architecture", T_STRING, "nemo"), ("general.name", T_STRING, name)]
    with open(path, "wb") as f:
        n_kv = 2 + 1 + (1 if legacy_vocab_blob else 0) + (2 if prompt_dict else 0) + len(hparams)
        f.write(MAGIC)
        f.write(struct.pack("<I", VERSION))
        f.write(struct.pack("<q", len(infos)))
        f.write(struct.pack("<q", n_kv))
        for k, t, v in kv:
            _wstr(f, k); f.write(struct.pack("<i", t)); _wstr(f, v)
        _wstr(f, "tokenizer.vocab_list"); f.write(struct.pack("<ii", T_ARRAY, T_STRING)); f.write(struct.pack("<Q", len(vocab)))
        for s in vocab:
            _wstr(f, s)
        if legacy_vocab_blob:   # fixed 8-byte NUL-padded records (src/nemo-ggml.cpp:156-165)
            _wstr(f, "tokenizer.vocab"); f.write(struct.pack("<i", T_STRING)); _wstr(f, blob)
        if prompt_dict:
            langs = list(prompt_dict)
            _wstr(f, "nemo.prompt_langs"); f.write(struct.pack("<ii", T_ARRAY, T_STRING)); f.write(struct.pack("<Q", len(langs)))
            for s in langs:
                _wstr(f, s)
            _wstr(f, "nemo.prompt_ids"); f.write(struct.pack("<ii", T_ARRAY, T_INT32)); f.write(struct.pack("<Q", len(langs)))
            for s in langs:
                f.write(struct.pack("<i", prompt_dict[s]))
        for k, v in hparams.items():
            _wstr(f, k); f.write(struct.pack("<i", T_UINT32)); f.write(struct.pack("<I", int(v)))
        for nm, dims, t, o, data in infos:
            _wstr(f, nm)
            f.write(struct.pack("<I", len(dims)))
            for d in dims:
                f.write(struct.pack("<q", d))
            f.write(struct.pack("<i", t))
            f.write(struct.pack("<Q", o))
        pos = f.tell()
        f.write(b"\0" * ((pos + ALIGN - 1) // ALIGN * ALIGN - pos))
        start = f.tell()
        for nm, dims, t, o, data in infos:
            cur = f.tell()
            f.write(b"\0" * (start + o - cur))
            f.write(data)


def read_gguf(path):
    """-> (kv dict, {name: (type, dims (ggml order), offset, nbytes)}, data_start)."""
    with open(path, "rb") as f:
        b = f.read()
    p = 0

    def rd(fmt):
        nonlocal p
        v = struct.unpack_from("<" + fmt, b, p)
        p += struct.calcsize("<" + fmt)
        return v[0] if len(v) == 1 else v

    def rstr():
        nonlocal p
        n = rd("Q")
        s = b[p:p + n]
        p += n
        return s

    assert b[:4] == MAGIC
    p = 4
    assert rd("I") == VERSION
    n_t, n_kv = rd("q"), rd("q")
    scal = {0: "B", 1: "b", 2: "H", 3: "h", 4: "I", 5: "i", 6: "f", 7: "?", 10: "Q", 11: "q", 12: "d"}
    kv = {}
    for _ in range(n_kv):
        k = rstr().decode()
        t = rd("i")
        if t == T_STRING:
            kv[k] = rstr()
        elif t == T_ARRAY:
            et, n = rd("i"), rd("Q")
            kv[k] = [rstr().decode() if et == T_STRING else rd(scal[et]) for _ in range(n)]
        else:
            kv[k] = rd(scal[t])
    tensors = {}
    for _ in range(n_t):
        nm = rstr().decode()
        nd = rd("I")
        dims = [rd("q") for _ in range(nd)]
        t, o = rd("i"), rd("Q")
        per, bs = _BLOCK[t]
        tensors[nm] = (t, dims, o, int(np.prod(dims)) // per * bs)
    start = (p + ALIGN - 1) // ALIGN * ALIGN
    return kv, tensors, start

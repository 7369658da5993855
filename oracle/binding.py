"""ctypes bindings for the CPU oracle (oracle/libnasr_oracle.so) and, when present, the
compiled reference (oracle/_ref/libnemo_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_SO = Path(os.environ["NASR_ORACLE_LIB"]) if os.environ.get("NASR_ORACLE_LIB") else HERE / "libnasr_oracle.so"     # override: the sanitizer build (tests/test_sanitizers.py)
REF_SO = HERE / "_ref" / "libnemo_ref.so"

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def _f(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_fp)


def build(force: bool = False) -> None:
    """Compile the oracle (and the reference shim when /root/reference exists)."""
    srcs = [HERE / n for n in ("nasr_oracle.c", "nasr_oracle.h", "diar_oracle.c", "diar_oracle.h")]
    if os.environ.get("NASR_ORACLE_LIB"):
        return
    if force or not ORACLE_SO.exists() or ORACLE_SO.stat().st_mtime < max(f.stat().st_mtime for f in srcs):
        subprocess.check_call(["make", "-C", str(HERE), "libnasr_oracle.so"], stdout=subprocess.DEVNULL)
    ref_root = Path(os.environ.get("NASR_REFERENCE", "/root/reference"))
    if (ref_root / "src").is_dir() and (force or not REF_SO.exists() or REF_SO.stat().st_mtime < (HERE / "ref_shim.cpp").stat().st_mtime):
        subprocess.check_call(["make", "-C", str(HERE), "ref", f"REF={ref_root}"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        # Bound the OpenMP team: GPU hosts expose hundreds of hardware threads but the job may be
        # pinned to a few, and an oversubscribed spinning team is orders of magnitude slower.
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = C.CDLL(str(ORACLE_SO))
        L.orc_set_num_threads.argtypes = [C.c_int]
        try:
            n_cpu = len(os.sched_getaffinity(0))
        except AttributeError:
            n_cpu = os.cpu_count() or 1
        L.orc_set_num_threads(int(os.environ.get("NASR_ORACLE_THREADS", min(n_cpu, 16))))
        L.orc_model_create.restype = C.c_void_p
        L.orc_model_create.argtypes = [C.c_int] * 4
        L.orc_model_set_tensor.argtypes = [C.c_void_p, C.c_char_p, _fp, C.c_int64]
        L.orc_model_set_tensor_q8_0.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
        L.orc_stream_reset_reference.argtypes = [C.c_void_p]
        L.orc_stream_enable_decision_log.argtypes = [C.c_void_p, C.c_int]
        L.orc_stream_decision_log.argtypes = [C.c_void_p, _ip, _ip, _ip, _ip, _fp, C.c_int]
        L.orc_model_finalize.argtypes = [C.c_void_p]
        L.orc_model_free.argtypes = [C.c_void_p]
        L.orc_preproc_create.restype = C.c_void_p
        L.orc_preproc_create.argtypes = [_fp, _fp]
        L.orc_preproc_free.argtypes = [C.c_void_p]
        L.orc_preproc_reset.argtypes = [C.c_void_p]
        L.orc_preproc_process.argtypes = [C.c_void_p, C.POINTER(C.c_int16), C.c_int, _fp, C.c_int]
        L.orc_subsampling.argtypes = [C.c_void_p, _fp, C.c_int, _fp]
        L.orc_pos_emb.argtypes = [C.c_int, _fp]
        L.orc_layer_chunk0.argtypes = [C.c_void_p, C.c_int, _fp, C.c_int, _fp]
        L.orc_decoder_joint.argtypes = [C.c_void_p, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]
        L.orc_stream_create.restype = C.c_void_p
        L.orc_stream_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
        for n in ("free", "reset"):
            getattr(L, f"orc_stream_{n}").argtypes = [C.c_void_p]
        for n in ("chunk_mel_frames", "chunk_len", "cache_valid_len", "total_chunks", "decode_iterations"):
            getattr(L, f"orc_stream_{n}").argtypes = [C.c_void_p]
        L.orc_stream_encode_chunk.argtypes = [C.c_void_p, _fp, _fp]
        L.orc_stream_decode.argtypes = [C.c_void_p, _fp, C.c_int, _ip, C.c_int]
        L.orc_stream_push_mel.argtypes = [C.c_void_p, _fp, C.c_int, _ip, C.c_int]
        L.orc_stream_process.argtypes = [C.c_void_p, C.POINTER(C.c_int16), C.c_int, _ip, C.c_int]
        L.orc_stream_finalize.argtypes = [C.c_void_p, _ip, C.c_int]
        L.orc_stream_set_taps.argtypes = [C.c_void_p, _fp, _fp]
        L.orc_stream_token_frames.argtypes = [C.c_void_p, _ip, C.c_int]
        L.orc_stream_set_prompt.argtypes = [C.c_void_p, C.c_int]
        L.orc_stream_get_cache.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp]
        L.orc_stream_get_decoder_state.argtypes = [C.c_void_p, _fp, _fp, _ip]
        L.orc_round_bf16.restype = C.c_float
        L.orc_round_bf16.argtypes = [C.c_float]
        _lib = L
    return _lib


EMU_BF16, EMU_Q8_ACT = 1, 2


class OracleModel:
    """weights: name -> float32 ndarray.  q8_blocks (with emulate_q8_act=True): name -> raw Q8_0 bytes of the encoder-layer
    matrices whose dequantised values are in `weights`: those are then multiplied with ggml-CPU semantics (activation rows
    quantised per 32, int8 dot products) instead of f32."""

    def __init__(self, weights: dict, n_layers: int, kernel_size: int = 9, num_prompts: int = 0,
                 emulate_bf16: bool = False, emulate_q8_act: bool = False, q8_blocks: dict = None):
        L = lib()
        self.n_layers, self.kernel_size = n_layers, kernel_size
        self._keep = {}
        self.h = L.orc_model_create(n_layers, kernel_size, num_prompts,
                                    (EMU_BF16 if emulate_bf16 else 0) | (EMU_Q8_ACT if emulate_q8_act else 0))
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            self._keep[name] = a
            rc = L.orc_model_set_tensor(self.h, name.encode(), _f(a), a.size)
            if rc != 0:
                raise ValueError(f"oracle rejected tensor {name} {a.shape}")
        for name, raw in (q8_blocks or {}).items():
            r = np.ascontiguousarray(raw).view(np.uint8)
            self._keep["q8:" + name] = r
            numel = r.size // 34 * 32
            if L.orc_model_set_tensor_q8_0(self.h, name.encode(), r.ctypes.data, numel) != 0:
                raise ValueError(f"oracle rejected Q8_0 blocks of {name}")
        if L.orc_model_finalize(self.h) != 0:
            raise ValueError("oracle model incomplete")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_model_free(self.h)
            self.h = None

    def subsampling(self, mel: np.ndarray) -> np.ndarray:
        mel = np.ascontiguousarray(mel, np.float32)
        n = mel.shape[0]
        out = np.zeros((n // 8 + 4, 1024), np.float32)
        k = lib().orc_subsampling(self.h, _f(mel), n, _f(out))
        return out[:k].copy()

    def layer_chunk0(self, layer: int, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros_like(x)
        lib().orc_layer_chunk0(self.h, layer, _f(x), x.shape[0], _f(out))
        return out

    def decoder_joint(self, prev_token, h, c, enc_frame):
        h = np.ascontiguousarray(h, np.float32); c = np.ascontiguousarray(c, np.float32)
        e = np.ascontiguousarray(enc_frame, np.float32)
        logits = np.zeros(1025, np.float32); ho = np.zeros(1280, np.float32); co = np.zeros(1280, np.float32)
        lib().orc_decoder_joint(self.h, int(prev_token), _f(h), _f(c), _f(e), _f(logits), _f(ho), _f(co))
        return logits, ho, co


class OraclePreproc:
    def __init__(self, fb, window):
        self._fb = np.ascontiguousarray(fb, np.float32)
        self._w = np.ascontiguousarray(window, np.float32)
        self.h = lib().orc_preproc_create(_f(self._fb), _f(self._w))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_preproc_free(self.h)
            self.h = None

    def process(self, pcm: np.ndarray) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, np.int16)
        cap = pcm.size // 160 + 8
        out = np.zeros((cap, 128), np.float32)
        n = lib().orc_preproc_process(self.h, pcm.ctypes.data_as(C.POINTER(C.c_int16)), pcm.size, _f(out), cap)
        assert n >= 0
        return out[:n].copy()


class OracleStream:
    def __init__(self, model: OracleModel, right_context: int = 0, prompt_index: int = -1):
        self.model = model
        self.h = lib().orc_stream_create(model.h, right_context, prompt_index)
        self.T = lib().orc_stream_chunk_len(self.h)
        self.chunk_mel = lib().orc_stream_chunk_mel_frames(self.h)
        self._taps = None

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_stream_free(self.h)
            self.h = None

    def reset(self, reference=False):
        """reference=True: nemo_stream_reset as the reference codes it (caches and preprocessor carry survive)"""
        (lib().orc_stream_reset_reference if reference else lib().orc_stream_reset)(self.h)

    def enable_decision_log(self, on=True):
        lib().orc_stream_enable_decision_log(self.h, int(on))

    def decision_log(self):
        """one record per LSTM+joint evaluation: dict of arrays frame, ntok_before, best, second, margin"""
        n = lib().orc_stream_decision_log(self.h, None, None, None, None, None, 0)
        a = {k: np.zeros(max(n, 1), np.int32) for k in ("frame", "ntok_before", "best", "second")}
        mg = np.zeros(max(n, 1), np.float32)
        lib().orc_stream_decision_log(self.h, *[a[k].ctypes.data_as(_ip) for k in ("frame", "ntok_before", "best", "second")], _f(mg), n)
        out = {k: v[:n] for k, v in a.items()}
        out["margin"] = mg[:n]
        return out

    def enable_taps(self):
        sub = np.zeros((self.T, 1024), np.float32)
        lay = np.zeros((self.model.n_layers, self.T, 1024), np.float32)
        self._taps = (sub, lay)
        lib().orc_stream_set_taps(self.h, _f(sub), _f(lay))
        return sub, lay

    def encode_chunk(self, mel_chunk: np.ndarray) -> np.ndarray:
        mel_chunk = np.ascontiguousarray(mel_chunk, np.float32)
        assert mel_chunk.shape == (self.chunk_mel, 128)
        out = np.zeros((self.T, 1024), np.float32)
        lib().orc_stream_encode_chunk(self.h, _f(mel_chunk), _f(out))
        return out

    def decode(self, enc: np.ndarray) -> list:
        enc = np.ascontiguousarray(enc, np.float32)
        cap = enc.shape[0] * 10 + 1
        toks = np.zeros(cap, np.int32)
        n = lib().orc_stream_decode(self.h, _f(enc), enc.shape[0], toks.ctypes.data_as(_ip), cap)
        return toks[:n].tolist()

    def push_mel(self, mel: np.ndarray) -> list:
        mel = np.ascontiguousarray(mel, np.float32)
        cap = (mel.shape[0] // 8 + self.T + 2) * 10
        toks = np.zeros(cap, np.int32)
        n = lib().orc_stream_push_mel(self.h, _f(mel), mel.shape[0], toks.ctypes.data_as(_ip), cap)
        return toks[:n].tolist()

    def process(self, pcm: np.ndarray) -> list:
        pcm = np.ascontiguousarray(pcm, np.int16)
        cap = (pcm.size // 1280 + self.T + 2) * 10
        toks = np.zeros(cap, np.int32)
        n = lib().orc_stream_process(self.h, pcm.ctypes.data_as(C.POINTER(C.c_int16)), pcm.size,
                                     toks.ctypes.data_as(_ip), cap)
        return toks[:n].tolist()

    def finalize(self) -> list:
        cap = self.T * 10 + 1
        toks = np.zeros(cap, np.int32)
        n = lib().orc_stream_finalize(self.h, toks.ctypes.data_as(_ip), cap)
        return toks[:n].tolist()

    @property
    def cache_valid_len(self):
        return lib().orc_stream_cache_valid_len(self.h)

    @property
    def total_chunks(self):
        return lib().orc_stream_total_chunks(self.h)

    @property
    def decode_iterations(self):
        return lib().orc_stream_decode_iterations(self.h)

    def set_prompt(self, prompt_index: int):
        lib().orc_stream_set_prompt(self.h, prompt_index)

    def token_frames(self) -> list:
        """absolute encoder-frame index (x 80 ms) of every token since create/reset"""
        n = lib().orc_stream_token_frames(self.h, None, 0)
        out = np.zeros(max(n, 1), np.int32)
        lib().orc_stream_token_frames(self.h, out.ctypes.data_as(_ip), n)
        return out[:n].tolist()

    def get_cache(self, which: int, layer: int) -> np.ndarray:
        rows = 70 if which < 2 else self.model.kernel_size - 1
        out = np.zeros((rows, 1024), np.float32)
        lib().orc_stream_get_cache(self.h, which, layer, _f(out))
        return out

    def decoder_state(self):
        h = np.zeros(1280, np.float32); c = np.zeros(1280, np.float32); p = C.c_int(0)
        lib().orc_stream_get_decoder_state(self.h, _f(h), _f(c), C.byref(p))
        return h, c, p.value


def first_divergence(log: dict, ref_tokens, ref_frames, got_tokens, got_frames):
    """Where a reduced-precision engine's greedy path leaves the oracle's.  Returns None when the two token streams
    (ids and emission frames) are identical, else a dict: index of the first differing token, the oracle decision (one
    LSTM+joint evaluation) at which the engine chose differently, and that decision's top-2 logit margin in the oracle.
    The engine only reports emissions; the decision is recovered from them: with i tokens in common, the engine emitted
    earlier than the oracle (=> the oracle's BLANK decision at that frame with i tokens out), later or never (=> the oracle's
    emission decision of token i), or another token at the same frame (=> that same emission decision)."""
    n = min(len(ref_tokens), len(got_tokens))
    i = 0
    while i < n and ref_tokens[i] == got_tokens[i] and ref_frames[i] == got_frames[i]:
        i += 1
    if i == len(ref_tokens) and i == len(got_tokens):
        return None
    if i < len(got_tokens) and (i >= len(ref_tokens) or got_frames[i] < ref_frames[i]):
        frame, want_blank = got_frames[i], True       # the engine emitted where the oracle said blank
    else:
        frame, want_blank = ref_frames[i], False      # the engine said blank / something else where the oracle emitted token i
    sel = np.nonzero((log["frame"] == frame) & (log["ntok_before"] == i))[0]
    if sel.size != 1:
        return dict(index=i, frame=int(frame), decision=-1, margin=float("nan"), oracle_best=-1, oracle_second=-1)
    k = int(sel[0])
    assert (log["best"][k] == BLANK_ID) == want_blank
    return dict(index=i, frame=int(frame), decision=k, margin=float(log["margin"][k]), oracle_best=int(log["best"][k]),
                oracle_second=int(log["second"][k]), engine_choice=(int(got_tokens[i]) if i < len(got_tokens) and got_frames[i] == frame else BLANK_ID))


BLANK_ID = 1024


def pos_emb(position: int) -> np.ndarray:
    out = np.zeros(1024, np.float32)
    lib().orc_pos_emb(position, _f(out))
    return out


def round_bf16(x: np.ndarray) -> np.ndarray:
    """Vectorised RNE f32 -> bf16 -> f32 (same as orc_round_bf16 for finite x)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


# ------------------------------------------------------------------------------------
# compiled reference (only where oracle/_ref/libnemo_ref.so exists)
# ------------------------------------------------------------------------------------
_ref = None


def have_ref() -> bool:
    build()
    return REF_SO.exists()


def ref():
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libnemo_ref.so not built (needs /root/reference)")
        R = C.CDLL(str(REF_SO))
        pp = C.POINTER(_fp)
        R.ref_preproc_create.restype = C.c_void_p
        R.ref_preproc_create.argtypes = [_fp, _fp]
        R.ref_preproc_free.argtypes = [C.c_void_p]
        R.ref_preproc_process.argtypes = [C.c_void_p, C.POINTER(C.c_int16), C.c_int, _fp, C.c_int]
        R.ref_subsampling.argtypes = [pp, _fp, C.c_int, _fp]
        R.ref_pos_emb.argtypes = [C.c_int, _fp]
        if hasattr(R, "ref_rel_shift"):
            R.ref_rel_shift.argtypes = [_fp, C.c_int, C.c_int, _fp]
        R.ref_conformer_layer.argtypes = [pp, _fp, C.c_int, _fp]
        R.ref_ffn.argtypes = [_fp, _fp, _fp, C.c_int, _fp]
        R.ref_layer_norm.argtypes = [_fp, _fp, _fp, C.c_int, _fp]
        R.ref_decoder_joint_seq.argtypes = [pp, _ip, C.c_int, _fp, _fp, _fp, _fp]
        R.ref_greedy.argtypes = [pp, _fp, C.c_int, _ip, C.c_int]
        _ref = R
    return _ref


def _ptr_array(arrs):
    keep = [np.ascontiguousarray(a, np.float32) for a in arrs]
    arr = (_fp * len(keep))(*[_f(a) for a in keep])
    return arr, keep


SUB_ORDER = ["conv.0.weight", "conv.0.bias", "conv.2.weight", "conv.2.bias", "conv.3.weight", "conv.3.bias",
             "conv.5.weight", "conv.5.bias", "conv.6.weight", "conv.6.bias", "out.weight", "out.bias"]
LAYER_ORDER = ["norm_feed_forward1.weight", "norm_feed_forward1.bias", "feed_forward1.linear1.weight",
               "feed_forward1.linear2.weight", "norm_self_att.weight", "norm_self_att.bias",
               "self_attn.linear_q.weight", "self_attn.linear_k.weight", "self_attn.linear_v.weight",
               "self_attn.linear_pos.weight", "self_attn.linear_out.weight", "self_attn.pos_bias_u",
               "self_attn.pos_bias_v", "norm_conv.weight", "norm_conv.bias", "conv.pointwise_conv1.weight",
               "conv.depthwise_conv.weight", "conv.batch_norm.weight", "conv.batch_norm.bias",
               "conv.pointwise_conv2.weight", "norm_feed_forward2.weight", "norm_feed_forward2.bias",
               "feed_forward2.linear1.weight", "feed_forward2.linear2.weight", "norm_out.weight", "norm_out.bias"]
DEC_ORDER = ["decoder.prediction.embed.weight"] + [
    f"decoder.prediction.dec_rnn.lstm.{k}_l{l}" for l in (0, 1) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")
] + ["joint.enc.weight", "joint.enc.bias", "joint.pred.weight", "joint.pred.bias",
     "joint.joint_net.2.weight", "joint.joint_net.2.bias"]


def ref_preproc(fb, window, pcm_pieces) -> np.ndarray:
    R = ref()
    fb = np.ascontiguousarray(fb, np.float32); window = np.ascontiguousarray(window, np.float32)
    h = R.ref_preproc_create(_f(fb), _f(window))
    outs = []
    for p in pcm_pieces:
        p = np.ascontiguousarray(p, np.int16)
        cap = p.size // 160 + 8
        o = np.zeros((cap, 128), np.float32)
        n = R.ref_preproc_process(h, p.ctypes.data_as(C.POINTER(C.c_int16)), p.size, _f(o), cap)
        assert n >= 0
        outs.append(o[:n].copy())
    R.ref_preproc_free(h)
    return np.concatenate(outs, axis=0) if outs else np.zeros((0, 128), np.float32)


def ref_subsampling(weights, mel) -> np.ndarray:
    arr, keep = _ptr_array([weights["encoder.pre_encode." + k] for k in SUB_ORDER])
    mel = np.ascontiguousarray(mel, np.float32)
    out = np.zeros((mel.shape[0] // 8 + 4, 1024), np.float32)
    n = ref().ref_subsampling(arr, _f(mel), mel.shape[0], _f(out))
    return out[:n].copy()


def ref_conformer_layer(weights, layer, x) -> np.ndarray:
    ws = []
    for k in LAYER_ORDER:
        a = weights[f"encoder.layers.{layer}.{k}"]
        if k == "conv.depthwise_conv.weight":   # GGUF (k, C) -> PyTorch [C][1][k]
            a = np.ascontiguousarray(a.T)
        ws.append(a)
    arr, keep = _ptr_array(ws)
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    ref().ref_conformer_layer(arr, _f(x), x.shape[0], _f(out))
    return out


def ref_pos_emb(seq_len) -> np.ndarray:
    out = np.zeros((2 * seq_len - 1, 1024), np.float32)
    ref().ref_pos_emb(seq_len, _f(out))
    return out


def ref_rel_shift(x: np.ndarray) -> np.ndarray:
    """x [heads][qlen][2 qlen - 1] -> [heads][qlen][qlen] through the compiled reference rel_shift"""
    x = np.ascontiguousarray(x, np.float32)
    H, q, pl = x.shape
    assert pl == 2 * q - 1
    out = np.zeros((H, q, q), np.float32)
    ref().ref_rel_shift(_f(x), H, q, _f(out))
    return out


def ref_decoder_joint_seq(weights, tokens, enc):
    arr, keep = _ptr_array([weights[k] for k in DEC_ORDER])
    tokens = np.ascontiguousarray(tokens, np.int32)
    enc = np.ascontiguousarray(enc, np.float32)
    n = tokens.size
    logits = np.zeros((n, 1025), np.float32); h = np.zeros(1280, np.float32); c = np.zeros(1280, np.float32)
    ref().ref_decoder_joint_seq(arr, tokens.ctypes.data_as(_ip), n, _f(enc), _f(logits), _f(h), _f(c))
    return logits, h, c


def ref_greedy(weights, enc) -> list:
    arr, keep = _ptr_array([weights[k] for k in DEC_ORDER])
    enc = np.ascontiguousarray(enc, np.float32)
    cap = enc.shape[0] * 10 + 1
    toks = np.zeros(cap, np.int32)
    n = ref().ref_greedy(arr, _f(enc), enc.shape[0], toks.ctypes.data_as(_ip), cap)
    return toks[:n].tolist()


def token_timing_report(log: dict, ref_tokens, ref_frames, got_tokens, got_frames):
    """Token-for-token comparison of an engine's greedy output with the oracle's, with the oracle decision behind every
    difference.  Returns dict(tokens_equal, n_ref, n_got, shifts=[...], first_divergence): `shifts` lists the tokens that
    are the same id at another frame (only meaningful when the id sequences are equal): the oracle decision at which the
    engine chose differently -- the oracle's emission (engine later) or the oracle's blank at the engine's frame (engine
    earlier) -- and its top-2 margin."""
    out = dict(tokens_equal=list(ref_tokens) == list(got_tokens), n_ref=len(ref_tokens), n_got=len(got_tokens), shifts=[],
               first_divergence=first_divergence(log, ref_tokens, ref_frames, got_tokens, got_frames))
    if not out["tokens_equal"]:
        return out
    for i, (fr, fg) in enumerate(zip(ref_frames, got_frames)):
        if fr == fg:
            continue
        frame = fg if fg < fr else fr
        sel = np.nonzero((log["frame"] == frame) & (log["ntok_before"] == i))[0]
        margin = float(log["margin"][int(sel[0])]) if sel.size == 1 else float("nan")
        out["shifts"].append(dict(index=i, token=int(ref_tokens[i]), ref_frame=int(fr), got_frame=int(fg), margin=margin))
    return out

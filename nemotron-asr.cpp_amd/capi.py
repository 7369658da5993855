"""ctypes binding of the C ABI in include/nemotron_asr_amd.h (the HIP engine).

This is plumbing only: every call goes straight into libnemotron_asr_amd.so.  There is no
CPU fallback: if the library is missing or no MI355X is visible the calls raise."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("NASR_LIB_PATH") or HERE / "libnemotron_asr_amd.so")   # NASR_LIB_PATH: a diagnostic build of the same ABI

DTYPE_F32, DTYPE_BF16 = 0, 1
TYPE_F32, TYPE_F16, TYPE_Q4_0, TYPE_Q8_0 = 0, 1, 2, 8
FLAG_PCM_DEVICE, FLAG_NO_SYNC = 1, 2
FLAG_AUDIO_S16 = 4
DIAR_VAD_BF16 = 0x100       # OR into Diar's dtype: MarbleNet on the bf16 MFMA
DIAR_VAD_F16 = 0x200        # ... on the f16 MFMA with IEEE-half activation planes
RESET_FRESH, RESET_REFERENCE = 0, 1
TAP_MEL, TAP_SUBSAMPLED, TAP_LAYER_OUT, TAP_ENCODER_OUT, TAP_K_CACHE, TAP_V_CACHE, TAP_CONV_CACHE, TAP_DEC_STATE = range(8)

EXPORTS = [
    "nasr_last_error", "nasr_abi_version", "nasr_tensor_to_f32", "nasr_engine_create", "nasr_engine_create_ex", "nasr_engine_destroy",
    "nasr_stream_create", "nasr_stream_reset", "nasr_stream_reset_ex", "nasr_stream_destroy", "nasr_stream_set_prompt",
    "nasr_stream_get_stats", "nasr_stream_get_progress", "nasr_stream_get_token_frames", "nasr_engine_step", "nasr_engine_step_mel", "nasr_engine_finalize",
    "nasr_engine_collect", "nasr_engine_set_option", "nasr_engine_set_debug", "nasr_stream_get_tap", "nasr_engine_profile",
    "nasr_engine_profile_read", "nasr_engine_hip_stream", "nasr_engine_lend_stream", "nasr_device_alloc", "nasr_device_free",
    "nasr_device_upload", "nasr_engine_synchronize", "nasr_engine_get_counter", "nasr_stream_debug_fill_kv",
    "nasr_diar_create", "nasr_diar_destroy", "nasr_diar_set_stream", "nasr_diar_vad", "nasr_diar_embed", "nasr_diar_logmel", "nasr_diar_last_gpu_ms",
]


class HParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_mels", "d_model", "n_heads", "d_head", "d_ff", "n_layers", "vocab_size", "decoder_dim",
        "joint_dim", "subsampling_factor", "att_left_context", "kernel_size", "num_prompts")]


class WeightDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_int32), ("n_dims", C.c_int32),
                ("ne", C.c_int64 * 4), ("data", C.c_void_p)]


class StreamStats(C.Structure):
    _fields_ = [("samples_in", C.c_int64), ("chunks", C.c_int32), ("decode_iterations", C.c_int32),
                ("tokens", C.c_int32), ("cache_valid_len", C.c_int32), ("mel_frames_buffered", C.c_int32),
                ("reserved", C.c_int32)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("bytes", C.c_double), ("flops", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} not built: run __graft_entry__.build() (no CPU fallback exists)")
        L = C.CDLL(str(LIB_PATH))
        vp, ip = C.c_void_p, C.POINTER(C.c_int32)
        L.nasr_last_error.restype = C.c_char_p
        L.nasr_engine_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(HParams), C.POINTER(WeightDesc), C.c_int, C.c_int]
        L.nasr_engine_create_ex.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(HParams), C.POINTER(WeightDesc), C.c_int, C.c_int, C.c_int]
        L.nasr_engine_destroy.argtypes = [vp]
        L.nasr_engine_destroy.restype = None
        L.nasr_tensor_to_f32.argtypes = [C.POINTER(WeightDesc), C.POINTER(C.c_float), C.c_int64]
        L.nasr_tensor_to_f32.restype = C.c_int64
        L.nasr_stream_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
        for n in ("reset", "destroy"):
            getattr(L, f"nasr_stream_{n}").argtypes = [vp]
        L.nasr_stream_set_prompt.argtypes = [vp, C.c_int]
        L.nasr_stream_reset_ex.argtypes = [vp, C.c_int]
        L.nasr_stream_debug_fill_kv.argtypes = [vp, C.c_float]
        L.nasr_stream_get_stats.argtypes = [vp, C.POINTER(StreamStats)]
        L.nasr_stream_get_progress.argtypes = [vp, C.POINTER(StreamStats)]
        L.nasr_diar_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(WeightDesc), C.c_int, C.c_int, C.c_int]
        L.nasr_diar_destroy.argtypes = [vp]
        L.nasr_diar_destroy.restype = None
        L.nasr_diar_vad.argtypes = [vp, C.c_int, C.POINTER(vp), ip, C.POINTER(vp), ip, ip, C.c_uint32]
        L.nasr_diar_embed.argtypes = [vp, C.c_int, C.POINTER(vp), ip, C.POINTER(C.c_float), C.c_uint32]
        L.nasr_diar_last_gpu_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
        L.nasr_diar_logmel.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.c_int32, C.c_int, C.POINTER(C.c_float), C.c_int64, ip]
        L.nasr_stream_get_token_frames.argtypes = [vp, C.c_int64, C.c_int32, C.POINTER(C.c_int32)]
        L.nasr_engine_step.argtypes = [vp, C.POINTER(vp), C.c_int, C.POINTER(vp), ip, C.POINTER(vp), ip, ip, C.c_uint32]
        L.nasr_engine_step_mel.argtypes = [vp, C.POINTER(vp), C.c_int, C.POINTER(vp), ip, C.POINTER(vp), ip, ip, C.c_uint32]
        L.nasr_engine_finalize.argtypes = [vp, C.POINTER(vp), C.c_int, C.POINTER(vp), ip, ip]
        L.nasr_engine_collect.argtypes = [vp, C.POINTER(vp), C.c_int, C.POINTER(vp), ip, ip]
        L.nasr_engine_set_debug.argtypes = [vp, C.c_int]
        L.nasr_engine_set_option.argtypes = [vp, C.c_char_p, C.c_int]
        L.nasr_stream_get_tap.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int64]
        L.nasr_stream_get_tap.restype = C.c_int64
        L.nasr_engine_profile.argtypes = [vp, C.c_int]
        L.nasr_engine_profile_read.argtypes = [vp, C.POINTER(KernelStat), C.c_int]
        L.nasr_engine_hip_stream.argtypes = [vp]
        L.nasr_engine_hip_stream.restype = vp
        L.nasr_engine_lend_stream.argtypes = [vp, C.POINTER(vp)]
        L.nasr_diar_set_stream.argtypes = [vp, vp]
        L.nasr_device_alloc.argtypes = [vp, C.POINTER(vp), C.c_int64]
        L.nasr_device_free.argtypes = [vp, vp]
        L.nasr_device_upload.argtypes = [vp, vp, vp, C.c_int64]
        L.nasr_engine_synchronize.argtypes = [vp]
        L.nasr_engine_get_counter.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def check_exports():
    """Every symbol include/nemotron_asr_amd.h declares is exported (no compute call)."""
    L = lib()
    for name in EXPORTS:
        getattr(L, name)
    assert L.nasr_abi_version() == 1
    return True


class NasrError(RuntimeError):
    pass


def _chk(rc):
    if rc < 0:
        raise NasrError(lib().nasr_last_error().decode())
    return rc


def default_hparams(n_layers=24, kernel_size=9, num_prompts=0) -> HParams:
    return HParams(n_mels=128, d_model=1024, n_heads=8, d_head=128, d_ff=4096, n_layers=n_layers,
                   vocab_size=1025, decoder_dim=640, joint_dim=640, subsampling_factor=8,
                   att_left_context=70, kernel_size=kernel_size, num_prompts=num_prompts)


class Stream:
    def __init__(self, engine: "Engine", right_context=0, prompt_index=-1):
        self.engine = engine
        h = C.c_void_p()
        _chk(lib().nasr_stream_create(engine.h, right_context, prompt_index, C.byref(h)))
        self.h = h
        self.R, self.T = right_context, 1 + right_context

    def reset(self, reference=False):
        """reference=True: nemo_stream_reset as the reference codes it (stale conv cache / preprocessor carry survive)"""
        _chk(lib().nasr_stream_reset_ex(self.h, RESET_REFERENCE if reference else RESET_FRESH))

    def set_prompt(self, prompt_index: int):
        _chk(lib().nasr_stream_set_prompt(self.h, prompt_index))

    def debug_fill_kv(self, value: float):
        """test hook: every K/V ring row of this stream's slot := +-value (stale rows must never reach a result)"""
        _chk(lib().nasr_stream_debug_fill_kv(self.h, float(value)))

    def destroy(self):
        if self.h:
            lib().nasr_stream_destroy(self.h)
            self.h = None

    def stats(self) -> StreamStats:
        s = StreamStats()
        _chk(lib().nasr_stream_get_stats(self.h, C.byref(s)))
        return s

    def progress(self) -> StreamStats:
        """host-mirror counters only: no device synchronisation"""
        s = StreamStats()
        _chk(lib().nasr_stream_get_progress(self.h, C.byref(s)))
        return s

    def token_frames(self, first=0, count=None) -> list:
        """absolute encoder-frame index (x 80 ms) of tokens [first, first + count) since create/reset"""
        if count is None:
            count = max(int(self.stats().tokens) - first, 0)
        out = np.zeros(max(count, 1), np.int32)
        n = _chk(lib().nasr_stream_get_token_frames(self.h, first, count, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out[:n].tolist()

    def tap(self, which, index=0, cap=None) -> np.ndarray:
        cap = cap or 1024 * 260          # up to MAXNEW = 256 encoder frames of one launch, or a 70-row cache
        out = np.zeros(cap, np.float32)
        n = _chk(lib().nasr_stream_get_tap(self.h, which, index, out.ctypes.data_as(C.POINTER(C.c_float)), cap))
        return out[:n].copy()


def weight_descs(weights: dict):
    """name -> ndarray (F32) or (ggml type id, packed bytes, logical shape)  ->  (keep-alive list, WeightDesc array)"""
    keep, descs = [], (WeightDesc * len(weights))()
    for i, (name, v) in enumerate(weights.items()):
        if isinstance(v, tuple):
            tid, raw, shape = v
            raw = np.ascontiguousarray(raw)
        else:
            tid, raw, shape = TYPE_F32, np.ascontiguousarray(v, np.float32), v.shape
        keep.append(raw)
        d = descs[i]
        d.name = name.encode()
        d.type = tid
        d.n_dims = len(shape)
        for j, s in enumerate(reversed(shape)):     # ggml order: ne[0] fastest
            d.ne[j] = s
        for j in range(len(shape), 4):
            d.ne[j] = 1
        d.data = raw.ctypes.data
    return keep, descs


def tensor_to_f32(type_id: int, raw: np.ndarray, shape) -> np.ndarray:
    """what the engine computes with for one GGUF tensor (host-side dequantisation, no GPU)"""
    keep, descs = weight_descs({"t": (type_id, raw, tuple(shape))})
    out = np.zeros(int(np.prod(shape)), np.float32)
    n = lib().nasr_tensor_to_f32(C.byref(descs[0]), out.ctypes.data_as(C.POINTER(C.c_float)), out.size)
    if n < 0:
        raise NasrError(lib().nasr_last_error().decode())
    return out[:n].reshape(shape)


class Diar:
    """Diarization side-car (MarbleNet VAD + TitaNet-L embeddings) through the C ABI."""

    def __init__(self, weights: dict, dtype=DTYPE_BF16, max_windows=8192, max_segments=64, device=0):
        L = lib()
        keep, descs = weight_descs(weights)
        h = C.c_void_p()
        _chk(L.nasr_diar_create(C.byref(h), device, dtype, descs, len(weights), max_windows, max_segments))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().nasr_diar_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_stream(self, hip_stream):
        """run on a stream the caller owns (Engine.lend_stream): a hardware queue no ASR lane uses"""
        _chk(lib().nasr_diar_set_stream(self.h, C.c_void_p(hip_stream)))

    def vad(self, audios: list) -> list:
        """P(speech) of every 0.63 s window (10 ms shift) of each buffer (float32 in [-1, 1], or int16 PCM) -> float32 arrays"""
        B = len(audios)
        s16 = all(np.asarray(a).dtype == np.int16 for a in audios)
        bufs = [np.ascontiguousarray(a, np.int16 if s16 else np.float32) for a in audios]
        return self._vad([b.ctypes.data for b in bufs], [b.size for b in bufs], FLAG_AUDIO_S16 if s16 else 0)

    def vad_device_s16(self, ptrs: list, n_samples: list) -> list:
        """same, on s16 PCM that is already in HBM (device pointers, e.g. Engine.upload): the ASR streams' own audio"""
        return self._vad(ptrs, n_samples, FLAG_AUDIO_S16 | FLAG_PCM_DEVICE)

    def _vad(self, ptrs, sizes, flags):
        B = len(ptrs)
        n = (C.c_int32 * B)(*sizes)
        outs = [np.zeros(max(1, 1 + (sz - 10080) // 160 if sz >= 10080 else 1), np.float32) for sz in sizes]
        ap = (C.c_void_p * B)(*ptrs)
        op = (C.c_void_p * B)(*[o.ctypes.data for o in outs])
        caps = (C.c_int32 * B)(*[o.size for o in outs])
        nw = (C.c_int32 * B)()
        _chk(lib().nasr_diar_vad(self.h, B, ap, n, op, caps, nw, flags))
        return [outs[b][:nw[b]] for b in range(B)]

    def embed_device_s16(self, ptrs: list, lens: list = None) -> np.ndarray:
        """192-d embeddings of sub-segments given as device pointers to 24 000 s16 samples each"""
        S = len(ptrs)
        ln = (C.c_int32 * S)(*(lens or [24000] * S))
        ap = (C.c_void_p * S)(*ptrs)
        out = np.zeros((S, 192), np.float32)
        _chk(lib().nasr_diar_embed(self.h, S, ap, ln, out.ctypes.data_as(C.POINTER(C.c_float)), FLAG_AUDIO_S16 | FLAG_PCM_DEVICE))
        return out

    def last_gpu_ms(self, which: str) -> float:
        """device time of the last vad() / embed() call's launch sequence (HIP events on the side-car's stream)"""
        ms = C.c_float(0.0)
        _chk(lib().nasr_diar_last_gpu_ms(self.h, 0 if which == "vad" else 1, C.byref(ms)))
        return float(ms.value)

    def logmel(self, audio: np.ndarray, which: str = "vad", normalize: bool = False):
        """parity tap: diarize_compute_logmel of one buffer on the device front end -> ([80][t_padded], t_valid)"""
        a = np.ascontiguousarray(audio, np.float32)
        t_valid = a.size // 160
        t_pad = (t_valid + 15) // 16 * 16
        out = np.empty((80, t_pad), np.float32)
        tv = C.c_int32(0)
        _chk(lib().nasr_diar_logmel(self.h, 0 if which == "vad" else 1, a.ctypes.data_as(C.POINTER(C.c_float)), a.size,
                                    int(normalize), out.ctypes.data_as(C.POINTER(C.c_float)), out.size, C.byref(tv)))
        return out, tv.value

    def embed(self, segments: list, lens: list = None) -> np.ndarray:
        """192-d embeddings of 1.5 s sub-segments (each zero padded to 24 000 samples) -> [S][192]"""
        S = len(segments)
        bufs = []
        for a in segments:
            b = np.zeros(24000, np.float32)
            a = np.asarray(a, np.float32)[:24000]
            b[:a.size] = a
            bufs.append(b)
        ln = (C.c_int32 * S)(*[(lens[i] if lens else min(len(segments[i]), 24000)) for i in range(S)])
        ap = (C.c_void_p * S)(*[b.ctypes.data for b in bufs])
        out = np.zeros((S, 192), np.float32)
        _chk(lib().nasr_diar_embed(self.h, S, ap, ln, out.ctypes.data_as(C.POINTER(C.c_float)), 0))
        return out


class Engine:
    """weights: dict name -> ndarray (float32, or (type_id, raw bytes ndarray, shape) for quantised)."""

    def __init__(self, weights: dict, n_layers=24, dtype=DTYPE_BF16, max_streams=1, kernel_size=9,
                 num_prompts=0, device=0):
        L = lib()
        self.n_layers = n_layers
        hp = default_hparams(n_layers, kernel_size, num_prompts)
        keep, descs = weight_descs(weights)
        h = C.c_void_p()
        _chk(L.nasr_engine_create(C.byref(h), device, dtype, C.byref(hp), descs, len(weights), max_streams))
        self.h = h
        self._dev_allocs = []

    def close(self):
        if getattr(self, "h", None):
            lib().nasr_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def lend_stream(self) -> int:
        """the last of the engine's side-by-side HIP streams for another GPU client (Diar.set_stream); close that client first"""
        out = C.c_void_p()
        _chk(lib().nasr_engine_lend_stream(self.h, C.byref(out)))
        return out.value

    def stream(self, right_context=0, prompt_index=-1) -> Stream:
        return Stream(self, right_context, prompt_index)

    def set_option(self, key: str, value: int):
        _chk(lib().nasr_engine_set_option(self.h, key.encode(), int(value)))

    def set_debug(self, on=True):
        _chk(lib().nasr_engine_set_debug(self.h, int(on)))

    def counter(self, name: str) -> int:
        v = C.c_int64()
        _chk(lib().nasr_engine_get_counter(self.h, name.encode(), C.byref(v)))
        return v.value

    # ---- batched calls ---------------------------------------------------------------
    @staticmethod
    def _handles(streams):
        return (C.c_void_p * len(streams))(*[s.h for s in streams])

    def _tok_bufs(self, B, cap):
        bufs = [np.zeros(cap, np.int32) for _ in range(B)]
        ptrs = (C.c_void_p * B)(*[b.ctypes.data for b in bufs])
        caps = (C.c_int32 * B)(*([cap] * B))
        n = (C.c_int32 * B)()
        return bufs, ptrs, caps, n

    def _gather(self, streams, bufs, tptrs, caps, n, cap):
        """tokens of the call just made; a full buffer means more may be queued on the stream: collect until drained"""
        B = len(streams)
        out = [bufs[b][:n[b]].tolist() for b in range(B)]
        while any(n[b] >= cap for b in range(B)):
            _chk(lib().nasr_engine_collect(self.h, self._handles(streams), B, tptrs, caps, n))
            for b in range(B):
                out[b] += bufs[b][:n[b]].tolist()
        return out

    def step(self, streams, pcms, flags=0, tok_cap=None):
        """pcms: list of int16 ndarrays (host) or list of (device_ptr, n) when FLAG_PCM_DEVICE."""
        B = len(streams)
        if flags & FLAG_PCM_DEVICE:
            ptrs = (C.c_void_p * B)(*[p for p, _ in pcms])
            ns = (C.c_int32 * B)(*[n for _, n in pcms])
            total = max(n for _, n in pcms)
        else:
            arrs = [np.ascontiguousarray(p, np.int16) for p in pcms]
            ptrs = (C.c_void_p * B)(*[a.ctypes.data for a in arrs])
            ns = (C.c_int32 * B)(*[a.size for a in arrs])
            total = max(a.size for a in arrs)
        cap = tok_cap or (total // 1280 + 16) * 10
        bufs, tptrs, caps, n = self._tok_bufs(B, cap)
        _chk(lib().nasr_engine_step(self.h, self._handles(streams), B, ptrs, ns, tptrs, caps, n, flags))
        return self._gather(streams, bufs, tptrs, caps, n, cap) if not flags & FLAG_NO_SYNC else [[] for _ in range(B)]

    def step_mel(self, streams, mels, flags=0):
        B = len(streams)
        arrs = [np.ascontiguousarray(m, np.float32) for m in mels]
        ptrs = (C.c_void_p * B)(*[a.ctypes.data for a in arrs])
        ns = (C.c_int32 * B)(*[a.shape[0] for a in arrs])
        cap = (max(a.shape[0] for a in arrs) // 8 + 16) * 10
        bufs, tptrs, caps, n = self._tok_bufs(B, cap)
        _chk(lib().nasr_engine_step_mel(self.h, self._handles(streams), B, ptrs, ns, tptrs, caps, n, flags))
        return self._gather(streams, bufs, tptrs, caps, n, cap) if not flags & FLAG_NO_SYNC else [[] for _ in range(B)]

    def finalize(self, streams):
        B = len(streams)
        bufs, tptrs, caps, n = self._tok_bufs(B, 256)
        _chk(lib().nasr_engine_finalize(self.h, self._handles(streams), B, tptrs, caps, n))
        return self._gather(streams, bufs, tptrs, caps, n, 256)

    def collect(self, streams, cap=4096):
        B = len(streams)
        bufs, tptrs, caps, n = self._tok_bufs(B, cap)
        _chk(lib().nasr_engine_collect(self.h, self._handles(streams), B, tptrs, caps, n))
        return self._gather(streams, bufs, tptrs, caps, n, cap)

    # ---- measurement ------------------------------------------------------------------
    def profile(self, on=True):
        _chk(lib().nasr_engine_profile(self.h, int(on)))

    def profile_read(self):
        arr = (KernelStat * 64)()
        n = _chk(lib().nasr_engine_profile_read(self.h, arr, 64))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms,
                     bytes=arr[i].bytes, flops=arr[i].flops) for i in range(min(n, 64))]

    def synchronize(self):
        _chk(lib().nasr_engine_synchronize(self.h))

    def upload(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        _chk(lib().nasr_device_alloc(self.h, C.byref(p), arr.nbytes))
        _chk(lib().nasr_device_upload(self.h, p, arr.ctypes.data, arr.nbytes))
        self._dev_allocs.append(p)
        return p.value

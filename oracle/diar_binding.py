"""ctypes binding of the diarization oracle (oracle/diar_oracle.c) and of the compiled reference front end
(oracle/_ref: src/diarize_audio.cpp).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import binding as _b

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_ready = False


def _f(a):
    return a.ctypes.data_as(_fp)


def lib():
    global _ready
    L = _b.lib()
    if not _ready:
        L.dorc_logmel.argtypes = [_fp, C.c_int, C.c_int, _fp, _fp, _fp, C.c_int, _ip]
        L.dorc_model_create.restype = C.c_void_p
        L.dorc_model_free.argtypes = [C.c_void_p]
        L.dorc_model_set_tensor.argtypes = [C.c_void_p, C.c_char_p, _fp, C.c_longlong]
        L.dorc_model_finalize.argtypes = [C.c_void_p]
        L.dorc_vad_window.restype = C.c_float
        L.dorc_vad_window.argtypes = [C.c_void_p, _fp, C.c_int]
        L.dorc_vad_batch.argtypes = [C.c_void_p, _fp, C.c_int, _fp, C.c_int]
        L.dorc_spk_embed.argtypes = [C.c_void_p, _fp, C.c_int, _fp]
        _ready = True
    return L


def logmel(audio: np.ndarray, fb: np.ndarray, window: np.ndarray, normalize: bool):
    """-> (mel [80][t_padded], t_valid)"""
    a = np.ascontiguousarray(audio, np.float32)
    cap = a.size // 160 + 32
    out = np.zeros((80, cap), np.float32)
    tv = C.c_int()
    tp = lib().dorc_logmel(_f(a), a.size, int(normalize), _f(np.ascontiguousarray(fb, np.float32)),
                           _f(np.ascontiguousarray(window, np.float32)), _f(out), cap, C.byref(tv))
    assert tp >= 0
    return out.reshape(-1)[:80 * tp].reshape(80, tp).copy(), tv.value


def ref_logmel(audio: np.ndarray, fb: np.ndarray, window: np.ndarray, normalize: bool):
    """the reference's own diarize_compute_logmel (oracle/_ref), same return convention"""
    R = _b.ref()
    R.ref_diar_logmel.argtypes = [_fp, C.c_int, C.c_int, _fp, _fp, _fp, C.c_int, _ip]
    a = np.ascontiguousarray(audio, np.float32)
    cap = a.size // 160 + 32
    out = np.zeros(80 * cap, np.float32)
    tv = C.c_int()
    tp = R.ref_diar_logmel(_f(a), a.size, int(normalize), _f(np.ascontiguousarray(fb, np.float32)),
                           _f(np.ascontiguousarray(window, np.float32)), _f(out), cap, C.byref(tv))
    assert tp >= 0
    return out[:80 * tp].reshape(80, tp).copy(), tv.value


class DiarModel:
    def __init__(self, weights: dict):
        L = lib()
        self.h = L.dorc_model_create()
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, np.float32)
            rc = L.dorc_model_set_tensor(self.h, name.encode(), _f(a), a.size)
            assert rc >= 0, name
        assert L.dorc_model_finalize(self.h) == 0, "diar oracle: tensors missing"

    def __del__(self):
        if getattr(self, "h", None):
            lib().dorc_model_free(self.h)
            self.h = None

    def vad_window(self, audio: np.ndarray, lens_samples: int = 10080) -> float:
        a = np.ascontiguousarray(audio, np.float32)
        assert a.size >= 10080
        return float(lib().dorc_vad_window(self.h, _f(a), lens_samples))

    def vad_batch(self, audio: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(audio, np.float32)
        n = max(0, 1 + (a.size - 10080) // 160) if a.size >= 10080 else 0
        out = np.zeros(max(n, 1), np.float32)
        got = lib().dorc_vad_batch(self.h, _f(a), a.size, _f(out), n)
        assert got == n
        return out[:n]

    def spk_embed(self, audio: np.ndarray, lens_samples: int = 24000) -> np.ndarray:
        a = np.ascontiguousarray(audio, np.float32)
        assert a.size >= 24000
        out = np.zeros(192, np.float32)
        assert lib().dorc_spk_embed(self.h, _f(a), lens_samples, _f(out)) == 0
        return out

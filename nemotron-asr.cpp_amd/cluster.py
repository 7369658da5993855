"""ctypes face of libnasr_cluster.so: NME-SC speaker clustering (host/diarize_cluster_amd.{h,cpp}), the CPU stage of the
diarization side-car (reference: src/diarize_cluster.h:14-51, nmesc_cluster / nmesc_cosine_affinity)."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = Path(__file__).resolve().parent / "libnasr_cluster.so"
        if not path.exists():
            raise RuntimeError(f"{path} not built: run `python __graft_entry__.py` (make -C nemotron-asr.cpp_amd/host)")
        L = C.CDLL(str(path))
        fp, ip, dp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        L.nasr_nmesc_affinity.argtypes = [fp, C.c_int, C.c_int, fp]
        L.nasr_nmesc_cluster.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_uint64, ip, ip, ip]
        L.nasr_sym_eigen.argtypes = [dp, C.c_int, dp, dp]
        _LIB = L
    return _LIB


def cosine_affinity(emb: np.ndarray) -> np.ndarray:
    e = np.ascontiguousarray(emb, np.float32)
    out = np.zeros((e.shape[0], e.shape[0]), np.float32)
    fp = C.POINTER(C.c_float)
    if lib().nasr_nmesc_affinity(e.ctypes.data_as(fp), e.shape[0], e.shape[1], out.ctypes.data_as(fp)) < 0:
        raise ValueError("nasr_nmesc_affinity: bad arguments")
    return out


def nmesc_cluster(emb: np.ndarray, max_num_speakers=8, max_rp_threshold=0.25, sparse_search_volume=30, nme_mat_size=512,
                  oracle_num_speakers=-1, kmeans_seed=0):
    """-> (labels [N] int32, est_num_speakers, p_hat)"""
    e = np.ascontiguousarray(emb, np.float32)
    labels = np.zeros(e.shape[0], np.int32)
    est, p_hat = C.c_int32(0), C.c_int32(0)
    rc = lib().nasr_nmesc_cluster(e.ctypes.data_as(C.POINTER(C.c_float)), e.shape[0], e.shape[1], max_num_speakers,
                                  max_rp_threshold, sparse_search_volume, nme_mat_size, oracle_num_speakers, kmeans_seed,
                                  labels.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(est), C.byref(p_hat))
    if rc < 0:
        raise ValueError("nasr_nmesc_cluster: bad arguments")
    return labels, est.value, p_hat.value


def sym_eigen(a: np.ndarray, vectors=True):
    m = np.ascontiguousarray(a, np.float64)
    n = m.shape[0]
    val = np.zeros(n)
    vec = np.zeros((n, n)) if vectors else None
    dp = C.POINTER(C.c_double)
    lib().nasr_sym_eigen(m.ctypes.data_as(dp), n, val.ctypes.data_as(dp), vec.ctypes.data_as(dp) if vectors else None)
    return val, vec

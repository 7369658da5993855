"""Seeded inputs shared by gen_golden.py and the tests (data, not reference code)."""
import numpy as np

from nemotron_asr_amd import synth

PIECES = [180, 333, 1, 2000, 512, 159, 4000, 8815]   # sums to 16000; 180 = tests/test_preprocessor.cpp:90
DEC_TOKENS = [1024, 0, 100, 500]                     # reference tests/test_compute.cpp:2407


def pcm():
    return synth.make_pcm(0, 1.0)


def split_pcm(p):
    out, o = [], 0
    for n in PIECES:
        out.append(p[o:o + n])
        o += n
    assert o == p.size
    return out


def mel_chunk(mel, n):
    """n mel frames: the golden mel, repeated if shorter."""
    reps = -(-n // mel.shape[0])
    return np.ascontiguousarray(np.concatenate([mel] * reps)[:n])


def _normal(seed, shape):
    n = int(np.prod(shape))
    u1 = synth.uniform01(seed, n, 0)
    u2 = synth.uniform01(seed, n, n)
    z = np.sqrt(-2.0 * np.log(np.maximum(u1, 2.0 ** -53))) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32).reshape(shape)


def layer_input(T):
    return _normal(0x1A7E0000 + T, (T, 1024))


def enc_frames(n):
    return _normal(0xE7C0DE, (n, 1024))


def rel_shift_input(heads, qlen):
    """the reference's own rel-shift test pattern (tests/test_compute.cpp:1028-1038): in[h][i][p] = 100 h + 10 i + p"""
    h, i, p = np.meshgrid(np.arange(heads), np.arange(qlen), np.arange(2 * qlen - 1), indexing="ij")
    return (100.0 * h + 10.0 * i + p).astype(np.float32)

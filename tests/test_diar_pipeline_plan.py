"""Host logic of the diarization pipeline without a GPU (host/diarize_pipeline_amd.cpp, nasr_diar_plan): the onset / offset
state machine and the sub-segment cursor over a given probability track, against a restatement of the reference's
frame-by-frame loop (src/diarize_pipeline.cpp:198-263, :341-363) fed one sample at a time -- many random tracks,
several parameter sets, edge cases (no speech, speech to the end, short blips, segments shorter than a sub-segment)."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

LIB = Path(__file__).resolve().parent.parent / "nemotron-asr.cpp_amd" / "libnasr_diarize_host.so"


def restatement(probs, total, onset, offset, min_off=60, shift=12000, window=24000, min_seg=8000):
    """try_advance / finalize_open_segment of the reference: when VAD frame f is evaluated the audio reaches the end of its
    window, f * 160 + 10080."""
    segs, subs = [], []
    in_speech, off_run, start_f, k, seg_id, next_id = False, 0, -1, 0, -1, 0

    def tail(seg_end, at_eof):
        nonlocal k
        seg_start = start_f * 160
        covered = seg_start + ((k - 1) * shift + window if k > 0 else 0)
        left = seg_end - covered
        if left >= min_seg and (k > 0 or at_eof):
            subs.append((seg_id, covered, min(left, 24000))); k += 1
        elif k == 0 and seg_end - seg_start >= min_seg:
            subs.append((seg_id, seg_start, min(seg_end - seg_start, 24000))); k += 1

    for f, p in enumerate(probs):
        if not in_speech:
            if p >= onset:
                in_speech, seg_id, start_f, k, off_run = True, next_id, f, 0, 0
                next_id += 1
        elif p < offset:
            off_run += 1
            if off_run >= min_off:
                end_f = max(f + 1 - off_run, start_f)
                tail(end_f * 160, False)
                segs.append((start_f, end_f))
                in_speech, off_run = False, 0
        else:
            off_run = 0
        if in_speech:
            while start_f * 160 + k * shift + window <= f * 160 + 10080:
                subs.append((seg_id, start_f * 160 + k * shift, window)); k += 1
    if in_speech:
        tail(min(len(probs) * 160, total), True)
        segs.append((start_f, len(probs)))
    return segs, subs


def plan(probs, total, onset, offset, min_off_sec=0.6, window_sec=1.5, shift_sec=0.75, min_seg_sec=0.5):
    L = C.CDLL(str(LIB))
    ll = C.POINTER(C.c_longlong)
    L.nasr_diar_plan.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_longlong, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_float, ll, C.c_int, C.POINTER(C.c_int), ll, C.c_int, C.POINTER(C.c_int)]
    p = np.ascontiguousarray(probs, np.float32)
    cap = p.size + 8
    segs, subs = np.zeros(2 * cap, np.int64), np.zeros(3 * cap, np.int64)
    ns, nb = C.c_int(), C.c_int()
    rc = L.nasr_diar_plan(p.ctypes.data_as(C.POINTER(C.c_float)), p.size, total, onset, offset, min_off_sec, window_sec, shift_sec,
                          min_seg_sec, segs.ctypes.data_as(ll), cap, C.byref(ns), subs.ctypes.data_as(ll), cap, C.byref(nb))
    assert rc == 0
    return ([tuple(int(v) for v in segs[2 * i:2 * i + 2]) for i in range(ns.value)],
            [tuple(int(v) for v in subs[3 * i:3 * i + 3]) for i in range(nb.value)])


def _track(seed, n):
    """speech-like: stretches of high / low probability of random length with noise, so that thresholds are crossed often"""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        level = rng.choice([0.05, 0.3, 0.7, 0.97])
        out += list(np.clip(level + 0.08 * rng.standard_normal(int(rng.integers(5, 400))), 0, 1))
    return np.asarray(out[:n], np.float32)


@pytest.mark.skipif(not LIB.exists(), reason="libnasr_diarize_host.so not built (python __graft_entry__.py)")
@pytest.mark.parametrize("seed", range(12))
def test_plan_matches_reference_loop_on_random_tracks(seed):
    n = 600 + 211 * seed
    probs = _track(seed, n)
    total = n * 160 + 10080 - 160 + (seed * 37) % 160                     # the samples that make n windows (+ a ragged tail)
    for onset, offset in ((0.9, 0.5), (0.5, 0.5), (0.6, 0.3)):
        want = restatement(probs, total, np.float32(onset), np.float32(offset))
        got = plan(probs, total, onset, offset)
        assert got[0] == want[0] and got[1] == want[1]
    assert len(plan(probs, total, 0.5, 0.5)[0]) >= 1


@pytest.mark.skipif(not LIB.exists(), reason="libnasr_diarize_host.so not built (python __graft_entry__.py)")
def test_plan_edge_cases_and_parameters():
    z = np.zeros(500, np.float32)
    assert plan(z, 500 * 160 + 9920, 0.9, 0.5) == ([], [])                 # no speech
    one = np.ones(500, np.float32)
    segs, subs = plan(one, 500 * 160 + 9920, 0.9, 0.5)                     # speech to the end: closed by finalize, tail sub-segment
    assert segs == [(0, 500)] and subs == restatement(one, 500 * 160 + 9920, 0.9, 0.5)[1] and subs[-1][1] + subs[-1][2] <= 500 * 160 + 9920
    # 0.3 s of speech: the segment stays open through the 0.6 s hang-over, long enough for the buffered audio to reach the end
    # of a full 1.5 s window from the segment's start -- the reference's loop emits it (src/diarize_pipeline.cpp:253-263)
    blip = z.copy(); blip[100:130] = 1.0
    assert plan(blip, 500 * 160 + 9920, 0.9, 0.5) == ([(100, 130)], [(0, 16000, 24000)]) == restatement(blip, 500 * 160 + 9920, 0.9, 0.5)
    # the same blip right before the end of the audio: no full window fits, the 0.3 s are below min_seg -> no sub-segment
    late = z.copy(); late[460:490] = 1.0
    assert plan(late, 500 * 160 + 9920, 0.9, 0.5) == ([(460, 500)], [(0, 73600, 6400)][:0] + restatement(late, 500 * 160 + 9920, 0.9, 0.5)[1])
    # 0.9 s of speech right before the end: one masked-pad sub-segment of what is left
    short = z.copy(); short[400:490] = 1.0
    assert plan(short, 500 * 160 + 9920, 0.9, 0.5) == restatement(short, 500 * 160 + 9920, 0.9, 0.5)
    assert len(plan(short, 500 * 160 + 9920, 0.9, 0.5)[1]) == 1
    probs = _track(99, 3000)
    total = 3000 * 160 + 9920
    for kw in (dict(shift_sec=0.5), dict(window_sec=1.0, shift_sec=0.25), dict(min_off_sec=0.2), dict(min_seg_sec=1.0)):
        r = dict(min_off=int(np.ceil(np.float32(kw.get("min_off_sec", 0.6)) / np.float32(0.01))), shift=round(kw.get("shift_sec", 0.75) * 16000),
                 window=round(kw.get("window_sec", 1.5) * 16000), min_seg=round(kw.get("min_seg_sec", 0.5) * 16000))
        assert plan(probs, total, 0.7, 0.4, **kw) == restatement(probs, total, np.float32(0.7), np.float32(0.4), **r)

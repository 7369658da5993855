"""Host-side restatement of the index arithmetic of round 6's TitaNet-L segment-tile kernels (nemotron-asr.cpp_amd/csrc/kernels_spk.hip): no GPU, no compute --
the maps the kernels' correctness rests on, checked for being bijections, staying inside their LDS allocation and waiting for the right chunk.  The constants are READ
from the source, so an edit of the kernel that breaks one of these relations fails here before it produces wrong embeddings on a GPU."""
import itertools
import re
from pathlib import Path

import numpy as np
import pytest

SRC = (Path(__file__).resolve().parent.parent / "nemotron-asr.cpp_amd" / "csrc" / "kernels_spk.hip").read_text()
HDR = (Path(__file__).resolve().parent.parent / "nemotron-asr.cpp_amd" / "csrc" / "nasr_internal.h").read_text()


def _named(text, name):
    m = re.search(rf"\b{name} = (\d+)\b", text)
    assert m, name
    return int(m.group(1))


SPK_T, SPK_TVALID = _named(HDR, "SPK_T"), _named(HDR, "SPK_TVALID")
SG_BN, SG_MT, SG_NT, SG_NW, SG_NS, SG_SLD = (_named(SRC, n) for n in ("SG_BN", "SG_MT", "SG_NT", "SG_NW", "SG_NS", "SG_SLD"))
SG_BM = SPK_T
SG_SLOT = (SG_BM + SG_BN) * 64
SG_NP, SG_PIECES = SG_BM // 16, SG_BM // 16 + SG_BN // 16
SG_DMA = (SG_PIECES + SG_NW - 1) // SG_NW
LDS = SG_NS * SG_SLOT


def test_tile_is_one_sub_segment_and_two_workgroups_fit_a_cu():
    assert SG_BM == 160 and SPK_TVALID <= SG_BM and SG_BN == 128 and SG_NW == 4
    assert SG_MT * 16 * 2 == SG_BM and SG_NT * 16 * 2 == SG_BN            # LOOP 0: 2 row halves x 2 column halves of 80 x 64 per wave
    assert 2 * LDS <= 160 * 1024                                           # two workgroups per CU (the point of the four-wave form)
    stage_and_red = (SG_BM * SG_SLD + 3 * 512) * 4
    assert stage_and_red <= LDS and SG_SLD >= 64 + 4
    assert "constexpr int SG_RED_FLOATS = 3 * 512;" in SRC and "__launch_bounds__(SG_THREADS, 2)" in SRC


def test_loop0_dma_pieces_cover_a_chunk_exactly():
    """instruction j = wave * SG_DMA + u clamped to the last piece: 10 panel pieces (16 rows x 64 B) + 8 weight tiles of 1 KiB, each written to its own place"""
    dst = {}
    for wave, u in itertools.product(range(SG_NW), range(SG_DMA)):
        j = min(wave * SG_DMA + u, SG_PIECES - 1)
        dst.setdefault(j, set()).add(j * 1024 if j < SG_NP else SG_BM * 64 + (j - SG_NP) * 1024)
    assert sorted(dst) == list(range(SG_PIECES)) and all(len(v) == 1 for v in dst.values())
    assert sorted(next(iter(v)) for v in dst.values()) == list(range(0, SG_SLOT, 1024))


def test_loop1_activation_pieces_and_weight_tiles():
    """LOOP 1: the LDS ring carries only the panel (10 pieces; wave w issues 3 w .. 3 w + 2 clamped to 9), seven slots of 10 KiB, six chunks ahead; a wave's two weight
    fragments are tiles ng * 8 + 2 w + j of the packed weights: the four waves cover the tile's eight 16-column weight tiles exactly once"""
    assert "constexpr int D = 6, NSA = 7, ASLOT = SG_BM * 64, GROUP = 5;" in SRC      # the values restated below are the kernel's
    D, NSA, ASLOT, GROUP = 6, 7, SG_BM * 64, 5
    pieces = {}
    for wave, u in itertools.product(range(4), range(3)):
        j = min(wave * 3 + u, SG_NP - 1)
        pieces.setdefault(j, set()).add(j * 1024)
    assert sorted(pieces) == list(range(SG_NP)) and NSA * ASLOT <= LDS and NSA == D + 1
    assert sorted(2 * w + j for w in range(4) for j in range(2)) == list(range(SG_BN // 16))
    # slot of chunk i + D == slot of chunk i - 1 (free after the barrier of iteration i); register set of chunk i + D is not one of the sets in use
    for i in range(1, 64):
        assert (i + D) % NSA == (i - 1) % NSA
        assert (i + D) & 7 not in {(i + k) & 7 for k in range(D)}
    # the eight SG_BODY invocations of an unrolled round name (set, nset = (set + D) & 7)
    body = re.search(r"SG_BODY\(i0, 0, (\d)\); SG_BODY\(i0 \+ 1, 1, (\d)\); SG_BODY\(i0 \+ 2, 2, (\d)\); SG_BODY\(i0 \+ 3, 3, (\d)\);\s*SG_BODY\(i0 \+ 4, 4, (\d)\); SG_BODY\(i0 \+ 5, 5, (\d)\); SG_BODY\(i0 \+ 6, 6, (\d)\); SG_BODY\(i0 \+ 7, 7, (\d)\);", SRC)
    assert body and [int(x) for x in body.groups()] == [(s + D) & 7 for s in range(8)]
    assert GROUP == 3 + 2


@pytest.mark.parametrize("loop,P,group,nchunks", [(0, SG_NS - 1, SG_DMA, 32), (0, SG_NS - 1, SG_DMA, 4), (0, SG_NS - 1, SG_DMA, 96), (1, 6, 5, 32), (1, 6, 5, 8), (1, 6, 5, 96)])
def test_counted_waits(loop, P, group, nchunks):
    """Both loops: P chunks issued in the prologue, chunk i + P during iteration i (after the barrier), every chunk = `group` VMEM instructions per wave in a fixed order.
    At the top of iteration i the wave waits for vmcnt(group * min(left, P - 1)), left = nchunks - 1 - i: chunk i must be among the completed ones, and the kernel's
    immediates must cover every value of min(left, P - 1) that occurs."""
    issued = min(P, nchunks)
    for i in range(nchunks):
        allow = min(nchunks - 1 - i, P - 1)
        assert issued - allow - 1 >= i, (i, issued, allow)                 # in-order completion: all but the youngest `allow` chunks are done
        assert allow <= issued - i - 1                                     # never waits for more than is outstanding
        if i + P < nchunks:
            issued += 1
    assert issued == nchunks
    if loop == 0:
        assert P - 1 == 2 and 'else if (left == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SG_DMA) : "memory");' in SRC
    else:
        for k in (4, 3, 2):
            assert f'else if (left == {k}) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GROUP * {k}) : "memory");' in SRC
        assert group * (P - 1) <= 63                                       # vmcnt is a 6-bit field


def _park(loop):
    """rows / columns of the 160 x 128 tile each (wave, accumulator, lane, register) parks in which column half: -> coverage count [2][160][64]"""
    seen = np.zeros((2, SG_BM, 64), np.int32)
    for wave, lane in itertools.product(range(4), range(64)):
        q, r = lane >> 4, lane & 15
        if loop == 0:
            nq, mh = wave & 1, wave >> 1
            cq, col0, row0, PJ, PM = nq, 0, mh * (SG_BM // 2), SG_NT, SG_MT
        else:
            cq, col0, row0, PJ, PM = wave >> 1, (wave & 1) * 32, 0, 2, 10
        for j, mt, i in itertools.product(range(PJ), range(PM), range(4)):
            seen[cq, row0 + mt * 16 + r, col0 + j * 16 + q * 4 + i] += 1
    return seen


@pytest.mark.parametrize("loop", [0, 1])
def test_parking_covers_the_tile_exactly_once(loop):
    assert (_park(loop) == 1).all()
    assert "constexpr int PJ = LOOP == 0 ? SG_NT : 2, PM = LOOP == 0 ? SG_MT : 10;" in SRC


@pytest.mark.parametrize("NT", [256, 512])
@pytest.mark.parametrize("KS", [3, 7, 11, 15])
def test_epilogue_thread_maps_cover_the_stage(NT, KS):
    """sg_dw_pass: thread = (channel pair, group of frames), 10 frames per round; sg_combine / the Y store: items of four columns; column reductions: 160 / (NT / 64) frames per group"""
    TG, rounds = 10, SG_BM // (NT // 32) // 10
    assert rounds * TG * (NT // 32) == SG_BM
    out = np.zeros((SG_BM, 64), np.int32)
    for tid in range(NT):
        cp = tid & 31
        for rd in range(rounds):
            t0 = ((tid >> 5) * rounds + rd) * TG
            pad = (KS - 1) // 2
            window = [t0 + j - pad for j in range(TG + KS - 1)]
            assert window[0] == t0 - pad and window[-1] == t0 + TG - 1 + pad           # every tap of every output frame is inside the register window
            for u in range(TG):
                out[t0 + u, 2 * cp:2 * cp + 2] += 1
    assert (out == 1).all()
    per = SG_BM * 16 // NT
    assert per % 5 == 0                                                                 # sg_combine reads Y five items at a time
    items = np.zeros((SG_BM, 16), np.int32)
    for tid, k in itertools.product(range(NT), range(per)):
        e = tid + k * NT
        items[e >> 4, e & 15] += 1
        assert (e & 15) == (tid & 15)                                                   # a thread keeps its four columns: the gate is loaded once per thread
    assert (items == 1).all()
    RG = SG_BM // (NT // 64)
    rows = sorted(t for g in range(NT // 64) for t in range(g * RG, g * RG + RG))
    assert rows == list(range(SG_BM))


def test_small_linear_k_slices():
    """spk_fc_slices: Z doubles while the launch has fewer than 256 workgroups and every wave keeps at least one 16-deep group; K < 1024 is never split"""
    def slices(M, K, N):
        KG, wgs = K // 16, (N // 16) * ((M + 63) // 64)
        if K < 1024:
            return 1
        Z = 1
        while Z < 16 and wgs * Z < 256 and KG % (8 * Z) == 0:
            Z *= 2
        return Z
    assert "if (K < 1024) return 1;" in SRC and "while (Z < 16 && wgs * Z < 256 && KG % (8 * Z) == 0) Z *= 2;" in SRC
    for M, K, N in [(96, 1024, 128), (96, 128, 1024), (96, 3072, 384), (96, 384, 3072), (96, 6144, 128), (96, 6144, 192), (1, 1024, 128), (300, 6144, 192)]:
        Z = slices(M, K, N)
        KG = K // 16
        assert KG % (4 * Z) == 0 and KG // (4 * Z) >= 1, (M, K, N, Z)                   # per_wave = KG / (4 Z) whole groups
        assert Z * M * N <= 16 * M * 512                                                # the scratch nasr_diar.hip allocates (16 x S x 512 floats)

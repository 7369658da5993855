"""Host-side restatement of the index arithmetic of round 5's GEMM kernels (nemotron-asr.cpp_amd/csrc/kernels_gemm.hip): no GPU, no compute -- the maps a kernel's
correctness rests on, checked for being bijections / staying inside their LDS allocation.  The bug this file would have caught in the round: the ring of the
224 x 192 tile (five slots = 133 120 B) was smaller than the eight wave-private epilogue regions (139 264 B) -- wrong tokens on the GPU, found by a token count."""
import itertools
import re
from pathlib import Path

import numpy as np
import pytest

SRC = (Path(__file__).resolve().parent.parent / "nemotron-asr.cpp_amd" / "csrc" / "kernels_gemm.hip").read_text()


def _const(name):
    m = re.search(rf"constexpr int {name} = (\d+);", SRC)
    assert m, name
    return int(m.group(1))


W2_NS, T3_NS, WE_LD, K32_SLOT = _const("W2_NS"), _const("T3_NS"), _const("WE_LD"), _const("K32_SLOT")


def wide_cfg(BN, MT):          # struct WideCfg
    BM = 32 * MT
    return dict(BM=BM, SLOT=(BM + BN) * 64, NT=BN // 64, NP=BM // 16, PIECES=BM // 16 + BN // 16, DMA=(BM // 16 + BN // 16 + 7) // 8)


@pytest.mark.parametrize("BN,MT", [(256, 7), (192, 7)])
def test_wide2_lds_holds_ring_and_epilogue_regions(BN, MT):
    c = wide_cfg(BN, MT)
    ring = W2_NS * c["SLOT"]
    regions = 8 * 64 * WE_LD * 4                      # eight waves x 64 rows x WE_LD floats (k_gemm_wide2's epilogue: stg = ring + wave * 64 * WE_LD)
    lds = max(ring, regions)                          # wide2_lds<BN, MT>()
    assert "return W2_NS * WideCfg<BN, MT>::SLOT > 8 * 64 * 68 * 4 ? W2_NS * WideCfg<BN, MT>::SLOT : 8 * 64 * 68 * 4;" in SRC and WE_LD == 68
    assert lds <= 160 * 1024                          # one workgroup per CU
    assert c["NT"] * 16 <= WE_LD - 4                  # a wave's block (NT x 16 columns) fits a staged row
    assert MT >= c["DMA"] + 0 and MT - 1 >= c["DMA"]  # one DMA instruction after each of the first DMA MFMA groups (static_assert in the kernel)


@pytest.mark.parametrize("BN,MT", [(256, 7), (192, 7), (256, 8)])
def test_dma_pieces_cover_a_chunk_exactly(BN, MT):
    """instruction j = wave * DMA + u, clamped to the last piece: every 1 KiB piece of a chunk (panel rows, then weight tiles) is written, to its own place."""
    c = wide_cfg(BN, MT)
    dst = {}
    for wave, u in itertools.product(range(8), range(c["DMA"])):
        j = min(wave * c["DMA"] + u, c["PIECES"] - 1)
        d = j * 1024 if j < c["NP"] else c["BM"] * 64 + (j - c["NP"]) * 1024
        dst.setdefault(j, set()).add(d)
    assert sorted(dst) == list(range(c["PIECES"])) and all(len(v) == 1 for v in dst.values())
    offs = sorted(next(iter(v)) for v in dst.values())
    assert offs == list(range(0, c["SLOT"], 1024))   # contiguous, no overlap, fills the slot


@pytest.mark.parametrize("COLS,rows", [(64, 64), (64, 48), (48, 64), (48, 48), (64, 32)])
def test_wave_epilogue_rows_touch_every_element_once(COLS, rows):
    """wave_epilogue_rows<COLS>: items of eight columns (16-bit outputs) / four columns (f32) / the residual form's batches of eight instructions."""
    for per, width in ((COLS // 8, 8), (COLS // 4, 4)):
        seen = np.zeros((rows, COLS), np.int32)
        for lane in range(64):
            for e in range(lane, rows * per, 64):
                row, c = e // per, (e - (e // per) * per) * width
                seen[row, c:c + width] += 1
        assert (seen == 1).all()
    per = COLS // 4
    seen = np.zeros((rows, COLS), np.int32)
    for lane in range(64):
        for e0 in range(lane, rows * per, 8 * 64):
            for k in range(8):
                e = e0 + k * 64
                if e >= rows * per:
                    continue
                row, c4 = e // per, (e - (e // per) * per) * 4
                seen[row, c4:c4 + 4] += 1
    assert (seen == 1).all()


def test_wave_blocks_tile_the_224_row_tile():
    """k_gemm_wide2: wave = (mh, nq); its block is rows mh * 112 + part * 64 + [0, 64 or 48), columns nq * NT * 16 + [0, NT * 16): the eight blocks x two parts cover the tile once."""
    for BN in (256, 192):
        NT, MT = BN // 64, 7
        seen = np.zeros((224, BN), np.int32)
        for wave in range(8):
            nq, mh = wave & 3, wave >> 2
            for part in range(2):
                nmt = MT - 4 if part else 4
                r0, c0 = mh * 112 + part * 64, nq * NT * 16
                seen[r0:r0 + nmt * 16, c0:c0 + NT * 16] += 1
        assert (seen == 1).all()


def tile_of(i, n_groups, m_chunks, bands):          # tile_of() of the kernels, one K slice
    if bands == 2 or (bands == 0 and m_chunks <= 4):
        return i % m_chunks, i // m_chunks
    w = 8 if n_groups % 8 == 0 else 4 if n_groups % 4 == 0 else 2 if n_groups % 2 == 0 else 1
    band, ib = divmod(i, w * m_chunks)
    return ib // w, band * w + ib % w


@pytest.mark.parametrize("n_groups,m_chunks", [(16, 32), (16, 16), (12, 32), (4, 32), (8, 7), (32, 7), (24, 7), (16, 7), (12, 17)])
def test_tile_order_and_xcd_remap_are_bijections(n_groups, m_chunks):
    n = n_groups * m_chunks
    for bands in (0, 1, 2):
        tiles = {tile_of(i, n_groups, m_chunks, bands) for i in range(n)}
        assert len(tiles) == n and all(0 <= mc < m_chunks and 0 <= ng < n_groups for mc, ng in tiles)
    qd, rm = n >> 3, n & 7                            # the remap at the top of every kernel: ids contiguous per XCD
    ids = set()
    for b in range(n):
        xcd, loc = b & 7, b >> 3
        ids.add((xcd * (qd + 1) if xcd < rm else rm * (qd + 1) + (xcd - rm) * qd) + loc)
    assert ids == set(range(n))


def test_tiled3_ring_holds_the_staged_tile_and_two_fit_a_cu():
    assert T3_NS * K32_SLOT >= 128 * 132 * 4          # staged_epilogue: f32 tile [128][STG_LD = 132]
    assert 2 * T3_NS * K32_SLOT <= 160 * 1024         # two workgroups per CU


@pytest.mark.parametrize("NS,DMA,nchunks", [(5, 4, 32), (5, 4, 8), (5, 2, 6), (5, 2, 128), (5, 4, 136)])
def test_counted_waits_of_the_new_loops(NS, DMA, nchunks):
    """k_gemm_wide2 / k_gemm_tiled3: at the top of iteration i (i + 1 < nchunks) a wave waits until all but `allow` of its DMA instructions are done; chunk i + 1 must be among
    the completed ones and the count must never exceed what is outstanding.  Issue order: chunks 0 .. NS - 1 in the prologue, chunk i + NS during iteration i."""
    issued = NS                                       # chunks issued so far (all DMA instructions of a chunk are issued together, in chunk order)
    for i in range(nchunks):
        if i + 1 < nchunks:
            left = nchunks - 2 - i
            allow_chunks = NS - 2 if left >= NS - 2 else left
            done_through = issued - allow_chunks - 1  # vmcnt(DMA * allow_chunks): every chunk up to this index has landed
            assert done_through >= i + 1, (i, issued, allow_chunks)
            assert allow_chunks <= issued - (i + 1) - 1 + 1
        if i + NS < nchunks:
            issued += 1
    assert issued == nchunks


def panel32_off(row, col):          # byte offset in a [rows][32] bf16 panel (64-byte rows), the kernels' swizzle
    return row * 64 + ((col ^ ((0 - (row >> 2)) & 3)) << 4)


def test_activation_fragment_reads_are_bank_conflict_free():
    """ds_read_b128 is served in four groups of 16 lanes (MI355X_MICROARCH.md, LDS table); bank of byte a = (a / 4) mod 64.  A fragment read of the 32-deep activation panel
    (lane = (q, r): row r of a 16-row tile, 16-byte column q, swizzled by panel32_off) must put the 16 lanes of a group on 64 distinct banks; the weight tiles are lane-linear."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g] for g in groups]
    assert sorted(sum(groups, [])) == list(range(64))
    for row0 in (0, 16, 112, 208):
        for g in groups:
            banks = []
            for lane in g:
                a = panel32_off(row0 + (lane & 15), lane >> 4)
                banks += [((a + 4 * d) // 4) % 64 for d in range(4)]
            assert len(set(banks)) == 64, (row0, g)
            lin = []
            for lane in g:
                lin += [((lane * 16 + 4 * d) // 4) % 64 for d in range(4)]
            assert len(set(lin)) == 64
    # the swizzle is an involution on the 16-byte column index: the DMA source applies it, the read applies it again
    for row in range(224):
        assert sorted(panel32_off(row, c) - row * 64 for c in range(4)) == [0, 16, 32, 48]

// kernels_gemm.hip -- the linear layers of the encoder (reference: every ggml_mul_mat of
// src/nemo-stream.cpp:485-487,:570,:599-601,:654,:677 and src/nemo-ggml.cpp:983,:999,:1020).
//
// out[m][n] = sum_k A[m][k] * W[n][k]   (W stored [out][in] like PyTorch / GGUF)
//
// bf16 path (gfx950 MFMA v_mfma_f32_16x16x32_bf16, f32 accumulate):
//   * weights are PRE-PACKED at upload into MFMA A-fragment order: tile (nt, kt) of
//     16 n x 32 k is 1 KiB, lane l = q*16 + r holds W[nt*16+r][kt*32+q*8 .. +8).  One
//     wave-load of a tile is a single fully coalesced 1 KiB read -- the layout is chosen
//     for the access pattern, not inherited from the file format.
//   * the weight tile is the MFMA A operand (rows = n), activations are the B operand
//     (cols = m), so D[n][m]: lane holds 4 consecutive n for one m -> row-major stores.
//   * k_gemm_skinny (M <= 32): weight-streaming kernel, HBM-bound.  One 16-column tile x one
//     64-row slab per workgroup, K split over the 4 waves (and over blockIdx.y for split-K);
//     activations go straight to registers (requested before the weights); no LDS in the loop.
//   * M > 32: LDS-tiled kernels, both operands by LDS-DMA into a ring, staged epilogue.  Deep rings, one workgroup per CU:
//     k_gemm_roles<4> (128 x 128 tile, 8 loader + 8 consumer waves; >= 8 chunks per workgroup), k_gemm_tiled2<4> (the same
//     tile, 8 waves that do both; short K), k_gemm_t64<4> (128 x 64 tile x 2 K-splits for the N = 1024 GEMMs).  Shallow rings,
//     TWO workgroups per CU (pipelined steps above 768 rows, every step from 1 792 rows; GemmParams::coresident):
//     k_gemm_tiled2_k32<4> (32-deep chunks, 4 x 16 KiB) and k_gemm_t64<3> (72 KiB).  All of them perform the same MFMAs in the
//     same order per accumulator: which one runs never changes a bit of the result (tests/micro/gemm_variant_identity.py).
// f32 path (parity mode): plain LDS-tiled FMA kernel, k ascending, deterministic.
#include "nasr_internal.h"
#include <type_traits>
#include "nasr_epilogue.h"
#include "nasr_wave.h"
#include "nasr_post.h"

namespace nasr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// ------------------------------------------------------------------------------------
// skinny: weight-streaming, 16 output columns x (16*MT rows) per workgroup.  grid = (N/16, splits, slabs of
// 16*MT rows), block = 256 (4 waves split K).  Used up to a few hundred rows: there the 128x128 tiles of the
// large-M kernel would leave most CUs idle (M = 128 -> 32 tiles), while here every slab re-reads its 32 KiB of
// weights from L2 and all CUs stream.
// ------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void k_gemm_skinny(GemmParams p) {
    __shared__ float red[4][MT][64][4];
    const int nt = blockIdx.x, split = blockIdx.y, m_base = blockIdx.z * 16 * MT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, r = lane & 15;
    const int KT = p.K >> 5;
    const int t0 = (int)((long)KT * split / p.splits), t1 = (int)((long)KT * (split + 1) / p.splits);
    const int nts = t1 - t0;
    const int w0 = t0 + nts * wave / 4, w1 = t0 + nts * (wave + 1) / 4;

    const u32x4 *wp = (const u32x4 *)p.W + (size_t)nt * KT * 64 + lane;
    // Rows >= M read a valid row (their MFMA columns are independent and never stored).
    const char *arow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int m = m_base + mt * 16 + r;
        arow[mt] = a_row_ptr(p, m < p.M ? m : p.M - 1, 2) + q * 16;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int U = 8;   // weight tiles in flight per wave (8 KiB)
    int kt = w0;
    for (; kt + U <= w1; kt += U) {
        // activations first: they were written by the previous kernel and are the critical path; the memory pipeline
        // serves a wave's requests in order, so they must not queue behind 8 KiB of weights
        uint4 av[U][MT];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) av[u][mt] = *(const uint4 *)(arow[mt] + (size_t)(kt + u) * 64);
        u32x4 wv[U];
#pragma unroll
        for (int u = 0; u < U; u++) wv[u] = __builtin_nontemporal_load(wp + (size_t)(kt + u) * 64);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bf16x8 wf = __builtin_bit_cast(bf16x8, wv[u]);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, av[u][mt]), acc[mt], 0, 0, 0);
        }
    }
    for (; kt < w1; kt++) {
        const u32x4 wv = __builtin_nontemporal_load(wp + (size_t)kt * 64);
        const bf16x8 wf = __builtin_bit_cast(bf16x8, wv);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const uint4 av = *(const uint4 *)(arow[mt] + (size_t)kt * 64);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, av), acc[mt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int j = 0; j < 4; j++) red[wave][mt][lane][j] = acc[mt][j];
    __syncthreads();
    // wave w finishes m-tile w (MT <= 4)
    if (wave < MT) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            v[j] = ((red[0][wave][lane][j] + red[1][wave][lane][j]) + red[2][wave][lane][j]) + red[3][wave][lane][j];
        epi_quad<true>(p, split, m_base + wave * 16 + r, nt * 16 + q * 4, v[0], v[1], v[2], v[3]);
    }
}

constexpr int TM = 128;
constexpr int TM_ROWS = 128;          // rows of a row chunk whose publication a chained launch counts (= TM)

__device__ __forceinline__ int panel_off(int row, int chunk) {  // byte offset in a [128][64] bf16 panel
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// ------------------------------------------------------------------------------------
// tiled, LDS-DMA pipelined (the large-M kernel): 128 x 128 tile, 8 waves = 4 n-tile pairs x 2 halves of
// the rows (32 x 64 per wave); BOTH operands arrive by global_load_lds into a 4-slot LDS ring
// (slot = 16 KiB activation panel + 16 KiB weight tiles per 64-deep K chunk) with three chunks in
// flight per workgroup and one raw s_barrier per chunk.  With ~1 workgroup per CU (M = 896 gives
// 224 tiles) there is no other wave to hide the ~1 us L2/HBM latency behind, so the depth has to
// come from the pipeline itself; the DMA needs no staging registers.
//  * weight tiles are lane-linear in HBM already -> LDS image is linear, ds_read_b128 conflict-free
//  * the activation panel is stored linear too; the bank swizzle is applied to the per-lane SOURCE
//    address and again on the read (same involution), cdna_hip_programming.md rule 21
//  * the DMA is issued from inline asm (M0 = wave-uniform LDS base), so hipcc does not see a pending
//    LDS write and does not drain vmcnt(0) in front of every ds_read; completion is counted by hand
//    (8 DMA instructions per wave per chunk).

// Which tile a workgroup computes.  ids are contiguous per XCD (the remap at the top of every kernel); split-K slices outermost.
// Rounds 1-3 ran the row chunk fastest: an XCD then sweeps ALL activation panels once per column group -- 14.7 MB at 7 168 rows x
// K = 1 024, 58 MB for W2, against 4 MB of L2 -- and every sweep comes from the Infinity Cache again (W2 at 7 168 rows: 470 MB per launch
// in 77 us = 6 TB/s: the kernel ran at the Infinity Cache's rate, not the MFMA's).  Now: bands of w column groups with the column group
// fastest; the ids of an XCD are a compact w x (tiles / 8 / w) block, and the tiles that run at a time share w weight panels and a few
// activation panels.  Cold operands, us per launch at 7 168 rows, rows-fastest -> bands: W2 77.8 -> 62.3 (965 TFLOP/s), Wo 25.6 -> 20.0,
// QKV 70.0 -> 56.5, W1 82.2 -> 73.8 (256-row tiles 76.0 -> 69.1); per step: 512 streams 17.8 -> 16.5-16.9 ms synchronous, 128 streams
// 4.81 -> 4.49 pipelined, 64 streams (7 row chunks) 2.49 -> 2.44; 32 streams (4 row chunks) 1.52 / 1.54: up to 4 row chunks the old order
// stays (profiles/r4_tile_order.md).  The order never changes a result (engine option "tile_bands", gemm_variant_identity.py).
__device__ __forceinline__ void tile_of(int id, int n_groups, int m_chunks, int bands, int &mc, int &ng, int &split) {      // bands = GemmParams::tile_bands
    const int per = n_groups * m_chunks;
    split = id / per;
    const int in = id - split * per;
    if (bands == 2 || (bands == 0 && m_chunks <= 4)) { mc = in % m_chunks; ng = in / m_chunks; return; }
    const int w = (n_groups & 7) == 0 ? 8 : (n_groups & 3) == 0 ? 4 : (n_groups & 1) == 0 ? 2 : 1;
    const int band = in / (w * m_chunks), ib = in - band * (w * m_chunks);
    mc = ib / w;
    ng = band * w + ib % w;
}

// ------------------------------------------------------------------------------------
constexpr int G2_SLOT = 32768;

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Epilogue of the 128 x 128 tile through LDS.  Straight from the accumulators a wave-store writes 16 rows x 64 B (f32) or
// 32 B (bf16): partial lines, and the epilogue cost 3.7-5.3 us of a 8-15 us GEMM (tests/micro/gemm_probe.hip, mode 8: W1
// 15.2 us with, 9.8 us without its stores).  The ring is free once the K loop is over: the accumulators are parked in it as
// an f32 tile [128][132] and every thread then handles 4 consecutive columns of one row, so that a wave writes two whole
// rows of the tile (2 x 512 B of f32 / 2 x 256 B of bf16) per instruction.  Same epi_quad, same values, other thread.
constexpr int STG_LD = 132;      // floats per staged row (128 + 4: the 16 rows of a float4 store spread over the banks)
__device__ __forceinline__ void stage_acc(float *stage, int m_local, int n_local, const f32x4 &a) {
    *(float4 *)(stage + m_local * STG_LD + n_local) = make_float4(a[0], a[1], a[2], a[3]);
}
// The two bulk outputs (split-K partials, SiLU activations: 7-15 MB per launch) are stored write-through (sc0 sc1): nothing
// dirty is left in L2 for the end of the kernel to write back (gemm_probe: -0.4 ... -1.3 us per launch).
// 16-byte stores for the 16-bit / paired outputs (round 5): with four columns per thread the SiLU rows, the K / V ring rows and the GLU pairs left as 8-byte
// stores -- write-through 8-byte stores cost 2.7 x a 16-byte store per byte (MI355X_MICROARCH.md, stores of each flavour), and k_post's move to 16-byte
// stores was most of what k_post_wave gained.  Eight columns per thread here: same values, half the store instructions, twice the bytes each.
__device__ __forceinline__ uint4 pack8_bf16(const float (&v)[8]) {
    uint4 r;
    r.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    r.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    r.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    r.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    return r;
}
// one item of eight consecutive columns n0 .. n0 + 7 of row m (n0 % 8 == 0) for the three epilogues with narrow elements; returns false for the others
__device__ __forceinline__ bool epi_oct(const GemmParams &p, int m, int n0, const float (&v)[8]) {
    if (p.epi == EPI_SILU_ACT) {
        float s[8];
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = silu_f(v[i]);
        store_u4((bf16_t *)p.out_act + (size_t)m * p.ldo_act + n0, pack8_bf16(s));
        return true;
    }
    if (p.epi == EPI_GLU) {                    // (value, gate) pairs: four outputs
        store_wt_f4(p.out_f32 + (size_t)m * p.ldo + (n0 >> 1), make_float4(v[0] * sigmoid_f(v[1]), v[2] * sigmoid_f(v[3]), v[4] * sigmoid_f(v[5]), v[6] * sigmoid_f(v[7])));
        return true;
    }
    if (p.epi == EPI_QKV && n0 >= 1024) {      // K / V ring rows (the query columns stay f32: epi_quad)
        const int which = n0 >> 10, col = n0 & 1023;
        const int b = m / p.T, i = m - b * p.T;
        const RowDesc rd = p.rows[b];
        int ring = rd.kv_head + LCTX + i;
        if (ring >= KVC) ring -= KVC;
        const size_t off = (size_t)rd.slot * p.kv_slot_stride + ((size_t)(which - 1) * KVC + ring) * D + col;
        *(uint4 *)((bf16_t *)p.kv_pool + off) = pack8_bf16(v);
        return true;
    }
    return false;
}
template <int NTHREADS>
__device__ __forceinline__ void staged_epilogue(const GemmParams &p, int split, int m0, int n_base, const float *stage) {
    if (!p.narrow_stores && (p.epi == EPI_SILU_ACT || p.epi == EPI_GLU || p.epi == EPI_QKV)) {
        for (int e = threadIdx.x; e < TM * 16; e += NTHREADS) {
            const int row = e >> 4, c8 = (e & 15) * 8, m = m0 + row, n0 = n_base + c8;
            if (m >= p.M) continue;
            const float4 a = *(const float4 *)(stage + row * STG_LD + c8), b = *(const float4 *)(stage + row * STG_LD + c8 + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            if (!epi_oct(p, m, n0, v)) { epi_quad<true>(p, split, m, n0, a.x, a.y, a.z, a.w); epi_quad<true>(p, split, m, n0 + 4, b.x, b.y, b.z, b.w); }
        }
        return;
    }
    for (int e = threadIdx.x; e < TM * 32; e += NTHREADS) {
        const int row = e >> 5, c4 = (e & 31) * 4, m = m0 + row, n0 = n_base + c4;
        const float4 v = *(const float4 *)(stage + row * STG_LD + c4);
        if (p.epi == EPI_PART_F32) {
            if (m < p.M) {
                store_wt_f4(p.out_f32 + ((size_t)split * p.M + m) * p.ldo + n0, v);
            }
        } else if (p.epi == EPI_SILU_ACT) {
            if (m < p.M) {
                store_wt_u2((bf16_t *)p.out_act + (size_t)m * p.ldo_act + n0, pack4_bf16(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w)));
            }
        } else {
            epi_quad<true>(p, split, m, n0, v.x, v.y, v.z, v.w);
        }
    }
}

// NS = slots of the LDS ring (NS - 1 chunks in flight); shipped with NS = 4: 128 KiB, one workgroup per CU.  (A 2-slot ring so that two
// workgroups could share a CU was slower -- profiles/lab_book_rounds_1_2.md; what does pay is k_gemm_tiled2_k32 below: the same ring depth
// in 32-deep chunks.)
template <int NS>
__global__ __launch_bounds__(512) void k_gemm_tiled2(GemmParams p, int n_groups, int m_chunks) {
    constexpr int P = NS - 1;
    // 8 waves = 2 per SIMD: wave w owns n-tile pair (w & 3) x m-tiles [(w >> 2) * 4, +4).  One wave's
    // LDS-DMA issue (expensive: ~100+ cycles per 1 KiB instruction) overlaps its SIMD partner's MFMAs.
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int nblk = gridDim.x;
    int id = blockIdx.x;
    {
        const int qd = nblk >> 3, rm = nblk & 7, xcd = id & 7, loc = id >> 3;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    }
    int mc, ng, split;
    tile_of(id, n_groups, m_chunks, p.tile_bands, mc, ng, split);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ng4 = wave & 3, mh = wave >> 2;
    const int q = lane >> 4, r = lane & 15;
    const int KT = p.K >> 5;
    const int kc_total = KT >> 1;
    const int c0 = (int)((long)kc_total * split / p.splits), c1 = (int)((long)kc_total * (split + 1) / p.splits);
    const int nchunks = c1 - c0;
    const int m0 = mc * TM;
    const int ntile0 = (ng * 4 + ng4) * 2;
    // this wave DMAs weight tile ntile0 + mh (both k-tiles of the chunk) and panel rows [wave*16, +16)
    const uint4 *wpd = (const uint4 *)p.W + (size_t)(ntile0 + mh) * KT * 64 + lane;
    const int prow = lane >> 3, pc = lane & 7;
    const char *asrc[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = wave * 16 + i * 8 + prow;
        int m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        asrc[i] = a_row_ptr(p, m, 2) + ((pc ^ ((row >> 1) & 7)) << 4);
    }
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)ring;   // LDS byte address
    auto issue = [&](int kc, int slot) {
        const unsigned sb = ring_base + slot * G2_SLOT;
#pragma unroll
        for (int i = 0; i < 2; i++) glds16(asrc[i] + (size_t)kc * 128, sb + (wave * 16 + i * 8) * 128);
        const unsigned wb = sb + 16384 + ng4 * 4096 + mh * 2048;
        glds16(wpd + (size_t)(2 * kc) * 64, wb);
        glds16(wpd + (size_t)(2 * kc + 1) * 64, wb + 1024);
    };

    f32x4 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) acc[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < P; i++)
        if (i < nchunks) issue(c0 + i, i);
    for (int i = 0; i < nchunks; i++) {
        const int rem = nchunks - 1 - i < P - 1 ? nchunks - 1 - i : P - 1;         // chunks allowed to stay in flight
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // 4 DMA instructions per wave per chunk
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave's part of chunk i has landed; chunk i-1 is fully consumed
        const char *sp = ring + (i % NS) * G2_SLOT;
        const char *wl = sp + 16384 + ng4 * 4096 + lane * 16;
        uint4 w[2][2];
        w[0][0] = *(const uint4 *)(wl);
        w[0][1] = *(const uint4 *)(wl + 1024);
        w[1][0] = *(const uint4 *)(wl + 2048);
        w[1][1] = *(const uint4 *)(wl + 3072);
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                const uint4 bv = *(const uint4 *)(sp + panel_off((mh * 4 + mt) * 16 + r, k2 * 4 + q));
                const bf16x8 bf = __builtin_bit_cast(bf16x8, bv);
                acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[0][k2]), bf, acc[0][mt], 0, 0, 0);
                acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[1][k2]), bf, acc[1][mt], 0, 0, 0);
            }
        }
        if (i + P < nchunks) issue(c0 + i + P, (i + P) % NS);
    }
    __syncthreads();                           // every wave is done with the ring
    float *stage = (float *)ring;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) stage_acc(stage, (mh * 4 + mt) * 16 + r, (ng4 * 2 + j) * 16 + q * 4, acc[j][mt]);
    __syncthreads();
    staged_epilogue<512>(p, split, m0, ng * 128, stage);
}

// ------------------------------------------------------------------------------------
// The same 128 x 128 tile with 32-deep chunks (ONE k-tile per ring slot: 8 KiB activation panel + 8 KiB weight tiles), for
// pipelined steps: NS slots of 16 KiB (+ the staged epilogue's 66 KiB) let TWO workgroups share a CU -- GEMMs of two launch
// chains, whose ring fills, barrier waits and epilogues then run under each other's MFMAs.  Same MFMAs in the same order per
// accumulator as k_gemm_tiled2 / k_gemm_roles (k ascending, split boundaries on 64-deep chunks): bit-identical results.
// Panel rows are 64 B here, so four rows share a 256-byte bank line: 16-byte column c of row r lives at column c ^ ((-(r >> 2)) & 3)
// (conflict-free for the four 16-lane groups a ds_read_b128 is served in: MI355X_MICROARCH.md, LDS).
// ------------------------------------------------------------------------------------
// ---- chained launches (ChainParams, nasr_internal.h): head workgroups = the k_post that produces this launch's A rows ----------------------
template <int NT>
__device__ __forceinline__ void chain_head_phase(const ChainParams &c, float *sh) {          // sh: NT / 256 x 8 floats of LDS
    constexpr int SIDE = NT / 256;                                                            // rows that go through the barriers side by side
    const int side = threadIdx.x >> 8, t256 = threadIdx.x & 255;
    const int row0 = (int)blockIdx.x * c.head_rows;
    for (int it = 0; it < c.head_rows; it += SIDE) {
        const int m = row0 + it + side;
        post_row(c.post, m < c.post.M ? m : c.post.M - 1, t256, sh + side * 8, m < c.post.M && it + side < c.head_rows);
    }
    // publish: the rows went out write-through (store_wt_*); every storing wave waits for its stores, the barrier collects the waves, then ONE
    // agent-scope add for the workgroup (MI355X_MICROARCH.md, inter-workgroup visibility: the signalling lane comes after EVERY wave's wait)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int rows = c.post.M - row0 < c.head_rows ? c.post.M - row0 : c.head_rows;
        if (rows > 0) __hip_atomic_fetch_add(c.flags + row0 / TM_ROWS, (unsigned)rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// tile workgroup of row chunk mc (chunk_rows rows per chunk, n_wait tile workgroups share the chunk): returns once the chunk's rows are published
__device__ __forceinline__ void chain_wait(const ChainParams &c, int mc, int chunk_rows, int n_wait) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0) {
        if (threadIdx.x == 0) {
            const int left = c.post.M - mc * chunk_rows;
            const unsigned want = (unsigned)(left < chunk_rows ? left : chunk_rows);
            unsigned spins = 0;
            while (__hip_atomic_load(c.flags + mc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {      // ONE relaxed poller per workgroup
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1u << 21)) { __hip_atomic_store(c.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }      // never hang the GPU
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // buffer_inv sc1: this CU's L1 forgets what it held of the rows
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // ... completed before the barrier lets the other waves load
    }
    __syncthreads();
    if (threadIdx.x == 0) {          // the last tile workgroup of the chunk to get here re-arms both counters for the next launch that uses them
        const unsigned passed = __hip_atomic_fetch_add(c.flags + 64 + mc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (passed + 1 == (unsigned)n_wait) {
            __hip_atomic_store(c.flags + mc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c.flags + 64 + mc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

constexpr int K32_SLOT = 16384;
__device__ __forceinline__ int panel32_off(int row, int col) { return row * 64 + ((col ^ ((0 - (row >> 2)) & 3)) << 4); }
template <int NS>
__global__ __launch_bounds__(512) void k_gemm_tiled2_k32(GemmParams p, int n_groups, int m_chunks) {
    constexpr int P = NS - 1;
    extern __shared__ __attribute__((aligned(16))) char ring[];
    if (p.prio & 1) __builtin_amdgcn_s_setprio(3);
    const int n_head = p.chain.head_wgs;            // chained launch: the first workgroups are the k_post that produces this launch's A rows
    if ((int)blockIdx.x < n_head) { chain_head_phase<512>(p.chain, (float *)ring); return; }
    const int nblk = gridDim.x - n_head;
    int id = blockIdx.x - n_head;                   // n_head is a multiple of 8: the round-robin over the XCDs stays in step
    {
        const int qd = nblk >> 3, rm = nblk & 7, xcd = id & 7, loc = id >> 3;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    }
    int mc, ng, split;
    tile_of(id, n_groups, m_chunks, p.tile_bands, mc, ng, split);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ng4 = wave & 3, mh = wave >> 2, q = lane >> 4, r = lane & 15;
    const int KT = p.K >> 5, kc_total = KT >> 1;
    const int t0 = 2 * (int)((long)kc_total * split / p.splits), t1 = 2 * (int)((long)kc_total * (split + 1) / p.splits);
    const int nchunks = t1 - t0, m0 = mc * TM;
    // this wave DMAs weight tile ng * 8 + wave (one k-tile per chunk) and panel rows [wave * 16, +16)
    const uint4 *wpd = (const uint4 *)p.W + (size_t)(ng * 8 + wave) * KT * 64 + lane;
    const char *asrc;
    {
        const int row = wave * 16 + (lane >> 2);
        int m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        asrc = a_row_ptr(p, m, 2) + (((lane & 3) ^ ((0 - (row >> 2)) & 3)) << 4);
    }
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)ring;
    auto issue = [&](int kt, int slot) {
        const unsigned sb = ring_base + slot * K32_SLOT;
        glds16(asrc + (size_t)kt * 64, sb + wave * 1024);
        glds16(wpd + (size_t)kt * 64, sb + 8192 + wave * 1024);
    };
    f32x4 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) acc[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (n_head) {
        // chained: the weight halves of the first P slots go out at once (they depend on nothing), the activation halves once the head workgroups
        // have published this row chunk; the prologue is then drained completely, so the loop's counted waits (which assume two DMA
        // instructions per chunk in issue order) hold from its first iteration on
#pragma unroll
        for (int i = 0; i < P; i++)
            if (i < nchunks) glds16(wpd + (size_t)(t0 + i) * 64, ring_base + i * K32_SLOT + 8192 + wave * 1024);
        chain_wait(p.chain, mc, TM, n_groups * p.splits);
#pragma unroll
        for (int i = 0; i < P; i++)
            if (i < nchunks) glds16(asrc + (size_t)(t0 + i) * 64, ring_base + i * K32_SLOT + wave * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
        for (int i = 0; i < P; i++)
            if (i < nchunks) issue(t0 + i, i);
    }
    int slot = 0;
    for (int i = 0; i < nchunks; i++) {
        const int left = nchunks - 1 - i;                                          // chunks allowed to stay in flight: min(left, P - 1)
        if (left >= P - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (P - 1)) : "memory");   // 2 DMA instructions per wave per chunk
        else if (left == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (left == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (left == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave's part of chunk i has landed; chunk i - 1 is fully consumed
        const char *sp = ring + slot * K32_SLOT;
        const char *wl = sp + 8192 + ng4 * 2048 + lane * 16;
        const uint4 w0 = *(const uint4 *)(wl), w1 = *(const uint4 *)(wl + 1024);
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            const uint4 bv = *(const uint4 *)(sp + panel32_off((mh * 4 + mt) * 16 + r, q));
            const bf16x8 bf = __builtin_bit_cast(bf16x8, bv);
            acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), bf, acc[0][mt], 0, 0, 0);
            acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1), bf, acc[1][mt], 0, 0, 0);
        }
        if (i + P < nchunks) issue(t0 + i + P, slot == 0 ? NS - 1 : slot - 1);     // the slot chunk i - 1 has just left
        slot = slot + 1 == NS ? 0 : slot + 1;
    }
    if (p.prio & 2) __builtin_amdgcn_s_setprio(0);
    __syncthreads();                           // every wave is done with the ring
    float *stage = (float *)ring;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) stage_acc(stage, (mh * 4 + mt) * 16 + r, (ng4 * 2 + j) * 16 + q * 4, acc[j][mt]);
    __syncthreads();
    staged_epilogue<512>(p, split, m0, ng * 128, stage);
}

// k_gemm_tiled3 (round 5): k_gemm_tiled2_k32's tile, MFMAs and order with k_gemm_wide2's loop (below): the fragments of chunk i + 1 are read into a second
// register set under chunk i's MFMAs, the two DMA instructions of chunk i + 5 go out between the MFMA pairs, five slots of 16 KiB (80 KiB: still two
// workgroups per CU).  Needs an even number of chunks >= 6 per K slice; no chained head phase.
#ifdef NASR_GEMM_STAMPS
#define GSTAMP(k) do { if (p.stamps && threadIdx.x == 0) { if ((k) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); p.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); if ((k) == 0 || (k) == 3) p.stamps[(size_t)blockIdx.x * 8 + 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define GSTAMP(k)
#endif
constexpr int T3_NS = 5;
__global__ __launch_bounds__(512) void k_gemm_tiled3(GemmParams p, int n_groups, int m_chunks) {
    constexpr int NS = T3_NS;
    GSTAMP(0);
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int nblk = gridDim.x;
    int id = blockIdx.x;
    {
        const int qd = nblk >> 3, rm = nblk & 7, xcd = id & 7, loc = id >> 3;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    }
    int mc, ng, split;
    tile_of(id, n_groups, m_chunks, p.tile_bands, mc, ng, split);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ng4 = wave & 3, mh = wave >> 2, q = lane >> 4, r = lane & 15;
    const int KT = p.K >> 5, kc_total = KT >> 1;
    const int t0 = 2 * (int)((long)kc_total * split / p.splits), t1 = 2 * (int)((long)kc_total * (split + 1) / p.splits);
    const int nchunks = t1 - t0, m0 = mc * TM;
    const char *wpd = (const char *)((const uint4 *)p.W + (size_t)(ng * 8 + wave) * KT * 64 + lane) + (size_t)t0 * 1024;
    const char *asrc;
    {
        const int row = wave * 16 + (lane >> 2);
        int m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        asrc = a_row_ptr(p, m, 2) + (((lane & 3) ^ ((0 - (row >> 2)) & 3)) << 4) + (size_t)t0 * 64;
    }
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)ring;
    f32x4 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) acc[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 fw[2][2], fa[2][4];
    const unsigned w_off = 8192 + ng4 * 2048 + lane * 16;
    unsigned a_off[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) a_off[mt] = panel32_off((mh * 4 + mt) * 16 + r, q);
#pragma unroll
    for (int c = 0; c < NS; c++) {
        glds16(asrc + (size_t)c * 64, ring_base + c * K32_SLOT + wave * 1024);
        glds16(wpd + (size_t)c * 1024, ring_base + c * K32_SLOT + 8192 + wave * 1024);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NS - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    GSTAMP(1);
    fw[0][0] = *(const uint4 *)(ring + w_off); fw[0][1] = *(const uint4 *)(ring + w_off + 1024);
#pragma unroll
    for (int mt = 0; mt < 4; mt++) fa[0][mt] = *(const uint4 *)(ring + a_off[mt]);
    int slot = 0;
    auto body = [&](int i, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value, nxt = 1 - cur;
        const bool has_next = i + 1 < nchunks;
        const int nslot = slot + 1 == NS ? 0 : slot + 1;
        if (has_next) {
            const int left = nchunks - 2 - i;
            if (left >= NS - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (NS - 2)) : "memory");
            else if (left == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else if (left == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // chunk i + 1 has landed; every wave holds chunk i in registers: its slot is free
        }
        const char *sp = ring + nslot * K32_SLOT;
        const bool more = i + NS < nchunks;
        const unsigned sb = ring_base + slot * K32_SLOT;
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[cur][0]), __builtin_bit_cast(bf16x8, fa[cur][mt]), acc[0][mt], 0, 0, 0);
            acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[cur][1]), __builtin_bit_cast(bf16x8, fa[cur][mt]), acc[1][mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) {
                if (mt == 0) { fw[nxt][0] = *(const uint4 *)(sp + w_off); fw[nxt][1] = *(const uint4 *)(sp + w_off + 1024); }
                if (mt == 1) { fa[nxt][0] = *(const uint4 *)(sp + a_off[0]); fa[nxt][1] = *(const uint4 *)(sp + a_off[1]); }
                if (mt == 2) { fa[nxt][2] = *(const uint4 *)(sp + a_off[2]); fa[nxt][3] = *(const uint4 *)(sp + a_off[3]); }
            }
            if (more) {
                if (mt == 0) glds16(asrc + (size_t)(i + NS) * 64, sb + wave * 1024);
                if (mt == 1) glds16(wpd + (size_t)(i + NS) * 1024, sb + 8192 + wave * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        slot = nslot;
    };
    for (int i = 0; i < nchunks; i += 2) {
        body(i, std::integral_constant<int, 0>{});
        body(i + 1, std::integral_constant<int, 1>{});
    }
    GSTAMP(2);
    __syncthreads();                           // every wave is done with the ring
    float *stage = (float *)ring;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) stage_acc(stage, (mh * 4 + mt) * 16 + r, (ng4 * 2 + j) * 16 + q * 4, acc[j][mt]);
    __syncthreads();
    staged_epilogue<512>(p, split, m0, ng * 128, stage);
    GSTAMP(3);
}

// (A four-wave form of this tile -- 64 x 64 per wave, 16 MFMAs and four DMA instructions per wave and chunk -- measured 1 092 cycles per chunk against 634 here
// (in-kernel stamps, 896 rows) and 2.58 against 2.41 ms per pipelined 64-stream step: a wave's LDS-DMA instructions cost it ~100+ cycles of issue each, and with one wave
// per SIMD nothing runs under them.  Removed; profiles/r5_gemm_tile_stamps.md.)

// ------------------------------------------------------------------------------------
// 128 (M) x 64 (N) tile of the same structure, for the split-K GEMMs with N = 1024 (W2, Wo, pw2) at M = 896: the 128 x 128
// tile gives 56 output tiles, so K had to be split four ways to fill the chip and every GEMM wrote 4 x 3.67 MB of f32
// partials that k_post read back (round 1: 59 MB per layer).  Half-width tiles fill the chip with TWO splits: half the
// partial bytes on both sides of the seam.  8 waves = 2 n-tile pairs x 4 quarters of the rows (32 x 32 per wave); slot =
// 16 KiB activation panel + 8 KiB weight tiles per 64-deep chunk, 3 DMA instructions per wave and chunk.
// ------------------------------------------------------------------------------------
constexpr int T64_SLOT = 24576, T64_STG_LD = 68;
template <int NS>
__global__ __launch_bounds__(512) void k_gemm_t64(GemmParams p, int n_groups, int m_chunks) {
    constexpr int P = NS - 1;
    extern __shared__ __attribute__((aligned(16))) char ring[];
    if (p.prio & 1) __builtin_amdgcn_s_setprio(3);
    const int nblk = gridDim.x;
    int id = blockIdx.x;
    {
        const int qd = nblk >> 3, rm = nblk & 7, xcd = id & 7, loc = id >> 3;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    }
    int mc, ng, split;
    tile_of(id, n_groups, m_chunks, p.tile_bands, mc, ng, split);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ng2 = wave & 1, mq = wave >> 1, q = lane >> 4, r = lane & 15;
    const int KT = p.K >> 5, kc_total = KT >> 1;
    const int c0 = (int)((long)kc_total * split / p.splits), c1 = (int)((long)kc_total * (split + 1) / p.splits);
    const int nchunks = c1 - c0, m0 = mc * TM, ntile0 = ng * 4;
    // this wave DMAs weight tile (ntile0 + (wave >> 1), k-tile wave & 1) and panel rows [wave * 16, +16)
    const uint4 *wpd = (const uint4 *)p.W + ((size_t)(ntile0 + (wave >> 1)) * KT + (wave & 1)) * 64 + lane;
    const int prow = lane >> 3, pc = lane & 7;
    const char *asrc[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = wave * 16 + i * 8 + prow;
        int m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        asrc[i] = a_row_ptr(p, m, 2) + ((pc ^ ((row >> 1) & 7)) << 4);
    }
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)ring;
    auto issue = [&](int kc, int slot) {
        const unsigned sb = ring_base + slot * T64_SLOT;
#pragma unroll
        for (int i = 0; i < 2; i++) glds16(asrc[i] + (size_t)kc * 128, sb + (wave * 16 + i * 8) * 128);
        glds16(wpd + (size_t)(2 * kc) * 64, sb + 16384 + wave * 1024);          // [n-tile 0..3][k-tile 0..1][1 KiB]
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 2; mt++) acc[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < P; i++)
        if (i < nchunks) issue(c0 + i, i);
    for (int i = 0; i < nchunks; i++) {
        const int rem = nchunks - 1 - i < P - 1 ? nchunks - 1 - i : P - 1;         // chunks allowed to stay in flight
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");             // 3 DMA instructions per wave per chunk
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char *sp = ring + (i % NS) * T64_SLOT;
        const char *wl = sp + 16384 + ng2 * 4096 + lane * 16;
        uint4 w[2][2];
        w[0][0] = *(const uint4 *)(wl);
        w[0][1] = *(const uint4 *)(wl + 1024);
        w[1][0] = *(const uint4 *)(wl + 2048);
        w[1][1] = *(const uint4 *)(wl + 3072);
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                const uint4 bv = *(const uint4 *)(sp + panel_off((mq * 2 + mt) * 16 + r, k2 * 4 + q));
                const bf16x8 bf = __builtin_bit_cast(bf16x8, bv);
                acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[0][k2]), bf, acc[0][mt], 0, 0, 0);
                acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[1][k2]), bf, acc[1][mt], 0, 0, 0);
            }
        }
        if (i + P < nchunks) issue(c0 + i + P, (i + P) % NS);
    }
    if (p.prio & 2) __builtin_amdgcn_s_setprio(0);
    __syncthreads();                           // every wave is done with the ring
    float *stage = (float *)ring;              // f32 tile [128][68]
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
            *(float4 *)(stage + ((mq * 2 + mt) * 16 + r) * T64_STG_LD + (ng2 * 2 + j) * 16 + q * 4) =
                make_float4(acc[j][mt][0], acc[j][mt][1], acc[j][mt][2], acc[j][mt][3]);
    __syncthreads();
    for (int e = threadIdx.x; e < TM * 16; e += 512) {                // a wave stores four whole 256-byte rows of the tile per instruction
        const int row = e >> 4, c4 = (e & 15) * 4, m = m0 + row, n0 = ng * 64 + c4;
        const float4 v = *(const float4 *)(stage + row * T64_STG_LD + c4);
        if (m >= p.M) continue;
        if (p.epi == EPI_PART_F32) store_wt_f4(p.out_f32 + ((size_t)split * p.M + m) * p.ldo + n0, v);
        else epi_quad<true>(p, split, m, n0, v.x, v.y, v.z, v.w);
    }
}

// ------------------------------------------------------------------------------------
// Round 5: the two K-halves of a 128 x 64 tile in ONE workgroup (16 waves: waves 0-7 = k_gemm_t64's eight waves on the first half of K,
// waves 8-15 the same on the second half, each half with its own 3-slot ring: 2 x 72 KiB = what two co-resident k_gemm_t64<3> workgroups hold).
// Rounds 1-4 wrote the two halves out as f32 partial slabs (2 x 3.67 MB per GEMM at 896 rows, write-through) and k_post read them back with the
// residual: 16.5 MB per k_post launch, 97 launches per step; leaving k_post out of a pipelined 64-stream step saved 0.41 of its 2.48 ms
// (profiles/r5_configs2_launch_structure.md, r5_ablation_b64_R13.json).  Here the halves meet in LDS, and the epilogue adds scale x (p0 + p1) to the residual stream in place
// (EPI_RESID_F32: the order and the fmaf of k_post -- o = t[0] + t[1]; x = fmaf(scale, o, x) -- so x keeps its bits): no partial slab leaves the
// CU, k_post is left with the LayerNorm.  Each half performs k_gemm_t64's MFMAs in k_gemm_t64's order (split = half of 2): p0 and p1 are
// the slabs' values.  The residual tile (32 KiB) is requested at kernel entry, long before it is needed.
// ------------------------------------------------------------------------------------
constexpr int T64W_NS = 3, T64W_HALF = T64W_NS * T64_SLOT;          // 73 728 B per K-half
__global__ __launch_bounds__(1024) void k_gemm_t64w(GemmParams p, int n_groups, int m_chunks) {
    constexpr int NS = T64W_NS, P = NS - 1;
    extern __shared__ __attribute__((aligned(16))) char ring_all[];
    const int nblk = gridDim.x;
    int id = blockIdx.x;
    {
        const int qd = nblk >> 3, rm = nblk & 7, xcd = id & 7, loc = id >> 3;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    }
    int mc, ng, split_unused;
    tile_of(id, n_groups, m_chunks, p.tile_bands, mc, ng, split_unused);
    const int wave16 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int half = wave16 >> 3, wave = wave16 & 7;
    char *ring = ring_all + half * T64W_HALF;
    const int ng2 = wave & 1, mq = wave >> 1, q = lane >> 4, r = lane & 15;
    const int KT = p.K >> 5, kc_total = KT >> 1;
    const int c0 = (int)((long)kc_total * half / 2), c1 = (int)((long)kc_total * (half + 1) / 2);     // k_gemm_t64's slice `half` of 2
    const int nchunks = c1 - c0, m0 = mc * TM, ntile0 = ng * 4;
    // the residual tile: two float4 per thread (row e >> 4, columns 4 (e & 15) of the 128 x 64 tile, e = tid and tid + 1024)
    float4 xin[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = threadIdx.x + i * 1024, m = m0 + (e >> 4);
        xin[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.epi == EPI_RESID_F32 && m < p.M) xin[i] = *(const float4 *)(p.resid + (size_t)m * p.ldo + ng * 64 + (e & 15) * 4);
    }
    const uint4 *wpd = (const uint4 *)p.W + ((size_t)(ntile0 + (wave >> 1)) * KT + (wave & 1)) * 64 + lane;
    const int prow = lane >> 3, pc = lane & 7;
    const char *asrc[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = wave * 16 + i * 8 + prow;
        int m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        asrc[i] = a_row_ptr(p, m, 2) + ((pc ^ ((row >> 1) & 7)) << 4);
    }
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)ring;
    auto issue = [&](int kc, int slot) {
        const unsigned sb = ring_base + slot * T64_SLOT;
#pragma unroll
        for (int i = 0; i < 2; i++) glds16(asrc[i] + (size_t)kc * 128, sb + (wave * 16 + i * 8) * 128);
        glds16(wpd + (size_t)(2 * kc) * 64, sb + 16384 + wave * 1024);
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 2; mt++) acc[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < P; i++)
        if (i < nchunks) issue(c0 + i, i);
    for (int i = 0; i < nchunks; i++) {                 // both halves run the same number of chunks (the launcher checks K % 128 == 0): the barriers pair up
        const int rem = nchunks - 1 - i < P - 1 ? nchunks - 1 - i : P - 1;
        if (rem >= 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");             // 3 DMA instructions per wave per chunk; the residual loads are older still
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char *sp = ring + (i % NS) * T64_SLOT;
        const char *wl = sp + 16384 + ng2 * 4096 + lane * 16;
        uint4 w[2][2];
        w[0][0] = *(const uint4 *)(wl);
        w[0][1] = *(const uint4 *)(wl + 1024);
        w[1][0] = *(const uint4 *)(wl + 2048);
        w[1][1] = *(const uint4 *)(wl + 3072);
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                const uint4 bv = *(const uint4 *)(sp + panel_off((mq * 2 + mt) * 16 + r, k2 * 4 + q));
                const bf16x8 bf = __builtin_bit_cast(bf16x8, bv);
                acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[0][k2]), bf, acc[0][mt], 0, 0, 0);
                acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[1][k2]), bf, acc[1][mt], 0, 0, 0);
            }
        }
        if (i + P < nchunks) issue(c0 + i + P, (i + P) % NS);
    }
    __syncthreads();                           // every wave is done with its ring
    float *stage = (float *)ring;              // this half's f32 tile [128][68]
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
            *(float4 *)(stage + ((mq * 2 + mt) * 16 + r) * T64_STG_LD + (ng2 * 2 + j) * 16 + q * 4) =
                make_float4(acc[j][mt][0], acc[j][mt][1], acc[j][mt][2], acc[j][mt][3]);
    __syncthreads();
    const float *s0 = (const float *)ring_all, *s1 = (const float *)(ring_all + T64W_HALF);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = threadIdx.x + i * 1024, row = e >> 4, c4 = (e & 15) * 4, m = m0 + row, n0 = ng * 64 + c4;
        if (m >= p.M) continue;
        const float4 a = *(const float4 *)(s0 + row * T64_STG_LD + c4), b = *(const float4 *)(s1 + row * T64_STG_LD + c4);
        const float4 o = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);             // k_post: o = t[0] + t[1]
        if (p.epi == EPI_RESID_F32)
            store_wt_f4(p.out_f32 + (size_t)m * p.ldo + n0, make_float4(__builtin_fmaf(p.resid_scale, o.x, xin[i].x), __builtin_fmaf(p.resid_scale, o.y, xin[i].y),
                                                                       __builtin_fmaf(p.resid_scale, o.z, xin[i].z), __builtin_fmaf(p.resid_scale, o.w, xin[i].w)));
        else epi_quad<true>(p, 0, m, n0, o.x, o.y, o.z, o.w);
    }
}

// ------------------------------------------------------------------------------------
// the same tile and ring with ROLE-SPECIALISED waves (tests/micro/gemm_probe.hip, mode 7): in k_gemm_tiled2 every wave
// issues its share of the LDS-DMA (4 instructions of ~100 cycles per chunk), reads its fragments and multiplies -- one
// after the other, and all eight waves in the same phase behind the per-chunk barrier: 0.5 us per chunk where the DMA
// alone needs 0.17 and the MFMAs 0.21.  Here 16 waves share one workgroup: 8 loaders (two per SIMD) do nothing but issue
// the DMA of the ring, 8 consumers (the tiling of k_gemm_tiled2: 32 n x 64 m each) do nothing but read fragments and issue
// MFMAs, half a chunk ahead in registers.  One s_barrier per chunk for all 16 waves: at barrier i chunk i has landed (every
// loader waited for its own pieces) and every consumer holds what it still needs of chunk i - 1 in registers (lgkmcnt(0)
// before the barrier), so the loaders overwrite slot (i - 1) & 3 with chunk i + 3 while the consumers work on chunk i.
// 13.0 instead of 15.2 us on the W1 shape, 12.5 instead of 13.6 on W2 (M = 896); slightly slower below 8 chunks per
// workgroup, where k_gemm_tiled2 stays.  1024 threads leave 128 VGPRs per wave: fragments of half a chunk per register set.
// ------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(1024) void k_gemm_roles(GemmParams p, int n_groups, int m_chunks) {
    constexpr int P = NS - 1;
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int nblk = gridDim.x;
    int id = blockIdx.x;
    {
        const int qd = nblk >> 3, rm = nblk & 7, xcd = id & 7, loc = id >> 3;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    }
    int mc, ng, split;
    tile_of(id, n_groups, m_chunks, p.tile_bands, mc, ng, split);
    const int wave16 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool loader = wave16 >= 8;
    const int wave = wave16 & 7;
    const int ng4 = wave & 3, mh = wave >> 2, q = lane >> 4, r = lane & 15;
    const int KT = p.K >> 5, kc_total = KT >> 1;
    const int c0 = (int)((long)kc_total * split / p.splits), c1 = (int)((long)kc_total * (split + 1) / p.splits);
    const int nchunks = c1 - c0, m0 = mc * TM, ntile0 = (ng * 4 + ng4) * 2;
    if (loader) {
        const uint4 *wpd = (const uint4 *)p.W + (size_t)(ntile0 + mh) * KT * 64 + lane;
        const int prow = lane >> 3, pc = lane & 7;
        const char *asrc[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = wave * 16 + i * 8 + prow;
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            asrc[i] = a_row_ptr(p, m, 2) + ((pc ^ ((row >> 1) & 7)) << 4);
        }
        const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)ring;
        auto issue = [&](int kc, int slot) {
            const unsigned sb = ring_base + slot * G2_SLOT;
#pragma unroll
            for (int i = 0; i < 2; i++) glds16(asrc[i] + (size_t)kc * 128, sb + (wave * 16 + i * 8) * 128);
            const unsigned wb = sb + 16384 + ng4 * 4096 + mh * 2048;
            glds16(wpd + (size_t)(2 * kc) * 64, wb);
            glds16(wpd + (size_t)(2 * kc + 1) * 64, wb + 1024);
        };
#pragma unroll
        for (int i = 0; i < P; i++)
            if (i < nchunks) issue(c0 + i, i);
        for (int i = 0; i <= nchunks; i++) {                 // barriers 0 .. nchunks (the consumers' last one closes the pipeline)
            if (i < nchunks) {
                const int rem = nchunks - 1 - i < P - 1 ? nchunks - 1 - i : P - 1;
                if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (i + P < nchunks) issue(c0 + i + P, (i + P) % NS);
        }
        __syncthreads();                       // (consumers: done with the ring) ...
        __syncthreads();                       // ... (consumers: accumulators parked in it): the loaders help to store them
        staged_epilogue<1024>(p, split, m0, ng * 128, (const float *)ring);
        return;
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) acc[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragments of HALF a chunk (one 32-deep k-tile: 2 weight + 4 activation fragments = 24 VGPRs) per register set, two sets:
    // while one half is multiplied the next one is being read (1024 threads leave 128 VGPRs per wave).  The reads are issued
    // from inline asm and waited for by hand (LDS returns in order: lgkmcnt(6) = the older set has arrived): the compiler's own
    // wait insertion puts lgkmcnt(0) in front of the second half's MFMAs and serialises it.
#define LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)ring;
    const unsigned w_addr = lds0 + 16384 + ng4 * 4096 + lane * 16;
    unsigned b_addr[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; k2++) b_addr[k2] = lds0 + panel_off(mh * 64 + r, k2 * 4 + q);     // + mt * 2048: the swizzle depends on r only
    uint4 wA[2], bA[4], wB[2], bB[4];
    auto mm = [&](uint4 (&w)[2], uint4 (&bv)[4]) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            const bf16x8 bf = __builtin_bit_cast(bf16x8, bv[mt]);
            acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[0]), bf, acc[0][mt], 0, 0, 0);
            acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[1]), bf, acc[1][mt], 0, 0, 0);
        }
    };
    // barrier i: chunk i has landed.  Then: read (i, half 0) -> A | multiply (i - 1, half 1) from B | read (i, half 1) -> B |
    // multiply (i, half 0) from A | all reads of chunk i complete before barrier i + 1
    for (int i = 0; i <= nchunks; i++) {
        __builtin_amdgcn_s_barrier();
        const unsigned so = (unsigned)(i % NS) * G2_SLOT;
        if (i < nchunks) {
            const unsigned wa = w_addr + so, ba = b_addr[0] + so;
            LDS_RD(wA[0], wa, 0); LDS_RD(wA[1], wa, 2048);
            LDS_RD(bA[0], ba, 0); LDS_RD(bA[1], ba, 2048); LDS_RD(bA[2], ba, 4096); LDS_RD(bA[3], ba, 6144);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (i > 0) mm(wB, bB);                             // set B arrived before the last barrier (lgkmcnt(0) below)
        __builtin_amdgcn_sched_barrier(0);
        if (i < nchunks) {
            const unsigned wa = w_addr + so + 1024, ba = b_addr[1] + so;
            LDS_RD(wB[0], wa, 0); LDS_RD(wB[1], wa, 2048);
            LDS_RD(bB[0], ba, 0); LDS_RD(bB[1], ba, 2048); LDS_RD(bB[2], ba, 4096); LDS_RD(bB[3], ba, 6144);
            asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");   // set A is in (6 younger reads in flight)
            __builtin_amdgcn_sched_barrier(0);
            mm(wA, bA);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of chunk i complete before the barrier frees its slot
        }
    }
    __syncthreads();
    float *stage = (float *)ring;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) stage_acc(stage, (mh * 4 + mt) * 16 + r, (ng4 * 2 + j) * 16 + q * 4, acc[j][mt]);
    __syncthreads();
    staged_epilogue<1024>(p, split, m0, ng * 128, stage);
}

// ------------------------------------------------------------------------------------
// PERSISTENT tile loop (round 4; probe: tests/micro/gemm_probe.hip mode 11) for GEMMs with several 128 x 128 tiles per CU (128+
// streams x R = 13, buffered audio, TitaNet-L: M >= 1 792).  One workgroup per CU walks its tiles (tile id = i * grid + block):
//   waves 0-7   consumers: k_gemm_roles' tiling and MFMA order (32 n x 64 m per wave, k ascending) -- same bits;
//   waves 8-11  loaders: 8 LDS-DMA instructions per 64-deep chunk each, running straight on into the next tile's chunks, so the ring
//               (3 x 32 KiB) never drains between tiles;
//   waves 12-15 storers: the consumers park a finished tile as raw f32 in a 64 KiB staging tile BESIDE the ring (96 + 64 = the CU's
//               160 KiB) and go on multiplying; the storers evaluate the epilogue (SiLU, GLU, bias, K/V scatter) and write the tile
//               out in 16 slices of 8 rows, one per chunk interval of the next tile (two in the fifteenth).
// Every wave executes G + 16 barriers (G = chunks of all tiles of the workgroup).  What the round-3 probe left on the table was in the
// storers (stamps: profiles/r4_persistent_gemm.md): their slice took 0.44-0.56 us -- the epilogue switch, 64-bit index arithmetic and a
// branchy bf16 pack in ONE wave's dependent instruction stream -- and set the length of every chunk interval (0.68 us against 0.42
// with idle storers); and SiLU in the consumers stalled the MFMAs 1.3 us per tile.  Here the epilogue is a template parameter, the
// per-tile pointers are computed once, the pack is branch-free and SiLU is the storers' work.
// ------------------------------------------------------------------------------------
constexpr int PS_NS = 3, PS_STAGE = PS_NS * G2_SLOT, PS_LDS = PS_STAGE + 65536;
__device__ __forceinline__ unsigned ps_stage_off(int row, int cg) { return PS_STAGE + row * 512 + ((cg ^ (row & 31)) << 4); }
__device__ __forceinline__ uint32_t bf16_rne_bits(float f) {          // f32_to_bf16 without a branch (same values, NaN kept NaN)
    const uint32_t u = __float_as_uint(f);
    const uint32_t r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    return (u & 0x7fffffffu) > 0x7f800000u ? ((u >> 16) | 0x40u) : r;
}
template <int EPI>
__global__ __launch_bounds__(1024) void k_gemm_persist(GemmParams p, int n_groups, int m_chunks) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int n_tiles = n_groups * m_chunks, grid = (int)gridDim.x, block = (int)blockIdx.x;
    const int my_tiles = (n_tiles - block + grid - 1) / grid;
    const int wave16 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int KT = p.K >> 5, CPT = KT >> 1;                      // 64-deep chunks per tile (>= 16: the launcher checks)
    const int G = my_tiles * CPT, NB = G + 16;                   // the last tile is parked in interval G and drained in G + 1 .. G + 15
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)ring;
    auto tile_mn = [&](int i, int &m0, int &ng) { const int id = i * grid + block; m0 = (id % m_chunks) * TM; ng = id / m_chunks; };
    if (wave16 >= 8 && wave16 < 12) {
        // ---------------- loaders ----------------
        const int lw = wave16 - 8, prow = lane >> 3, pc = lane & 7;
        const char *asrc[4];
        const uint4 *wpd[2];
        auto set_tile = [&](int i) {
            int m0, ng;
            tile_mn(i, m0, ng);
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const int row = lw * 32 + a * 8 + prow;
                int m = m0 + row;
                if (m >= p.M) m = p.M - 1;
                asrc[a] = a_row_ptr(p, m, 2) + ((pc ^ ((row >> 1) & 7)) << 4);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) wpd[j] = (const uint4 *)p.W + (size_t)(ng * 8 + 2 * lw + j) * KT * 64 + lane;
        };
        int it = 0, ikc = 0;                                     // tile / chunk-in-tile of the next chunk to issue
        if (G > 0) set_tile(0);
        auto issue_next = [&](int g) {
            const unsigned sb = lds0 + (g % PS_NS) * G2_SLOT;
#pragma unroll
            for (int a = 0; a < 4; a++) glds16(asrc[a] + (size_t)ikc * 128, sb + (lw * 32 + a * 8) * 128);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const unsigned wb = sb + 16384 + lw * 4096 + j * 2048;
                glds16(wpd[j] + (size_t)(2 * ikc) * 64, wb);
                glds16(wpd[j] + (size_t)(2 * ikc + 1) * 64, wb + 1024);
            }
            if (++ikc == CPT) { ikc = 0; if (++it < my_tiles) set_tile(it); }
        };
        for (int g = 0; g < PS_NS - 1 && g < G; g++) issue_next(g);
        for (int g = 0; g < NB; g++) {
            if (g < G) {
                if (G - 1 - g >= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // chunk g landed, chunk g + 1 may be in flight
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (g + PS_NS - 1 < G) issue_next(g + PS_NS - 1);
        }
        return;
    }
    if (wave16 >= 12) {
        // ---------------- storers: the tile parked in the staging area, 8 rows per slice ----------------
        const int sw = wave16 - 12, half = lane >> 5, cg = lane & 31;
        int ti = -1, m0 = 0, ng = 0, n0 = 0;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        int kv_which = 0;
        int rd_slot = 0, rd_head = 0;                            // EPI_QKV: the slice's row descriptor, requested one interval ahead
        auto qkv_fetch = [&](int m) {
            if (EPI != EPI_QKV) return;
            if (m >= p.M) m = p.M - 1;
            const RowDesc *rd = p.rows + m / p.T;
            rd_slot = rd->slot; rd_head = rd->kv_head;
        };
        auto emit = [&](int row, int fetch_row) {
            const float4 v = *(const float4 *)(ring + ps_stage_off(row, cg));
            const int m = m0 + row;
            const int my_slot = rd_slot, my_head = rd_head;
            if (fetch_row >= 0) qkv_fetch(m0 + fetch_row);       // the next slice's descriptor travels while this slice is stored
            if (m >= p.M) return;
            if (EPI == EPI_PART_F32) store_wt_f4(p.out_f32 + (size_t)m * p.ldo + n0, v);
            else if (EPI == EPI_SILU_ACT) {
                const float a = silu_f(v.x), b = silu_f(v.y), c = silu_f(v.z), d = silu_f(v.w);
                store_wt_u2((bf16_t *)p.out_act + (size_t)m * p.ldo_act + n0, make_uint2(bf16_rne_bits(a) | (bf16_rne_bits(b) << 16), bf16_rne_bits(c) | (bf16_rne_bits(d) << 16)));
            } else if (EPI == EPI_GLU) {
                *(float2 *)(p.out_f32 + (size_t)m * p.ldo + (n0 >> 1)) = make_float2(v.x * sigmoid_f(v.y), v.z * sigmoid_f(v.w));
            } else if (EPI == EPI_BIAS_F32) {
                *(float4 *)(p.out_f32 + (size_t)m * p.ldo + n0) = make_float4(v.x + bias4.x, v.y + bias4.y, v.z + bias4.z, v.w + bias4.w);
            } else if (EPI == EPI_BIAS_RELU_F32) {
                *(float4 *)(p.out_f32 + (size_t)m * p.ldo + n0) = make_float4(fmaxf(v.x + bias4.x, 0.f), fmaxf(v.y + bias4.y, 0.f), fmaxf(v.z + bias4.z, 0.f), fmaxf(v.w + bias4.w, 0.f));
            } else if (EPI == EPI_QKV) {
                if (kv_which == 0) *(float4 *)(p.q_out + (size_t)m * D + n0) = v;
                else {
                    const int i = m - (m / p.T) * p.T;
                    int r = my_head + LCTX + i;
                    if (r >= KVC) r -= KVC;
                    const size_t off = (size_t)my_slot * p.kv_slot_stride + ((size_t)(kv_which - 1) * KVC + r) * D + (n0 & 1023);
                    *(uint2 *)((bf16_t *)p.kv_pool + off) = make_uint2(bf16_rne_bits(v.x) | (bf16_rne_bits(v.y) << 16), bf16_rne_bits(v.z) | (bf16_rne_bits(v.w) << 16));
                }
            }
        };
        for (int g = 0; g < NB; g++) {
            __builtin_amdgcn_s_barrier();
            if (g < 1) continue;
            const int u = g - 1, t = u / CPT - 1, sl = u - (u / CPT) * CPT;
            if (t < 0 || t >= my_tiles || sl >= 15) continue;
            if (t != ti) {                                       // first slice of a tile: its constants
                ti = t;
                tile_mn(t, m0, ng);
                n0 = ng * 128 + cg * 4;
                if (EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_RELU_F32) bias4 = *(const float4 *)(p.bias + n0);
                if (EPI == EPI_QKV) { kv_which = n0 >> 10; qkv_fetch(m0 + 2 * sw + half); }
            }
            const int row = 8 * sl + 2 * sw + half;
            emit(row, row + 8);
            if (sl == 14) emit(row + 8, -1);
        }
        return;
    }
    // ---------------- consumers (k_gemm_roles' loop; a finished tile is parked raw and the accumulators restart from zero) ----------------
    const int wave = wave16, ng4 = wave & 3, mh = wave >> 2, q = lane >> 4, r = lane & 15;
    f32x4 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) acc[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned w_addr = lds0 + 16384 + ng4 * 4096 + lane * 16;
    unsigned b_addr[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; k2++) b_addr[k2] = lds0 + panel_off(mh * 64 + r, k2 * 4 + q);
    uint4 wA[2], bA[4], wB[2], bB[4];
    auto mm = [&](uint4 (&w)[2], uint4 (&bv)[4]) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            const bf16x8 bf = __builtin_bit_cast(bf16x8, bv[mt]);
            acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[0]), bf, acc[0][mt], 0, 0, 0);
            acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[1]), bf, acc[1][mt], 0, 0, 0);
        }
    };
    int in_tile = 0;                                             // chunks of the current tile multiplied so far (second halves pending)
    for (int g = 0; g < NB; g++) {
        __builtin_amdgcn_s_barrier();
        const unsigned so = (unsigned)(g % PS_NS) * G2_SLOT;
        if (g < G) {
            const unsigned wa = w_addr + so, ba = b_addr[0] + so;
            LDS_RD(wA[0], wa, 0); LDS_RD(wA[1], wa, 2048);
            LDS_RD(bA[0], ba, 0); LDS_RD(bA[1], ba, 2048); LDS_RD(bA[2], ba, 4096); LDS_RD(bA[3], ba, 6144);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g > 0 && g <= G) mm(wB, bB);                         // second half of chunk g - 1
        __builtin_amdgcn_sched_barrier(0);
        if (g > 0 && g <= G && in_tile == CPT) {                 // the tile is complete: park it, start the next one from zero
            in_tile = 0;
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int mt = 0; mt < 4; mt++) {
                    const int row = (mh * 4 + mt) * 16 + r, cgw = (ng4 * 2 + j) * 4 + q;
                    *(float4 *)(ring + ps_stage_off(row, cgw)) = make_float4(acc[j][mt][0], acc[j][mt][1], acc[j][mt][2], acc[j][mt][3]);
                    acc[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g < G) {
            const unsigned wa = w_addr + so + 1024, ba = b_addr[1] + so;
            LDS_RD(wB[0], wa, 0); LDS_RD(wB[1], wa, 2048);
            LDS_RD(bB[0], ba, 0); LDS_RD(bB[1], ba, 2048); LDS_RD(bB[2], ba, 4096); LDS_RD(bB[3], ba, 6144);
            asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mm(wA, bA);
            __builtin_amdgcn_sched_barrier(0);
            in_tile++;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ------------------------------------------------------------------------------------
// 256-row tiles for GEMMs with more than one wave of them (round 4; M >= 1 792: 128+ streams x R = 13, buffered audio, TitaNet-L).
// At 128 x 128 a 64-deep chunk moves 32 KiB for 2.1 MFLOP: at the MFMA rate the chip sustains that is ~19 TB/s of operand traffic out
// of L2 / MALL / HBM, and with weights that come from HBM (every GEMM of the engine: a layer's matrices are read once per step) the
// loop waits for memory however its fills and epilogues are hidden (profiles/r4_persistent_gemm.md).  A 256 (m) x BN (n) tile moves
// (256 + BN) / (2 BN) as many operand bytes per flop: 0.5 at BN = 256 (the form that ships; BN = 128, 0.75, measured worse than the
// per-tile pair with cold operands and is not launched: profiles/r4_wide_tiles.md).
//   8 waves = 2 halves of the rows x 4 quarters of the columns: 128 m x BN / 4 n per wave (BN = 256: 32 accumulators of 16 x 16);
//   32-deep chunks (256 x 64 B activation panel + BN / 16 weight tiles of 1 KiB) by LDS-DMA into a 4-slot ring, three in flight;
//   per chunk a wave reads BN / 64 weight and 8 activation fragments for 8 BN / 64 MFMAs (0.375 ds_read_b128 per MFMA at BN = 256);
//   epilogue staged through the ring in four 64-row quarters, then the same per-row stores as the other kernels.
// Same v_mfma_f32_16x16x32_bf16, k ascending from zero per accumulator: the bits of every other kernel of this file
// (tests/micro/gemm_variant_identity.py, engine option "wide_tiles" = 0 / 1).
// ------------------------------------------------------------------------------------
constexpr int WD_NS = 4;
// MT = 16-row m-tiles per wave: the tile has BM = 32 MT rows.  MT = 8 (256 rows) and MT = 7 (224 rows = 16 streams x R = 13: 7 168 rows
// are 32 of them, so that N = 4096 gives 512 tiles = two FULL rounds of the chip where 256-row tiles give 1.75, and N = 2048 one
// round of 256 smaller tiles instead of 224 larger ones) -- launch_gemm_bf16 takes the one whose rounds x rows is smaller.
template <int BN, int MT> struct WideCfg {
    static constexpr int BM = 32 * MT;
    static constexpr int SLOT = (BM + BN) * 64;               // bytes per 32-deep chunk
    static constexpr int NT = BN / 64;                        // weight fragments (16-row tiles) per wave and chunk
    static constexpr int NP = BM / 16;                        // LDS-DMA pieces of the activation panel (16 rows x 64 B each)
    static constexpr int PIECES = NP + BN / 16;               // + the weight tiles of 1 KiB
    static constexpr int DMA = (PIECES + 7) / 8;              // LDS-DMA instructions per wave and chunk (a wave without a piece of its own repeats the last one)
    static constexpr int STG_LD = BN + 4;                     // floats per staged row
    static constexpr size_t LDS = (size_t)WD_NS * SLOT > (size_t)64 * (BN + 4) * 4 ? (size_t)WD_NS * SLOT : (size_t)64 * (BN + 4) * 4;
};
template <int BN, int MT>
__global__ __launch_bounds__(512) void k_gemm_wide(GemmParams p, int n_groups, int m_chunks) {
    using C = WideCfg<BN, MT>;
    constexpr int P = WD_NS - 1, NT = C::NT, DMA = C::DMA, BM = C::BM;
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int nblk = gridDim.x;
    int id = blockIdx.x;
    {
        const int qd = nblk >> 3, rm = nblk & 7, xcd = id & 7, loc = id >> 3;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    }
    int mc, ng, split_unused;
    tile_of(id, n_groups, m_chunks, p.tile_bands == 2 ? 2 : 1, mc, ng, split_unused);          // bands of column groups whatever the row count (tile_of() above)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nq = wave & 3, mh = wave >> 2, q = lane >> 4, r = lane & 15;
    const int KT = p.K >> 5, m0 = mc * BM;
    // this wave's share of a chunk's DMA: instruction j = wave * DMA + u; j < NP: panel rows [16 j, 16 j + 16), else weight tile j - NP
    const char *src[DMA];
    unsigned dst[DMA];
    int step[DMA];
#pragma unroll
    for (int u = 0; u < DMA; u++) {
        int j = wave * DMA + u;
        if (j >= C::PIECES) j = C::PIECES - 1;          // the same bytes to the same place a second time: keeps every wave's vmcnt arithmetic equal
        if (j < C::NP) {
            const int row = j * 16 + (lane >> 2);
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            src[u] = a_row_ptr(p, m, 2) + (((lane & 3) ^ ((0 - (row >> 2)) & 3)) << 4);
            dst[u] = (unsigned)(j * 1024);
            step[u] = 64;
        } else {
            const int t = j - C::NP;
            src[u] = (const char *)p.W + (size_t)(ng * (BN / 16) + t) * KT * 1024 + lane * 16;
            dst[u] = (unsigned)(BM * 64 + t * 1024);
            step[u] = 1024;
        }
    }
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)ring;
    auto issue = [&](int kt, int slot) {
        const unsigned sb = ring_base + slot * C::SLOT;
#pragma unroll
        for (int u = 0; u < DMA; u++) glds16(src[u] + (size_t)kt * step[u], sb + dst[u]);
    };
    f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < P; i++)
        if (i < KT) issue(i, i);
    int slot = 0;
    for (int i = 0; i < KT; i++) {
        const int left = KT - 1 - i;                            // chunks allowed to stay in flight: min(left, P - 1)
        if (left >= P - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA * (P - 1)) : "memory");
        else if (left == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave's part of chunk i has landed; chunk i - 1 is fully consumed
        if (i + P < KT) issue(i + P, slot == 0 ? WD_NS - 1 : slot - 1);          // into the slot chunk i - 1 has just left
        const char *sp = ring + slot * C::SLOT;
        uint4 wf[NT];
#pragma unroll
        for (int j = 0; j < NT; j++) wf[j] = *(const uint4 *)(sp + BM * 64 + (nq * NT + j) * 1024 + lane * 16);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const uint4 bv = *(const uint4 *)(sp + panel32_off(mh * (BM / 2) + mt * 16 + r, q));
            const bf16x8 bf = __builtin_bit_cast(bf16x8, bv);
#pragma unroll
            for (int j = 0; j < NT; j++) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]), bf, acc[j][mt], 0, 0, 0);
        }
        slot = slot + 1 == WD_NS ? 0 : slot + 1;
    }
    // epilogue: each half of the rows in two parts (64 rows, then the rest: 64 or 48) through an f32 tile [64][BN + 4] in the ring
    float *stage = (float *)ring;
#pragma unroll
    for (int qr = 0; qr < 4; qr++) {
        constexpr int HALF = BM / 2;
        const int part = qr & 1, rows0 = (qr >> 1) * HALF + part * 64, nrows = part ? HALF - 64 : 64;
        // EPI_RESID_F32: the residual values this pass adds to are requested BEFORE the pass's two barriers and its staging, not inside the store
        // loop (round 5: there the round trip sat in each of the four passes: 512 streams, k_gemm +2.5 us per launch with the fold, k_post -5.6)
        float4 xq[8];
        if (p.epi == EPI_RESID_F32) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = threadIdx.x + k * 512, row = e / (BN / 4), m = m0 + rows0 + row;
                xq[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < nrows * (BN / 4) && m < p.M) xq[k] = *(const float4 *)(p.resid + (size_t)m * p.ldo + ng * BN + (e - row * (BN / 4)) * 4);
            }
        }
        __syncthreads();                       // the ring (or the previous part) is no longer read
        if (mh == (qr >> 1)) {
#pragma unroll
            for (int j = 0; j < NT; j++)
#pragma unroll
                for (int mt4 = 0; mt4 < 4; mt4++) {
                    if (part * 4 + mt4 >= MT) continue;
                    const f32x4 &a = acc[j][part * 4 + mt4];
                    *(float4 *)(stage + (mt4 * 16 + r) * C::STG_LD + (nq * NT + j) * 16 + q * 4) = make_float4(a[0], a[1], a[2], a[3]);
                }
        }
        __syncthreads();
        if (!p.narrow_stores && (p.epi == EPI_SILU_ACT || p.epi == EPI_GLU || p.epi == EPI_QKV)) {          // eight columns per thread: 16-byte stores (epi_oct)
            for (int e = threadIdx.x; e < nrows * (BN / 8); e += 512) {
                const int row = e / (BN / 8), c8 = (e - row * (BN / 8)) * 8, m = m0 + rows0 + row, n0 = ng * BN + c8;
                if (m >= p.M) continue;
                const float4 a = *(const float4 *)(stage + row * C::STG_LD + c8), b = *(const float4 *)(stage + row * C::STG_LD + c8 + 4);
                const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                if (!epi_oct(p, m, n0, v)) { epi_quad<true>(p, 0, m, n0, a.x, a.y, a.z, a.w); epi_quad<true>(p, 0, m, n0 + 4, b.x, b.y, b.z, b.w); }
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {          // 64 rows x BN / 4 quads = 8 per thread at BN = 256
            const int e = threadIdx.x + k * 512;
            if (e >= nrows * (BN / 4)) break;
            const int row = e / (BN / 4), c4 = (e - row * (BN / 4)) * 4, m = m0 + rows0 + row, n0 = ng * BN + c4;
            if (m >= p.M) continue;
            const float4 v = *(const float4 *)(stage + row * C::STG_LD + c4);
            if (p.epi == EPI_PART_F32) store_wt_f4(p.out_f32 + (size_t)m * p.ldo + n0, v);
            else if (p.epi == EPI_SILU_ACT) store_wt_u2((bf16_t *)p.out_act + (size_t)m * p.ldo_act + n0, pack4_bf16(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w)));
            else if (p.epi == EPI_RESID_F32)
                store_wt_f4(p.out_f32 + (size_t)m * p.ldo + n0, make_float4(__builtin_fmaf(p.resid_scale, v.x, xq[k].x), __builtin_fmaf(p.resid_scale, v.y, xq[k].y),
                                                                           __builtin_fmaf(p.resid_scale, v.z, xq[k].z), __builtin_fmaf(p.resid_scale, v.w, xq[k].w)));
            else epi_quad<true>(p, 0, m, n0, v.x, v.y, v.z, v.w);
        }
    }
}

// Row-wise output of one wave's block parked in its own LDS region (stg[row][col], WE_LD floats per row): COLS = 64 or 48 columns, `rows` rows starting at
// (m_base, n_base) of the GEMM.  16-bit outputs: eight columns per lane, eight rows per wave instruction (128-byte segments); f32 outputs: four columns
// per lane, four rows per instruction (256-byte segments); the residual form requests eight instructions' worth of residual values ahead.
constexpr int WE_LD = 68;
template <int COLS>
__device__ __forceinline__ void wave_epilogue_rows(const GemmParams &p, const float *stg, int rows, int m_base, int n_base, int lane) {
    static_assert(COLS == 64 || COLS == 48, "a wave's block is 64 or 48 columns wide");
    if (!p.narrow_stores && (p.epi == EPI_SILU_ACT || p.epi == EPI_GLU || p.epi == EPI_QKV)) {
        constexpr int PER = COLS / 8;          // items of eight columns per row
        for (int e = lane; e < rows * PER; e += 64) {
            const int row = e / PER, c8 = (e - row * PER) * 8, m = m_base + row, n0 = n_base + c8;
            const float4 a = *(const float4 *)(stg + row * WE_LD + c8), b = *(const float4 *)(stg + row * WE_LD + c8 + 4);
            if (m >= p.M) continue;
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            if (!epi_oct(p, m, n0, v)) { epi_quad<true>(p, 0, m, n0, a.x, a.y, a.z, a.w); epi_quad<true>(p, 0, m, n0 + 4, b.x, b.y, b.z, b.w); }
        }
        return;
    }
    constexpr int PER = COLS / 4;              // items of four columns per row
    if (p.epi == EPI_RESID_F32) {
        for (int e0 = lane; e0 < rows * PER; e0 += 8 * 64) {
            float4 xq[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = e0 + k * 64, row = e / PER, m = m_base + row;
                xq[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < rows * PER && m < p.M) xq[k] = *(const float4 *)(p.resid + (size_t)m * p.ldo + n_base + (e - row * PER) * 4);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = e0 + k * 64, row = e / PER, c4 = (e - row * PER) * 4, m = m_base + row;
                if (e >= rows * PER || m >= p.M) continue;
                const float4 v = *(const float4 *)(stg + row * WE_LD + c4);
                store_wt_f4(p.out_f32 + (size_t)m * p.ldo + n_base + c4, make_float4(__builtin_fmaf(p.resid_scale, v.x, xq[k].x), __builtin_fmaf(p.resid_scale, v.y, xq[k].y),
                                                                                  __builtin_fmaf(p.resid_scale, v.z, xq[k].z), __builtin_fmaf(p.resid_scale, v.w, xq[k].w)));
            }
        }
        return;
    }
    for (int e = lane; e < rows * PER; e += 64) {
        const int row = e / PER, c4 = (e - row * PER) * 4, m = m_base + row, n0 = n_base + c4;
        const float4 v = *(const float4 *)(stg + row * WE_LD + c4);
        if (m >= p.M) continue;
        if (p.epi == EPI_PART_F32) store_wt_f4(p.out_f32 + (size_t)m * p.ldo + n0, v);
        else if (p.epi == EPI_SILU_ACT) store_wt_u2((bf16_t *)p.out_act + (size_t)m * p.ldo_act + n0, pack4_bf16(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w)));
        else epi_quad<true>(p, 0, m, n0, v.x, v.y, v.z, v.w);
    }
}

// ------------------------------------------------------------------------------------
// k_gemm_wide2 (round 5): the same 224 x 256 tile, the same MFMAs in the same order (k ascending per accumulator: the bits of k_gemm_wide), another
// loop.  k_gemm_wide's iteration is barrier -> 4 DMA instructions -> 6 + 5 ds_read_b128 -> wait -> 28 MFMAs: both waves of a SIMD stand in the same
// phase at the top of every chunk.  Here the fragments of chunk i + 1 are read into a SECOND register set while chunk i's MFMAs run (the barrier of
// iteration i certifies chunk i + 1, one ahead), and the DMA instructions of chunk i + 5 go out one by one between the MFMA groups.  Five ring slots
// (150 KiB) keep the DMA look-ahead at four iterations.  Needs K / 32 even and >= 8.
// Measured (in-kernel stamps, tests/micro/wide_stamps.hip, profiles/r5_gemm_tile_stamps.md): the K loop stayed at ~1 300 cycles per chunk (896 of MFMA) -- what bounds it
// is the issue of the LDS-DMA instructions themselves (~100+ cycles each), not the order around them; what the kernel gains (W2 at 7 168 rows 94 -> 84 us, W1 71 -> 68,
// a 512-stream step 13.8 -> 13.5 ms) comes from the deeper ring, the reads off the critical path and the epilogue below.
// ------------------------------------------------------------------------------------
constexpr int W2_NS = 5;
template <int BN, int MT> constexpr int wide2_lds() {          // the ring, or the eight wave-private epilogue regions (64 rows x WE_LD floats each), whichever is larger
    return W2_NS * WideCfg<BN, MT>::SLOT > 8 * 64 * 68 * 4 ? W2_NS * WideCfg<BN, MT>::SLOT : 8 * 64 * 68 * 4;
}
template <int BN, int MT>
__global__ __launch_bounds__(512) void k_gemm_wide2(GemmParams p, int n_groups, int m_chunks) {
    using C = WideCfg<BN, MT>;
    constexpr int NS = W2_NS, NT = C::NT, DMA = C::DMA, BM = C::BM;
    GSTAMP(0);
    static_assert(MT >= DMA + 1, "one DMA instruction after each of the first DMA MFMA groups");
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int nblk = gridDim.x;
    int id = blockIdx.x;
    {
        const int qd = nblk >> 3, rm = nblk & 7, xcd = id & 7, loc = id >> 3;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    }
    int mc, ng, split_unused;
    tile_of(id, n_groups, m_chunks, p.tile_bands == 2 ? 2 : 1, mc, ng, split_unused);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nq = wave & 3, mh = wave >> 2, q = lane >> 4, r = lane & 15;
    const int KT = p.K >> 5, m0 = mc * BM;
    const char *src[DMA];
    unsigned dst[DMA];
    int step[DMA];
#pragma unroll
    for (int u = 0; u < DMA; u++) {
        int j = wave * DMA + u;
        if (j >= C::PIECES) j = C::PIECES - 1;
        if (j < C::NP) {
            const int row = j * 16 + (lane >> 2);
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            src[u] = a_row_ptr(p, m, 2) + (((lane & 3) ^ ((0 - (row >> 2)) & 3)) << 4);
            dst[u] = (unsigned)(j * 1024);
            step[u] = 64;
        } else {
            const int t = j - C::NP;
            src[u] = (const char *)p.W + (size_t)(ng * (BN / 16) + t) * KT * 1024 + lane * 16;
            dst[u] = (unsigned)(BM * 64 + t * 1024);
            step[u] = 1024;
        }
    }
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)ring;
    f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 fw[2][NT], fa[2][MT];
    const unsigned w_off = BM * 64 + nq * NT * 1024 + lane * 16;
    unsigned a_off[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) a_off[mt] = panel32_off(mh * (BM / 2) + mt * 16 + r, q);
    // prologue: chunks 0 .. NS - 1 on their way, chunk 0 landed, its fragments into set 0
#pragma unroll
    for (int c = 0; c < NS; c++)
#pragma unroll
        for (int u = 0; u < DMA; u++) glds16(src[u] + (size_t)c * step[u], ring_base + c * C::SLOT + dst[u]);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA * (NS - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    GSTAMP(1);
#pragma unroll
    for (int j = 0; j < NT; j++) fw[0][j] = *(const uint4 *)(ring + w_off + j * 1024);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) fa[0][mt] = *(const uint4 *)(ring + a_off[mt]);
    int slot = 0;          // slot of chunk i
    auto body = [&](int i, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value, nxt = 1 - cur;
        const bool has_next = i + 1 < KT;
        const int nslot = slot + 1 == NS ? 0 : slot + 1;
        if (has_next) {
            const int left = KT - 2 - i;          // chunks after i + 1 that may still be in flight: min(left, NS - 2)
            if (left >= NS - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMA * (NS - 2)) : "memory");
            else if (left == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMA * 2) : "memory");
            else if (left == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // chunk i + 1 has landed (every wave's pieces); every wave holds chunk i in registers: its slot is free
        }
        const char *sp = ring + nslot * C::SLOT;
        const bool more = i + NS < KT;
        const unsigned sb = ring_base + slot * C::SLOT;
        // the first MFMA group goes out before any new read: hipcc's own lgkmcnt wait in front of it (it cannot see the asm wait above) then waits for nothing
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
#pragma unroll
            for (int j = 0; j < NT; j++) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[cur][j]), __builtin_bit_cast(bf16x8, fa[cur][mt]), acc[j][mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) {
                if (mt == 0) {
#pragma unroll
                    for (int j = 0; j < NT; j++) fw[nxt][j] = *(const uint4 *)(sp + w_off + j * 1024);
                } else {
#pragma unroll
                    for (int k = 3 * (mt - 1); k < 3 * mt; k++)
                        if (k < MT) fa[nxt][k] = *(const uint4 *)(sp + a_off[k]);
                }
            }
            if (mt < DMA && more) glds16(src[mt] + (size_t)(i + NS) * step[mt], sb + dst[mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        slot = nslot;
    };
    for (int i = 0; i < KT; i += 2) {
        body(i, std::integral_constant<int, 0>{});
        body(i + 1, std::integral_constant<int, 1>{});
    }
    GSTAMP(2);
    // epilogue, wave-private (round 5: in-kernel stamps of the staged four-pass form of k_gemm_wide: 23 000 cycles of a 71 500-cycle W1 tile, eight
    // workgroup barriers with half of the waves idle in each staging step).  Every wave parks ITS 112 x 64 block in its own 17 KiB of the ring, 64 rows then
    // 48, and writes it out row-wise: 128-byte (16-bit outputs) or 256-byte (f32) row segments, no barrier after the one that ends the K loop
    // (a wave's LDS accesses execute in order).  Same epi_oct / epi_quad per item: same values.
    __syncthreads();                           // every wave is done with the ring
    float *stg = (float *)ring + wave * (64 * WE_LD);
#pragma unroll
    for (int part = 0; part < 2; part++) {
        constexpr int HALF = BM / 2;
        const int nmt = part ? MT - 4 : 4;
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int mt4 = 0; mt4 < 4; mt4++) {
                if (part * 4 + mt4 >= MT) continue;
                const f32x4 &a = acc[j][part * 4 + mt4];
                *(float4 *)(stg + (mt4 * 16 + r) * WE_LD + j * 16 + q * 4) = make_float4(a[0], a[1], a[2], a[3]);
            }
        wave_epilogue_rows<NT * 16>(p, stg, nmt * 16, m0 + mh * HALF + part * 64, ng * BN + nq * (NT * 16), lane);
    }
    GSTAMP(3);
}

static size_t gemm_lds_bytes(int ns) {     // the ring, or the f32 tile the epilogue parks in it, whichever is larger
    const size_t ring = (size_t)ns * G2_SLOT, stage = (size_t)TM * STG_LD * 4;
    return ring > stage ? ring : stage;
}

static size_t gemm_k32_lds_bytes(int ns) {
    const size_t ring = (size_t)ns * K32_SLOT, stage = (size_t)TM * STG_LD * 4;
    return ring > stage ? ring : stage;
}

constexpr int F32M_KC = 32, F32M_NS = 3;
template <int BM, int BN> __global__ void k_gemm_f32_mfma(GemmParams p, int n_groups, int m_chunks);      // the f32 engine's MFMA kernel, below

static int g_num_cus = 256;      // MI355X; refreshed from the device below
void init_gemm_kernel_attributes() {
    hipFuncSetAttribute((const void *)k_gemm_tiled2<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(4));
    hipFuncSetAttribute((const void *)k_gemm_roles<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(4));
    hipFuncSetAttribute((const void *)k_gemm_t64<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * T64_SLOT);
    hipFuncSetAttribute((const void *)k_gemm_t64<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * T64_SLOT);
    hipFuncSetAttribute((const void *)k_gemm_t64w, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * T64W_HALF);
    hipFuncSetAttribute((const void *)k_gemm_tiled2_k32<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_k32_lds_bytes(4));
    hipFuncSetAttribute((const void *)k_gemm_tiled3, hipFuncAttributeMaxDynamicSharedMemorySize, T3_NS * K32_SLOT);
    hipFuncSetAttribute((const void *)k_gemm_persist<EPI_PART_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS);
    hipFuncSetAttribute((const void *)k_gemm_persist<EPI_SILU_ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS);
    hipFuncSetAttribute((const void *)k_gemm_persist<EPI_QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS);
    hipFuncSetAttribute((const void *)k_gemm_persist<EPI_GLU>, hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS);
    hipFuncSetAttribute((const void *)k_gemm_persist<EPI_BIAS_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS);
    hipFuncSetAttribute((const void *)k_gemm_persist<EPI_BIAS_RELU_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS);
    hipFuncSetAttribute((const void *)k_gemm_wide<256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WideCfg<256, 8>::LDS);
    hipFuncSetAttribute((const void *)k_gemm_wide<256, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WideCfg<256, 7>::LDS);
    hipFuncSetAttribute((const void *)k_gemm_wide2<256, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, wide2_lds<256, 7>());
    hipFuncSetAttribute((const void *)k_gemm_wide2<192, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, wide2_lds<192, 7>());
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) g_num_cus = cus;
    hipFuncSetAttribute((const void *)k_gemm_f32_mfma<128, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, F32M_NS * 256 * 128);
    hipFuncSetAttribute((const void *)k_gemm_f32_mfma<64, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, F32M_NS * 128 * 128);
}

// 128 x 64 tiles (k_gemm_t64): for the split-K GEMMs with N = 1024 when that halves the split factor, and for any other GEMM
// whose 128 x 128 tiling gives at most 64 workgroups (a quarter of the CUs).  With pipelined steps "fill the chip" is the wrong
// rule for the in-between sizes: CUs one launch leaves idle run another chain's kernels.  Measured with three lanes, half-width
// tiles wherever the 128 x 128 tiling had <= 128 workgroups against the split-K form only: 16 streams x R = 13 (32-64 tiles)
// 1.22 vs 1.32 ms per step, 32 streams (64-128 tiles) 1.82 vs 1.77, 64 streams (pw1: 112 tiles -> 224) 2.66 vs 2.65 -- although
// alone the 112-tile launch takes 13.3 us and the vendor library's MT128x64 kernel 10.6 (tests/prof_gemm_shapes.sh).
// Round 4: for the split-K GEMMs only up to 64 tiles of 128 x 128 (M <= 1 024).  Above that two K-halves of 128 x 128 tiles fill the chip by themselves
// (72-120 tiles x 2 <= 256 workgroups) where the half-width form makes 288-480 workgroups of a shape that moves 1.5 x the operand bytes per flop: 112 streams
// x R = 13 (104 tiles) synchronous 5.92 -> 5.30 ms, pipelined 3.82 -> 3.77; 128 streams (112 tiles) pipelined 4.21 -> 4.14, synchronous 5.52 -> 5.69
// (profiles/r4_tile_order.md).  Engine option "t64_tiles" (the choice is made in two places that must agree: both read GemmParams::t64_tiles_p1).
// t64_p1 = GemmParams::t64_tiles_p1 (the engine's option + 1; 0 = the default of 64): carried per GEMM so that several engines in one
// process cannot change each other's choice between the two places that must agree (round-4 advisor: it was a process-wide global).
bool gemm_use_t64(int M, int N, int epi, int t64_p1) {
    if (M <= gemm_skinny_max_m()) return false;
    const int tiles = (N / 128) * ((M + 127) / 128);
    if (epi == EPI_PART_F32) return N == 1024 && tiles <= (t64_p1 > 0 ? t64_p1 - 1 : 64);
    return tiles <= 64;
}
int gemm_tile_n(int M, int N, int epi, int t64_p1) { return gemm_use_t64(M, N, epi, t64_p1) ? 64 : 128; }
// EPI_RESID_F32 needs the complete K sum in one workgroup: the welded two-slice 128 x 64 form where pick_splits chose two slices of
// half-width tiles (256 < M <= 1 024 at the default "t64_tiles"), or any launch without split-K
static bool gemm_welded(int M, int N, int K, int splits, int t64_p1) {
    return splits == 2 && (K & 127) == 0 && gemm_use_t64(M, N, EPI_PART_F32, t64_p1);
}
// a GEMM whose A rows come out of a k_post can carry that k_post as its head phase (ChainParams) when it runs on the 128 x 128 tiles of
// k_gemm_tiled2_k32: more than 32 rows, no split-K, N a multiple of 128, at most 64 row chunks (the counters)
bool gemm_chain_ok(int M, int N, int K, int splits) {
    return M > gemm_skinny_max_m() && splits == 1 && N % 128 == 0 && (K & 63) == 0 && (M + TM - 1) / TM <= 64;
}
bool gemm_resid_foldable(int M, int N, int K, int splits, int t64_p1) {
    if (M <= gemm_skinny_max_m()) return false;
    return splits == 1 || gemm_welded(M, N, K, splits, t64_p1);
}

// Largest M served by the weight-streaming ("skinny") kernel; above it the LDS-tiled kernels take over.  Round 1 had 128 (chosen
// on synchronous steps).  Re-measured in round 2 (ms per step, <= 32 / <= 64 / <= 128 rows skinny):
//   three lanes: 64 streams x R = 0 (M = 64) 1.15 / 1.33 / 1.33; 40 / 48 streams x R = 0 1.04 / 1.13 and 1.06 / 1.19 / -;
//                64 x R = 1 (M = 128) 1.29 / 1.30 / 1.89; 8 x R = 13 and 16 x R = 6 (M = 112) 1.13 / 1.13 / 1.62;
//                32 rows and fewer: skinny wins (32 streams x R = 0 0.94 against 1.00 tiled; 16 x R = 1 0.89 / 0.99)
//   synchronous: M = 64 2.66 / - / 2.56, M = 48 2.56 / - / 2.41, M = 112 2.66 / - / 2.81, M = 128 2.95 / - / 3.17
// -> 32 rows.  ONE threshold for both modes: which kernel a GEMM runs on must not depend on the mode, or pipelined steps would
// stop being bit-identical to synchronous ones (the synchronous step pays <= 6 % for it between 33 and 64 rows and gains above).
int gemm_skinny_max_m() {
    return 32;
}

// Pipelined steps (GemmParams::coresident): four launch chains advance in lock-step rounds, so the GEMM launches of a round start
// together and, with one 96-128 KiB workgroup per CU, run one after the other -- a round costs the SUM of its GEMMs.  With rings of
// <= 72 KiB two of them share every CU: one workgroup's ring fill and barrier waits run under the other's MFMAs (64 streams x
// R = 13: 2.64 -> 2.48 ms per step; alone on the chip the shallower rings cost 8 %, so synchronous steps keep the deep ones).
// From seven 128-row tiles up (M > 768), where every GEMM of the step covers most of the chip: measured per step with four lanes,
// 64 streams x R = 13 (M = 896) 2.61 -> 2.47 ms, 48 streams (M = 672) 2.05 -> 2.03, 40 streams (M = 560) 1.76 -> 1.81, 32 streams 1.51 -> 1.59.
static bool gemm_coresident(const GemmParams &p) {
    constexpr int min_m = 769;
    if (p.coresident >= 2) return p.coresident == 2;          // engine option "gemm_cores" (A/B runs, the bit-identity test)
    // more than one wave of tiles (M >= 1 792): workgroups of ONE launch start as earlier ones finish, so the two on a CU are out of
    // phase by themselves -- synchronous steps gain as well (128 streams x R = 13: 6.51 -> 5.95 ms, 512 streams 19.9 -> 18.6 ms).
    // Only with more tiles than CUs: a launch that puts at most one workgroup on a CU has nothing to pair and keeps the deep rings
    // (cold operands, us per launch, deep / shallow: 1 792 rows pw1 224 tiles 12.6 / 16.7, W2 112 tiles 28.3 / 32.4; 3 584 rows W2 224 tiles
    // 34.7 / 45.1, Wo 12.0 / 14.8 -- profiles/r4_tile_order.md)
    if (p.coresident == 1 && p.M >= min_m) return true;
    const long tiles = (long)(p.N / (gemm_use_t64(p.M, p.N, p.epi, p.t64_tiles_p1) ? 64 : 128)) * ((p.M + TM - 1) / TM) * (p.splits < 1 ? 1 : p.splits);
    return p.M >= 1792 && tiles > g_num_cus;
}

// round 5's loops (k_gemm_wide2, k_gemm_tiled3) unless GemmParams::prio >> 2 == 5: rounds 1-4's (engine option "gemm_prio" = 20: A/B runs, gemm_variant_identity.py).
// The probes this field also selected during the round (s_setprio around the MFMA cluster, "every fragment first", DMA between the MFMA groups on the old kernels) are
// gone from the tree: profiles/r5_gemm_tile_stamps.md, r5_gemm_loops_probe_{7168,896}.txt.
static bool gemm_new_loops(const GemmParams &p) { return (p.prio >> 2) == 0 || (p.prio >> 2) == 4; }
void launch_gemm_bf16(const GemmParams &p0, hipStream_t st) {
    GemmParams p = p0;
    if (p.splits < 1) p.splits = 1;
    if (p.M <= gemm_skinny_max_m()) {
        dim3 grid(p.N / 16, p.splits, (p.M + 63) / 64);
        if (p.M <= 16) hipLaunchKernelGGL(k_gemm_skinny<1>, grid, dim3(256), 0, st, p);
        else if (p.M <= 32) hipLaunchKernelGGL(k_gemm_skinny<2>, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(k_gemm_skinny<4>, grid, dim3(256), 0, st, p);
    } else {
        int n_groups = p.N / 128, m_chunks = (p.M + TM - 1) / TM;
        if (p.chain.head_wgs > 0) {                        // chained launch (the caller asked gemm_chain_ok()): k_gemm_tiled2_k32 is the kernel that carries a head phase
            hipLaunchKernelGGL(k_gemm_tiled2_k32<4>, dim3(p.chain.head_wgs + n_groups * m_chunks * p.splits), dim3(512), gemm_k32_lds_bytes(4), st, p, n_groups, m_chunks);
            return;
        }
        if (p.epi == EPI_RESID_F32 && p.splits == 2) {      // the caller asked gemm_resid_foldable(): both K slices in one 16-wave workgroup
            GemmParams w = p;
            w.splits = 1;                                  // tile_of(): one workgroup per tile
            hipLaunchKernelGGL(k_gemm_t64w, dim3((p.N / 64) * m_chunks), dim3(1024), 2 * T64W_HALF, st, w, p.N / 64, m_chunks);
            return;
        }
        if (gemm_use_t64(p.M, p.N, p.epi, p.t64_tiles_p1)) {       // half-width tiles: the caller chose splits for N / 64 column groups (gemm_tile_n)
            n_groups = p.N / 64;
            if (gemm_coresident(p))      // 3 slots = 72 KiB: two workgroups per CU
                hipLaunchKernelGGL(k_gemm_t64<3>, dim3(n_groups * m_chunks * p.splits), dim3(512), 3 * T64_SLOT, st, p, n_groups, m_chunks);
            else
                hipLaunchKernelGGL(k_gemm_t64<4>, dim3(n_groups * m_chunks * p.splits), dim3(512), 4 * T64_SLOT, st, p, n_groups, m_chunks);
            return;
        }
        // more than one wave of tiles: 256 x 256 tiles (half the operand bytes per flop) where their rounds fill the chip -- the last round at
        // least 5 / 8 full, or three rounds and more (persist_probe, cold operands, us per launch against the per-tile pair: 7 168 rows W1 448
        // tiles 78 / 83, pw1 224 tiles 38 / 45, QKV 336 tiles 69 / 70: a wash, left alone; 15 360 rows N = 1024 240 tiles 109 / 160).  The
        // 256 x 128 form measured worse than the per-tile kernels with cold operands (W2 at 7 168 rows 92 / 77) and is not used.
        // Synchronous steps only -- alone on the chip the QKV launch at 7 168 rows takes 52 us instead of 62, but a pipelined 512-stream step got SLOWER with it
        // (13.65 against 13.50 ms, same box, three-way A/B): the half-empty round is where the other lanes' launches run.
        // N = 3072 (QKV) on 224 x 192 tiles where 224 x 256 leaves a half-empty last round: 7 168 rows 384 tiles = 1.5 rounds -> 512 = two full rounds of
        // 3 / 4-size tiles, 3 584 rows 192 tiles (0.75 of the chip) -> 256.  k_gemm_wide2 only (its wave-private epilogue takes 48-column blocks).
        if (!p.no_wide && gemm_new_loops(p) && p.coresident != 1 && p.splits == 1 && p.M >= 1792 && (p.K & 63) == 0 && p.K >= 256 && p.N % 192 == 0 && p.N % 256 == 0) {
            const long mw = (p.M + 223) / 224, t192 = (long)(p.N / 192) * mw, t256 = (long)(p.N / 256) * mw;
            const long c192 = (t192 + g_num_cus - 1) / g_num_cus * 192, c256 = (t256 + g_num_cus - 1) / g_num_cus * 256;
            if (c192 < c256 && t192 >= (long)g_num_cus * 7 / 8) {
                hipLaunchKernelGGL((k_gemm_wide2<192, 7>), dim3((unsigned)t192), dim3(512), (wide2_lds<192, 7>()), st, p, p.N / 192, (int)mw);
                return;
            }
        }
        if (!p.no_wide && p.splits == 1 && p.M >= (p.coresident == 1 ? (p.wide_min_rows > 0 ? p.wide_min_rows : 1344) : 1792) && (p.K & 31) == 0 && p.N % 256 == 0) {
            // 256- or 224-row tiles: whichever needs fewer rounds x rows (7 168 rows: N = 4096 two full rounds of 224-row tiles instead of
            // 1.75 of 256-row ones, N = 2048 one round of 256 smaller tiles; 15 360 rows stay at 256).  Cold operands, us per launch, 256 / 224 rows:
            // W1 at 7 168 rows 78.8 / 74.9, pw1 38.4 / 35.4, W1 at 3 584 rows 43.8 / 41.0; synchronous steps 512 streams 18.08 -> 17.82 ms, 256
            // streams 9.56 -> 9.42; pipelined steps (three pieces) 384 streams 10.95 -> 10.82, 512 streams 14.32 = (profiles/r4_wide_tiles.md).
            int best_mt = 0;
            long best_cost = 0, best_tiles = 0;
            for (int mt = 8; mt >= 7; mt--) {
                const int bm = 32 * mt, mw = (p.M + bm - 1) / bm;
                const long tiles = (long)(p.N / 256) * mw, last = tiles % g_num_cus;
                if (!(tiles >= (long)g_num_cus * 7 / 8 && (last == 0 || last * 8 >= (long)g_num_cus * 5 || tiles >= (long)g_num_cus * 3))) continue;
                const long cost = (tiles + g_num_cus - 1) / g_num_cus * bm;
                if (!best_mt || cost < best_cost) { best_mt = mt; best_cost = cost; best_tiles = tiles; }
            }
            // Pipelined steps (other lanes' workgroups fill the CUs a launch leaves idle): 224-row tiles from 32 of them (round 4: 96; round 5, profiles/r5_gemm_tile_stamps.md section 3:
            // what a pipelined step pays for a GEMM is its CU-time, and a 224 x 256 tile costs 40 % less of it than four 128 x 128 ones -- 256 streams 7.4 -> 7.2 ms), where the rule above finds
            // too few to fill the chip.  What a pipelined step is short of is operand delivery -- at 64 streams the LDS fills of a step's
            // 128 x 128 tiles add up to 23 GB = 9.5 TB/s, between what the Infinity Cache (8.6) and an XCD's L2 (17-19) deliver -- and a
            // 224 x 256 tile moves 0.54 of the bytes per flop.  ms per step, four lanes, without / with: 96 streams 3.37 / 3.33, 128 streams
            // 4.30 / 4.19, 192 streams 6.22 / 6.02; from 64 tiles: 4.21 (128 streams), 3.35 (96); at 64 streams (64 / 48 / 32 tiles) 2.42 -> 2.54, W1's 64 tiles alone 2.415 -> 2.449: not taken.
            // With it: 256 streams 8.00 -> 7.93, 384 streams 11.75 -> 11.43, 512 streams (every GEMM of the layer on these tiles) 15.31 -> 14.49.
            if (!best_mt && p.coresident == 1 && p.wide_rows != 2) {
                const long tiles = (long)(p.N / 256) * ((p.M + 223) / 224);
                if (tiles >= (p.wide_min_tiles > 0 ? p.wide_min_tiles : 32)) { best_mt = 7; best_tiles = tiles; }          // engine option "wide_min_tiles"
            }
            if (p.wide_rows == 256 && best_mt) { best_mt = 8; best_tiles = (long)(p.N / 256) * ((p.M + 255) / 256); }      // engine option "wide_tiles" = 256: round 4's first form only
            if (best_mt == 8) {
                hipLaunchKernelGGL((k_gemm_wide<256, 8>), dim3((unsigned)best_tiles), dim3(512), (WideCfg<256, 8>::LDS), st, p, p.N / 256, (p.M + 255) / 256);
                return;
            }
            if (best_mt == 7) {
                if (gemm_new_loops(p) && (p.K & 63) == 0 && p.K >= 256) hipLaunchKernelGGL((k_gemm_wide2<256, 7>), dim3((unsigned)best_tiles), dim3(512), (wide2_lds<256, 7>()), st, p, p.N / 256, (p.M + 223) / 224);
                else hipLaunchKernelGGL((k_gemm_wide<256, 7>), dim3((unsigned)best_tiles), dim3(512), (WideCfg<256, 7>::LDS), st, p, p.N / 256, (p.M + 223) / 224);
                return;
            }
        }
        // several 128 x 128 tiles per CU: the persistent tile loop (one workgroup per CU; ring fills and epilogues off the critical path).
        // From 1.75 tiles per CU: below that a workgroup has no second tile to hide anything under.
        if (!p.no_persist && p.splits == 1 && p.K >= 1024 && (p.K & 63) == 0 && (long)n_groups * m_chunks * 4 >= (long)g_num_cus * 7) {
            const dim3 pgrid(g_num_cus), pblock(1024);
            switch (p.epi) {
            case EPI_PART_F32: hipLaunchKernelGGL(k_gemm_persist<EPI_PART_F32>, pgrid, pblock, PS_LDS, st, p, n_groups, m_chunks); return;
            case EPI_SILU_ACT: hipLaunchKernelGGL(k_gemm_persist<EPI_SILU_ACT>, pgrid, pblock, PS_LDS, st, p, n_groups, m_chunks); return;
            case EPI_QKV: hipLaunchKernelGGL(k_gemm_persist<EPI_QKV>, pgrid, pblock, PS_LDS, st, p, n_groups, m_chunks); return;
            case EPI_GLU: hipLaunchKernelGGL(k_gemm_persist<EPI_GLU>, pgrid, pblock, PS_LDS, st, p, n_groups, m_chunks); return;
            case EPI_BIAS_F32: hipLaunchKernelGGL(k_gemm_persist<EPI_BIAS_F32>, pgrid, pblock, PS_LDS, st, p, n_groups, m_chunks); return;
            case EPI_BIAS_RELU_F32: hipLaunchKernelGGL(k_gemm_persist<EPI_BIAS_RELU_F32>, pgrid, pblock, PS_LDS, st, p, n_groups, m_chunks); return;
            default: break;                        // the act-dtype bias epilogues (subsampling) stay on the per-tile kernels
            }
        }
        dim3 grid(n_groups * m_chunks * p.splits);
        if (gemm_coresident(p) && gemm_new_loops(p) && ((p.K >> 6) / p.splits) * 2 >= 6 && (p.K >> 6) % p.splits == 0) {
            hipLaunchKernelGGL(k_gemm_tiled3, grid, dim3(512), T3_NS * K32_SLOT, st, p, n_groups, m_chunks);
            return;
        }
        if (gemm_coresident(p)) {                  // 4 x 16 KiB ring (+ the staged tile: 66 KiB): two workgroups per CU
            hipLaunchKernelGGL(k_gemm_tiled2_k32<4>, grid, dim3(512), gemm_k32_lds_bytes(4), st, p, n_groups, m_chunks);
            return;
        }
        constexpr int roles_min_chunks = 8;
        const bool roles = (p.K >> 6) / p.splits >= roles_min_chunks;
        const size_t lds = gemm_lds_bytes(4);
        if (roles) hipLaunchKernelGGL(k_gemm_roles<4>, grid, dim3(1024), lds, st, p, n_groups, m_chunks);
        else hipLaunchKernelGGL(k_gemm_tiled2<4>, grid, dim3(512), lds, st, p, n_groups, m_chunks);
    }
}

// ------------------------------------------------------------------------------------
// f32 parity kernel: 64 n x 16 m per workgroup, K chunks of 32, k ascending.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void epi_elem_f32(const GemmParams &p, int m, int n, float v, float vpair) {
    if (m >= p.M) return;
    switch (p.epi) {
    case EPI_PART_F32: p.out_f32[(size_t)m * p.ldo + n] = v; break;
    case EPI_SILU_ACT: ((float *)p.out_act)[(size_t)m * p.ldo_act + n] = silu_f(v); break;
    case EPI_QKV: {
        int which = n >> 10, col = n & 1023;
        if (which == 0) p.q_out[(size_t)m * D + col] = v;
        else {
            int b = m / p.T, i = m - b * p.T;
            RowDesc rd = p.rows[b];
            int ring = rd.kv_head + LCTX + i;
            if (ring >= KVC) ring -= KVC;
            ((float *)p.kv_pool)[(size_t)rd.slot * p.kv_slot_stride + ((size_t)(which - 1) * KVC + ring) * D + col] = v;
        }
    } break;
    case EPI_GLU: if ((n & 1) == 0) p.out_f32[(size_t)m * p.ldo + (n >> 1)] = v * sigmoid_f(vpair); break;
    case EPI_BIAS_F32: p.out_f32[(size_t)m * p.ldo + n] = v + p.bias[n]; break;
    case EPI_BIAS_RELU_F32: p.out_f32[(size_t)m * p.ldo + n] = fmaxf(v + p.bias[n], 0.f); break;
    case EPI_BIAS_RELU_ACT: ((float *)p.out_act)[(size_t)m * p.ldo_act + n] = fmaxf(v + p.bias[n], 0.f); break;
    case EPI_BIAS_ACT: ((float *)p.out_act)[(size_t)m * p.ldo_act + n] = v + p.bias[n]; break;
    }
}

__global__ __launch_bounds__(256) void k_gemm_f32(GemmParams p) {
    __shared__ float As[16][33];
    __shared__ float Ws[64][33];
    const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 16;
    const int tn = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const float *W = (const float *)p.W;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < p.K; k0 += 32) {
        // A tile: 16 x 32 = 512 elements, 2 per thread
        for (int e = threadIdx.x; e < 512; e += 256) {
            int mm = e >> 5, kk = e & 31;
            float v = 0.f;
            if (m0 + mm < p.M) v = ((const float *)a_row_ptr(p, m0 + mm, 4))[k0 + kk];
            As[mm][kk] = v;
        }
        for (int e = threadIdx.x; e < 2048; e += 256) {
            int nn = e >> 5, kk = e & 31;
            Ws[nn][kk] = W[(size_t)(n0 + nn) * p.K + k0 + kk];
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; kk++) {
            float w = Ws[tn][kk];
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = fmaf(As[tg * 4 + j][kk], w, acc[j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float vp = __shfl_xor(acc[j], 1);
        epi_elem_f32(p, m0 + tg * 4 + j, n0 + tn, acc[j], vp);
    }
}

// The f32 engine at one to four rows (a live stream, a handful of streams): k_gemm_f32 above puts 64 weight rows into a
// workgroup, i.e. 16-64 workgroups walk 8-16 MB of f32 weights 8 KiB at a time -- 52 ms per 24-layer step at batch 1.  This form
// gives a workgroup 4 weight rows (256-1024 workgroups), streams them through LDS 1008 k at a time with the next chunk's loads in
// flight under the current chunk's arithmetic, and keeps k_gemm_f32's arithmetic EXACTLY: every output is one chain
// acc = fmaf(a[k], w[k], acc) over ascending k from 0.f -- bit-identical results (the f32 engine is the configuration whose
// tokens and frames equal the oracle's), one lane per (row, weight row).
constexpr int F32R_NR = 4, F32R_KC = 1008, F32R_PAD = 4, F32R_MMAX = 4;     // 2 x 8 rows x 1012 floats = 64 768 B of static LDS
constexpr int F32R_LDS = 2 * (F32R_NR + F32R_MMAX) * (F32R_KC + F32R_PAD);
__global__ __launch_bounds__(256) void k_gemm_f32_rows(GemmParams p) {
    __shared__ __attribute__((aligned(16))) float f32r_lds[F32R_LDS];
    constexpr int LD = F32R_KC + F32R_PAD;
    const int tid = threadIdx.x, n0 = blockIdx.x * F32R_NR, M = p.M;
    const float *W = (const float *)p.W;
    auto Ws = [&](int buf, int r) { return f32r_lds + (size_t)(buf * (F32R_NR + F32R_MMAX) + r) * LD; };
    auto As = [&](int buf, int m) { return f32r_lds + (size_t)(buf * (F32R_NR + F32R_MMAX) + F32R_NR + m) * LD; };
    const int n_chunks = (p.K + F32R_KC - 1) / F32R_KC;
    float4 wreg[F32R_NR], areg[F32R_MMAX];
    auto fetch = [&](int c) {
        const int k = c * F32R_KC + tid * 4;
        const bool in = tid * 4 < F32R_KC && k < p.K;
#pragma unroll
        for (int r = 0; r < F32R_NR; r++) wreg[r] = in ? *(const float4 *)(W + (size_t)(n0 + r) * p.K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int m = 0; m < F32R_MMAX; m++)
            areg[m] = (in && m < M) ? *(const float4 *)((const float *)a_row_ptr(p, m, 4) + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stash = [&](int buf) {
        if (tid * 4 >= F32R_KC) return;
#pragma unroll
        for (int r = 0; r < F32R_NR; r++) *(float4 *)(Ws(buf, r) + tid * 4) = wreg[r];
#pragma unroll
        for (int m = 0; m < F32R_MMAX; m++) *(float4 *)(As(buf, m) + tid * 4) = areg[m];
    };
    fetch(0);
    stash(0);
    __syncthreads();
    const int r = tid & 3, m = tid >> 2;
    const bool active = tid < 4 * M;
    float acc = 0.f;
    for (int c = 0; c < n_chunks; c++) {
        if (c + 1 < n_chunks) fetch(c + 1);
        if (active) {
            const int len = min(F32R_KC, p.K - c * F32R_KC);
            const float4 *w4 = (const float4 *)Ws(c & 1, r), *a4 = (const float4 *)As(c & 1, m);
#pragma unroll 4
            for (int k4 = 0; k4 < len / 4; k4++) {
                const float4 w = w4[k4], a = a4[k4];
                acc = fmaf(a.x, w.x, acc); acc = fmaf(a.y, w.y, acc); acc = fmaf(a.z, w.z, acc); acc = fmaf(a.w, w.w, acc);
            }
        }
        if (c + 1 < n_chunks) stash((c + 1) & 1);
        __syncthreads();
    }
    const float vp = __shfl_xor(acc, 1);
    if (active) epi_elem_f32(p, m, n0 + r, acc, vp);
}

// ------------------------------------------------------------------------------------
// The f32 engine above four rows on the f32-input MFMA (round 4).  v_mfma_f32_32x32x2_f32 multiplies f32 operands exactly and adds the
// products to the accumulator one k after the other -- the fmaf chain of k_gemm_f32 (MI355X_MICROARCH.md, matrix-core table: "exact
// f32 (= fmaf chain, bitwise)") -- so a kernel that feeds it k ascending from a zero accumulator returns k_gemm_f32's bits at the f32
// matrix rate (155 TFLOP/s against the ~10 the FMA tile reaches): the configuration whose tokens equal the oracle's on ANY
// checkpoint becomes usable at batch 64 (tests/test_gpu_parity.py::test_f32_mfma_gemm_is_bit_identical_to_the_fma_tile).
//   tile BM (rows m) x BN (weight rows n), 4 waves = 2 x 2 sub-tiles of (BM/2) x (BN/2), 32-deep K chunks (128 B per row) of both
//   operands by LDS-DMA into a 3-slot ring, 16-byte columns XOR-swizzled by the row on the SOURCE side.  Weights are the MFMA's A
//   operand (rows = n), activations its B operand (columns = m): lane l feeds row / column l & 31 at k = k0 + (l >> 5) and holds
//   D[n = 8 (i / 4) + 4 (l / 32) + i % 4][m = l & 31] in accumulator register i: four consecutive n per quad -> epi_quad<false>.
// ------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int BM, int BN>
__global__ __launch_bounds__(256) void k_gemm_f32_mfma(GemmParams p, int n_groups, int m_chunks) {
    constexpr int ROWS = BM + BN, SLOT = ROWS * 128, DMA = ROWS / 32;      // DMA instructions per wave and chunk (8 rows each)
    constexpr int IM = BM / 64, IN = BN / 64;                              // 32 x 32 blocks per wave along m / n
    extern __shared__ __attribute__((aligned(16))) char ring[];
    int id = blockIdx.x;
    {
        const int nblk = gridDim.x, qd = nblk >> 3, rm = nblk & 7, xcd = id & 7, loc = id >> 3;
        id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    }
    const int mc = id % m_chunks, ng = id / m_chunks;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, kh = lane >> 5;
    const int m0 = mc * BM, n0 = ng * BN, nchunks = p.K / F32M_KC;
    // this wave fills rows [wave * ROWS / 4, + ROWS / 4) of every slot: activation rows first, weight rows behind them
    const char *src[DMA];
#pragma unroll
    for (int j = 0; j < DMA; j++) {
        const int row = wave * (ROWS / 4) + j * 8 + (lane >> 3), c4 = lane & 7;
        const char *base;
        if (row < BM) { int m = m0 + row; if (m >= p.M) m = p.M - 1; base = a_row_ptr(p, m, 4); }
        else base = (const char *)p.W + (size_t)(n0 + row - BM) * p.K * 4;
        src[j] = base + (((c4 ^ (row >> 1)) & 7) << 4);
    }
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)ring;
    auto issue = [&](int kc, int slot) {
        const unsigned sb = ring_base + slot * SLOT + wave * (ROWS / 4) * 128;
#pragma unroll
        for (int j = 0; j < DMA; j++) glds16(src[j] + (size_t)kc * 128, sb + j * 1024);
    };
    f32x16 acc[IN][IM];
#pragma unroll
    for (int a = 0; a < IN; a++)
#pragma unroll
        for (int b = 0; b < IM; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
    issue(0, 0);
    if (nchunks > 1) issue(1, 1);
    int slot = 0;
    for (int i = 0; i < nchunks; i++) {
        if (i + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA) : "memory");     // chunk i has landed, chunk i + 1 may be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // ... for every wave; and every wave is done with chunk i - 1
        if (i + 2 < nchunks) issue(i + 2, slot == 0 ? 2 : slot - 1);     // into the slot chunk i - 1 has just left
        const char *sp = ring + slot * SLOT;
        const char *arow = sp + (wm * (BM / 2) + l31) * 128, *wrow = sp + (BM + wn * (BN / 2) + l31) * 128;
        const int swa = ((wm * (BM / 2) + l31) >> 1) & 7, sww = ((BM + wn * (BN / 2) + l31) >> 1) & 7;     // + 32 rows: the same swizzle ((row >> 1) & 7 has period 16)
#pragma unroll
        for (int k2 = 0; k2 < F32M_KC / 2; k2++) {
            const int k = 2 * k2 + kh;
            const int oa = (((k >> 2) ^ swa) << 4) + ((k & 3) << 2), ow = (((k >> 2) ^ sww) << 4) + ((k & 3) << 2);
            float av[IM], wv[IN];
#pragma unroll
            for (int b = 0; b < IM; b++) av[b] = *(const float *)(arow + b * 32 * 128 + oa);
#pragma unroll
            for (int a = 0; a < IN; a++) wv[a] = *(const float *)(wrow + a * 32 * 128 + ow);
#pragma unroll
            for (int a = 0; a < IN; a++)
#pragma unroll
                for (int b = 0; b < IM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[a], av[b], acc[a][b], 0, 0, 0);
        }
        slot = slot + 1 == F32M_NS ? 0 : slot + 1;
    }
#pragma unroll
    for (int a = 0; a < IN; a++)
#pragma unroll
        for (int b = 0; b < IM; b++) {
            const int m = m0 + wm * (BM / 2) + b * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = n0 + wn * (BN / 2) + a * 32 + g * 8 + kh * 4;
                epi_quad<false>(p, 0, m, n, acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
            }
        }
}
void launch_gemm_f32(const GemmParams &p0, hipStream_t st) {
    GemmParams p = p0;
    p.splits = 1;
    if (p.M <= F32R_MMAX && p.N % F32R_NR == 0 && p.K % 4 == 0 && p.lda % 4 == 0) {
        hipLaunchKernelGGL(k_gemm_f32_rows, dim3(p.N / F32R_NR), dim3(256), 0, st, p);
        return;
    }
    if (!p.f32_fma_tile && p.M > F32R_MMAX && p.N % 64 == 0 && p.K % F32M_KC == 0 && p.lda % 4 == 0 && ((size_t)p.A & 15) == 0 && ((size_t)p.W & 15) == 0 &&
        (p.rows_per_batch == 0 || (p.batch_stride % 4 == 0))) {
        // 128 x 128 tiles when they fill the chip (or the width allows nothing else to), 64 x 64 otherwise: more, smaller workgroups, three per CU
        const long big = (long)(p.N / 128) * ((p.M + 127) / 128);
        if (p.N % 128 == 0 && big >= 192) {
            const int n_groups = p.N / 128, m_chunks = (p.M + 127) / 128;
            hipLaunchKernelGGL((k_gemm_f32_mfma<128, 128>), dim3(n_groups * m_chunks), dim3(256), F32M_NS * 256 * 128, st, p, n_groups, m_chunks);
        } else {
            const int n_groups = p.N / 64, m_chunks = (p.M + 63) / 64;
            hipLaunchKernelGGL((k_gemm_f32_mfma<64, 64>), dim3(n_groups * m_chunks), dim3(256), F32M_NS * 128 * 128, st, p, n_groups, m_chunks);
        }
        return;
    }
    dim3 grid(p.N / 64, (p.M + 15) / 16);
    hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------
// upload-time layout kernels
// ------------------------------------------------------------------------------------
__global__ void k_pack_weight_bf16(const float *w, bf16_t *packed, int N, int K) {
    const int KT = K >> 5;
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one 8-element group
    size_t total = (size_t)(N >> 4) * KT * 64;
    if (g >= total) return;
    int lane = (int)(g & 63);
    size_t tile = g >> 6;
    int kt = (int)(tile % KT), nt = (int)(tile / KT);
    int q = lane >> 4, r = lane & 15;
    const float *src = w + (size_t)(nt * 16 + r) * K + kt * 32 + q * 8;
    bf16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = f32_to_bf16(src[j]);
    *(uint4 *)(packed + g * 8) = *(const uint4 *)o;
}

void launch_pack_weight_bf16(const float *w_f32, bf16_t *packed, int N, int K, hipStream_t st) {
    size_t total = (size_t)(N >> 4) * (K >> 5) * 64;
    hipLaunchKernelGGL(k_pack_weight_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_f32, packed, N, K);
}

__global__ void k_f32_to_bf16(const float *in, bf16_t *out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f32_to_bf16(in[i]);
}
void launch_f32_to_bf16(const float *in, bf16_t *out, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
}

__global__ void k_fill_f32(float *p, float v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
void launch_fill_f32(float *p, float v, int64_t n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, v, n);
}

__global__ void k_relu(float *x, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = fmaxf(x[i], 0.f);
}
void launch_relu(float *x, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_relu, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n);
}

}  // namespace nasr

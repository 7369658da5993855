// kernels_fused.hip -- the batch-1 / small-M (M <= 16 rows) form of the conformer layer.
//
// At small M every linear layer is a weight-streaming problem (HBM-bound) and the step is
// dominated by kernel count, not bytes: the unfused layer is 14 launches.  Here the
// element-wise / reduction stages ride in the PROLOGUE of the GEMM that consumes them, so a
// layer is 8 launches:
//   PRO_LN     residual add of the previous GEMM's split-K partials + LayerNorm (+ the previous
//              layer's norm_out) -> bf16 A panel in LDS               (k_post folded in)
//   PRO_ATTN   cached rel-pos attention of ONE head (blockIdx.y) -> that head's 128 ctx columns;
//              the out-projection is split-K over the 8 heads         (k_attention folded in)
//   PRO_DWCONV cached depthwise conv + LayerNorm + SiLU -> A panel     (k_dwconv folded in)
//   PRO_PLAIN  activations straight from global (FFN second linear)
// Every workgroup recomputes the tiny prologue redundantly (M x 1024 elements, L2 resident);
// block (0,0) alone writes state that must persist (updated residual stream, conv cache).
// The GEMM body / weight layout / epilogues are those of k_gemm_skinny (kernels_gemm.hip).
#include "nasr_internal.h"
#include "nasr_epilogue.h"
#include "nasr_wave.h"

namespace nasr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ float wsum_f(float v) { return wave_sum(v); }
__device__ __forceinline__ float wmax_f(float v) { return wave_max(v); }

// LDS A panel: [16 rows][KP] bf16, 16-byte chunks XOR-swizzled with the row so that the MFMA
// B-fragment read (16 rows x same chunk) is bank-conflict free.
__device__ __forceinline__ int panel_byte(int row, int k, int KP) {   // k multiple of 4 -> 8-byte aligned
    const int chunk = k >> 3;
    return row * KP * 2 + (((chunk ^ (row & 15)) << 4) | ((k & 7) << 1));
}

__device__ __forceinline__ void store4_panel(char *panel, int row, int k, int KP, float a, float b, float c, float d) {
    uint2 r;
    r.x = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
    r.y = (uint32_t)f32_to_bf16(c) | ((uint32_t)f32_to_bf16(d) << 16);
    *(uint2 *)(panel + panel_byte(row, k, KP)) = r;
}

// LayerNorm of a 1024-row held as 16 values per lane (4 float4 at e = lane*4 + 256*i)
__device__ __forceinline__ void wave_ln(float4 v[4], const float *w, const float *b, int lane) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wsum_f(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float inv = 1.0f / sqrtf(wsum_f(q) * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int e = lane * 4 + 256 * i;
        const float4 ww = *(const float4 *)(w + e), bb = *(const float4 *)(b + e);
        v[i] = make_float4(v[i].x * inv * ww.x + bb.x, v[i].y * inv * ww.y + bb.y, v[i].z * inv * ww.z + bb.z, v[i].w * inv * ww.w + bb.w);
    }
}

// sum over the 256-thread block; sh = 4 floats of LDS scratch that the previous bsum did not use (two halves
// alternate: one barrier per sum, see block_sum in kernels_layer.hip)
__device__ __forceinline__ float bsum(float v, float *sh) {
    v = wsum_f(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// LayerNorm of one 1024-row spread over the block (4 channels per thread), weights already loaded; sh = 8 floats
__device__ __forceinline__ float4 block_ln(float4 v, float4 ww, float4 bb, float *sh) {
    const float mean = bsum((v.x + v.y) + (v.z + v.w), sh) * (1.0f / D);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    const float var = bsum((dx * dx + dy * dy) + (dz * dz + dw * dw), sh + 4) * (1.0f / D);
    const float inv = 1.0f / sqrtf(var + 1e-5f);
    return make_float4(dx * inv * ww.x + bb.x, dy * inv * ww.y + bb.y, dz * inv * ww.z + bb.z, dw * inv * ww.w + bb.w);
}

template <bool BF16>
__device__ __forceinline__ float dot32(const float *qs, const void *row) {   // 32 elements
    float s = 0.0f;
    if (BF16) {
        const uint4 *r = (const uint4 *)row;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint4 u = r[c];
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                s += qs[c * 8 + 2 * e] * __uint_as_float(w[e] << 16);
                s += qs[c * 8 + 2 * e + 1] * __uint_as_float(w[e] & 0xffff0000u);
            }
        }
    }
    return s;
}

#ifdef NASR_STAMPS
// diagnostic build (never shipped): wave 0 of the first and of the last workgroup stamp the 100 MHz real-time counter
#define STAMP(i)                                                                                                         \
    do {                                                                                                                 \
        if (p.stamps && threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))          \
            p.stamps[(blockIdx.x == 0 ? 0 : 16) + (i)] = __builtin_amdgcn_s_memrealtime();                                \
    } while (0)
#else
#define STAMP(i)
#endif

#define PIN_SGPR(x) asm volatile("" ::"s"(x))

// MMAX = compile-time bound of M: 1 (one row: the chunk-by-chunk single stream), 2, or 16 (any M the small-M path takes).
// The M <= 2 forms keep every prologue input in registers at once; giving them their own instantiation keeps the
// row-staging form's 144 registers of partials out of their allocation: the one-row kernels fit 128 VGPRs = 4 waves per
// SIMD, so the kernels of up to four launch chains (pipelined steps + decode) are resident on a CU together -- at 228
// VGPRs two were the limit and a third chain queued behind them.
#ifndef GRP_OCC
#define GRP_OCC 4
#endif
// The kernel body is a device function with two entry points: k_fused_skinny (one problem per launch) and k_fused_skinny_grp (up
// to FUSED_GROUP problems per launch, blockIdx.z = problem: the same layer type of several steps in flight, each with its own
// weights, activations and state -- round 3, tests/micro/dual_probe.hip: two chains of 4-problem launches stream 4.1 TB/s where
// four chains of 1-problem launches stream 2.45).  Same code per problem: results are bit-identical whichever entry runs it.
template <int PRO, int MMAX>
__device__ __forceinline__ void fused_skinny_body(const FusedParams &p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemmParams &g = p.g;
    const int nt = blockIdx.x, split = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, r = lane & 15;
    const int KT = g.K >> 5;
    const int t0 = (int)((long)KT * split / g.splits), t1 = (int)((long)KT * (split + 1) / g.splits);
    const int KP = (t1 - t0) * 32;                 // K range of this workgroup
    float *red = (float *)smem;                    // [4][64][4] floats = 4 KiB
    char *panel = smem + 4096;                     // [16][KP] bf16
    const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
    const int M = MMAX == 1 ? 1 : g.M;
    constexpr bool SMALL = MMAX <= 2;              // M <= 2: block-per-row prologues
    constexpr int MR = MMAX == 1 ? 1 : 2;
    STAMP(0);

    // The weight stream does not depend on the prologue: this wave's tiles (<= 8 KiB) are requested at the top of
    // the kernel so that their 2.2-2.4 us run under the prologue -- but AFTER the prologue's own first loads: the
    // memory pipeline serves a wave's requests in order, and the prologue inputs are the critical path.
    constexpr int U = (PRO == PRO_DWCONV && MMAX == 1) ? 2 : 8;   // k-tiles per wave (pw2 at split-K 4: 2; the launcher checks)
    const int nts = t1 - t0;
    const int w0 = t0 + nts * wave / 4, w1 = t0 + nts * (wave + 1) / 4;   // host guarantees w1 - w0 <= U
    const u32x4 *wp = (const u32x4 *)g.W + (size_t)nt * KT * 64 + lane;
    u32x4 wv[U];
    auto issue_weights = [&]() {
        if (PRO == PRO_ATTN) {
            // blockIdx.x = group of 128 output columns, blockIdx.y = head: wave w owns column tiles 2w, 2w+1 of the group
            // over the head's four k-tiles (K = 128, no cross-wave reduction).  64 workgroups redo the head's attention
            // 8 times, not 64: the 512-workgroup form spent ~3 us per launch with every workgroup pulling the same K/V/P rows.
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int ntile = blockIdx.x * 8 + wave * 2 + (u >> 2), kt = t0 + (u & 3);
                wv[u] = __builtin_nontemporal_load((const u32x4 *)g.W + ((size_t)ntile * KT + kt) * 64 + lane);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int kt = w0 + u < w1 ? w0 + u : w1 - 1;
                wv[u] = __builtin_nontemporal_load(wp + (size_t)kt * 64);
            }
        }
        // hipcc sinks the scalar loads of kernel-argument fields to their first use: the epilogue's output pointers
        // and strides were fetched after the last barrier, ~0.3 us of exposed latency per kernel (seen in the ISA and
        // in the stamps).  Naming them here makes them resident before the vector loads above return.
        PIN_SGPR(g.epi); PIN_SGPR(g.out_f32); PIN_SGPR(g.ldo); PIN_SGPR(g.out_act); PIN_SGPR(g.ldo_act); PIN_SGPR(g.N);
        PIN_SGPR(g.q_out); PIN_SGPR(g.kv_pool); PIN_SGPR(g.kv_slot_stride); PIN_SGPR(g.rows); PIN_SGPR(g.T);
    };
    if (!((PRO == PRO_LN && SMALL) || PRO == PRO_ATTN || PRO == PRO_PLAIN || (PRO == PRO_DWCONV && SMALL))) issue_weights();
    STAMP(1);

    if (PRO == PRO_LN && SMALL) {
        // block-per-row: every global load of the prologue is issued up front (one memory round trip)
        const int c4 = threadIdx.x * 4;
        const float4 lw = *(const float4 *)(p.ln_w + c4), lb = *(const float4 *)(p.ln_b + c4);
        float4 ow = make_float4(0.f, 0.f, 0.f, 0.f), ob = ow;
        if (p.lno_w) { ow = *(const float4 *)(p.lno_w + c4); ob = *(const float4 *)(p.lno_b + c4); }
        float4 xv[MR], t[MR][8];
#pragma unroll
        for (int m = 0; m < MR; m++) {
            if (m < M) {
                xv[m] = *(const float4 *)(p.x_in + (size_t)m * D + c4);
#pragma unroll
                for (int sI = 0; sI < 8; sI++)
                    t[m][sI] = sI < p.part_splits ? *(const float4 *)(p.part + ((size_t)sI * M + m) * D + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        issue_weights();
#pragma unroll
        for (int m = 0; m < MR; m++) {
            if (m < M) {
                float4 v = xv[m];
                if (p.part_splits > 0) {
                    float4 o = t[m][0];
#pragma unroll
                    for (int sI = 1; sI < 8; sI++)
                        if (sI < p.part_splits) { o.x += t[m][sI].x; o.y += t[m][sI].y; o.z += t[m][sI].z; o.w += t[m][sI].w; }
                    v.x += p.scale * o.x; v.y += p.scale * o.y; v.z += p.scale * o.z; v.w += p.scale * o.w;
                }
                STAMP(2);
                if (p.lno_w) v = block_ln(v, ow, ob, red);                 // previous layer's norm_out (:687)
                if (writer && p.x_out) *(float4 *)(p.x_out + (size_t)m * D + c4) = v;
                v = block_ln(v, lw, lb, red);
                STAMP(3);
                store4_panel(panel, m, c4, KP, v.x, v.y, v.z, v.w);
            }
        }
        __syncthreads();
    } else if (PRO == PRO_LN) {
        // M > 2, two phases.  A: every thread forms x + scale * sum(partials) for a strip of all rows (all
        // loads independent) and parks it in LDS; B: one wave per row does the LayerNorm(s) out of LDS.
        float *xs = (float *)(panel + 16 * KP * 2);            // [M][1024] f32
        const int c4 = threadIdx.x * 4;
        for (int m0 = 0; m0 < M; m0 += 4) {                    // 4 rows per round: all their loads in flight together
            float4 xv[4], t[4][8];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int m = m0 + u < M ? m0 + u : M - 1;
                xv[u] = *(const float4 *)(p.x_in + (size_t)m * D + c4);
#pragma unroll
                for (int sI = 0; sI < 8; sI++)
                    t[u][sI] = sI < p.part_splits ? *(const float4 *)(p.part + ((size_t)sI * M + m) * D + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int m = m0 + u;
                if (m < M) {
                    float4 v = xv[u];
                    if (p.part_splits > 0) {
                        float4 o = t[u][0];
#pragma unroll
                        for (int sI = 1; sI < 8; sI++)
                            if (sI < p.part_splits) { o.x += t[u][sI].x; o.y += t[u][sI].y; o.z += t[u][sI].z; o.w += t[u][sI].w; }
                        v.x += p.scale * o.x; v.y += p.scale * o.y; v.z += p.scale * o.z; v.w += p.scale * o.w;
                    }
                    *(float4 *)(xs + (size_t)m * D + c4) = v;
                    if (writer && p.x_out && !p.lno_w) *(float4 *)(p.x_out + (size_t)m * D + c4) = v;
                }
            }
        }
        __syncthreads();
        for (int m = wave; m < M; m += 4) {
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = *(const float4 *)(xs + (size_t)m * D + lane * 4 + 256 * i);
            if (p.lno_w) {
                wave_ln(v, p.lno_w, p.lno_b, lane);                // previous layer's norm_out (:687)
                if (writer && p.x_out) {
#pragma unroll
                    for (int i = 0; i < 4; i++) *(float4 *)(p.x_out + (size_t)m * D + lane * 4 + 256 * i) = v[i];
                }
            }
            wave_ln(v, p.ln_w, p.ln_b, lane);
#pragma unroll
            for (int i = 0; i < 4; i++) store4_panel(panel, m, lane * 4 + 256 * i, KP, v[i].x, v[i].y, v[i].z, v[i].w);
        }
        __syncthreads();
    } else if (PRO == PRO_ATTN) {
        // one head (= split) of the cached rel-pos attention for all M <= 2 rows (src/nemo-stream.cpp:463-573).
        // For the first 128 (row, key) pairs and for row 0 -- all there is at M = 1 -- everything the arithmetic needs
        // is requested up front, in two memory round trips instead of five: (a) row descriptors, queries,
        // relative-position rows; (b) once the descriptors are in, the K and V ring rows.  A second row (M = 2) is
        // handled by a second pass that loads as it goes (keeps the kernel under the register limit).
        const AttnParams &a = p.at;
        const int h = split, T = a.T, KV = LCTX + T, TS = a.TS > 0 ? a.TS : a.T;
        float *qu = (float *)(panel + 16 * KP * 2);                  // [2][128]
        float *qv = qu + 16 * DH;                                    // [2][128]
        float *sc = qv + 16 * DH;                                    // [2][KVC]
        const int sub = threadIdx.x & 1;                             // 2 lanes per (row, key) pair, 64 dims each
        const int npair = M * KV;                                    // <= 144: at most two passes of 128 pairs
        const float scale = 0.08838834764831845f;
        float qq = 0.f, bu = 0.f, bv = 0.f;
        if ((int)threadIdx.x < M * DH) {
            const int m = threadIdx.x >> 7, d = threadIdx.x & 127;
            qq = a.q[(size_t)m * D + h * DH + d]; bu = a.bias_u[h * DH + d]; bv = a.bias_v[h * DH + d];
        }
        RowDesc rds[2];
        rds[0] = a.rows[0];
        rds[1] = a.rows[(M - 1) / TS];                               // stream of row 1 (= stream 0 unless B = 2)
        auto pair_rows = [&](int pr, int &m, int &j, const bf16_t *&prow) {
            const bool ok = pr < npair;
            m = ok ? pr / KV : 0;
            j = ok ? pr - m * KV : 0;
            const int il = m - (m / TS) * TS, gch = il / T, i = il - gch * T;
            prow = (const bf16_t *)a.posproj + (size_t)(j + T - 1 - i) * D + h * DH + sub * 64;
            return ok;
        };
        auto key_row = [&](int m, int j) {
            const int b = m / TS, gch = (m - b * TS) / T;
            const RowDesc rd = b == 0 ? rds[0] : rds[1];
            int ring = rd.kv_head + gch * T + j;
            if (ring >= KVC) ring -= KVC;
            if (ring >= KVC) ring -= KVC;
            return (const bf16_t *)a.kv_pool + (size_t)rd.slot * a.kv_slot_stride + (size_t)ring * D + h * DH + sub * 64;
        };
        // q + bias live in LDS as bf16 (rounded as in k_attention_mfma) so that a score is 64 v_dot2_f32_bf16 per operand pair
        // instead of 128 unpack + 128 FMA instructions (the scalar form spent 1.5 us of the kernel's 5.4 here, by the stamps)
        auto score = [&](const uint4 *kk, const uint4 *pp, int m, int j, bool ok) {
            const uint4 *qa = (const uint4 *)((const bf16_t *)qu + m * DH + sub * 64), *qb = (const uint4 *)((const bf16_t *)qv + m * DH + sub * 64);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const uint4 a4 = qa[c], b4 = qb[c];
                const uint32_t kw[4] = {kk[c].x, kk[c].y, kk[c].z, kk[c].w}, pw[4] = {pp[c].x, pp[c].y, pp[c].z, pp[c].w};
                const uint32_t aw[4] = {a4.x, a4.y, a4.z, a4.w}, bw[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int e2 = 0; e2 < 4; e2++) {
                    s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, aw[e2]), __builtin_bit_cast(bf16x2, kw[e2]), s1, false);
                    s2 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, bw[e2]), __builtin_bit_cast(bf16x2, pw[e2]), s2, false);
                }
            }
            float v = s1 + s2;
            v += dpp_mov<0xB1>(v);                                   // the other half of the pair (lane ^ 1)
            if (ok && sub == 0) {
                const int b = m / TS, gch = (m - b * TS) / T;
                const RowDesc rd = b == 0 ? rds[0] : rds[1];
                v *= scale;
                const int valid = rd.valid_len + gch * T < LCTX ? rd.valid_len + gch * T : LCTX;
                if (j < LCTX - valid) v += -1e9f;
                sc[m * KVC + j] = v;
            }
        };
        auto value_rows = [&](int m, uint2 *vv) {
            const int b = m / TS, gch = (m - b * TS) / T;
            const RowDesc rd = b == 0 ? rds[0] : rds[1];
            const int head0 = rd.kv_head + gch * T;
            const bf16_t *vbase = (const bf16_t *)a.kv_pool + (size_t)rd.slot * a.kv_slot_stride + (size_t)KVC * D + h * DH + (threadIdx.x & 31) * 4;
#pragma unroll
            for (int u = 0; u < 11; u++) {                           // KV <= 84 -> at most 11 keys per group
                const int jj = (threadIdx.x >> 5) + 8 * u;
                int ring = head0 + (jj < KV ? jj : 0);
                if (ring >= KVC) ring -= KVC;
                if (ring >= KVC) ring -= KVC;
                vv[u] = *(const uint2 *)(vbase + (size_t)ring * D);
            }
        };
        if (MMAX == 1 || (T == 1 && TS == 1)) {
            // One new row per stream (T = 1 and one chunk per stream, KV = 71; two streams: one after the other).  The scattered form below (a thread pair per key, 128 contiguous bytes per thread as eight
            // 16-byte loads) keeps the CU's address unit busy for 2.5 us before the last of its 35 load instructions per wave has
            // even been issued (stamps: "all loads issued" at 2.57 us; each instruction touches 64 different 64-byte segments).
            // Here a load instruction of a wave covers 4 whole head rows (256 contiguous bytes each): lane = (sub-row sr, chunk
            // c of 8 dims), key j = 16 t + 4 wave + sr in pass t = 0..4.  A lane forms the partial dot of its 8 dims (v_dot2 on
            // bf16 q + bias, as before), the 16 lanes of a row add up on the DPP path, and P.V runs in the same layout (each lane
            // accumulates its 8 dims over its 5 keys).  No q staging through LDS, no softmax barrier: every wave computes the
            // statistics of the 71 scores itself.
            const int sr = lane >> 4, c = lane & 15;
            for (int m = 0; m < M; m++) {
            const RowDesc rd = m == 0 ? rds[0] : rds[1];
            float *scm = sc + m * KVC;
            const float4 *qp = (const float4 *)(a.q + (size_t)m * D + h * DH + c * 8);
            const float4 *up = (const float4 *)(a.bias_u + h * DH + c * 8), *vp = (const float4 *)(a.bias_v + h * DH + c * 8);
            const float4 q0 = qp[0], q1 = qp[1], u0 = up[0], u1 = up[1], b0 = vp[0], b1 = vp[1];
            uint4 kk[5], pp[5], vr[5];
            int jk[5];
#pragma unroll
            for (int t = 0; t < 5; t++) {
                jk[t] = t * 16 + wave * 4 + sr;
                const int jc = jk[t] < KV ? jk[t] : 0;
                pp[t] = *(const uint4 *)((const bf16_t *)a.posproj + (size_t)(jc + T - 1) * D + h * DH + c * 8);
            }
            if (m == 0) issue_weights();
#pragma unroll
            for (int t = 0; t < 5; t++) {
                const int jc = jk[t] < KV ? jk[t] : 0;
                int ring = rd.kv_head + jc;
                if (ring >= KVC) ring -= KVC;
                if (ring >= KVC) ring -= KVC;
                const bf16_t *krow = (const bf16_t *)a.kv_pool + (size_t)rd.slot * a.kv_slot_stride + (size_t)ring * D + h * DH + c * 8;
                kk[t] = *(const uint4 *)krow;
                vr[t] = *(const uint4 *)(krow + (size_t)KVC * D);
            }
            STAMP(2);
            auto pk2 = [](float x, float y) { return (uint32_t)f32_to_bf16(x) | ((uint32_t)f32_to_bf16(y) << 16); };
            const uint32_t qu8[4] = {pk2(q0.x + u0.x, q0.y + u0.y), pk2(q0.z + u0.z, q0.w + u0.w), pk2(q1.x + u1.x, q1.y + u1.y), pk2(q1.z + u1.z, q1.w + u1.w)};
            const uint32_t qv8[4] = {pk2(q0.x + b0.x, q0.y + b0.y), pk2(q0.z + b0.z, q0.w + b0.w), pk2(q1.x + b1.x, q1.y + b1.y), pk2(q1.z + b1.z, q1.w + b1.w)};
            const int valid = rd.valid_len < LCTX ? rd.valid_len : LCTX;
#pragma unroll
            for (int t = 0; t < 5; t++) {
                const uint32_t kw[4] = {kk[t].x, kk[t].y, kk[t].z, kk[t].w}, pw[4] = {pp[t].x, pp[t].y, pp[t].z, pp[t].w};
                float s1 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, qu8[e]), __builtin_bit_cast(bf16x2, kw[e]), s1, false);
                    s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, qv8[e]), __builtin_bit_cast(bf16x2, pw[e]), s1, false);
                }
                s1 += dpp_mov<0xB1>(s1);                                // the 16 lanes of the row
                s1 += dpp_mov<0x4E>(s1);
                s1 += dpp_mov<0x141>(s1);
                s1 += dpp_mov<0x140>(s1);
                if (c == 0 && jk[t] < KV) {
                    float v = s1 * scale;
                    if (jk[t] < LCTX - valid) v += -1e9f;
                    scm[jk[t]] = v;
                }
            }
            STAMP(3);
            __syncthreads();
            const float v0 = lane < KV ? scm[lane] : -INFINITY, v1 = lane + 64 < KV ? scm[lane + 64] : -INFINITY;
            const float mx = wmax_f(fmaxf(v0, v1));
            const float inv = 1.0f / wsum_f((lane < KV ? __expf(v0 - mx) : 0.0f) + (lane + 64 < KV ? __expf(v1 - mx) : 0.0f));
            float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 5; t++) {
                const float w = jk[t] < KV ? __expf(scm[jk[t]] - mx) * inv : 0.0f;
                const uint32_t vw[4] = {vr[t].x, vr[t].y, vr[t].z, vr[t].w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    acc8[2 * e] += w * __uint_as_float(vw[e] << 16);
                    acc8[2 * e + 1] += w * __uint_as_float(vw[e] & 0xffff0000u);
                }
            }
            STAMP(6);
            float *pv = qu;                                             // [16 row groups][128]
            *(float4 *)(pv + (wave * 4 + sr) * DH + c * 8) = make_float4(acc8[0], acc8[1], acc8[2], acc8[3]);
            *(float4 *)(pv + (wave * 4 + sr) * DH + c * 8 + 4) = make_float4(acc8[4], acc8[5], acc8[6], acc8[7]);
            __syncthreads();
            if ((int)threadIdx.x < DH) {
                const int d = threadIdx.x;
                float o = 0.f;
#pragma unroll
                for (int gI = 0; gI < 16; gI++) o += pv[gI * DH + d];
                *(bf16_t *)(panel + m * KP * 2 + ((((d >> 3) ^ (m & 15)) << 4) | ((d & 7) << 1))) = f32_to_bf16(o);
            }
            if (m + 1 < M) __syncthreads();                              // the next row reuses the P.V scratch
            }
        } else {
        // (a) + (b) for pass 0 / row 0
        int m0, j0;
        const bf16_t *prow0;
        const bool ok0 = pair_rows(threadIdx.x >> 1, m0, j0, prow0);
        uint4 kk[8], pp[8];
#pragma unroll
        for (int c = 0; c < 8; c++) pp[c] = ((const uint4 *)prow0)[c];
        issue_weights();
        const bf16_t *krow0 = key_row(m0, j0);
#pragma unroll
        for (int c = 0; c < 8; c++) kk[c] = ((const uint4 *)krow0)[c];
        uint2 vv[11];
        value_rows(0, vv);
        if ((int)threadIdx.x < M * DH) { ((bf16_t *)qu)[threadIdx.x] = f32_to_bf16(qq + bu); ((bf16_t *)qv)[threadIdx.x] = f32_to_bf16(qq + bv); }
        STAMP(2);
        __syncthreads();
        score(kk, pp, m0, j0, ok0);
        STAMP(3);
        if (MMAX > 1 && npair > 128) {                               // M = 2: the remaining pairs
            int m1, j1;
            const bf16_t *prow1;
            const bool ok1 = pair_rows(128 + (threadIdx.x >> 1), m1, j1, prow1);
            const bf16_t *krow1 = key_row(m1, j1);
#pragma unroll
            for (int c = 0; c < 8; c++) { pp[c] = ((const uint4 *)prow1)[c]; kk[c] = ((const uint4 *)krow1)[c]; }
            score(kk, pp, m1, j1, ok1);
        }
        __syncthreads();
        for (int m = wave; m < M; m += 4) {
            const float v0 = lane < KV ? sc[m * KVC + lane] : -INFINITY;
            const float v1 = lane + 64 < KV ? sc[m * KVC + lane + 64] : -INFINITY;
            const float mx = wmax_f(fmaxf(v0, v1));
            const float e0 = lane < KV ? __expf(v0 - mx) : 0.0f;
            const float e1 = lane + 64 < KV ? __expf(v1 - mx) : 0.0f;
            const float inv = 1.0f / wsum_f(e0 + e1);
            if (lane < KV) sc[m * KVC + lane] = e0 * inv;
            if (lane + 64 < KV) sc[m * KVC + lane + 64] = e1 * inv;
        }
        __syncthreads();
        STAMP(6);
        // P.V: thread = (key group kgp of 8, 4 consecutive d); partial sums of the 8 groups reduced through LDS
        {
            float *pv = qu;                                          // reuse: [2 rows][8 groups][128]
            const int dq = threadIdx.x & 31, kgp = threadIdx.x >> 5;
            for (int m = 0; m < M; m++) {
                if (m > 0) value_rows(m, vv);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 11; u++) {
                    const int jj = kgp + 8 * u;
                    const float w = jj < KV ? sc[m * KVC + jj] : 0.0f;
                    acc.x += w * __uint_as_float(vv[u].x << 16); acc.y += w * __uint_as_float(vv[u].x & 0xffff0000u);
                    acc.z += w * __uint_as_float(vv[u].y << 16); acc.w += w * __uint_as_float(vv[u].y & 0xffff0000u);
                }
                *(float4 *)(pv + (m * 8 + kgp) * DH + dq * 4) = acc;
            }
            __syncthreads();
            if ((int)threadIdx.x < M * DH) {
                const int m = threadIdx.x >> 7, d = threadIdx.x & 127;
                float o = 0.f;
#pragma unroll
                for (int gI = 0; gI < 8; gI++) o += pv[(m * 8 + gI) * DH + d];
                *(bf16_t *)(panel + m * KP * 2 + ((((d >> 3) ^ (m & 15)) << 4) | ((d & 7) << 1))) = f32_to_bf16(o);
            }
        }
        }
        __syncthreads();
    } else if (PRO == PRO_DWCONV) {
        const ConvParams &c = p.cv;
        const int T = c.T, ks1 = c.ks - 1;
        const int k0 = t0 * 32;
        if (SMALL) {
            // block-per-row, 4 channels per thread; taps unrolled so all loads are in flight together
            const int c4 = threadIdx.x * 4;
            const float4 lw = *(const float4 *)(c.ln_w + c4), lb = *(const float4 *)(c.ln_b + c4);
            const RowDesc rd_first = c.rows[0];             // the cache rows below depend on it: request it first,
            float4 w[9];                                    // then what does not (taps weights, the GEMM weight stream)
            if (c.ks == 9) {
#pragma unroll
                for (int k = 0; k < 9; k++) w[k] = *(const float4 *)(c.dw + (size_t)k * D + c4);
            }
            issue_weights();
            for (int m = 0; m < M; m++) {
                const int b = m / T, i = m - b * T;
                const RowDesc rd = b == 0 ? rd_first : c.rows[b];
                const float *cc_in = c.cc_pool + (size_t)rd.slot * c.cc_slot_stride + (size_t)rd.cc_par * ks1 * D;
                float *cc_out = c.cc_pool + (size_t)rd.slot * c.cc_slot_stride + (size_t)(rd.cc_par ^ 1) * ks1 * D;
                const float *gl = c.glu + (size_t)b * T * D;
                float4 acc;
                bool cache_written = false;
                if (c.ks == 9) {
                    float4 z[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        const int rr = i + k;
                        z[k] = rr < 8 ? *(const float4 *)(cc_in + (size_t)rr * D + c4) : *(const float4 *)(gl + (size_t)(rr - 8) * D + c4);
                    }
                    acc = make_float4(z[0].x * w[0].x, z[0].y * w[0].y, z[0].z * w[0].z, z[0].w * w[0].w);
#pragma unroll
                    for (int k = 1; k < 9; k++) { acc.x += z[k].x * w[k].x; acc.y += z[k].y * w[k].y; acc.z += z[k].z * w[k].z; acc.w += z[k].w * w[k].w; }
                    if (writer && i == 0 && T == 1) {      // the stream's new cache = window rows 1..8: already in registers
#pragma unroll                                             // (the load-then-store loop below made block (0,0) 1.3 us the kernel's last)
                        for (int r2 = 0; r2 < 8; r2++) *(float4 *)(cc_out + (size_t)r2 * D + c4) = z[r2 + 1];
                        cache_written = true;
                    }
                } else {
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int k = 0; k < c.ks; k++) {
                        const int rr = i + k;
                        const float4 z = rr < ks1 ? *(const float4 *)(cc_in + (size_t)rr * D + c4) : *(const float4 *)(gl + (size_t)(rr - ks1) * D + c4);
                        const float4 w = *(const float4 *)(c.dw + (size_t)k * D + c4);
                        if (k == 0) acc = make_float4(z.x * w.x, z.y * w.y, z.z * w.z, z.w * w.w);
                        else { acc.x += z.x * w.x; acc.y += z.y * w.y; acc.z += z.z * w.z; acc.w += z.w * w.w; }
                    }
                }
                float4 n = block_ln(acc, lw, lb, red);
                if (c4 >= k0 && c4 < k0 + KP) {
                    n.x = n.x / (1.0f + __expf(-n.x)); n.y = n.y / (1.0f + __expf(-n.y));
                    n.z = n.z / (1.0f + __expf(-n.z)); n.w = n.w / (1.0f + __expf(-n.w));
                    store4_panel(panel, m, c4 - k0, KP, n.x, n.y, n.z, n.w);
                }
                if (writer && i == 0 && !cache_written) {
                    for (int r2 = 0; r2 < ks1; r2++) {
                        const int rr = T + r2;
                        *(float4 *)(cc_out + (size_t)r2 * D + c4) =
                            rr < ks1 ? *(const float4 *)(cc_in + (size_t)rr * D + c4) : *(const float4 *)(gl + (size_t)(rr - ks1) * D + c4);
                    }
                }
            }
        } else {
            // M > 2, two phases: A = taps for a 4-channel strip of every row (independent loads) -> LDS,
            // B = one wave per row: LayerNorm + SiLU out of LDS.
            float *cs = (float *)(panel + 16 * KP * 2);        // [M][1024] f32
            const int c4 = threadIdx.x * 4;
            for (int m = 0; m < M; m++) {
                const int b = m / T, i = m - b * T;
                const RowDesc rd = c.rows[b];
                const float *cc_in = c.cc_pool + (size_t)rd.slot * c.cc_slot_stride + (size_t)rd.cc_par * ks1 * D;
                float *cc_out = c.cc_pool + (size_t)rd.slot * c.cc_slot_stride + (size_t)(rd.cc_par ^ 1) * ks1 * D;
                const float *gl = c.glu + (size_t)b * T * D;
                float4 acc;
                if (c.ks == 9) {
                    float4 z[9], w[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        const int rr = i + k;
                        z[k] = rr < 8 ? *(const float4 *)(cc_in + (size_t)rr * D + c4) : *(const float4 *)(gl + (size_t)(rr - 8) * D + c4);
                        w[k] = *(const float4 *)(c.dw + (size_t)k * D + c4);
                    }
                    acc = make_float4(z[0].x * w[0].x, z[0].y * w[0].y, z[0].z * w[0].z, z[0].w * w[0].w);
#pragma unroll
                    for (int k = 1; k < 9; k++) { acc.x += z[k].x * w[k].x; acc.y += z[k].y * w[k].y; acc.z += z[k].z * w[k].z; acc.w += z[k].w * w[k].w; }
                } else {
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int k = 0; k < c.ks; k++) {
                        const int rr = i + k;
                        const float4 z = rr < ks1 ? *(const float4 *)(cc_in + (size_t)rr * D + c4) : *(const float4 *)(gl + (size_t)(rr - ks1) * D + c4);
                        const float4 w = *(const float4 *)(c.dw + (size_t)k * D + c4);
                        if (k == 0) acc = make_float4(z.x * w.x, z.y * w.y, z.z * w.z, z.w * w.w);
                        else { acc.x += z.x * w.x; acc.y += z.y * w.y; acc.z += z.z * w.z; acc.w += z.w * w.w; }
                    }
                }
                *(float4 *)(cs + (size_t)m * D + c4) = acc;
                if (writer && i == 0) {
                    for (int r2 = 0; r2 < ks1; r2++) {
                        const int rr = T + r2;
                        *(float4 *)(cc_out + (size_t)r2 * D + c4) =
                            rr < ks1 ? *(const float4 *)(cc_in + (size_t)rr * D + c4) : *(const float4 *)(gl + (size_t)(rr - ks1) * D + c4);
                    }
                }
            }
            __syncthreads();
            for (int m = wave; m < M; m += 4) {
                float4 v[4];
#pragma unroll
                for (int ii = 0; ii < 4; ii++) v[ii] = *(const float4 *)(cs + (size_t)m * D + lane * 4 + 256 * ii);
                wave_ln(v, c.ln_w, c.ln_b, lane);
#pragma unroll
                for (int ii = 0; ii < 4; ii++) {
                    const int e = lane * 4 + 256 * ii;
                    if (e >= k0 && e < k0 + KP) {
                        float4 n = v[ii];
                        n.x = n.x / (1.0f + __expf(-n.x)); n.y = n.y / (1.0f + __expf(-n.y));
                        n.z = n.z / (1.0f + __expf(-n.z)); n.w = n.w / (1.0f + __expf(-n.w));
                        store4_panel(panel, m, e - k0, KP, n.x, n.y, n.z, n.w);
                    }
                }
            }
        }
        __syncthreads();
    }

    STAMP(4);
    // ---- GEMM over this workgroup's k-tiles (weights already in registers) -------------------------
    uint4 avp[U];
    if (PRO == PRO_PLAIN) {       // activations (written by the previous kernel) first, then the weight stream
        const char *arow = (const char *)g.A + ((size_t)(r < M ? r : M - 1) * g.lda) * 2 + q * 16;
#pragma unroll
        for (int u = 0; u < U; u++) avp[u] = *(const uint4 *)(arow + (size_t)(w0 + u < w1 ? w0 + u : w1 - 1) * 64);
        issue_weights();
    }
    if (PRO == PRO_ATTN) {
        f32x4 a2[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint4 av = *(const uint4 *)(panel + r * KP * 2 + (((((u & 3) << 2) | q) ^ r) << 4));
            a2[u >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[u]), __builtin_bit_cast(bf16x8, av), a2[u >> 2], 0, 0, 0);
        }
        STAMP(5);
#pragma unroll
        for (int j = 0; j < 2; j++)
            epi_quad<true>(g, split, r, (blockIdx.x * 8 + wave * 2 + j) * 16 + q * 4, a2[j][0], a2[j][1], a2[j][2], a2[j][3]);
        STAMP(7);
        return;
    }
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int kt = w0 + u;
        if (kt < w1) {
            uint4 av;
            if (PRO == PRO_PLAIN) av = avp[u];
            else av = *(const uint4 *)(panel + r * KP * 2 + (((((kt - t0) << 2) | q) ^ r) << 4));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[u]), __builtin_bit_cast(bf16x8, av), acc, 0, 0, 0);
        }
    }
    STAMP(5);
#pragma unroll
    for (int j = 0; j < 4; j++) red[(wave * 64 + lane) * 4 + j] = acc[j];
    __syncthreads();
    STAMP(6);
    if (wave == 0) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            v[j] = ((red[lane * 4 + j] + red[(64 + lane) * 4 + j]) + red[(128 + lane) * 4 + j]) + red[(192 + lane) * 4 + j];
        epi_quad<true>(g, split, r, nt * 16 + q * 4, v[0], v[1], v[2], v[3]);
    }
    STAMP(7);
}

template <int PRO, int MMAX>
__global__ __launch_bounds__(256, (MMAX == 1 && PRO != PRO_ATTN) ? 4 : 1) void k_fused_skinny(FusedParams p) {
    fused_skinny_body<PRO, MMAX>(p);
}
template <int PRO, int MMAX>
__global__ __launch_bounds__(256, (MMAX == 1 && PRO != PRO_ATTN) ? GRP_OCC : 1) void k_fused_skinny_grp(FusedParamsGroup pp) {
    const FusedParams &p = pp.p[blockIdx.z];
    if (p.g.M <= 0) return;                       // a stage of the pipeline that holds no step (fill / drain)
    fused_skinny_body<PRO, MMAX>(p);
}

void init_fused_kernel_attributes() {   // LN / dwconv row staging can exceed the 64 KiB default dynamic-LDS limit at M = 16
    hipFuncSetAttribute((const void *)k_fused_skinny<PRO_LN, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 + 32768 + 16 * D * 4);
    hipFuncSetAttribute((const void *)k_fused_skinny<PRO_DWCONV, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 + 32768 + 16 * D * 4);
}

template <int MMAX>
static void launch_fused_m(const FusedParams &p, hipStream_t st, size_t lds) {
    const GemmParams &g = p.g;
    dim3 grid(g.N / 16, g.splits);
    switch (p.pro) {
    case PRO_LN: hipLaunchKernelGGL((k_fused_skinny<PRO_LN, MMAX>), grid, dim3(256), lds, st, p); break;
    case PRO_PLAIN: hipLaunchKernelGGL((k_fused_skinny<PRO_PLAIN, MMAX>), grid, dim3(256), 4096, st, p); break;
    case PRO_ATTN: hipLaunchKernelGGL((k_fused_skinny<PRO_ATTN, MMAX>), dim3(g.N / 128, g.splits), dim3(256), lds, st, p); break;   // (column group, head)
    case PRO_DWCONV: hipLaunchKernelGGL((k_fused_skinny<PRO_DWCONV, MMAX>), grid, dim3(256), lds, st, p); break;
    }
}

template <int MMAX>
static void launch_fused_grp_m(const FusedParamsGroup &pp, const FusedParams &p, int n, hipStream_t st, size_t lds) {
    const GemmParams &g = p.g;
    dim3 grid(g.N / 16, g.splits, n);
    switch (p.pro) {
    case PRO_LN: hipLaunchKernelGGL((k_fused_skinny_grp<PRO_LN, MMAX>), grid, dim3(256), lds, st, pp); break;
    case PRO_PLAIN: hipLaunchKernelGGL((k_fused_skinny_grp<PRO_PLAIN, MMAX>), grid, dim3(256), 4096, st, pp); break;
    case PRO_ATTN: hipLaunchKernelGGL((k_fused_skinny_grp<PRO_ATTN, MMAX>), dim3(g.N / 128, g.splits, n), dim3(256), lds, st, pp); break;
    case PRO_DWCONV: hipLaunchKernelGGL((k_fused_skinny_grp<PRO_DWCONV, MMAX>), grid, dim3(256), lds, st, pp); break;
    }
}

// n <= FUSED_GROUP problems of the SAME kind (pro, M, N, K, splits) in one launch; a problem with g.M == 0 is skipped
void launch_fused_skinny_group(const FusedParamsGroup &pp, int n, hipStream_t st) {
    int first = -1;
    for (int i = 0; i < n; i++)
        if (pp.p[i].g.M > 0) { first = i; break; }
    if (first < 0) return;
    const FusedParams &p = pp.p[first];
    const GemmParams &g = p.g;
    const int KP = (g.K / g.splits);
    size_t lds = 4096 + (size_t)16 * KP * 2;
    if (p.pro == PRO_ATTN) lds += (size_t)(2 * 16 * DH + 16 * KVC) * 4;
    const bool fits_u2 = p.pro != PRO_DWCONV || (g.K / 32 / g.splits + 3) / 4 <= 2;
    if (g.M == 1 && fits_u2) launch_fused_grp_m<1>(pp, p, n, st, lds);
    else launch_fused_grp_m<2>(pp, p, n, st, lds);          // grouped launches exist for M <= 2 only (the engine checks)
}

void launch_fused_skinny(const FusedParams &p, hipStream_t st) {
    const GemmParams &g = p.g;
    const int KP = (g.K / g.splits);
    size_t lds = 4096 + (size_t)16 * KP * 2;
    if (p.pro == PRO_ATTN) lds += (size_t)(2 * 16 * DH + 16 * KVC) * 4;
    if ((p.pro == PRO_LN || p.pro == PRO_DWCONV) && g.M > 2) lds += (size_t)g.M * D * 4;   // f32 row staging
    const bool fits_u2 = p.pro != PRO_DWCONV || (g.K / 32 / g.splits + 3) / 4 <= 2;   // the one-row dwconv form holds 2 k-tiles per wave
    if (g.M == 1 && fits_u2) launch_fused_m<1>(p, st, lds);
    else if (g.M <= 2) launch_fused_m<2>(p, st, lds);      // also one row whose k-slice does not fit the one-row form: the SMALL forms need no row staging
    else launch_fused_m<16>(p, st, lds);
}

}  // namespace nasr

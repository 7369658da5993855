// kernels_layer.hip -- the non-GEMM parts of one cached conformer layer
// (reference src/nemo-stream.cpp:605-690):
//   k_post      residual add of the split-K partials + LayerNorm(s)     (:580-591, :633-634, :687)
//   k_attention cached relative-position attention with the rel-shift, the validity mask,
//               softmax and P.V in one kernel, K/V read straight from the ring (:463-573)
//   k_dwconv    cached causal depthwise conv + LayerNorm + SiLU + conv-cache update (:336-412, :671-674)
#include "nasr_internal.h"
#include "nasr_wave.h"
#include "nasr_post.h"

namespace nasr {

// ---- post: one workgroup per row ------------------------------------------------------------
__global__ __launch_bounds__(256) void k_post(PostParams p) {
    __shared__ float sh[8];
    post_row(p, blockIdx.x, threadIdx.x, sh, true);          // nasr_post.h: the same body runs in the head phase of a chained GEMM launch
}
void launch_post(const PostParams &p, hipStream_t st) {
    // (also measured at 7 168 rows: a LayerNorm-only form with four rows per workgroup, all loads first, full occupancy: 16.2 against 14.3 us; one WAVE per
    // row with 16-byte stores and no barrier: 13.51 against 13.58 ms per 512-stream step, 2.43 against 2.40 at 64 streams -- nothing: profiles/r5_configs2_launch_structure.md)
    // (round 5: two rows per workgroup with both rows' loads in flight before the first reduction measured SLOWER at 7 168 rows: 16.5 against 14.0 us)
    hipLaunchKernelGGL(k_post, dim3(p.M), dim3(256), 0, st, p);
}

// ---- attention: one workgroup per (head, stream) ----------------------------------------------
template <bool BF16>
__device__ __forceinline__ float dot128(const float *qs, const void *row) {
    float s = 0.0f;
    if (BF16) {
        const uint4 *r = (const uint4 *)row;
#pragma unroll 4
        for (int c = 0; c < 16; c++) {
            const uint4 u = r[c];
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                s += qs[c * 8 + 2 * e] * __uint_as_float(w[e] << 16);
                s += qs[c * 8 + 2 * e + 1] * __uint_as_float(w[e] & 0xffff0000u);
            }
        }
    } else {
        const float4 *r = (const float4 *)row;
#pragma unroll 4
        for (int c = 0; c < 32; c++) {
            const float4 u = r[c];
            s += qs[c * 4] * u.x; s += qs[c * 4 + 1] * u.y; s += qs[c * 4 + 2] * u.z; s += qs[c * 4 + 3] * u.w;
        }
    }
    return s;
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_attention(AttnParams p) {
    // One workgroup per (head, stream, chunk).  A launch carries TS = G * T rows per stream (G chunks of T frames,
    // G = 1 unless a multi-chunk step): the T rows of chunk g attend the 70 + T keys that start g*T rows into the
    // ring, with the cache validity that chunk would have seen (valid_len + g*T).
    __shared__ float qu[TMAX][DH], qv[TMAX][DH];
    __shared__ float sc[TMAX][KVC];
    const int h = blockIdx.x, b = blockIdx.y, g = blockIdx.z, T = p.T, KV = LCTX + T, TS = p.TS > 0 ? p.TS : p.T;
    const RowDesc rd = p.rows[b];
    const int esz = BF16 ? 2 : 4;
    const char *kbase = (const char *)p.kv_pool + ((size_t)rd.slot * p.kv_slot_stride) * esz;
    const char *vbase = kbase + (size_t)KVC * D * esz;
    const size_t row0 = (size_t)b * TS + (size_t)g * T;
    int head0 = rd.kv_head + g * T;
    while (head0 >= KVC) head0 -= KVC;
    for (int e = threadIdx.x; e < T * DH; e += 256) {
        const int i = e >> 7, d = e & 127;
        const float q = p.q[(row0 + i) * D + h * DH + d];
        qu[i][d] = q + p.bias_u[h * DH + d];     // src/nemo-stream.cpp:531-535
        qv[i][d] = q + p.bias_v[h * DH + d];
    }
    __syncthreads();
    const float scale = 0.08838834764831845f;    // 1/sqrt(128), :545
    const int valid = rd.valid_len + g * T < LCTX ? rd.valid_len + g * T : LCTX;
    for (int e = threadIdx.x; e < T * KV; e += 256) {
        const int i = e / KV, j = e - i * KV;
        int ring = head0 + j;
        if (ring >= KVC) ring -= KVC;
        const char *krow = kbase + ((size_t)ring * D + h * DH) * esz;
        // rel-shift folded into indexing: slice row j + T - 1 - i  <->  rel = (70 + i) - j  (:419-461)
        const char *prow = (const char *)p.posproj + ((size_t)(j + T - 1 - i) * D + h * DH) * esz;
        const float s1 = dot128<BF16>(qu[i], krow);   // :538
        const float s2 = dot128<BF16>(qv[i], prow);   // :541-542
        float v = (s1 + s2) * scale;                   // :546-547
        if (j < LCTX - valid) v += -1e9f;              // :552-556, :1037-1043
        sc[i][j] = v;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = wave; i < T; i += 4) {               // softmax, :559
        float v0 = lane < KV ? sc[i][lane] : -INFINITY;
        float v1 = lane + 64 < KV ? sc[i][lane + 64] : -INFINITY;
        const float mx = wave_max(fmaxf(v0, v1));
        const float e0 = lane < KV ? __expf(v0 - mx) : 0.0f;
        const float e1 = lane + 64 < KV ? __expf(v1 - mx) : 0.0f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        if (lane < KV) sc[i][lane] = e0 * inv;
        if (lane + 64 < KV) sc[i][lane + 64] = e1 * inv;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * DH; e += 256) { // context = P.V, :563
        const int i = e >> 7, d = e & 127;
        float acc = 0.0f;
        int ring = head0;
#pragma unroll 4
        for (int j = 0; j < KV; j++) {
            float vv;
            if (BF16) vv = bf16_to_f32(((const bf16_t *)vbase)[(size_t)ring * D + h * DH + d]);
            else vv = ((const float *)vbase)[(size_t)ring * D + h * DH + d];
            acc += sc[i][j] * vv;
            if (++ring == KVC) ring = 0;
        }
        const size_t o = (row0 + i) * D + h * DH + d;
        if (BF16) ((bf16_t *)p.ctx_out)[o] = f32_to_bf16(acc);
        else ((float *)p.ctx_out)[o] = acc;
    }
}
// ---- attention on the matrix cores (bf16, T <= 16): one workgroup per (head, stream, group of query rows) ----
// A workgroup takes QB consecutive rows of a stream: one chunk (QB = T) for T >= 4, sixteen rows = 16/T chunks for
// T <= 2 (multi-chunk steps at 80/160 ms lookahead).  Its keys are the SPAN of ring rows those rows can see:
// 70 + (chunks in the group) * T <= 86 rows; row (chunk g, frame i) attends the window of 70 + T keys that starts
// (g - g0) * T rows into the span, everything else gets weight 0.
// scores^T tiles  S[j][i] = K[j].(q_i+u)   (A = 16 span keys straight from the ring, B = queries)
// pos-score tiles P[r][i] = Pp[r].(q_i+v)  for every relative row r, rel-shift = an LDS gather
// context^T       O[d][i] = sum_j V^T[d][j] w[i][j]   (V tile transposed into LDS once per workgroup)
// q+u / q+v and the softmax weights are rounded to bf16 for the MFMA (two extra rounding points vs the
// VALU kernel; within the stated bf16 tolerance).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#ifdef ATTN_STAMPS          // tests/micro/attn_probe.hip: 8 real-time stamps (100 MHz) per workgroup
__device__ unsigned long long *g_attn_stamps;
#define ASTAMP(i) do { if (threadIdx.x == 0 && g_attn_stamps) g_attn_stamps[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(256) void k_attention_mfma(AttnParams p) {
    // LDS (round 5): the V^T image (32 KiB) is needed in phase 3 only, the query images and the score tiles in phases 0-2 only: they share one region,
    // the V rows wait in registers until the scores are done (one more barrier).  58 -> 36 KiB per workgroup: four workgroups per CU instead of two --
    // the kernel is one load round trip, then three short compute phases; what hides a workgroup's round trip is the other workgroups of its CU
    // (512 streams x R = 13: 77 us per launch at 2.9 TB/s before).
    constexpr int SKP = 100, SPP = 116;                    // row pitches (floats) of the score tiles: 96 / 112 + 4, so the 16 query rows of a float4 store spread over the banks
    __shared__ __attribute__((aligned(16))) char lds[128 * 256 + 16 * 256];
    char *const vt_s = lds;                                            // [128 d][256 B]                       phase 3
    char *const qu_s = lds, *const qv_s = lds + 4096;                  // [16][256 B] each                     phases 0-1
    float *const sk = (float *)(lds + 8192), *const sp = (float *)(lds + 8192 + 16 * SKP * 4);      // score tiles   phases 1-2 (ends at 22 016 < 32 768)
    char *const w_s = lds + 128 * 256;                                 // softmax weights [16][256 B]          phases 2-3
    ASTAMP(0);
    const int h = blockIdx.x, b = blockIdx.y, T = p.T, KV = LCTX + T, n_rel = KV + T - 1;
    const int TS = p.TS > 0 ? p.TS : p.T;
    const int QB = T <= 2 ? 16 : T;                          // query rows per workgroup
    const int r0 = blockIdx.z * QB, nrows = TS - r0 < QB ? TS - r0 : QB;
    const int g0 = r0 / T;                                   // first chunk of the group
    const int span = LCTX + ((nrows + T - 1) / T) * T;       // ring rows the group can see (<= 86 < 96)
    const RowDesc rd = p.rows[b];
    const size_t row0 = (size_t)b * TS + (size_t)r0;
    int head0 = rd.kv_head + g0 * T;
    while (head0 >= KVC) head0 -= KVC;
    const bf16_t *kbase = (const bf16_t *)p.kv_pool + (size_t)rd.slot * p.kv_slot_stride + h * DH;
    const bf16_t *vbase = kbase + (size_t)KVC * D;
    const bf16_t *pbase = (const bf16_t *)p.posproj + h * DH;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, r = lane & 15;
    // ---- every global read of the workgroup is requested up front (they depend only on the row descriptor): the V
    // rows, this wave's K / relative-position tiles and the queries arrive in ONE round trip instead of three ----
    float qreg[8];                                         // the queries first: the previous kernel wrote them
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int e = threadIdx.x + it * 256, i = e >> 7, d = e & 127;
        qreg[it] = i < nrows ? p.q[(row0 + i) * D + h * DH + d] : 0.0f;
    }
    uint4 vreg[6];
#pragma unroll
    for (int it = 0; it < 6; it++) {
        const int e = threadIdx.x + it * 256, c = e & 15, j = e >> 4;      // 8 consecutive d of span key j
        vreg[it] = make_uint4(0, 0, 0, 0);
        if (j < span) {
            int ring = head0 + j;
            if (ring >= KVC) ring -= KVC;
            vreg[it] = *(const uint4 *)(vbase + (size_t)ring * D + c * 8);
        }
    }
    uint4 av[4][4];                                        // tiles wave, wave + 4, wave + 8, wave + 12 of the 6 key + 7 position tiles
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const int t = wave + 4 * n;
        if (t < 13) {
            const bf16_t *arow;
            if (t < 6) {
                int j = t * 16 + r;
                if (j >= span) j = span - 1;
                int ring = head0 + j;
                if (ring >= KVC) ring -= KVC;
                arow = kbase + (size_t)ring * D;
            } else {
                int rr = (t - 6) * 16 + r;
                if (rr >= n_rel) rr = n_rel - 1;
                arow = pbase + (size_t)rr * D;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ks++) av[n][ks] = *(const uint4 *)(arow + ks * 32 + q * 8);
        }
    }
    // ---- phase 0: queries (+u, +v) -> bf16 LDS (the V rows stay in registers until the scores are done) ----
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int e = threadIdx.x + it * 256, i = e >> 7, d = e & 127;
        const float qq = qreg[it];
        const int off = i * 256 + ((((d >> 3) ^ i) & 15) << 4) + (d & 7) * 2;
        *(bf16_t *)(qu_s + off) = f32_to_bf16(qq + p.bias_u[h * DH + d]);
        *(bf16_t *)(qv_s + off) = f32_to_bf16(qq + p.bias_v[h * DH + d]);
    }
    __syncthreads();
    ASTAMP(1);
    // ---- phase 1: 6 key tiles + 7 relative-position tiles on the MFMA -------------------------------
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const int t = wave + 4 * n;
        if (t < 13) {
            const bool isk = t < 6;
            const char *bq = isk ? qu_s : qv_s;
            f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const uint4 bv = *(const uint4 *)(bq + r * 256 + ((((ks << 2) | q) ^ r) << 4));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[n][ks]), __builtin_bit_cast(bf16x8_t, bv), acc, 0, 0, 0);
            }
            // D[j][i]: lane holds query i = r, rows 4q + reg
            float *dst = isk ? sk + r * SKP + t * 16 + q * 4 : sp + r * SPP + (t - 6) * 16 + q * 4;
            *(float4 *)dst = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
    __syncthreads();
    ASTAMP(2);
    // ---- phase 2: rel-shift gather + mask + softmax -> bf16 weights over the span (zero outside the row's window) ----
    const float scale = 0.08838834764831845f;
    for (int i = wave; i < 16; i += 4) {
        const int il = r0 + i, g = il / T, ic = il - g * T;      // (chunk, frame in chunk) of this row
        const int wo = (g - g0) * T;                             // its window starts wo rows into the span
        const int valid = rd.valid_len + g * T < LCTX ? rd.valid_len + g * T : LCTX;
        const int mask_upto = LCTX - valid;
        const int j0 = lane - wo, j1 = lane + 64 - wo;           // window index of span positions lane, lane + 64
        const bool in0 = i < nrows && j0 >= 0 && j0 < KV, in1 = i < nrows && j1 >= 0 && j1 < KV;
        float v0 = -INFINITY, v1 = -INFINITY;
        if (in0) { v0 = (sk[i * SKP + lane] + sp[i * SPP + j0 + T - 1 - ic]) * scale; if (j0 < mask_upto) v0 += -1e9f; }
        if (in1) { v1 = (sk[i * SKP + lane + 64] + sp[i * SPP + j1 + T - 1 - ic]) * scale; if (j1 < mask_upto) v1 += -1e9f; }
        const float mx = wave_max(fmaxf(v0, v1));
        const float e0 = in0 ? __expf(v0 - mx) : 0.0f;
        const float e1 = in1 ? __expf(v1 - mx) : 0.0f;
        const float sum = wave_sum(e0 + e1);
        const float inv = i < nrows ? 1.0f / sum : 0.0f;
        const int p0 = lane, p1 = lane + 64;
        *(bf16_t *)(w_s + i * 256 + ((((p0 >> 3) ^ i) & 15) << 4) + (p0 & 7) * 2) = f32_to_bf16(e0 * inv);
        *(bf16_t *)(w_s + i * 256 + ((((p1 >> 3) ^ i) & 15) << 4) + (p1 & 7) * 2) = f32_to_bf16(e1 * inv);
    }
    __syncthreads();                          // every read of the query images and the score tiles is done: their region becomes the V^T image
    ASTAMP(3);
    // V^T image: row d = 256 B = 16 chunks of 8 keys; chunk (j >> 3) sits at ((j >> 3) ^ d ^ (d >> 4)) & 15.  The 16 lanes
    // that hold the 16 d-groups of one key write 16 different chunks (d & 15 alone takes two values there: 16-way
    // conflicts, SQ_LDS_BANK_CONFLICT 85 % of the LDS cycles of this kernel before)
    if (!(p.ablate & 1))
#pragma unroll
    for (int it = 0; it < 6; it++) {
        const int e = threadIdx.x + it * 256, c = e & 15, j = e >> 4;
        const uint32_t w4[4] = {vreg[it].x, vreg[it].y, vreg[it].z, vreg[it].w};
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int d = c * 8 + u;
            const bf16_t val = (bf16_t)((u & 1) ? (w4[u >> 1] >> 16) : (w4[u >> 1] & 0xffffu));
            *(bf16_t *)(vt_s + d * 256 + ((((j >> 3) ^ d ^ (d >> 4)) & 15) << 4) + (j & 7) * 2) = val;
        }
    }
    __syncthreads();
    ASTAMP(4);
    // ---- phase 3: O^T[d][i] = V^T . w^T, 8 d-tiles x 3 k-steps ---------------------------------------
    for (int dt = wave * 2; dt < wave * 2 + 2; dt++) {
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            const int d = dt * 16 + r;
            const uint4 av = *(const uint4 *)(vt_s + d * 256 + (((((ks << 2) | q) ^ d ^ (d >> 4)) & 15) << 4));
            const uint4 bv = *(const uint4 *)(w_s + r * 256 + (((((ks << 2) | q) ^ r) & 15) << 4));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bv), acc, 0, 0, 0);
        }
        if (r < nrows) {   // D[d][i]: lane holds query i = r, d = dt*16 + 4q + reg
            uint2 o;
            o.x = (uint32_t)f32_to_bf16(acc[0]) | ((uint32_t)f32_to_bf16(acc[1]) << 16);
            o.y = (uint32_t)f32_to_bf16(acc[2]) | ((uint32_t)f32_to_bf16(acc[3]) << 16);
            store_wt_u2((bf16_t *)p.ctx_out + (row0 + r) * D + h * DH + dt * 16 + q * 4, o);
        }
    }
    ASTAMP(5);
}

// (round 6, measured and removed: a form of this kernel in which a workgroup walks four streams of a head with the NEXT stream's rows in flight in a second
// register set and the relative-position tiles loaded once -- bit-identical, 252 VGPRs, two workgroups per CU.  Alone with cold rings 86.9 -> 67.9 us at 512
// streams (tests/micro/attn_probe.hip), inside the engine SLOWER: 512 streams pipelined 13.05 -> 13.23 ms, synchronous 14.90 -> 15.20, 256 streams 6.70 -> 6.77
// on the same box: there four workgroups per CU win over two with a deeper queue.  profiles/r6_attention.md)
// ---- one new row per stream (T = 1: every R = 0 batch), bf16 caches ------------------------------------------------------------
// k_attention_mfma feeds its MFMAs with 16-byte loads in the operand layout (16 lanes = 16 different rows): at T = 1 there is
// next to no arithmetic and the kernel is bound by its address unit (12.4 us per launch at 64 streams, the same as at T = 14).
// Here a wave's load instruction covers 4 whole 256-byte head rows: lane = (sub-row sr, chunk c of 8 dims), key
// j = 16 t + 4 wave + sr in pass t = 0..4; partial v_dot2 sums of a row's 16 lanes add up on the DPP path; P.V in the same
// layout.  The arithmetic is that of the one-row prologue of k_fused_skinny<PRO_ATTN, 1> (kernels_fused.hip), operation for
// operation: a stream gets the same context row whether it is stepped alone (fused layer) or in a batch.
typedef __attribute__((ext_vector_type(2))) __bf16 attn_bf16x2;
__global__ __launch_bounds__(256) void k_attention_row1(AttnParams a) {
    __shared__ float sc[96];
    __shared__ __attribute__((aligned(16))) float pv[16 * DH];
    const int h = blockIdx.x, b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sr = lane >> 4, c = lane & 15;
    constexpr int KV = LCTX + 1;
    const float scale = 0.08838834764831845f;
    const RowDesc rd = a.rows[b];
    const float4 *qp = (const float4 *)(a.q + (size_t)b * D + h * DH + c * 8);
    const float4 *up = (const float4 *)(a.bias_u + h * DH + c * 8), *vp = (const float4 *)(a.bias_v + h * DH + c * 8);
    const float4 q0 = qp[0], q1 = qp[1], u0 = up[0], u1 = up[1], b0 = vp[0], b1 = vp[1];
    uint4 kk[5], pp[5], vr[5];
    int jk[5];
#pragma unroll
    for (int t = 0; t < 5; t++) {
        jk[t] = t * 16 + wave * 4 + sr;
        const int jc = jk[t] < KV ? jk[t] : 0;
        pp[t] = *(const uint4 *)((const bf16_t *)a.posproj + (size_t)jc * D + h * DH + c * 8);      // row j + T - 1 - i = j
    }
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
            s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(attn_bf16x2, qu8[e]), __builtin_bit_cast(attn_bf16x2, kw[e]), s1, false);
            s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(attn_bf16x2, qv8[e]), __builtin_bit_cast(attn_bf16x2, pw[e]), s1, false);
        }
        s1 += dpp_mov<0xB1>(s1);
        s1 += dpp_mov<0x4E>(s1);
        s1 += dpp_mov<0x141>(s1);
        s1 += dpp_mov<0x140>(s1);
        if (c == 0 && jk[t] < KV) {
            float v = s1 * scale;
            if (jk[t] < LCTX - valid) v += -1e9f;
            sc[jk[t]] = v;
        }
    }
    __syncthreads();
    const float v0 = lane < KV ? sc[lane] : -INFINITY, v1 = lane + 64 < KV ? sc[lane + 64] : -INFINITY;
    const float mx = wave_max(fmaxf(v0, v1));
    const float inv = 1.0f / wave_sum((lane < KV ? __expf(v0 - mx) : 0.0f) + (lane + 64 < KV ? __expf(v1 - mx) : 0.0f));
    float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 5; t++) {
        const float w = jk[t] < KV ? __expf(sc[jk[t]] - mx) * inv : 0.0f;
        const uint32_t vw[4] = {vr[t].x, vr[t].y, vr[t].z, vr[t].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            acc8[2 * e] += w * __uint_as_float(vw[e] << 16);
            acc8[2 * e + 1] += w * __uint_as_float(vw[e] & 0xffff0000u);
        }
    }
    *(float4 *)(pv + (wave * 4 + sr) * DH + c * 8) = make_float4(acc8[0], acc8[1], acc8[2], acc8[3]);
    *(float4 *)(pv + (wave * 4 + sr) * DH + c * 8 + 4) = make_float4(acc8[4], acc8[5], acc8[6], acc8[7]);
    __syncthreads();
    if ((int)threadIdx.x < DH) {
        const int d = threadIdx.x;
        float o = 0.f;
#pragma unroll
        for (int gI = 0; gI < 16; gI++) o += pv[gI * DH + d];
        ((bf16_t *)a.ctx_out)[(size_t)b * D + h * DH + d] = f32_to_bf16(o);
    }
}

void launch_attention(const AttnParams &p, hipStream_t st) {
    const int TS = p.TS > 0 ? p.TS : p.T;
    if (p.act_bf16 && p.T == 1 && TS == 1) {
        hipLaunchKernelGGL(k_attention_row1, dim3(NH, p.B), dim3(256), 0, st, p);
        return;
    }
    if (p.act_bf16 && p.T <= 16) {
        const int QB = p.T <= 2 ? 16 : p.T;
        hipLaunchKernelGGL(k_attention_mfma, dim3(NH, p.B, (TS + QB - 1) / QB), dim3(256), 0, st, p);
        return;
    }
    const dim3 grid(NH, p.B, TS / p.T);
    if (p.act_bf16) hipLaunchKernelGGL(k_attention<true>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(k_attention<false>, grid, dim3(256), 0, st, p);
}

// ---- depthwise conv + LN + SiLU: one workgroup per (frame, stream) ------------------------------
// Grid = (streams, frames): workgroups are dealt round-robin over the 8 XCDs by their linear id = frame * B + stream, so with
// B a multiple of 8 every frame of a stream lands on the same XCD and the 9-row window it shares with its neighbours is
// fetched into that XCD's L2 once (PMC, round 2: with frames fastest the kernel moved 37 MB per launch for 5.5 MB of
// algorithmic bytes at 64 streams x 14 frames -- each XCD re-fetched the rows).  Placement changes speed only.
__global__ __launch_bounds__(256) void k_dwconv(ConvParams p) {
    __shared__ float sh[8];
    const int i = blockIdx.y, b = blockIdx.x, T = p.T, ks1 = p.ks - 1;
    const int c4 = threadIdx.x * 4;
    const RowDesc rd = p.rows[b];
    const float *cc_in = p.cc_pool + (size_t)rd.slot * p.cc_slot_stride + (size_t)rd.cc_par * ks1 * D;
    float *cc_out = p.cc_pool + (size_t)rd.slot * p.cc_slot_stride + (size_t)(rd.cc_par ^ 1) * ks1 * D;
    const float *g = p.glu + (size_t)b * T * D;
    // z = [conv cache (ks-1 rows) ; GLU(new T rows)], out[t] = sum_k z[t+k] * w[k]  (:368-388)
    const float4 lw = *(const float4 *)(p.ln_w + c4), lb = *(const float4 *)(p.ln_b + c4);
    float4 acc;
    if (p.ks == 9) {           // the model's kernel size: taps unrolled, all 18 loads in flight together
        float4 z[9], w[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int r = i + k;
            z[k] = r < 8 ? *(const float4 *)(cc_in + (size_t)r * D + c4) : *(const float4 *)(g + (size_t)(r - 8) * D + c4);
            w[k] = *(const float4 *)(p.dw + (size_t)k * D + c4);
        }
        acc = make_float4(z[0].x * w[0].x, z[0].y * w[0].y, z[0].z * w[0].z, z[0].w * w[0].w);
#pragma unroll
        for (int k = 1; k < 9; k++) { acc.x += z[k].x * w[k].x; acc.y += z[k].y * w[k].y; acc.z += z[k].z * w[k].z; acc.w += z[k].w * w[k].w; }
    } else {
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < p.ks; k++) {
            const int r = i + k;
            const float4 z = r < ks1 ? *(const float4 *)(cc_in + (size_t)r * D + c4)
                                     : *(const float4 *)(g + (size_t)(r - ks1) * D + c4);
            const float4 w = *(const float4 *)(p.dw + (size_t)k * D + c4);
            if (k == 0) acc = make_float4(z.x * w.x, z.y * w.y, z.z * w.z, z.w * w.w);
            else { acc.x += z.x * w.x; acc.y += z.y * w.y; acc.z += z.z * w.z; acc.w += z.w * w.w; }
        }
    }
    float4 n = ln4(acc, lw, lb, sh, threadIdx.x);                                               // :671-673
    n.x = n.x / (1.0f + __expf(-n.x)); n.y = n.y / (1.0f + __expf(-n.y));          // SiLU :674
    n.z = n.z / (1.0f + __expf(-n.z)); n.w = n.w / (1.0f + __expf(-n.w));
    store_act4(p.c_out, ((size_t)b * T + i) * D + c4, n, p.act_bf16);
    if (i == 0) {   // new conv cache = last ks-1 rows of z (:396-408), written to the other buffer
        for (int r2 = 0; r2 < ks1; r2++) {
            const int r = T + r2;
            const float4 z = r < ks1 ? *(const float4 *)(cc_in + (size_t)r * D + c4)
                                     : *(const float4 *)(g + (size_t)(r - ks1) * D + c4);
            *(float4 *)(cc_out + (size_t)r2 * D + c4) = z;
        }
    }
}
// ---- the same, one workgroup per STREAM (or per half of its frames: below) (round 5; kernel size 9) ------------------------------------------
// k_dwconv's grid is (streams, frames): every GLU row is fetched by nine workgroups and the conv cache by eight, 7 168 workgroups of one
// load round trip + two barriers at 512 streams x R = 13 (30 us per launch for 78 MB of algorithmic bytes; with attention 2.3 ms of a
// 14.6 ms pipelined step, profiles/r5_configs2_launch_structure.md, r5_ablation_b64_R13.json).  Here a workgroup keeps the stream's whole window -- 8 cached rows + T new ones, its 4
// channels per thread -- in registers: every row is read ONCE, the T outputs are formed in k_dwconv's order, their T LayerNorms share two
// barriers (per row the sums are k_dwconv's: wave_sum, then (w0 + w1) + (w2 + w3)), and the new cache is the window's last 8 rows.
// Same bits as k_dwconv (engine option "dwconv_stream" = 0; tests/micro/gemm_variant_identity.py).
// NF frames per workgroup (7: a stream's 14 frames at 1.12 s lookahead are two workgroups, whose load and compute phases overlap on a CU -- four
// fit -- where ONE workgroup per stream with all 22 rows in registers was a single round of 2 per CU: 26.2 us per launch at 512 streams).  The second
// workgroup re-reads the 8 rows in front of its frames; the one that holds the stream's last frame writes the new cache.
template <int NF>
__global__ __launch_bounds__(256) void k_dwconv_stream(ConvParams p) {
    __shared__ float sh[2][NF][4];
    const int b = blockIdx.x, T = p.T, f0 = blockIdx.y * NF, nf = T - f0 < NF ? T - f0 : NF;
    const int c4 = threadIdx.x * 4, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const RowDesc rd = p.rows[b];
    const float *cc_in = p.cc_pool + (size_t)rd.slot * p.cc_slot_stride + (size_t)rd.cc_par * 8 * D;
    float *cc_out = p.cc_pool + (size_t)rd.slot * p.cc_slot_stride + (size_t)(rd.cc_par ^ 1) * 8 * D;
    const float *g = p.glu + (size_t)b * T * D;
    // local row j = row f0 + j of z = [conv cache (8 rows) ; GLU(new T rows)]
    float4 z[8 + NF], w[9];
#pragma unroll
    for (int j = 0; j < 8 + NF; j++) {
        const int r = f0 + j;
        z[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < 8) z[j] = *(const float4 *)(cc_in + (size_t)r * D + c4);
        else if (r - 8 < T) z[j] = *(const float4 *)(g + (size_t)(r - 8) * D + c4);
    }
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = *(const float4 *)(p.dw + (size_t)k * D + c4);
    const float4 lw = *(const float4 *)(p.ln_w + c4), lb = *(const float4 *)(p.ln_b + c4);
    float4 acc[NF];
    float s[NF];
#pragma unroll
    for (int i = 0; i < NF; i++) {
        float4 a = make_float4(z[i].x * w[0].x, z[i].y * w[0].y, z[i].z * w[0].z, z[i].w * w[0].w);
#pragma unroll
        for (int k = 1; k < 9; k++) { a.x += z[i + k].x * w[k].x; a.y += z[i + k].y * w[k].y; a.z += z[i + k].z * w[k].z; a.w += z[i + k].w * w[k].w; }
        acc[i] = a;
        s[i] = wave_sum((a.x + a.y) + (a.z + a.w));
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NF; i++) sh[0][i][wave] = s[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NF; i++) {
        const float mean = ((sh[0][i][0] + sh[0][i][1]) + (sh[0][i][2] + sh[0][i][3])) * (1.0f / D);
        acc[i].x -= mean; acc[i].y -= mean; acc[i].z -= mean; acc[i].w -= mean;
        s[i] = wave_sum((acc[i].x * acc[i].x + acc[i].y * acc[i].y) + (acc[i].z * acc[i].z + acc[i].w * acc[i].w));
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NF; i++) sh[1][i][wave] = s[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NF; i++) {
        if (i < nf) {
            const float var = ((sh[1][i][0] + sh[1][i][1]) + (sh[1][i][2] + sh[1][i][3])) * (1.0f / D);
            const float inv = 1.0f / sqrtf(var + 1e-5f);
            float4 n = make_float4(acc[i].x * inv * lw.x + lb.x, acc[i].y * inv * lw.y + lb.y, acc[i].z * inv * lw.z + lb.z, acc[i].w * inv * lw.w + lb.w);
            n.x = n.x / (1.0f + __expf(-n.x)); n.y = n.y / (1.0f + __expf(-n.y));          // SiLU :674
            n.z = n.z / (1.0f + __expf(-n.z)); n.w = n.w / (1.0f + __expf(-n.w));
            store_act4(p.c_out, ((size_t)b * T + f0 + i) * D + c4, n, p.act_bf16);
        }
    }
    // new conv cache = the last 8 rows of z (:396-408), written to the other buffer by the workgroup that holds the stream's last frame: local rows nf + r2
    if (f0 + nf == T) {
#pragma unroll
        for (int r2 = 0; r2 < 8; r2++) {
            float4 v = z[r2];
#pragma unroll
            for (int i = 1; i <= NF; i++)
                if (i == nf) v = z[i + r2];
            *(float4 *)(cc_out + (size_t)r2 * D + c4) = v;
        }
    }
}
void launch_dwconv(const ConvParams &p, hipStream_t st) {
    // from 256 streams: below that too few workgroups (64 streams: 64 of 256 CUs busy, a synchronous step 4.05 -> 4.24 ms); 512 streams x R = 13: 30.3 -> 26.2 us
    // per launch, a pipelined step 14.46 -> 14.18 ms
    if (p.stream_form && p.ks == 9 && p.T >= 7 && p.T % 7 == 0 && p.B >= 256) {
        hipLaunchKernelGGL(k_dwconv_stream<7>, dim3(p.B, p.T / 7), dim3(256), 0, st, p);
        return;
    }
    hipLaunchKernelGGL(k_dwconv, dim3(p.B, p.T), dim3(256), 0, st, p);
}

// ---- relative-position sinusoid rows (computed on the host like the reference does at load,
// src/nemo-ggml.cpp:17-32; see engine) -- nothing to do on the device. --------------------------

// ---- prompt fusion helper: h[m][n] = relu(h[m][n] + w1p[prompt(m)][n])  (src/nemo-ggml.cpp:1097-1101;
// the one-hot column of the concatenated input selects one column of fc1) --------------------------
__global__ void k_prompt_add_relu(float *h, const float *w1p /*[P][2048]*/, const RowDesc *rows, int T, int P) {
    const int m = blockIdx.x;
    int idx = rows[m / T].prompt;
    if (idx < 0 || idx >= P) idx = 0;                      // src/nemo-stream.cpp:1052-1053
    for (int n = threadIdx.x; n < 2048; n += 256) {
        float v = h[(size_t)m * 2048 + n] + w1p[(size_t)idx * 2048 + n];
        h[(size_t)m * 2048 + n] = fmaxf(v, 0.0f);
    }
}
void launch_prompt_add_relu(float *h, const float *w1p, const RowDesc *rows, int M, int T, int P, hipStream_t st) {
    hipLaunchKernelGGL(k_prompt_add_relu, dim3(M), dim3(256), 0, st, h, w1p, rows, T, P);
}

}  // namespace nasr

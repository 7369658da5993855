// kernels_layer.hip -- the non-GEMM parts of one cached conformer layer
// (reference src/nemo-stream.cpp:605-690):
//   k_post      residual add of the split-K partials + LayerNorm(s)     (:580-591, :633-634, :687)
//   k_attention cached relative-position attention with the rel-shift, the validity mask,
//               softmax and P.V in one kernel, K/V read straight from the ring (:463-573)
//   k_dwconv    cached causal depthwise conv + LayerNorm + SiLU + conv-cache update (:336-412, :671-674)
#include "nasr_internal.h"

namespace nasr {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// sum over a 256-thread block; every thread gets the result. `sh` = 4 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float *sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// LayerNorm of the 4 elements each thread holds (1024 = 256 x 4), biased variance, eps 1e-5
__device__ __forceinline__ float4 ln4(float4 v, const float *w, const float *b, int c4, float *sh) {
    float mean = block_sum((v.x + v.y) + (v.z + v.w), sh) * (1.0f / D);
    float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    float var = block_sum((dx * dx + dy * dy) + (dz * dz + dw * dw), sh) * (1.0f / D);
    float inv = 1.0f / sqrtf(var + 1e-5f);
    const float4 ww = *(const float4 *)(w + c4), bb = *(const float4 *)(b + c4);
    return make_float4(dx * inv * ww.x + bb.x, dy * inv * ww.y + bb.y, dz * inv * ww.z + bb.z, dw * inv * ww.w + bb.w);
}

__device__ __forceinline__ void store_act4(void *base, size_t off, float4 v, int bf16) {
    if (bf16) {
        uint2 r;
        r.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
        r.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
        *(uint2 *)((bf16_t *)base + off) = r;
    } else {
        *(float4 *)((float *)base + off) = v;
    }
}

// ---- post: one workgroup per row ------------------------------------------------------------
__global__ __launch_bounds__(256) void k_post(PostParams p) {
    __shared__ float sh[4];
    const int m = blockIdx.x, c4 = threadIdx.x * 4;
    float4 v = *(const float4 *)(p.x + (size_t)m * D + c4);
    if (p.splits > 0) {
        float4 o = *(const float4 *)(p.part + (size_t)m * D + c4);
        for (int s = 1; s < p.splits; s++) {
            const float4 t = *(const float4 *)(p.part + ((size_t)s * p.M + m) * D + c4);
            o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
        }
        v.x += p.scale * o.x; v.y += p.scale * o.y; v.z += p.scale * o.z; v.w += p.scale * o.w;
    }
    if (p.ln_out) v = ln4(v, p.ln1_w, p.ln1_b, c4, sh);
    if (p.splits > 0 || p.ln_out) *(float4 *)(p.x + (size_t)m * D + c4) = v;
    if (p.copy_out) *(float4 *)(p.copy_out + (size_t)m * D + c4) = v;
    if (p.ln2_w) {
        float4 a = ln4(v, p.ln2_w, p.ln2_b, c4, sh);
        store_act4(p.a_out, (size_t)m * D + c4, a, p.act_bf16);
    }
}
void launch_post(const PostParams &p, hipStream_t st) {
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
    __shared__ float qu[TMAX][DH], qv[TMAX][DH];
    __shared__ float sc[TMAX][KVC];
    const int h = blockIdx.x, b = blockIdx.y, T = p.T, KV = LCTX + T;
    const RowDesc rd = p.rows[b];
    const int esz = BF16 ? 2 : 4;
    const char *kbase = (const char *)p.kv_pool + ((size_t)rd.slot * p.kv_slot_stride) * esz;
    const char *vbase = kbase + (size_t)KVC * D * esz;
    for (int e = threadIdx.x; e < T * DH; e += 256) {
        const int i = e >> 7, d = e & 127;
        const float q = p.q[((size_t)b * T + i) * D + h * DH + d];
        qu[i][d] = q + p.bias_u[h * DH + d];     // src/nemo-stream.cpp:531-535
        qv[i][d] = q + p.bias_v[h * DH + d];
    }
    __syncthreads();
    const float scale = 0.08838834764831845f;    // 1/sqrt(128), :545
    const int mask_upto = LCTX - rd.valid_len;   // :1037-1043
    for (int e = threadIdx.x; e < T * KV; e += 256) {
        const int i = e / KV, j = e - i * KV;
        int ring = rd.kv_head + j;
        if (ring >= KVC) ring -= KVC;
        const char *krow = kbase + ((size_t)ring * D + h * DH) * esz;
        // rel-shift folded into indexing: slice row j + T - 1 - i  <->  rel = (70 + i) - j  (:419-461)
        const char *prow = (const char *)p.posproj + ((size_t)(j + T - 1 - i) * D + h * DH) * esz;
        const float s1 = dot128<BF16>(qu[i], krow);   // :538
        const float s2 = dot128<BF16>(qv[i], prow);   // :541-542
        float v = (s1 + s2) * scale;                  // :546-547
        if (j < mask_upto) v += -1e9f;                // :552-556
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
        int ring = rd.kv_head;
        for (int j = 0; j < KV; j++) {
            float vv;
            if (BF16) vv = bf16_to_f32(((const bf16_t *)vbase)[(size_t)ring * D + h * DH + d]);
            else vv = ((const float *)vbase)[(size_t)ring * D + h * DH + d];
            acc += sc[i][j] * vv;
            if (++ring == KVC) ring = 0;
        }
        const size_t o = ((size_t)b * T + i) * D + h * DH + d;
        if (BF16) ((bf16_t *)p.ctx_out)[o] = f32_to_bf16(acc);
        else ((float *)p.ctx_out)[o] = acc;
    }
}
void launch_attention(const AttnParams &p, hipStream_t st) {
    if (p.act_bf16) hipLaunchKernelGGL(k_attention<true>, dim3(NH, p.B), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(k_attention<false>, dim3(NH, p.B), dim3(256), 0, st, p);
}

// ---- depthwise conv + LN + SiLU: one workgroup per (frame, stream) ------------------------------
__global__ __launch_bounds__(256) void k_dwconv(ConvParams p) {
    __shared__ float sh[4];
    const int i = blockIdx.x, b = blockIdx.y, T = p.T, ks1 = p.ks - 1;
    const int c4 = threadIdx.x * 4;
    const RowDesc rd = p.rows[b];
    const float *cc_in = p.cc_pool + (size_t)rd.slot * p.cc_slot_stride + (size_t)rd.cc_par * ks1 * D;
    float *cc_out = p.cc_pool + (size_t)rd.slot * p.cc_slot_stride + (size_t)(rd.cc_par ^ 1) * ks1 * D;
    const float *g = p.glu + (size_t)b * T * D;
    // z = [conv cache (ks-1 rows) ; GLU(new T rows)], out[t] = sum_k z[t+k] * w[k]  (:368-388)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < p.ks; k++) {
        const int r = i + k;
        const float4 z = r < ks1 ? *(const float4 *)(cc_in + (size_t)r * D + c4)
                                 : *(const float4 *)(g + (size_t)(r - ks1) * D + c4);
        const float4 w = *(const float4 *)(p.dw + (size_t)k * D + c4);
        if (k == 0) acc = make_float4(z.x * w.x, z.y * w.y, z.z * w.z, z.w * w.w);
        else { acc.x += z.x * w.x; acc.y += z.y * w.y; acc.z += z.z * w.z; acc.w += z.w * w.w; }
    }
    float4 n = ln4(acc, p.ln_w, p.ln_b, c4, sh);                                   // :671-673
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
void launch_dwconv(const ConvParams &p, hipStream_t st) {
    hipLaunchKernelGGL(k_dwconv, dim3(p.T, p.B), dim3(256), 0, st, p);
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

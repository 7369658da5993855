// nasr_epilogue.h -- GEMM epilogues shared by kernels_gemm.hip and kernels_fused.hip.
#pragma once
#include "nasr_internal.h"

namespace nasr {

// 1 / (1 + e^-x) on v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions): a 128 x 128 SiLU epilogue is 32 of
// these per thread (round 4: 1.3 us per tile in the persistent kernel's stamps).  1 + e^-x = inf gives 0, as the division did.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }

__device__ __forceinline__ const char *a_row_ptr(const GemmParams &p, int m, int elt) {
    size_t off;
    if (p.rows_per_batch > 0) {
        int b = m / p.rows_per_batch, i = m - b * p.rows_per_batch;
        off = (size_t)b * p.batch_stride + (size_t)(p.row_offset + i) * p.lda;
    } else {
        off = (size_t)m * p.lda;
    }
    return (const char *)p.A + off * elt;
}

__device__ __forceinline__ uint2 pack4_bf16(float a, float b, float c, float d) {
    uint2 r;
    r.x = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
    r.y = (uint32_t)f32_to_bf16(c) | ((uint32_t)f32_to_bf16(d) << 16);
    return r;
}

// epilogue for 4 consecutive n (n0 % 4 == 0) of row m; ACT_BF16 selects the act dtype
template <bool ACT_BF16>
__device__ __forceinline__ void epi_quad(const GemmParams &p, int split, int m, int n0, float v0,
                                         float v1, float v2, float v3) {
    if (m >= p.M) return;
    switch (p.epi) {
    case EPI_PART_F32: {
        float *o = p.out_f32 + ((size_t)split * p.M + m) * p.ldo + n0;
        *(float4 *)o = make_float4(v0, v1, v2, v3);
    } break;
    case EPI_SILU_ACT: {
        v0 = silu_f(v0); v1 = silu_f(v1); v2 = silu_f(v2); v3 = silu_f(v3);
        if (ACT_BF16) *(uint2 *)((bf16_t *)p.out_act + (size_t)m * p.ldo_act + n0) = pack4_bf16(v0, v1, v2, v3);
        else *(float4 *)((float *)p.out_act + (size_t)m * p.ldo_act + n0) = make_float4(v0, v1, v2, v3);
    } break;
    case EPI_QKV: {
        int which = n0 >> 10, col = n0 & 1023;
        if (which == 0) {
            *(float4 *)(p.q_out + (size_t)m * D + col) = make_float4(v0, v1, v2, v3);
        } else {
            int b = m / p.T, i = m - b * p.T;
            RowDesc rd = p.rows[b];
            int ring = rd.kv_head + LCTX + i;
            if (ring >= KVC) ring -= KVC;
            size_t off = (size_t)rd.slot * p.kv_slot_stride + ((size_t)(which - 1) * KVC + ring) * D + col;
            if (ACT_BF16) *(uint2 *)((bf16_t *)p.kv_pool + off) = pack4_bf16(v0, v1, v2, v3);
            else *(float4 *)((float *)p.kv_pool + off) = make_float4(v0, v1, v2, v3);
        }
    } break;
    case EPI_GLU: {   // rows were interleaved at upload: (2c, 2c+1) = (value c, gate c)
        float *o = p.out_f32 + (size_t)m * p.ldo + (n0 >> 1);
        *(float2 *)o = make_float2(v0 * sigmoid_f(v1), v2 * sigmoid_f(v3));
    } break;
    case EPI_BIAS_F32: {
        const float4 b = *(const float4 *)(p.bias + n0);
        *(float4 *)(p.out_f32 + (size_t)m * p.ldo + n0) = make_float4(v0 + b.x, v1 + b.y, v2 + b.z, v3 + b.w);
    } break;
    case EPI_RESID_F32: {
        const float4 x = *(const float4 *)(p.resid + (size_t)m * p.ldo + n0);
        *(float4 *)(p.out_f32 + (size_t)m * p.ldo + n0) = make_float4(__builtin_fmaf(p.resid_scale, v0, x.x), __builtin_fmaf(p.resid_scale, v1, x.y),
                                                                        __builtin_fmaf(p.resid_scale, v2, x.z), __builtin_fmaf(p.resid_scale, v3, x.w));
    } break;
    case EPI_BIAS_RELU_F32: {
        const float4 b = *(const float4 *)(p.bias + n0);
        *(float4 *)(p.out_f32 + (size_t)m * p.ldo + n0) =
            make_float4(fmaxf(v0 + b.x, 0.f), fmaxf(v1 + b.y, 0.f), fmaxf(v2 + b.z, 0.f), fmaxf(v3 + b.w, 0.f));
    } break;
    case EPI_BIAS_RELU_ACT:
    case EPI_BIAS_ACT: {
        const float4 b = *(const float4 *)(p.bias + n0);
        v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
        if (p.epi == EPI_BIAS_RELU_ACT) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        if (ACT_BF16) *(uint2 *)((bf16_t *)p.out_act + (size_t)m * p.ldo_act + n0) = pack4_bf16(v0, v1, v2, v3);
        else *(float4 *)((float *)p.out_act + (size_t)m * p.ldo_act + n0) = make_float4(v0, v1, v2, v3);
    } break;
    }
}


}  // namespace nasr

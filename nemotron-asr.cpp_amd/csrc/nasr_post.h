// nasr_post.h -- the residual add + LayerNorm(s) of ONE row by 256 threads (reference src/nemo-stream.cpp:580-591, :633-634, :687): the body of
// k_post (kernels_layer.hip, one workgroup per row) and of the head phase of a chained GEMM launch (kernels_gemm.hip, round 5: the workgroups
// at the front of the grid produce the A rows of the launch's own tiles).  One definition, so a row has the same bits wherever it is computed.
#pragma once
#include "nasr_internal.h"
#include "nasr_wave.h"

namespace nasr {

// sum over the 256 threads that work on a row; every one of them gets the result.  `sh` = 4 floats of LDS that the PREVIOUS block_sum of these
// threads did not use (callers alternate between two 4-float halves, so one barrier per sum is enough: a wave can only be one barrier ahead,
// and by then every wave has read the half that is being rewritten).  t256 = the thread's index among the 256 (the barrier is the workgroup's:
// in a 512-thread workgroup two rows go through it side by side, each with its own sh).
__device__ __forceinline__ float block_sum(float v, float *sh, int t256) {
    v = wave_sum(v);
    if ((t256 & 63) == 0) sh[t256 >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// LayerNorm of the 4 elements each thread holds (1024 = 256 x 4), biased variance, eps 1e-5; the affine parameters are passed in so that the
// caller can load them before the reductions; sh = 8 floats of LDS
__device__ __forceinline__ float4 ln4(float4 v, float4 ww, float4 bb, float *sh, int t256) {
    float mean = block_sum((v.x + v.y) + (v.z + v.w), sh, t256) * (1.0f / D);
    float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    float var = block_sum((dx * dx + dy * dy) + (dz * dz + dw * dw), sh + 4, t256) * (1.0f / D);
    float inv = 1.0f / sqrtf(var + 1e-5f);
    return make_float4(dx * inv * ww.x + bb.x, dy * inv * ww.y + bb.y, dz * inv * ww.z + bb.z, dw * inv * ww.w + bb.w);
}

__device__ __forceinline__ void store_act4(void *base, size_t off, float4 v, int bf16) {      // write-through (nasr_wave.h)
    if (bf16) {
        uint2 r;
        r.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
        r.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
        store_wt_u2((bf16_t *)base + off, r);
    } else {
        store_wt_f4((float *)base + off, v);
    }
}

// row m of PostParams p by the 256 threads t256 = 0 .. 255; live = false: the threads only keep the barriers company (a row past M in a
// workgroup that handles several rows: every thread of the workgroup must execute the same barriers)
__device__ __forceinline__ void post_row(const PostParams &p, int m, int t256, float *sh, bool live) {
    const int c4 = t256 * 4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // every load of the row is independent of the arithmetic: issue them all first
    float4 v = live ? *(const float4 *)(p.x + (size_t)m * D + c4) : z4;
    float4 t[8];
#pragma unroll
    for (int s = 0; s < 8; s++)
        if (s < p.splits) t[s] = live ? *(const float4 *)(p.part + ((size_t)s * p.M + m) * D + c4) : z4;
    float4 w1 = z4, b1 = z4, w2 = z4, b2 = z4;
    if (p.ln_out) { w1 = *(const float4 *)(p.ln1_w + c4); b1 = *(const float4 *)(p.ln1_b + c4); }
    if (p.ln2_w) { w2 = *(const float4 *)(p.ln2_w + c4); b2 = *(const float4 *)(p.ln2_b + c4); }
    if (p.splits > 0) {
        float4 o = t[0];
#pragma unroll
        for (int s = 1; s < 8; s++)
            if (s < p.splits) { o.x += t[s].x; o.y += t[s].y; o.z += t[s].z; o.w += t[s].w; }
        // an explicit fma: the same bits as the GEMM epilogues that fold this add (EPI_RESID_F32, __builtin_fmaf) whatever -ffp-contract says
        v.x = __builtin_fmaf(p.scale, o.x, v.x); v.y = __builtin_fmaf(p.scale, o.y, v.y); v.z = __builtin_fmaf(p.scale, o.z, v.z); v.w = __builtin_fmaf(p.scale, o.w, v.w);
    }
    if (p.ln_out) v = ln4(v, w1, b1, sh, t256);
    if (live && (p.splits > 0 || p.ln_out)) store_wt_f4(p.x + (size_t)m * D + c4, v);
    if (live && p.copy_out) *(float4 *)(p.copy_out + (size_t)m * D + c4) = v;
    if (p.ln2_w) {
        float4 a = ln4(v, w2, b2, sh, t256);
        if (live) store_act4(p.a_out, (size_t)m * D + c4, a, p.act_bf16);
    }
}

}  // namespace nasr

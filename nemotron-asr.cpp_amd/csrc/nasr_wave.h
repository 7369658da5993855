// nasr_wave.h -- wave64 all-reduce on the DPP path.
//
// __shfl_xor compiles to ds_bpermute_b32 (a trip through the LDS crossbar, ~100 cycles each, six dependent ones
// per reduction): in-kernel stamps of the batch-1 layer showed one LayerNorm (two block sums) costing ~0.9 us of a
// 4 us kernel.  Here the 16 lanes of a DPP row are combined by four VALU adds with lane-permuting operands
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: every lane ends with its row's total) and the four rows by
// v_readlane.  The summation order differs from the xor butterfly (a fixed order either way).
#pragma once
#include <hip/hip_runtime.h>

namespace nasr {

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_value(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);     // row_half_mirror
    v += dpp_mov<0x140>(v);     // row_mirror
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// Write-through stores (sc0 sc1) for bulk outputs that the NEXT kernel reads: the bytes go to memory as they are produced
// instead of sitting dirty in this XCD's L2 until the end-of-kernel write-back (tests/micro/gemm_probe.hip, mode 9).
typedef __attribute__((ext_vector_type(4))) float wt_f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned wt_u32x2;
__device__ __forceinline__ void store_wt_f4(float *p, float4 v) {
    const wt_f32x4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
}
// Sixteen bytes of packed 16-bit values: a PLAIN store.  The write-through asm form of it (global_store_dwordx4 ... sc0 sc1 with a uint vector operand)
// put wrong words into memory for lanes 12-15 of every 16 (round 5, tests/micro/post_probe.hip: 1 element in 32 of k_post_wave's bf16 rows; a wait
// state in front of the asm did not help, the compiler's own store of the same registers is right) -- not understood, not used.
__device__ __forceinline__ void store_u4(void *p, uint4 v) { *(uint4 *)p = v; }
__device__ __forceinline__ void store_wt_u2(void *p, uint2 v) {
    const wt_u32x2 w = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
}

}  // namespace nasr

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
//
// HAZARD (CDNA3/4 ISA, "VMEM store of more than 64 bits followed by a write of its data VGPRs": 2 wait states; cdna_hip_programming.md
// section 5.7 item 1): hipcc's hazard recognizer does not look inside an asm string, so every > 64-bit store below ENDS with `s_nop 1`
// inside the string -- whatever hipcc schedules behind the statement then finds the wait states already served.  Round 5 shipped the x4 form
// without the pad (correct only by scheduling accident) and saw the uint4 form put wrong words into lanes 12-15 when the compiler's next
// instruction (the v_perm/v_lshl_or packing of the following 16-bit group) rewrote the data registers one state after the store issued;
// a wait state IN FRONT of the store cannot help with that.  tests/test_store_hazards.py disassembles every built code object and fails on
// an asm-emitted dwordx3/x4 store whose data registers are written within two wait states.
typedef __attribute__((ext_vector_type(4))) float wt_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned wt_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned wt_u32x2;
__device__ __forceinline__ void store_wt_f4(float *p, float4 v) {
    const wt_f32x4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}
// Sixteen bytes of packed 16-bit values, write-through (see HAZARD above: the pad is what round 5's form lacked).
__device__ __forceinline__ void store_wt_u4(void *p, uint4 v) {
    const wt_u32x4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}
// Sixteen bytes of packed 16-bit values: a PLAIN store (the compiler's own; it pads its hazards itself).
__device__ __forceinline__ void store_u4(void *p, uint4 v) { *(uint4 *)p = v; }
__device__ __forceinline__ void store_wt_u2(void *p, uint2 v) {      // 64 bits: no data hazard (the rule is for > 64-bit stores)
    const wt_u32x2 w = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
}

}  // namespace nasr

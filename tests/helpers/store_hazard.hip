// Test helper (not product code): the ">64-bit VMEM store, then a write of its data VGPRs" hazard of CDNA3/4, measured on the hardware.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -I../../nemotron-asr.cpp_amd/csrc -o libstore_hazard.so store_hazard.hip   (built by __graft_entry__.build())
//
// Round 5 saw `global_store_dwordx4 ... sc0 sc1` from inline asm put wrong words into memory for lanes 12-15 of every 16 and could not say why.
// Mode 0 / 1 pin the rule itself with explicit registers inside ONE asm statement (so hipcc has no say in what follows the store):
//   mode 0: store, then IMMEDIATELY v_mov into the four data registers           -> the ISA says: undefined (the store may read the new values)
//   mode 1: store, s_nop 1, then the same v_movs                                  -> must be right
// Mode 2 is the product's helper (nasr_wave.h store_wt_u4, pad inside its string) followed by C++ that rewrites the packed words at once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "nasr_wave.h"

#define POISON 0xDEADBEEFu

template <int PAD>
__global__ void k_store_then_clobber(uint4 *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned a = 4u * i, b = 4u * i + 1, c = 4u * i + 2, d = 4u * i + 3;
    uint4 *p = out + i;
    if (PAD)
        asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, %3\n\tv_mov_b32 v23, %4\n\t"
                     "global_store_dwordx4 %0, v[20:23], off sc0 sc1\n\ts_nop 1\n\t"
                     "v_mov_b32 v20, %5\n\tv_mov_b32 v21, %5\n\tv_mov_b32 v22, %5\n\tv_mov_b32 v23, %5"
                     ::"v"(p), "v"(a), "v"(b), "v"(c), "v"(d), "v"(POISON) : "memory", "v20", "v21", "v22", "v23");
    else
        asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, %3\n\tv_mov_b32 v23, %4\n\t"
                     "global_store_dwordx4 %0, v[20:23], off sc0 sc1\n\t"
                     "v_mov_b32 v20, %5\n\tv_mov_b32 v21, %5\n\tv_mov_b32 v22, %5\n\tv_mov_b32 v23, %5"
                     ::"v"(p), "v"(a), "v"(b), "v"(c), "v"(d), "v"(POISON) : "memory", "v20", "v21", "v22", "v23");
}

// the product helper in the situation that broke in round 5: a run of 16-byte stores of freshly packed 16-bit values, the packing of the
// next group reusing the registers of the one just stored
__global__ void k_helper_pack_chain(uint4 *out, const float *in, int n, int groups) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = in[(size_t)i * 8 + k];
    for (int g = 0; g < groups; g++) {
        uint4 w;
        w.x = (__float_as_uint(v[0]) >> 16) | (__float_as_uint(v[1]) & 0xffff0000u);
        w.y = (__float_as_uint(v[2]) >> 16) | (__float_as_uint(v[3]) & 0xffff0000u);
        w.z = (__float_as_uint(v[4]) >> 16) | (__float_as_uint(v[5]) & 0xffff0000u);
        w.w = (__float_as_uint(v[6]) >> 16) | (__float_as_uint(v[7]) & 0xffff0000u);
        nasr::store_wt_u4(out + (size_t)g * n + i, w);
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = v[k] * 2.0f + (float)k;       // x2 is exact: one rounding whether or not the compiler contracts to an fma
    }
}

// returns the number of 32-bit words that differ from what was stored (>= 0), or < 0 on a HIP error; *first_bad_lane = lane (i % 64) of the first one
extern "C" int store_hazard_probe(int device, int mode, int n, int *first_bad_lane, unsigned *bad_lane_mask16) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    const int groups = 6;
    const size_t words = (size_t)n * 4 * (mode == 2 ? groups : 1);
    uint32_t *d = nullptr;
    float *din = nullptr;
    if (hipMalloc((void **)&d, words * 4) != hipSuccess) return -2;
    (void)hipMemset(d, 0xA5, words * 4);
    uint32_t *h = (uint32_t *)malloc(words * 4);
    float *hin = nullptr;
    if (mode == 2) {
        hin = (float *)malloc((size_t)n * 8 * 4);
        for (size_t k = 0; k < (size_t)n * 8; k++) hin[k] = (float)((k * 2654435761u) % 1000u) * 0.37f - 150.f;
        if (hipMalloc((void **)&din, (size_t)n * 8 * 4) != hipSuccess) return -3;
        (void)hipMemcpy(din, hin, (size_t)n * 8 * 4, hipMemcpyHostToDevice);
    }
    const dim3 grid((n + 255) / 256), block(256);
    for (int rep = 0; rep < 8; rep++) {
        if (mode == 0) hipLaunchKernelGGL(k_store_then_clobber<0>, grid, block, 0, 0, (uint4 *)d, n);
        else if (mode == 1) hipLaunchKernelGGL(k_store_then_clobber<1>, grid, block, 0, 0, (uint4 *)d, n);
        else hipLaunchKernelGGL(k_helper_pack_chain, grid, block, 0, 0, (uint4 *)d, din, n, groups);
    }
    if (hipDeviceSynchronize() != hipSuccess) return -4;
    (void)hipMemcpy(h, d, words * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    *first_bad_lane = -1;
    *bad_lane_mask16 = 0;
    if (mode != 2) {
        for (size_t k = 0; k < words; k++)
            if (h[k] != (uint32_t)k) {
                if (!bad) *first_bad_lane = (int)((k / 4) % 64);
                *bad_lane_mask16 |= 1u << ((k / 4) % 16);
                bad++;
            }
    } else {
        for (int i = 0; i < n; i++) {
            float v[8];
            for (int k = 0; k < 8; k++) v[k] = hin[(size_t)i * 8 + k];
            for (int g = 0; g < groups; g++) {
                for (int q = 0; q < 4; q++) {
                    uint32_t lo, hi;
                    memcpy(&lo, &v[2 * q], 4);
                    memcpy(&hi, &v[2 * q + 1], 4);
                    const uint32_t want = (lo >> 16) | (hi & 0xffff0000u);
                    if (h[((size_t)g * n + i) * 4 + q] != want) {
                        if (!bad) *first_bad_lane = i % 64;
                        *bad_lane_mask16 |= 1u << (i % 16);
                        bad++;
                    }
                }
                for (int k = 0; k < 8; k++) v[k] = v[k] * 2.0f + (float)k;
            }
        }
    }
    free(h);
    free(hin);
    (void)hipFree(d);
    if (din) (void)hipFree(din);
    return bad;
}

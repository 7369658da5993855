// micro-benchmark: does a read of 8 MiB that another kernel has just streamed hit on-die (L2 / Infinity Cache)?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void k_read(const u32x4 *p, size_t n16, unsigned *sink) {
    u32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { size_t j = i + u * stride; v[u] = j < n16 ? (NT ? __builtin_nontemporal_load(p + j) : p[j]) : acc; }
#pragma unroll
    for (int u = 0; u < 8; u++) acc ^= v[u];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const size_t MB = 1 << 20, total = 2048 * MB;
    char *buf; hipMalloc(&buf, total); hipMemset(buf, 1, total);
    unsigned *sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t sz : {8 * MB, 32 * MB}) {
        const size_t n16 = sz / 16;
        dim3 grid((unsigned)(n16 / 8 / 256)), blk(256);
        for (int nt = 0; nt < 2; nt++) {
            float cold = 0, warm = 0, warm_after_nt = 0;
            const int reps = 20;
            for (int r = 0; r < reps; r++) {
                // cold: a region not touched for > 1 GiB of traffic
                const u32x4 *pc = (const u32x4 *)(buf + ((size_t)(r * 3 + 1) * 64 * MB) % (total - sz));
                hipEventRecord(e0, st);
                if (nt) hipLaunchKernelGGL(k_read<true>, grid, blk, 0, st, pc, n16, sink); else hipLaunchKernelGGL(k_read<false>, grid, blk, 0, st, pc, n16, sink);
                hipEventRecord(e1, st); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); cold += ms;
                // warm: same region again (previous read was of kind `nt`)
                hipEventRecord(e0, st);
                if (nt) hipLaunchKernelGGL(k_read<true>, grid, blk, 0, st, pc, n16, sink); else hipLaunchKernelGGL(k_read<false>, grid, blk, 0, st, pc, n16, sink);
                hipEventRecord(e1, st); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1); warm += ms;
                // warm read with plain loads of a region first touched with plain loads, read back nt
                const u32x4 *pd = (const u32x4 *)(buf + ((size_t)(r * 3 + 2) * 64 * MB) % (total - sz));
                hipLaunchKernelGGL(k_read<false>, grid, blk, 0, st, pd, n16, sink);
                hipEventRecord(e0, st);
                hipLaunchKernelGGL(k_read<true>, grid, blk, 0, st, pd, n16, sink);
                hipEventRecord(e1, st); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1); warm_after_nt += ms;
            }
            printf("%zu MiB %s: cold %.2f us, warm(same kind) %.2f us, plain-then-nt %.2f us\n", sz / MB, nt ? "nt" : "plain",
                   cold / reps * 1e3, warm / reps * 1e3, warm_after_nt / reps * 1e3);
        }
    }
    return 0;
}

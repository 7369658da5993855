// Test helper (not product code): fills the LDS of every CU with a bit pattern, so that a kernel launched afterwards that reads
// LDS it never wrote meets that pattern instead of a friendly leftover (round-3 advisor finding on k_vad_marblenet_bf16).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o liblds_poison.so lds_poison.hip        (built by __graft_entry__.build())
#include <hip/hip_runtime.h>
__global__ void k_lds_poison(unsigned pattern, int words, unsigned *sink) {
    extern __shared__ unsigned l[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) l[i] = pattern;
    __syncthreads();
    if (l[(threadIdx.x * 2654435761u) % (unsigned)words] != pattern) *sink = 1u;      // keeps the stores alive; never true
}
extern "C" int lds_poison(int device, unsigned pattern) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    int bytes = 0, cus = 0;
    if (hipDeviceGetAttribute(&bytes, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) return -2;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return -3;
    if (hipFuncSetAttribute((const void *)k_lds_poison, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -4;
    unsigned *sink = nullptr;
    if (hipMalloc((void **)&sink, 4) != hipSuccess) return -5;
    (void)hipMemset(sink, 0, 4);
    // a workgroup with the whole LDS owns its CU: 8 rounds over the CU count reach every CU
    hipLaunchKernelGGL(k_lds_poison, dim3(8 * cus), dim3(256), bytes, 0, pattern, bytes / 4, sink);
    const hipError_t e = hipDeviceSynchronize();
    unsigned hit = 0;
    (void)hipMemcpy(&hit, sink, 4, hipMemcpyDeviceToHost);
    (void)hipFree(sink);
    return e == hipSuccess && hit == 0 ? bytes : -6;
}

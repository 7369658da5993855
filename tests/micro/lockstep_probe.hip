// micro-benchmark (round 2): does a LONG kernel on one HIP stream hold up the dependent launches of another stream?
// Stream A replays a graph of 200 dependent short kernels (256 workgroups streaming 8 MB, ~3 us); stream B meanwhile runs
// dependent kernels that spin for `long_us` each on ONE workgroup (they leave 255 CUs free).  If A's chain takes the same time
// with and without B, launches of A do not wait for B's running kernel; if A's time grows by about B's busy time, a dependent
// launch waits for everything that is running on the chip (the lock-step rounds of profiles/r2_stamps_timeline.md, section 7).
//   hipcc --offload-arch=gfx950 -O3 -o lockstep_probe lockstep_probe.hip && ./lockstep_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ __launch_bounds__(256) void k_stream(const uint4 *w, float *out) {
    const uint4 *p = w + (size_t)blockIdx.x * 2048 + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < 8; i++) { const u32x4 t = __builtin_nontemporal_load((const u32x4 *)(p + (size_t)i * 256)); acc.x ^= t[0]; acc.y += t[1]; acc.z ^= t[2]; acc.w += t[3]; }
    if (threadIdx.x < 16) out[blockIdx.x * 16 + threadIdx.x] = (float)(acc.x ^ acc.y ^ acc.z ^ acc.w);
}
__global__ void k_spin(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
int main() {
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    uint4 *w; float *o;
    hipMalloc(&w, (size_t)8 << 20 << 3); hipMemset(w, 1, (size_t)8 << 20 << 3); hipMalloc(&o, 256 * 16 * 4);
    const int N = 200;
    hipGraph_t g; hipGraphExec_t ga;
    hipStreamBeginCapture(a, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_stream, dim3(256), dim3(256), 0, a, w + (size_t)(i % 8) * ((8 << 20) / 16), o);
    hipStreamEndCapture(a, &g); hipGraphInstantiate(&ga, g, nullptr, nullptr, 0);
    for (int long_us : {0, 20, 50, 100}) {
        double best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            hipGraphLaunch(ga, a);
            if (long_us) for (int i = 0; i < 4; i++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, (unsigned long long)long_us * 100);
            hipStreamSynchronize(a);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            hipStreamSynchronize(b);
            if (us < best) best = us;
        }
        printf("stream B: 4 dependent kernels of %3d us on one workgroup -> stream A's 200-kernel chain takes %.1f us\n", long_us, best);
    }
    return 0;
}

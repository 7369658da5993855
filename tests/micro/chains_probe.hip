// micro-benchmark (round 2): how many dependent launch chains can the chip run side by side?
// K HIP streams, each replaying a hipGraph of N dependent kernels; kernels are (a) empty 256-workgroup launches, (b) the
// shape of the batch-1 layer kernels: 256 workgroups x 256 threads, each workgroup streams `kb` KiB from HBM (once-read,
// nontemporal) and writes 64 bytes, (c) the same with 36 KiB of LDS per workgroup.  Prints the time per kernel of ONE chain
// (wall / N) for K = 1, 2, 3, 4, 6, 8: if K chains cost what one costs, their dependent-launch gaps overlap.
//   hipcc --offload-arch=gfx950 -O3 -o chains_probe chains_probe.hip && ./chains_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k_empty(int *p) { if (p == (int *)1) *p = 0; }

template <int LDS_KB>
__global__ __launch_bounds__(256) void k_stream(const uint4 *w, size_t stride_wg, int kb, float *out) {
    extern __shared__ char lds[];
    const uint4 *p = w + (size_t)blockIdx.x * stride_wg + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const int n = kb * 1024 / 16 / 256;            // 16-byte loads per thread
    uint4 v[8];
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) { const u32x4 t = __builtin_nontemporal_load((const u32x4 *)(p + (size_t)(i + u) * 256)); v[u] = make_uint4(t[0], t[1], t[2], t[3]); }
#pragma unroll
        for (int u = 0; u < 8; u++) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
    }
    if (LDS_KB > 0) { ((unsigned *)lds)[threadIdx.x] = acc.x; __syncthreads(); acc.y += ((unsigned *)lds)[(threadIdx.x + 1) & 255]; }
    if (threadIdx.x < 16) out[blockIdx.x * 16 + threadIdx.x] = (float)(acc.x ^ acc.y ^ acc.z ^ acc.w);
}

int main() {
    const int KMAX = 8, N = 200;
    std::vector<hipStream_t> st(KMAX);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t wbytes = (size_t)8 << 20;                       // 8 MiB per kernel (256 workgroups x 32 KiB)
    std::vector<uint4 *> w(KMAX);
    std::vector<float *> o(KMAX);
    for (int k = 0; k < KMAX; k++) {
        hipMalloc(&w[k], wbytes * 24);                           // 24 different slices per chain (like 24 layers: no cache reuse)
        hipMemset(w[k], k + 1, wbytes * 24);
        hipMalloc(&o[k], 256 * 16 * 4);
    }
    int *d; hipMalloc(&d, 64);
    for (int mode = 0; mode < 4; mode++) {
        const char *names[] = {"empty 256-WG kernels", "256 WG x 32 KiB streamed", "256 WG x 32 KiB streamed, 36 KiB LDS", "256 WG x 8 KiB streamed"};
        std::vector<hipGraphExec_t> ex(KMAX);
        for (int k = 0; k < KMAX; k++) {
            hipGraph_t g;
            hipStreamBeginCapture(st[k], hipStreamCaptureModeThreadLocal);
            for (int i = 0; i < N; i++) {
                const uint4 *wp = w[k] + (size_t)(i % 24) * (wbytes / 16);
                if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st[k], d);
                else if (mode == 1) hipLaunchKernelGGL(k_stream<0>, dim3(256), dim3(256), 0, st[k], wp, (size_t)2048, 32, o[k]);
                else if (mode == 2) hipLaunchKernelGGL(k_stream<36>, dim3(256), dim3(256), 36 * 1024, st[k], wp, (size_t)2048, 32, o[k]);
                else hipLaunchKernelGGL(k_stream<0>, dim3(256), dim3(256), 0, st[k], wp, (size_t)2048, 8, o[k]);
            }
            hipStreamEndCapture(st[k], &g);
            hipGraphInstantiate(&ex[k], g, nullptr, nullptr, 0);
            hipGraphDestroy(g);
        }
        printf("%s:", names[mode]);
        for (int K : {1, 2, 3, 4, 6, 8}) {
            double best = 1e9;
            for (int rep = 0; rep < 8; rep++) {
                hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                for (int k = 0; k < K; k++) hipGraphLaunch(ex[k], st[k]);
                for (int k = 0; k < K; k++) hipStreamSynchronize(st[k]);
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (us < best) best = us;
            }
            printf("  K=%d %.2f us/kernel/chain (%.2f us per kernel overall)", K, best / N, best / N / K);
        }
        printf("\n");
        for (auto e : ex) hipGraphExecDestroy(e);
    }
    return 0;
}

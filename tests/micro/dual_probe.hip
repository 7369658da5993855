// Round 3: would batch 1 gain from carrying TWO in-flight steps per lane in ONE launch ("dual" kernels: grid x 2, two weight
// matrices, two activation vectors) -- 8 steps in flight on the 4 hardware queues instead of 4?  Stand-in: 4 streams, each a chain of
// dependent weight-streaming links (8 MiB of distinct bf16 weights per 256 workgroups, as flagchain_probe), one hipGraph per stream.
//   single: 4 chains x L links, 256 workgroups per link            (what the four-lane pipeline does today)
//   dual:   4 chains x L/2 links, 512 workgroups per link (2 x 8 MiB)   (the same bytes and dependent work per chain, half the links)
// Prints the wall time of the four graphs side by side and the aggregate weight bandwidth.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int N = 4096, K = 1024;
typedef unsigned short bf16_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ inline float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
// blockIdx.y selects the problem (weights W + y * wstride, vectors + y * 8192 floats)
__global__ __launch_bounds__(256) void k_link(const bf16_t *__restrict__ W, size_t wstride, const float *__restrict__ part_in, float *__restrict__ part_out) {
    __shared__ float xs[K];
    __shared__ float red[8];
    const int tid = threadIdx.x, wg = blockIdx.x, y = blockIdx.y;
    W += (size_t)y * wstride; part_in += y * 8192; part_out += y * 8192;
    const int row = tid >> 4, kq = (tid & 15) * 64;
    const u32x4 *wp = (const u32x4 *)(W + ((size_t)(wg * 16 + row) * K + kq));
    u32x4 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = wp[i];
    float x4[4];
    {
        float4 a = ((const float4 *)part_in)[tid], b = ((const float4 *)part_in)[256 + tid], c = ((const float4 *)part_in)[512 + tid], d = ((const float4 *)part_in)[768 + tid];
        x4[0] = a.x + b.x + c.x + d.x; x4[1] = a.y + b.y + c.y + d.y; x4[2] = a.z + b.z + c.z + d.z; x4[3] = a.w + b.w + c.w + d.w;
    }
    float s = x4[0] + x4[1] + x4[2] + x4[3], q = x4[0] * x4[0] + x4[1] * x4[1] + x4[2] * x4[2] + x4[3] * x4[3];
    for (int o = 32; o; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if ((tid & 63) == 0) { red[tid >> 6] = s; red[4 + (tid >> 6)] = q; }
    __syncthreads();
    s = red[0] + red[1] + red[2] + red[3]; q = red[4] + red[5] + red[6] + red[7];
    const float mean = s / K, rstd = rsqrtf(q / K - mean * mean + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; i++) xs[tid * 4 + i] = (x4[i] - mean) * rstd;
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const bf16_t *h = (const bf16_t *)&w[i];
#pragma unroll
        for (int j = 0; j < 8; j++) acc += bf2f(h[j]) * xs[kq + i * 8 + j];
    }
    for (int o = 8; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 15) == 0) part_out[(size_t)((wg * 16 + row) & 3) * 1024 + ((wg * 16 + row) >> 2)] = acc * 0.05f;
}
int main(int argc, char **argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 96, NCH = 4;
    const int extra_lds = argc > 2 ? atoi(argv[2]) : 0;          // dynamic LDS per workgroup: the engine's one-row kernels declare 36 KiB (16-row panel)
    printf("extra dynamic LDS per workgroup: %d bytes\n", extra_lds);
    const size_t wbytes = (size_t)N * K * 2;
    bf16_t *W; float *parts;
    CHK(hipMalloc(&W, wbytes * L * NCH));                       // distinct weights for every link of every chain
    CHK(hipMemset(W, 0x3c, wbytes * L * NCH));
    CHK(hipMalloc(&parts, NCH * 16 * 8192 * sizeof(float) * 2));
    CHK(hipMemset(parts, 0, NCH * 16 * 8192 * sizeof(float) * 2));
    hipStream_t st[NCH];
    for (int c = 0; c < NCH; c++) CHK(hipStreamCreateWithFlags(&st[c], hipStreamNonBlocking));
    // (chains side by side, problems per launch): the same bytes per configuration where possible
    const int cfgs[][2] = {{1, 1}, {4, 1}, {1, 2}, {4, 2}, {1, 4}, {2, 4}, {4, 4}, {1, 8}, {2, 8}, {1, 16}};
    for (auto &cf : cfgs) {
        const int lanes = cf[0], Y = cf[1];
        const int links = L / Y;                          // launches per chain; a chain touches L weight matrices in all
        hipGraphExec_t ex[NCH];
        for (int c = 0; c < lanes; c++) {
            hipGraph_t g;
            CHK(hipStreamBeginCapture(st[c], hipStreamCaptureModeThreadLocal));
            float *pa = parts + (size_t)c * 4 * 8192 * 4, *pb = pa + 16 * 8192 / 2;
            for (int k = 0; k < links; k++) {
                const bf16_t *w = W + ((size_t)c * L + (size_t)Y * k) * N * K;
                hipLaunchKernelGGL(k_link, dim3(256, Y), dim3(256), extra_lds, st[c], w, (size_t)N * K, (k & 1) ? pb : pa, (k & 1) ? pa : pb);
            }
            CHK(hipStreamEndCapture(st[c], &g)); CHK(hipGraphInstantiate(&ex[c], g, nullptr, nullptr, 0)); CHK(hipGraphDestroy(g));
        }
        double best = 1e18;
        for (int rep = 0; rep < 12; rep++) {
            for (int c = 0; c < NCH; c++) CHK(hipStreamSynchronize(st[c]));
            auto t0 = std::chrono::steady_clock::now();
            for (int c = 0; c < lanes; c++) CHK(hipGraphLaunch(ex[c], st[c]));
            for (int c = 0; c < lanes; c++) CHK(hipStreamSynchronize(st[c]));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            best = us < best ? us : best;
        }
        const double bytes = (double)lanes * links * Y * wbytes;
        printf("%d chain(s) x %2d problem(s) per launch (%2d steps in flight): %8.1f us, %.2f TB/s, %.2f us per 8 MiB problem\n", lanes, Y, lanes * Y, best,
               bytes / best * 1e-6, best / (lanes * links * Y));
        for (int c = 0; c < lanes; c++) CHK(hipGraphExecDestroy(ex[c]));
    }
    return 0;
}

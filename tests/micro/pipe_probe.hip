// micro-benchmark (round 2): the launch pattern of the pipelined step -- E lanes, each replaying a hipGraph of N dependent
// kernels per "step", lane k of step s waiting (hipStreamWaitEvent) for lane k-1 of step s, the host waiting for the last lane
// of step s-E before it enqueues step s+1 -- with the synthetic kernels of chains_probe (256 workgroups x 256 threads streaming
// 32 KiB each).  Isolates the stream / event / graph plumbing from the engine's kernels: if E = 3 scales here (time per step
// = N x per-kernel time of 3 concurrent chains) the plumbing is fine and the engine's kernels are what three chains cannot share.
//   hipcc --offload-arch=gfx950 -O3 -o pipe_probe pipe_probe.hip && ./pipe_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int LDS_KB>
__global__ __launch_bounds__(256) void k_stream(const uint4 *w, size_t stride_wg, int kb, float *out) {
    extern __shared__ char lds[];
    const uint4 *p = w + (size_t)blockIdx.x * stride_wg + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const int n = kb * 1024 / 16 / 256;
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
    const int EMAX = 4, NTOT = 192, STEPS = 300, NSLOT = EMAX + 1;
    std::vector<hipStream_t> lane(EMAX);
    for (auto &s : lane) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStream_t dec;
    hipStreamCreateWithFlags(&dec, hipStreamNonBlocking);
    const size_t wbytes = (size_t)8 << 20;
    uint4 *w;
    hipMalloc(&w, wbytes * 24);
    hipMemset(w, 1, wbytes * 24);
    std::vector<float *> o(NSLOT);
    for (auto &p : o) hipMalloc(&p, 256 * 16 * 4);
    // hostwait: 0 = dependent graphs are queued behind hipStreamWaitEvent (a barrier packet that may find the event pending);
    //           1 = the host waits for the event and queues the dependent graph only then (no cross-queue packet at all)
    for (int hostwait = 0; hostwait < 2; hostwait++)
    for (int with_dec = 0; with_dec < 2; with_dec++)
        for (int E = 1; E <= EMAX; E++) {
            // graphs: per slot, per piece
            std::vector<std::vector<hipGraphExec_t>> ex(NSLOT, std::vector<hipGraphExec_t>(E));
            std::vector<hipGraphExec_t> dex(NSLOT);
            std::vector<std::vector<hipEvent_t>> done(NSLOT, std::vector<hipEvent_t>(E));
            std::vector<hipEvent_t> ddone(NSLOT);
            for (int p = 0; p < NSLOT; p++) {
                for (int k = 0; k < E; k++) {
                    hipGraph_t g;
                    hipStreamBeginCapture(lane[0], hipStreamCaptureModeThreadLocal);
                    for (int i = NTOT * k / E; i < NTOT * (k + 1) / E; i++)
                        hipLaunchKernelGGL(k_stream<36>, dim3(256), dim3(256), 36 * 1024, lane[0], w + (size_t)(i % 24) * (wbytes / 16), (size_t)2048, 32, o[p]);
                    hipStreamEndCapture(lane[0], &g);
                    hipGraphInstantiate(&ex[p][k], g, nullptr, nullptr, 0);
                    hipGraphDestroy(g);
                    hipEventCreateWithFlags(&done[p][k], hipEventDisableTiming);
                }
                hipGraph_t g;
                hipStreamBeginCapture(dec, hipStreamCaptureModeThreadLocal);
                for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_stream<0>, dim3(64), dim3(256), 0, dec, w, (size_t)2048, 8, o[p]);
                hipStreamEndCapture(dec, &g);
                hipGraphInstantiate(&dex[p], g, nullptr, nullptr, 0);
                hipGraphDestroy(g);
                hipEventCreateWithFlags(&ddone[p], hipEventDisableTiming);
            }
            std::vector<int> stage(NSLOT, 0);
            auto step = [&](long s) {
                const int p = (int)(s % NSLOT);
                hipGraphLaunch(ex[p][0], lane[0]);
                hipEventRecord(done[p][0], lane[0]);
                stage[p] = 1;
                for (int k = 1; k <= E && k <= s; k++) {
                    const int q = (int)((s - k) % NSLOT);
                    if (stage[q] < E) {
                        const int kk = stage[q];
                        if (hostwait) hipEventSynchronize(done[q][kk - 1]); else hipStreamWaitEvent(lane[kk], done[q][kk - 1], 0);
                        hipGraphLaunch(ex[q][kk], lane[kk]);
                        hipEventRecord(done[q][kk], lane[kk]);
                        stage[q] = kk + 1;
                    } else {
                        if (with_dec) {
                            if (hostwait) hipEventSynchronize(done[q][E - 1]); else hipStreamWaitEvent(dec, done[q][E - 1], 0);
                            hipGraphLaunch(dex[q], dec);
                            hipEventRecord(ddone[q], dec);
                            hipEventSynchronize(ddone[q]);
                        } else hipEventSynchronize(done[q][E - 1]);
                        stage[q] = 0;
                    }
                }
            };
            long s = 0;
            for (; s < 20; s++) step(s);
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (; s < 20 + STEPS; s++) step(s);
            hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("%s E=%d %s: %.1f us per step of %d kernels = %.2f us per kernel\n", hostwait ? "host waits  " : "stream waits", E, with_dec ? "decode graph on its own stream" : "no decode graph          ", us / STEPS, NTOT, us / STEPS / NTOT);
            for (auto &v : ex) for (auto e : v) hipGraphExecDestroy(e);
            for (auto e : dex) hipGraphExecDestroy(e);
        }
    return 0;
}

// micro-benchmark: per-node cost of a dependent kernel chain (hipGraph vs eager) on this box
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char pad[512]; int *p; };
__global__ void k_empty(int *p) { if (p == (int *)1) *p = 0; }
__global__ void k_big(Big b) { if (b.p == (int *)1) *b.p = 0; }
__global__ void k_load(int *p) { if (*p == 12345) p[1] = 1; }
__global__ void k_wide(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0 && *p == 12345) p[1] = 1; }
template <typename F> double run(F enqueue, hipStream_t st, int n, bool graph) {
    hipGraphExec_t ex = nullptr;
    if (graph) {
        hipGraph_t g;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < n; i++) enqueue();
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
    }
    double best = 1e9;
    for (int rep = 0; rep < 20; rep++) {
        hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        if (graph) hipGraphLaunch(ex, st); else for (int i = 0; i < n; i++) enqueue();
        hipStreamSynchronize(st);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < best) best = us;
    }
    if (ex) hipGraphExecDestroy(ex);
    return best / n;
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int *d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    Big b; b.p = d;
    const int n = 400;
    for (int graph = 0; graph < 2; graph++) {
        printf("%s: empty %.2f us/node, 512B-kernarg %.2f, one-load %.2f, 256wg-one-load %.2f\n", graph ? "graph" : "eager",
               run([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, d); }, st, n, graph),
               run([&] { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, b); }, st, n, graph),
               run([&] { hipLaunchKernelGGL(k_load, dim3(1), dim3(64), 0, st, d); }, st, n, graph),
               run([&] { hipLaunchKernelGGL(k_wide, dim3(256), dim3(256), 0, st, d); }, st, n, graph));
    }
    return 0;
}

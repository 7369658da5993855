// micro-benchmark: host cost of hipGraphExecKernelNodeSetParams (could per-step values ride in kernel arguments instead of a
// descriptor block the kernels have to fetch?)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Args { int a, b, c, d; int *p; };
__global__ void k(Args x) { if (x.p == (int *)1) *x.p = x.a; }
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int *d; hipMalloc(&d, 4096);
    const int n = 200;
    hipGraph_t g; hipGraphCreate(&g, 0);
    std::vector<hipGraphNode_t> nodes(n);
    Args a{1, 2, 3, 4, d};
    void *kargs[] = {&a};
    hipKernelNodeParams kp = {};
    kp.func = (void *)k; kp.gridDim = dim3(256); kp.blockDim = dim3(256); kp.kernelParams = kargs;
    for (int i = 0; i < n; i++) hipGraphAddKernelNode(&nodes[i], g, i ? &nodes[i - 1] : nullptr, i ? 1 : 0, &kp);
    hipGraphExec_t ex; hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    hipGraphLaunch(ex, st); hipStreamSynchronize(st);
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 48; i++) { a.a = rep * 100 + i; hipError_t e = hipGraphExecKernelNodeSetParams(ex, nodes[i * 4], &kp); if (e != hipSuccess) { printf("error %s\n", hipGetErrorString(e)); return 1; } }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        auto t1 = std::chrono::steady_clock::now();
        hipGraphLaunch(ex, st);
        double lus = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
        hipStreamSynchronize(st);
        double tot = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
        printf("48 hipGraphExecKernelNodeSetParams: %.1f us (%.2f us each); following hipGraphLaunch %.1f us, graph of %d nodes done after %.1f us\n", us, us / 48, lus, n, tot);
    }
    return 0;
}

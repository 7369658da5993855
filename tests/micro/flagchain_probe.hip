// can a chain of dependent weight-streaming kernels run faster when the dependency is carried by a
// release/acquire flag instead of the queue's barrier bit?  With the barrier bit, link k+1 cannot issue a single load before
// link k has retired (boundary 1.6-2.2 us) -- although 2.2-2.4 us of its life is the arrival of its weight slice, which does
// not depend on link k.  Variants (200 links, 8 MB of distinct bf16 weights per link, 256 workgroups x 256 threads,
// 4 split-K partial vectors handed from link to link, like k_fused_skinny<PRO_LN, 1> -> <PRO_PLAIN, 1>):
//   barrier/graph   one stream, plain launches captured into ONE hipGraph           (what ships)
//   barrier/eager   one stream, plain eager launches
//   anyorder        one stream, hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch), dependency by flag
//                   (hip_ext.h: "not supported on AMD GFX9xx boards" -- measured anyway)
//   flag2/graph     links alternate between TWO streams (two hardware queues), each stream's links captured into its own
//                   graph; link k+1 (other queue) is dispatched beside link k, requests its weights, then spins on link k's flag
//   flag2/eager     the same with eager launches
// A link's workgroup: (1) request its 32 KiB weight slice (registers), (2) [flag variants] wait for the previous link's flag,
// (3) read the 4 partial vectors of the previous link (16 KiB), reduce + normalise, (4) 16 x 256 dot products, (5) write its
// partial, (6) [flag variants] release-increment its flag.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int N = 4096, K = 1024, SPLITS = 1;      // W1-shaped link: 4096 x 1024 bf16 = 8 MiB, grid 256 x 1
constexpr int WG = 256;
typedef unsigned short bf16_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ inline float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }

template <int FLAG>
__global__ __launch_bounds__(256) void k_link(const bf16_t *__restrict__ W, const float *__restrict__ part_in, float *__restrict__ part_out,
                                              const int *flag_prev, int expect_prev, int *flag_mine, unsigned long long *stamps, int expect_mine = 0) {
    __shared__ float xs[K];
    __shared__ float red[8];
    const int tid = threadIdx.x, wg = blockIdx.x;
    // (1) weights: workgroup wg owns output rows [16 wg, 16 wg + 16): 16 x 1024 bf16 = 32 KiB, thread t: row t / 16, k in [64 (t % 16), +64)
    const int row = tid >> 4, kq = (tid & 15) * 64;
    const u32x4 *wp = (const u32x4 *)(W + ((size_t)(wg * 16 + row) * K + kq));
    u32x4 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = __builtin_nontemporal_load(wp + i);
    if (stamps && tid == 0) stamps[wg * 4 + 0] = __builtin_amdgcn_s_memrealtime();
    // (2) dependency
    if (FLAG) {
        if (FLAG == 3) {
            // one flag word per producer workgroup (no atomics): every thread watches one word, a coalesced 1 KiB coherent load per poll
            const int *fw = flag_prev + tid;
            for (;;) {
                int v;
                asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(fw) : "memory");
                if (__syncthreads_and(v >= expect_prev)) break;
                __builtin_amdgcn_s_sleep(1);
            }
        } else {
        if (tid == 0) {
            while (__hip_atomic_load(flag_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect_prev) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        }
        if (FLAG == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);       // agent scope: invalidates the non-coherent lines of this XCD's L2
    }
    if (stamps && tid == 0) stamps[wg * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    // (3) inputs: 4 partial vectors of 1024 floats -> x, normalised (a LayerNorm-like reduction: two block sums)
    float x4[4];
    {
        float4 a, b, c, d;
        if (FLAG >= 2) {            // FLAG 2, 3: no cache maintenance at all -- the hand-over bytes bypass the caches in both directions
            const float4 *pi = (const float4 *)part_in;
            asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\tglobal_load_dwordx4 %2, %6, off sc0 sc1\n\t"
                         "global_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(pi + tid), "v"(pi + 256 + tid), "v"(pi + 512 + tid), "v"(pi + 768 + tid) : "memory");
        } else {
            a = ((const float4 *)part_in)[tid]; b = ((const float4 *)part_in)[256 + tid]; c = ((const float4 *)part_in)[512 + tid]; d = ((const float4 *)part_in)[768 + tid];
        }
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
    // (4) 64 products per thread, reduce over the 16 threads of a row
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const bf16_t *h = (const bf16_t *)&w[i];
#pragma unroll
        for (int j = 0; j < 8; j++) acc += bf2f(h[j]) * xs[kq + i * 8 + j];
    }
    for (int o = 8; o; o >>= 1) acc += __shfl_xor(acc, o);
    // (5) the link's output: 4096 values -> folded into 4 "partials" of 1024 for the next link (keeps the hand-over at 16 KiB)
    if ((tid & 15) == 0) {
        float *po = part_out + (size_t)((wg * 16 + row) & 3) * 1024 + ((wg * 16 + row) >> 2);
        const float v = acc * 0.05f;
        if (FLAG >= 2) asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(po), "v"(v) : "memory");
        else *po = v;
    }
    if (stamps && tid == 0) stamps[wg * 4 + 2] = __builtin_amdgcn_s_memrealtime();
    // (6) release
    if (FLAG) {
        __syncthreads();
        if (FLAG == 3) {
            if (tid == 0) { int *fm = flag_mine + wg; const int v = expect_mine; asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(fm), "v"(v) : "memory"); }
        } else if (tid == 0) __hip_atomic_fetch_add(flag_mine, 1, FLAG == 1 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct Chain {
    int links; bf16_t *W; float *part[2]; int *flags; hipStream_t st[2];
};

static void enqueue_barrier(const Chain &c, int k, hipStream_t st) {
    hipLaunchKernelGGL(k_link<0>, dim3(WG), dim3(256), 0, st, c.W + (size_t)k * N * K, c.part[k & 1], c.part[(k & 1) ^ 1], nullptr, 0, nullptr, nullptr);
}
template <int MODE> static void enqueue_flag(const Chain &c, int k, hipStream_t st, int epoch, bool anyorder) {
    // flags[k] counts completed workgroups of link k over all epochs: link k of epoch e may start when flags[k-1] >= WG * (e + 1);
    // link 0 waits for the LAST link of the previous epoch (flags[links-1] >= WG * e): the chain is a ring, like consecutive steps
    const int stride = MODE == 3 ? 256 : 1, unit = MODE == 3 ? 1 : WG;         // mode 3: every flag word holds the epoch + 1 of its workgroup's last completion
    const int *fp = c.flags + (size_t)(k == 0 ? c.links - 1 : k - 1) * stride;
    const int expect = k == 0 ? unit * epoch : unit * (epoch + 1);
    const int mine = epoch + 1;
    if (anyorder)
        hipExtLaunchKernelGGL(k_link<MODE>, dim3(WG), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, c.W + (size_t)k * N * K, c.part[k & 1], c.part[(k & 1) ^ 1], fp, expect,
                              c.flags + (size_t)k * stride, (unsigned long long *)nullptr, mine);
    else
        hipLaunchKernelGGL(k_link<MODE>, dim3(WG), dim3(256), 0, st, c.W + (size_t)k * N * K, c.part[k & 1], c.part[(k & 1) ^ 1], fp, expect, c.flags + (size_t)k * stride, nullptr, mine);
}

int main(int argc, char **argv) {
    const int links = argc > 1 ? atoi(argv[1]) : 200, reps = 12;
    Chain c; c.links = links;
    CHK(hipMalloc(&c.W, (size_t)links * N * K * 2));
    CHK(hipMalloc(&c.part[0], 4096 * 4)); CHK(hipMalloc(&c.part[1], 4096 * 4)); CHK(hipMalloc(&c.flags, (size_t)links * 256 * 4));
    {
        std::vector<bf16_t> h((size_t)N * K);
        unsigned r = 12345;
        for (auto &v : h) { r = r * 1664525u + 1013904223u; v = (bf16_t)(0x3c00 + ((r >> 20) & 0x3ff) - ((r >> 31) ? 0x8000 : 0) ); }
        for (int k = 0; k < links; k++) CHK(hipMemcpy(c.W + (size_t)k * N * K, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        std::vector<float> x(4096, 0.25f);
        for (int i = 0; i < 4096; i++) x[i] = 0.01f * (i % 97) - 0.3f;
        CHK(hipMemcpy(c.part[0], x.data(), 4096 * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(c.part[1], x.data(), 4096 * 4, hipMemcpyHostToDevice));
    }
    CHK(hipStreamCreateWithFlags(&c.st[0], hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&c.st[1], hipStreamNonBlocking));
    auto sync = [&] { CHK(hipStreamSynchronize(c.st[0])); CHK(hipStreamSynchronize(c.st[1])); };
    auto timeit = [&](const char *name, auto body) {
        double best = 1e18, sum = 0;
        for (int r = 0; r < reps; r++) {
            sync();
            auto t0 = std::chrono::steady_clock::now();
            body(r);
            sync();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            best = us < best ? us : best; if (r >= 2) sum += us;
        }
        printf("%-16s %7.3f us per link (best), %7.3f (mean of %d)   chain of %d: %.1f us\n", name, best / links, sum / (reps - 2) / links, reps - 2, links, best);
        fflush(stdout);
    };
    // --- barrier / graph
    {
        hipGraph_t g; hipGraphExec_t ex;
        CHK(hipStreamBeginCapture(c.st[0], hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < links; k++) enqueue_barrier(c, k, c.st[0]);
        CHK(hipStreamEndCapture(c.st[0], &g)); CHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        timeit("barrier/graph", [&](int) { CHK(hipGraphLaunch(ex, c.st[0])); });
        CHK(hipGraphExecDestroy(ex)); CHK(hipGraphDestroy(g));
    }
    timeit("barrier/eager", [&](int) { for (int k = 0; k < links; k++) enqueue_barrier(c, k, c.st[0]); });
    float ref[8];
    {
        std::vector<float> x(4096);
        for (int i = 0; i < 4096; i++) x[i] = 0.01f * (i % 97) - 0.3f;
        CHK(hipMemcpy(c.part[0], x.data(), 4096 * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(c.part[1], x.data(), 4096 * 4, hipMemcpyHostToDevice));
        for (int k = 0; k < links; k++) enqueue_barrier(c, k, c.st[0]);
        sync();
        CHK(hipMemcpy(ref, c.part[links & 1], sizeof(ref), hipMemcpyDeviceToHost));
    }
    auto reset = [&] {
        std::vector<float> x(4096);
        for (int i = 0; i < 4096; i++) x[i] = 0.01f * (i % 97) - 0.3f;
        CHK(hipMemcpy(c.part[0], x.data(), 4096 * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(c.part[1], x.data(), 4096 * 4, hipMemcpyHostToDevice));
        CHK(hipMemset(c.flags, 0, (size_t)links * 256 * 4));
    };
    auto variants = [&](auto mode_tag, const char *suffix) {
        constexpr int MODE = decltype(mode_tag)::value;
        char name[64];
        if (!getenv("SKIP_ANYORDER")) {
            reset();
            int epoch = 0;
            snprintf(name, sizeof(name), "anyorder%s", suffix);
            timeit(name, [&](int) { for (int k = 0; k < links; k++) enqueue_flag<MODE>(c, k, c.st[0], epoch, true); epoch++; });
        }
        {
            reset();
            int epoch = 0;
            snprintf(name, sizeof(name), "flag1q/eager%s", suffix);          // one queue, barrier bit AND flag: the cost of the protocol alone
            timeit(name, [&](int) { for (int k = 0; k < links; k++) enqueue_flag<MODE>(c, k, c.st[0], epoch, false); epoch++; });
        }
        {
            reset();
            int epoch = 0;
            snprintf(name, sizeof(name), "flag2q/eager%s", suffix);
            timeit(name, [&](int) { for (int k = 0; k < links; k++) enqueue_flag<MODE>(c, k, c.st[k & 1], epoch, false); epoch++; });
        }
        {
            hipGraph_t g[2]; hipGraphExec_t ex[2];
            for (int q = 0; q < 2; q++) {
                CHK(hipStreamBeginCapture(c.st[q], hipStreamCaptureModeThreadLocal));
                for (int k = q; k < links; k += 2) enqueue_flag<MODE>(c, k, c.st[q], 0, false);
                CHK(hipStreamEndCapture(c.st[q], &g[q])); CHK(hipGraphInstantiate(&ex[q], g[q], nullptr, nullptr, 0));
            }
            snprintf(name, sizeof(name), "flag2q/graph%s", suffix);
            timeit(name, [&](int) {
                reset();
                CHK(hipGraphLaunch(ex[0], c.st[0])); CHK(hipGraphLaunch(ex[1], c.st[1]));
            });
            for (int q = 0; q < 2; q++) { CHK(hipGraphExecDestroy(ex[q])); CHK(hipGraphDestroy(g[q])); }
        }
    };
    variants(std::integral_constant<int, 1>{}, " (fences)");
    variants(std::integral_constant<int, 2>{}, " (sc0sc1)");
    variants(std::integral_constant<int, 3>{}, " (sc0sc1, flag words)");
    float got[8];
    CHK(hipMemcpy(got, c.part[links & 1], sizeof(got), hipMemcpyDeviceToHost));
    int same = 1;
    for (int i = 0; i < 8; i++) same &= ref[i] == got[i];
    printf("flag chain result %s the barrier chain's (%.6g %.6g | %.6g %.6g)\n", same ? "==" : "!=", ref[0], ref[1], got[0], got[1]);
    return same ? 0 : 2;
}

// Round 3 probe: ONE launch per step in which every XCD runs its own stage of the layer pipeline ("XCD teams").
// Today a batch-1 step is 192 dependent launches per lane, four lanes side by side: 2.9 TB/s of weights (0.42 ms per step),
// grouped launches (2 chains x 4 problems) 3.5 TB/s in the engine.  A kernel boundary drains the chip's memory pipeline for its
// chain; the lanes fill the holes with other steps' kernels.  Candidate: 8 steps in flight as 8 STAGES (3 layers = 24 phases each),
// stage s = the workgroups that landed on XCD s (s_getreg HW_REG_XCC_ID), all 8 stages in ONE launch.  A phase's output vector
// is handed over INSIDE the XCD (plain stores -> vmcnt(0) -> one atomic add per workgroup; the consumers poll with sc1 loads), the
// next phase's first weight slice is requested BEFORE the wait (weights never depend on the input), so the HBM stream does not stop
// at a dependency.  Stand-in link = dual_probe's (8 MiB of bf16 weights per phase, LayerNorm-like prologue over 4 partial vectors,
// 256 column slices of 16 rows).  Every phase has buffers of its own: no line is read by a CU before it was written in this launch.
//   ./team_probe [phases per stage = 24] [workgroups per XCD = 96] [sc1 loads of handed-over vectors 0|1]
// Prints: the launch-chain baselines (4 chains x 1 problem, 2 chains x 4 problems) and the team launch: us per step-equivalent
// (8 x phases links), TB/s, and whether the team launch's final vectors equal those of the launch chains (stale reads would not).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int N = 4096, K = 1024, NXCD = 8;
typedef unsigned short bf16_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ inline float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }

template <bool SC1>
__device__ __forceinline__ f32x4 load_vec(const float *p) {
    f32x4 v;
    if (SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else v = *(const f32x4 *)p;
    return v;
}

// one 16-row column slice `vb` of one link: out = W[vb*16 .. +16][:] . LN(sum of 4 partial vectors); w[] already requested
template <bool SC1>
__device__ __forceinline__ void link_compute(const u32x4 (&w)[8], const float *__restrict__ part_in, float *__restrict__ part_out, int vb, float *xs, float *red) {
    const int tid = threadIdx.x, row = tid >> 4, kq = (tid & 15) * 64;
    float x4[4];
    {
        const f32x4 a = load_vec<SC1>(part_in + tid * 4), b = load_vec<SC1>(part_in + 1024 + tid * 4), c = load_vec<SC1>(part_in + 2048 + tid * 4),
                    d = load_vec<SC1>(part_in + 3072 + tid * 4);
        for (int i = 0; i < 4; i++) x4[i] = a[i] + b[i] + c[i] + d[i];
    }
    float s = x4[0] + x4[1] + x4[2] + x4[3], q = x4[0] * x4[0] + x4[1] * x4[1] + x4[2] * x4[2] + x4[3] * x4[3];
    for (int o = 32; o; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    __syncthreads();                                   // xs / red of the previous slice are no longer read
    if ((tid & 63) == 0) { red[tid >> 6] = s; red[4 + (tid >> 6)] = q; }
    __syncthreads();
    s = red[0] + red[1] + red[2] + red[3]; q = red[4] + red[5] + red[6] + red[7];
    const float mean = s / K, rstd = rsqrtf(q / K - mean * mean + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; i++) xs[tid * 4 + i] = (x4[i] - mean) * rstd + 0.25f;
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const bf16_t *h = (const bf16_t *)&w[i];
#pragma unroll
        for (int j = 0; j < 8; j++) acc += bf2f(h[j]) * xs[kq + i * 8 + j];
    }
    for (int o = 8; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 15) == 0) part_out[(size_t)((vb * 16 + row) & 3) * 1024 + ((vb * 16 + row) >> 2)] = acc * 0.05f;
}
__device__ __forceinline__ void link_request(u32x4 (&w)[8], const bf16_t *__restrict__ W, int vb) {
    const int tid = threadIdx.x, row = tid >> 4, kq = (tid & 15) * 64;
    const u32x4 *wp = (const u32x4 *)(W + ((size_t)(vb * 16 + row) * K + kq));
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = __builtin_nontemporal_load(wp + i);
}

// the launch-chain form: blockIdx.y = problem
__global__ __launch_bounds__(256) void k_link(const bf16_t *__restrict__ W, size_t wstride, const float *__restrict__ part_in, float *__restrict__ part_out, size_t pstride) {
    __shared__ float xs[K];
    __shared__ float red[8];
    const int y = blockIdx.y;
    u32x4 w[8];
    link_request(w, W + (size_t)y * wstride, blockIdx.x);
    link_compute<false>(w, part_in + (size_t)y * pstride, part_out + (size_t)y * pstride, blockIdx.x, xs, red);
}

struct TeamCtl {
    unsigned team_count[NXCD];     // workgroups that landed on every XCD
    unsigned arrived;              // all workgroups of the launch
    unsigned abort_flag;
    unsigned pad[6];
    unsigned done[NXCD][64];       // per stage: workgroups that have finished phase p (own cache lines per stage)
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// spins until *p >= want; false when the launch was aborted (a workgroup waited too long: nothing in this kernel may hang the box)
template <int SLEEP>
__device__ __forceinline__ bool wait_for(const unsigned *p, unsigned want, unsigned *abort_flag) {
    for (unsigned it = 0;; it++) {
        if (ld_sc1(p) >= want) return true;
        if ((it & 63) == 63 && ld_sc1(abort_flag)) return false;
        if (it > (1u << 20)) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        __builtin_amdgcn_s_sleep(SLEEP);
    }
}

// parts: [stage][phase + 1][4096] floats (buffer p = input of phase p, buffer phases = the stage's output); W: [stage][phase][N][K]
template <bool SC1, int SLEEP, bool DEEP>
__global__ __launch_bounds__(256) void k_team(const bf16_t *__restrict__ W, float *__restrict__ parts, TeamCtl *ctl, int phases) {
    __shared__ float xs[K];
    __shared__ float red[8];
    __shared__ unsigned sh[4];
    const int tid = threadIdx.x;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;         // HW_REG_XCC_ID[3:0]
    if (tid == 0) {
        sh[0] = __hip_atomic_fetch_add(&ctl->team_count[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh[2] = wait_for<8>(&ctl->arrived, gridDim.x, &ctl->abort_flag) ? 1u : 0u;   // every workgroup is resident and has its rank
        sh[1] = ld_sc1(&ctl->team_count[xcc]);
    }
    __syncthreads();
    if (!sh[2]) return;
    const int rank = sh[0], n = sh[1], stage = xcc;
    const bf16_t *Ws = W + (size_t)stage * phases * N * K;
    float *ps = parts + (size_t)stage * (phases + 1) * 4096;
    // slices of this workgroup in execution order over the whole stage: (p, vb) with vb = rank, rank + n, ...; DEEP keeps TWO slices'
    // weights in flight (the second one possibly of the next phase), otherwise one
    u32x4 w[8], w2[8];
    const int per_phase = rank < 256 ? (256 - rank + n - 1) / n : 0;
    auto request = [&](u32x4 (&dst)[8], int idx) {           // idx-th slice of the stage
        const int p = idx / per_phase, k = idx - p * per_phase;
        link_request(dst, Ws + (size_t)p * N * K, rank + k * n);
    };
    const int total = per_phase * phases;
    if (total > 0) request(w, 0);
    if (DEEP && total > 1) request(w2, 1);
    int idx = 0;
    for (int p = 0; p < phases; p++) {
        if (p > 0 && SLEEP >= 0) {                      // the phase's input: every workgroup of the team has published its slices of phase p - 1
            if (tid == 0) sh[2] = wait_for<(SLEEP < 0 ? 0 : SLEEP)>(&ctl->done[stage][p - 1], n, &ctl->abort_flag) ? 1u : 0u;
            __syncthreads();
            if (!sh[2]) return;
        }
        const float *pin = ps + (size_t)p * 4096;
        float *pout = ps + (size_t)(p + 1) * 4096;
        for (int k = 0; k < per_phase; k++, idx++) {
            if (DEEP) {
                if (idx & 1) { link_compute<SC1>(w2, pin, pout, rank + k * n, xs, red); if (idx + 2 < total) request(w2, idx + 2); }
                else { link_compute<SC1>(w, pin, pout, rank + k * n, xs, red); if (idx + 2 < total) request(w, idx + 2); }
            } else {
                link_compute<SC1>(w, pin, pout, rank + k * n, xs, red);
                if (idx + 1 < total) request(w, idx + 1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have left the CU (the weight loads too: they are consumed after the wait anyway)
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&ctl->done[stage][p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char **argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 24;
    const int T = argc > 2 ? atoi(argv[2]) : 96;
    const int sc1 = argc > 3 ? atoi(argv[3]) : 0;
    const int variant = argc > 4 ? atoi(argv[4]) : 0;     // 0: poll sleep 2, one slice in flight; 1: sleep 32; 2: two slices in flight; 3: both; 4: sleep 127 + two slices
    if (P > 64 || P < 1 || T < 1) { fprintf(stderr, "phases 1..64\n"); return 1; }
    const size_t wbytes = (size_t)N * K * 2;
    bf16_t *W; float *parts, *parts_ref; TeamCtl *ctl;
    CHK(hipMalloc(&W, wbytes * P * NXCD));
    {   // weights: a few distinct bf16 values so that a slice read from the wrong place or a stale vector changes the result
        std::vector<bf16_t> h((size_t)N * K);
        for (int l = 0; l < P * NXCD; l++) {
            for (size_t i = 0; i < h.size(); i++) h[i] = (bf16_t)(0x3c00 + ((i * 2654435761u + l * 40503u) >> 27 & 0x3f));
            CHK(hipMemcpy(W + (size_t)l * N * K, h.data(), wbytes, hipMemcpyHostToDevice));
        }
    }
    const size_t pfloats = (size_t)NXCD * (P + 1) * 4096;
    CHK(hipMalloc(&parts, pfloats * 4)); CHK(hipMalloc(&parts_ref, pfloats * 4)); CHK(hipMalloc(&ctl, sizeof(TeamCtl)));
    std::vector<float> seed(pfloats, 0.f), got(pfloats), ref(pfloats);
    auto set_seed = [&](float scale) {
        for (int s = 0; s < NXCD; s++)
            for (int i = 0; i < 4096; i++) seed[(size_t)s * (P + 1) * 4096 + i] = scale * (float)((i * 37 + s * 11) % 101 - 50) / 50.f;
    };
    hipStream_t st[4];
    for (auto &s : st) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

    // ---- baselines: launch chains over the same weights (stage s = chain problem) ----
    auto chains = [&](int lanes, int Y, float *pbuf) {      // lanes x Y = 8 stages; every chain runs P launches of Y problems
        hipGraphExec_t ex[4];
        for (int c = 0; c < lanes; c++) {
            hipGraph_t g;
            CHK(hipStreamBeginCapture(st[c], hipStreamCaptureModeThreadLocal));
            for (int p = 0; p < P; p++)
                hipLaunchKernelGGL(k_link, dim3(256, Y), dim3(256), 0, st[c], W + ((size_t)(c * Y) * P + p) * N * K, (size_t)P * N * K,
                                   pbuf + ((size_t)(c * Y) * (P + 1) + p) * 4096, pbuf + ((size_t)(c * Y) * (P + 1) + p + 1) * 4096, (size_t)(P + 1) * 4096);
            CHK(hipStreamEndCapture(st[c], &g)); CHK(hipGraphInstantiate(&ex[c], g, nullptr, nullptr, 0)); CHK(hipGraphDestroy(g));
        }
        double best = 1e18;
        for (int rep = 0; rep < 10; rep++) {
            for (auto &s : st) CHK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int c = 0; c < lanes; c++) CHK(hipGraphLaunch(ex[c], st[c]));
            for (int c = 0; c < lanes; c++) CHK(hipStreamSynchronize(st[c]));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            best = us < best ? us : best;
        }
        for (int c = 0; c < lanes; c++) CHK(hipGraphExecDestroy(ex[c]));
        return best;
    };
    const double bytes = (double)NXCD * P * wbytes;
    set_seed(1.f);
    CHK(hipMemcpy(parts_ref, seed.data(), pfloats * 4, hipMemcpyHostToDevice));
    if (NXCD % 4 == 0) {
        const double a = chains(4, 2, parts_ref);
        printf("launch chains 4 x 2 problems (8 stages x %d phases): %8.1f us, %.2f TB/s, %.2f us per link\n", P, a, bytes / a * 1e-6, a / (NXCD * P));
        const double b = chains(2, 4, parts_ref);
        printf("launch chains 2 x 4 problems (8 stages x %d phases): %8.1f us, %.2f TB/s, %.2f us per link\n", P, b, bytes / b * 1e-6, b / (NXCD * P));
    }

    // ---- the team launch ----
    int bad_runs = 0;
    double best = 1e18;
    unsigned counts[NXCD] = {0};
    for (int rep = 0; rep < 12; rep++) {
        const float scale = 1.f + 0.125f * rep;                       // new values every launch: a stale line of the previous launch would show
        set_seed(scale);
        CHK(hipMemcpy(parts_ref, seed.data(), pfloats * 4, hipMemcpyHostToDevice));
        chains(2, 4, parts_ref);
        CHK(hipMemcpy(ref.data(), parts_ref, pfloats * 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(parts, seed.data(), pfloats * 4, hipMemcpyHostToDevice));
        CHK(hipMemset(ctl, 0, sizeof(TeamCtl)));
        CHK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
#define TEAM(S, SL, D) hipLaunchKernelGGL((k_team<S, SL, D>), dim3(NXCD * T), dim3(256), 0, st[0], W, parts, ctl, P)
        if (sc1) TEAM(true, 8, true);
        else if (variant == 0) TEAM(false, 2, false);
        else if (variant == 1) TEAM(false, 32, false);
        else if (variant == 2) TEAM(false, 2, true);
        else if (variant == 3) TEAM(false, 32, true);
        else if (variant == 4) TEAM(false, 127, true);
        else if (variant == 5) TEAM(false, -1, false);      // NO waits: the loop's streaming rate (results are wrong by construction)
        else TEAM(false, -1, true);
        CHK(hipStreamSynchronize(st[0]));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        TeamCtl h;
        CHK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
        if (h.abort_flag) { printf("team launch ABORTED (rep %d): arrived %u of %d, teams", rep, h.arrived, NXCD * T); for (unsigned c : h.team_count) printf(" %u", c); printf("\n"); return 2; }
        memcpy(counts, h.team_count, sizeof(counts));
        CHK(hipMemcpy(got.data(), parts, pfloats * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < pfloats; i++) bad += memcmp(&got[i], &ref[i], 4) != 0;
        if (bad && variant < 5) { bad_runs++; printf("rep %d: %zu of %zu floats differ from the launch chains' (stale or misplaced reads)\n", rep, bad, pfloats); }
        if (rep >= 2) best = us < best ? us : best;
    }
    printf("XCD teams, ONE launch, variant %d (%d workgroups per XCD, %s loads of handed-over vectors): %8.1f us, %.2f TB/s, %.2f us per link, %d of 12 runs differ; teams",
           variant, T, sc1 ? "sc1" : "plain", best, bytes / best * 1e-6, best / (NXCD * P), bad_runs);
    for (unsigned c : counts) printf(" %u", c);
    printf("\n");
    return bad_runs ? 3 : 0;
}

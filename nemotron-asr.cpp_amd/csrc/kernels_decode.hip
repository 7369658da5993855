// kernels_decode.hip -- device-resident batched RNN-T greedy decode (stage a-12..a-14).
//
// Reference: src/nemo-stream.cpp:840-930 runs, per symbol, 4 host->device copies, one graph
// launch and 1-3 device->host copies for ONE stream, and evaluates LSTMx2 + joint once per
// (frame, symbol).  Here the loop lives on the device for all B streams of a step and is
// re-scheduled around one fact of the reference's rules: a blank leaves the decoder state
// untouched (:908-911), so the prediction network's output only changes when a symbol is emitted.
//   * per stream the LSTM candidate (h', c') and g = joint.pred(h1') + b_pred are CACHED (they persist
//     across steps) and recomputed only for streams whose committed state changed ("dirty");
//   * one iteration evaluates the joint for EVERY frame a stream still has to decode against that
//     cached g -- all (stream, frame) rows of the batch in one f32-MFMA GEMM with the arg-max in the
//     epilogue -- and the commit kernel walks each stream's frames in order up to its first
//     non-blank: the blanks before it are final (state unchanged), the symbol is emitted there
//     (prev_token = best, commit h'/c' :921-926, at most 10 symbols per frame :849), and the frames
//     after it are re-evaluated in the next iteration with the new g.
// A step therefore needs (max symbols emitted by one stream) + 1 iterations instead of
// (frames + symbols); every (frame, state) pair the reference evaluates is evaluated with the same
// inputs, so tokens, the iteration statistic and the committed state are identical.  Arg-max
// keeps the first maximum (:899-906).
//
// All arithmetic is f32 (the reference keeps decoder/joint weights F32 in every GGUF flavour):
// the mat-vec stages are batched over rows and run on the f32-input MFMA
// (v_mfma_f32_16x16x4_f32, bit-exact fmaf chains).  Weights are pre-packed at upload into
// MFMA A-fragment order -- tile (nt, kg) = 16 rows x 16 k is 1 KiB, lane l = q*16 + r holds
// W[row(nt, r)][kg*16 + 4q .. +4) as one float4 = its operand for 4 consecutive MFMAs -- so a
// wave-load is one contiguous 1 KiB read.  The row vectors are the B operand: lane (q, j)
// loads x[row j][kg*16 + 4q .. +4), contiguous as well.  For the LSTM the 16 rows of a tile
// are ordered (unit u, gate g) = (r/4, r%4), which lands the four gates i,f,g,o of one hidden
// unit in the four accumulator registers of one lane: the cell update needs no data movement.
#include "nasr_internal.h"

namespace nasr {

typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int KG640 = HID / 16;     // 40 k-groups of 16
constexpr int MT_MAX = 4;           // m-tiles (16 rows each) per pass

// ---- work lists: streams to recompute, (stream, frame) rows to evaluate ---------------------------
// One 256-thread block.  Order inside the lists is arbitrary (shared-memory tickets); every output
// element is an independent accumulation chain, so the order cannot change a result.
__device__ __forceinline__ void build_lists(const DecParams &p, int *sh) {
    __syncthreads();
    if (threadIdx.x < 3) sh[threadIdx.x] = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < p.B; b += 256) {
        const DecCtrl ct = p.ctrl[p.rows[b].slot];
        if (!ct.active) continue;
        atomicAdd(&sh[0], 1);
        if (ct.dirty) p.dlist[atomicAdd(&sh[1], 1)] = b;
        const int n = ct.n_frames - ct.t;
        const int base = atomicAdd(&sh[2], n);
        for (int i = 0; i < n; i++) p.rowmap[base + i] = ((unsigned)(ct.t + i) << 16) | (unsigned)b;
    }
    __syncthreads();
    if (threadIdx.x == 0) { *p.n_active = sh[0]; *p.n_dirty = sh[1]; *p.n_rows = sh[2]; }
}

__global__ __launch_bounds__(256) void k_dec_begin(DecParams p) {
    __shared__ int sh[4];
    for (int b = threadIdx.x; b < p.B; b += 256) {
        const RowDesc rd = p.rows[b];
        DecCtrl *ct = &p.ctrl[rd.slot];
        ct->t = 0;
        ct->n_frames = rd.n_dec;
        ct->symbols = 0;
        ct->row = b;
        ct->active = rd.n_dec > 0 ? 1 : 0;
        ct->frame0 = ct->frame_next;
        ct->frame_next += rd.n_dec;
    }
    for (int i = threadIdx.x; i < p.B * p.T; i += 256) p.key[i] = 0ull;
    __threadfence_block();
    build_lists(p, sh);
}

// acc[mt] += W_tile(nt, kg range) . X   for MT m-tiles; xrow[mt] = this lane's row vector.
// JOINT: the operand is relu(xrow + grow) formed on the fly (src/nemo-ggml.cpp:1210-1217).
// The k-groups are processed in blocks of 5 with every load of the block (weights + row vectors)
// issued before its MFMAs, so a wave has ~25 independent 16-byte loads in flight instead of one.
template <int MT, bool JOINT>
__device__ __forceinline__ void mfma_range(const float4 *wt, int kg0, int kg1, const float *const *xrow,
                                           const float *const *grow, int q, f32x4 *acc) {
    constexpr int KB = JOINT ? (MT > 2 ? 2 : 5) : 5;
    auto fetch = [&](int mt, int kg) -> float4 {
        float4 x = *(const float4 *)(xrow[mt] + kg * 16 + q * 4);
        if (JOINT) {
            const float4 g = *(const float4 *)(grow[mt] + kg * 16 + q * 4);
            x = make_float4(fmaxf(x.x + g.x, 0.0f), fmaxf(x.y + g.y, 0.0f), fmaxf(x.z + g.z, 0.0f), fmaxf(x.w + g.w, 0.0f));
        }
        return x;
    };
    int kg = kg0;
    for (; kg + KB <= kg1; kg += KB) {
        float4 w[KB], x[KB][MT];
#pragma unroll
        for (int u = 0; u < KB; u++) {
            w[u] = wt[(size_t)(kg + u) * 64];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) x[u][mt] = fetch(mt, kg + u);
        }
#pragma unroll
        for (int u = 0; u < KB; u++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].x, x[u][mt].x, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].y, x[u][mt].y, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].z, x[u][mt].z, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].w, x[u][mt].w, acc[mt], 0, 0, 0);
            }
    }
    for (; kg < kg1; kg++) {
        const float4 w = wt[(size_t)kg * 64];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const float4 x = fetch(mt, kg);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.z, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.w, acc[mt], 0, 0, 0);
        }
    }
}

// ---- LSTM layer L over the dirty rows: grid = 160 (4 hidden units per workgroup), 4 waves split K ----
template <int L, int MT>
__device__ __forceinline__ void lstm_pass(const DecParams &p, int i0, int nd, float (*red)[2][MT_MAX][64][4]) {
    const int nt = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, r = lane & 15;
    const float4 *w_ih = (const float4 *)p.w_ih[L] + (size_t)nt * KG640 * 64 + lane;
    const float4 *w_hh = (const float4 *)p.w_hh[L] + (size_t)nt * KG640 * 64 + lane;
    const int kg0 = wave * (KG640 / 4), kg1 = kg0 + KG640 / 4;
    const float *xr[MT], *hr[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int i = i0 + mt * 16 + r;
        if (i >= nd) i = i0;                                       // valid memory, result discarded
        const int slot = p.rows[p.dlist[i]].slot;
        const DecCtrl ct = p.ctrl[slot];
        const float *hcom = p.h + (((size_t)slot * 2 + ct.cur) * 2) * HID;
        const float *hnew = p.h + (((size_t)slot * 2 + (ct.cur ^ 1)) * 2) * HID;
        xr[mt] = L == 0 ? p.embed + (size_t)ct.prev_token * HID : hnew;   // layer 1 input = h0'
        hr[mt] = hcom + L * HID;
    }
    f32x4 ai[MT], ah[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) { ai[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; ah[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    mfma_range<MT, false>(w_ih, kg0, kg1, xr, nullptr, q, ai);     // src/nemo-ggml.cpp:595
    mfma_range<MT, false>(w_hh, kg0, kg1, hr, nullptr, q, ah);     // :596
    __syncthreads();                                               // previous pass done with `red`
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int j = 0; j < 4; j++) { red[wave][0][mt][lane][j] = ai[mt][j]; red[wave][1][mt][lane][j] = ah[mt][j]; }
    __syncthreads();
    // wave mt finishes m-tile mt: lane (q, r) holds gates i,f,g,o of unit nt*4+q for its row
    const int mt = wave;
    const int i = i0 + mt * 16 + r;
    if (mt < MT && i < nd) {
        const int slot = p.rows[p.dlist[i]].slot;
        const DecCtrl ct = p.ctrl[slot];
        const int unit = nt * 4 + q;
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float gi = ((red[0][0][mt][lane][j] + red[1][0][mt][lane][j]) + red[2][0][mt][lane][j]) + red[3][0][mt][lane][j];
            const float gh = ((red[0][1][mt][lane][j] + red[1][1][mt][lane][j]) + red[2][1][mt][lane][j]) + red[3][1][mt][lane][j];
            const int row = j * HID + unit;
            g[j] = ((gi + gh) + p.b_ih[L][row]) + p.b_hh[L][row];     // :597-599
        }
        const float cprev = p.c[(((size_t)slot * 2 + ct.cur) * 2 + L) * HID + unit];
        const float cnew = sigm(g[1]) * cprev + sigm(g[0]) * tanhf(g[2]);   // :615
        p.c[(((size_t)slot * 2 + (ct.cur ^ 1)) * 2 + L) * HID + unit] = cnew;
        p.h[(((size_t)slot * 2 + (ct.cur ^ 1)) * 2 + L) * HID + unit] = sigm(g[3]) * tanhf(cnew);   // :618
    }
}

template <int L>
__global__ __launch_bounds__(256) void k_dec_lstm(DecParams p) {
    const int nd = *p.n_dirty;
    if (nd == 0) return;
    __shared__ float red[4][2][MT_MAX][64][4];
    for (int i0 = 0; i0 < nd; i0 += 16 * MT_MAX) {
        const int left = nd - i0;
        if (left <= 16) lstm_pass<L, 1>(p, i0, nd, red);
        else if (left <= 32) lstm_pass<L, 2>(p, i0, nd, red);
        else lstm_pass<L, 4>(p, i0, nd, red);
    }
}

// ---- g = W_pred . h1' + b_pred for the dirty rows, grid = 40 (src/nemo-ggml.cpp:1207-1208) ---------
template <int MT>
__device__ __forceinline__ void pred_pass(const DecParams &p, int i0, int nd, float (*red)[MT_MAX][64][4]) {
    const int nt = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, r = lane & 15;
    const float4 *w = (const float4 *)p.pred_w + (size_t)nt * KG640 * 64 + lane;
    const int kg0 = wave * (KG640 / 4), kg1 = kg0 + KG640 / 4;
    const float *xr[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int i = i0 + mt * 16 + r;
        if (i >= nd) i = i0;
        const int slot = p.rows[p.dlist[i]].slot;
        xr[mt] = p.h + (((size_t)slot * 2 + (p.ctrl[slot].cur ^ 1)) * 2 + 1) * HID;   // h1'
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_range<MT, false>(w, kg0, kg1, xr, nullptr, q, acc);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int j = 0; j < 4; j++) red[wave][mt][lane][j] = acc[mt][j];
    __syncthreads();
    const int mt = wave;
    const int i = i0 + mt * 16 + r;
    if (mt < MT && i < nd) {
        const int slot = p.rows[p.dlist[i]].slot;
        float4 o;
        float *op = (float *)&o;
#pragma unroll
        for (int j = 0; j < 4; j++)
            op[j] = (((red[0][mt][lane][j] + red[1][mt][lane][j]) + red[2][mt][lane][j]) + red[3][mt][lane][j]) + p.pred_b[nt * 16 + q * 4 + j];
        *(float4 *)(p.predg + (size_t)slot * JNT + nt * 16 + q * 4) = o;
    }
}

__global__ __launch_bounds__(256) void k_dec_pred(DecParams p) {
    const int nd = *p.n_dirty;
    if (nd == 0) return;
    __shared__ float red[4][MT_MAX][64][4];
    for (int i0 = 0; i0 < nd; i0 += 16 * MT_MAX) {
        const int left = nd - i0;
        if (left <= 16) pred_pass<1>(p, i0, nd, red);
        else if (left <= 32) pred_pass<2>(p, i0, nd, red);
        else pred_pass<4>(p, i0, nd, red);
    }
}

__device__ __forceinline__ unsigned long long pack_key(float v, int idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                 // order-preserving
    return ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (uint32_t)idx);
}
__device__ __forceinline__ unsigned long long kmax(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    const unsigned lo = __shfl_xor((unsigned)v, m), hi = __shfl_xor((unsigned)(v >> 32), m);
    return ((unsigned long long)hi << 32) | lo;
}

// ---- joint, few rows (<= 64 rows in the step): logits = W_out . relu(encproj[row] + g) + b_out and
// arg-max; grid = 65 (1040 padded vocab rows), 4 waves split K; first maximum wins (:899-906, :1220-1221)
template <int MT>
__device__ __forceinline__ void joint_pass(const DecParams &p, int i0, int nr, float (*red)[MT_MAX][64][4]) {
    const int nt = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, r = lane & 15;
    const float4 *w = (const float4 *)p.out_w + (size_t)nt * KG640 * 64 + lane;
    const int kg0 = wave * (KG640 / 4), kg1 = kg0 + KG640 / 4;
    const float *er[MT], *gr[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int i = i0 + mt * 16 + r;
        if (i >= nr) i = i0;
        const unsigned rm = p.rowmap[i];
        const int b = (int)(rm & 0xffffu), f = (int)(rm >> 16);
        er[mt] = p.encproj + ((size_t)b * p.T + f) * JNT;
        gr[mt] = p.predg + (size_t)p.rows[b].slot * JNT;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_range<MT, true>(w, kg0, kg1, er, gr, q, acc);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int j = 0; j < 4; j++) red[wave][mt][lane][j] = acc[mt][j];
    __syncthreads();
    const int mt = wave;
    const int i = i0 + mt * 16 + r;
    if (mt < MT) {
        unsigned long long best = 0ull;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int v = nt * 16 + q * 4 + j;
            if (v < VOCAB) {
                const float lg = (((red[0][mt][lane][j] + red[1][mt][lane][j]) + red[2][mt][lane][j]) + red[3][mt][lane][j]) + p.out_b[v];
                best = kmax(best, pack_key(lg, v));
            }
        }
        best = kmax(best, shfl_xor_u64(best, 16));
        best = kmax(best, shfl_xor_u64(best, 32));
        if (q == 0 && i < nr && best) {
            const unsigned rm = p.rowmap[i];
            atomicMax(&p.key[(size_t)(rm & 0xffffu) * p.T + (rm >> 16)], best);
        }
    }
}

__global__ __launch_bounds__(256) void k_dec_joint(DecParams p) {
    const int nr = *p.n_rows;
    if (nr == 0) return;
    __shared__ float red[4][MT_MAX][64][4];
    for (int i0 = 0; i0 < nr; i0 += 16 * MT_MAX) {
        const int left = nr - i0;
        if (left <= 16) joint_pass<1>(p, i0, nr, red);
        else if (left <= 32) joint_pass<2>(p, i0, nr, red);
        else joint_pass<4>(p, i0, nr, red);
    }
}

// ---- joint, many rows: 64 rows x 64 vocab entries per workgroup, grid = (17, ceil(B*T / 64)).
// Wave w owns vocab tile 4*blockIdx.x + w (weights straight from global, already in fragment order)
// and all four 16-row tiles; the relu(encproj + g) operand of a 64-deep K chunk is formed once per
// workgroup and staged in LDS (double-buffered, 16-byte chunks XOR-swizzled with the row).  The
// kernel is bound by the f32 MFMA (32 cycles per 16x16x4), 640 of them per wave.
constexpr int JT_KC = 64;                       // K per LDS chunk = 4 k-groups
__global__ __launch_bounds__(256) void k_dec_joint_tiled(DecParams p) {
    const int nr = *p.n_rows;
    const int m0 = blockIdx.y * 64;
    if (m0 >= nr) return;
    __shared__ __attribute__((aligned(16))) float xs[2][64 * JT_KC];
    __shared__ unsigned long long bests[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, r = lane & 15;
    const int nt = blockIdx.x * 4 + wave;
    const bool has_tile = nt * 16 < VOCAB;
    const float4 *w = (const float4 *)p.out_w + (size_t)(has_tile ? nt : 0) * KG640 * 64 + lane;
    int nmt = (nr - m0 + 15) >> 4;
    if (nmt > 4) nmt = 4;
    // staging role: thread -> (row, four 16-byte chunks)
    const int srow = threadIdx.x >> 2, sc0 = (threadIdx.x & 3) * 4;
    const unsigned srm = p.rowmap[m0 + srow < nr ? m0 + srow : m0];
    const float *erow = p.encproj + ((size_t)(srm & 0xffffu) * p.T + (srm >> 16)) * JNT + sc0 * 4;
    const float *grow = p.predg + (size_t)p.rows[srm & 0xffffu].slot * JNT + sc0 * 4;
    float4 ev[4], gv[4], wv[4], wn[4];
    auto gload = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 4; i++) { ev[i] = *(const float4 *)(erow + kc * JT_KC + i * 4); gv[i] = *(const float4 *)(grow + kc * JT_KC + i * 4); }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float4 x = make_float4(fmaxf(ev[i].x + gv[i].x, 0.0f), fmaxf(ev[i].y + gv[i].y, 0.0f),
                                         fmaxf(ev[i].z + gv[i].z, 0.0f), fmaxf(ev[i].w + gv[i].w, 0.0f));
            *(float4 *)(&xs[buf][srow * JT_KC + (((sc0 + i) ^ (srow & 15)) << 2)]) = x;
        }
    };
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gload(0);
#pragma unroll
    for (int u = 0; u < 4; u++) wv[u] = w[(size_t)u * 64];
    sstore(0);
    __syncthreads();
    constexpr int NKC = JNT / JT_KC;            // 10
    for (int kc = 0; kc < NKC; kc++) {
        const int cur = kc & 1;
        if (kc + 1 < NKC) {
            gload(kc + 1);
#pragma unroll
            for (int u = 0; u < 4; u++) wn[u] = w[(size_t)((kc + 1) * 4 + u) * 64];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                if (mt < nmt) {
                    const float4 x = *(const float4 *)(&xs[cur][(mt * 16 + r) * JT_KC + (((u * 4 + q) ^ r) << 2)]);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].x, x.x, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].y, x.y, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].z, x.z, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].w, x.w, acc[mt], 0, 0, 0);
                }
            }
        }
        if (kc + 1 < NKC) {
            sstore(cur ^ 1);
#pragma unroll
            for (int u = 0; u < 4; u++) wv[u] = wn[u];
        }
        __syncthreads();
    }
    // arg-max: in-lane over the 4 vocab entries, across the 4 lane groups, across the 4 waves, then one
    // atomic per row
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
        unsigned long long best = 0ull;
        if (has_tile && mt < nmt) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int v = nt * 16 + q * 4 + j;
                if (v < VOCAB) best = kmax(best, pack_key(acc[mt][j] + p.out_b[v], v));
            }
        }
        best = kmax(best, shfl_xor_u64(best, 16));
        best = kmax(best, shfl_xor_u64(best, 32));
        if (q == 0) bests[wave][mt * 16 + r] = best;
    }
    __syncthreads();
    if (threadIdx.x < 64 && m0 + threadIdx.x < nr) {
        const unsigned long long best = kmax(kmax(bests[0][threadIdx.x], bests[1][threadIdx.x]), kmax(bests[2][threadIdx.x], bests[3][threadIdx.x]));
        const unsigned rm = p.rowmap[m0 + threadIdx.x];
        if (best) atomicMax(&p.key[(size_t)(rm & 0xffffu) * p.T + (rm >> 16)], best);
    }
}

// ---- commit: walk each stream's evaluated frames up to its first non-blank -------------------------
__global__ __launch_bounds__(256) void k_dec_commit(DecParams p) {
    __shared__ int sh[4];
    if (*p.n_active == 0) return;
    const int nd = *p.n_dirty;
    for (int i = threadIdx.x; i < nd; i += 256) p.ctrl[p.rows[p.dlist[i]].slot].dirty = 0;   // candidates are fresh now
    __syncthreads();
    for (int b = threadIdx.x; b < p.B; b += 256) {
        const int slot = p.rows[b].slot;
        DecCtrl *ct = &p.ctrl[slot];
        if (!ct->active) continue;
        const int t0 = ct->t, nf = ct->n_frames;
        int f = t0, best = BLANK;
        for (; f < nf; f++) {
            const unsigned long long k = p.key[(size_t)b * p.T + f];
            best = (int)(0xffffffffu - (uint32_t)(k & 0xffffffffull));
            if (best != BLANK) break;                  // blank: next frame, state untouched (src/nemo-stream.cpp:908-911)
        }
        for (int g = t0; g < nf; g++) p.key[(size_t)b * p.T + g] = 0ull;
        if (f >= nf) {
            ct->iterations += nf - t0;
            ct->t = nf;
            ct->symbols = 0;
            ct->active = 0;
        } else {                                       // :921-926
            ct->iterations += f - t0 + 1;
            int sym = f > t0 ? 0 : ct->symbols;
            const int n = ct->n_tok;
            p.tok_ring[(size_t)slot * TOK_CAP + (n & (TOK_CAP - 1))] = best;
            p.tok_frame[(size_t)slot * TOK_CAP + (n & (TOK_CAP - 1))] = ct->frame0 + f;
            ct->n_tok = n + 1;
            ct->prev_token = best;
            ct->cur ^= 1;
            ct->dirty = 1;
            int t = f;
            if (++sym >= MAX_SYMBOLS) { t++; sym = 0; }   // :849, :865
            ct->t = t;
            ct->symbols = sym;
            if (t >= nf) ct->active = 0;
        }
    }
    __threadfence_block();
    build_lists(p, sh);
}

// ---- joint.enc projection hoisted out of the symbol loop: out[m][n] = W[n].x[m] + b[n] -------------
// (src/nemo-ggml.cpp:1204-1205; the reference recomputes it per symbol).  f32 MFMA, packed weights,
// grid = (N/16, ceil(M/64)), 4 waves split K.
__global__ __launch_bounds__(256) void k_encproj(const float *x, const float *wpk, const float *bias, float *out, int M, int K, int N) {
    __shared__ float red[4][MT_MAX][64][4];
    const int nt = blockIdx.x, b0 = blockIdx.y * 16 * MT_MAX;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, r = lane & 15;
    const int KG = K / 16;
    const float4 *w = (const float4 *)wpk + (size_t)nt * KG * 64 + lane;
    const int kg0 = wave * (KG / 4), kg1 = kg0 + KG / 4;
    const float *xr[MT_MAX];
#pragma unroll
    for (int mt = 0; mt < MT_MAX; mt++) {
        int m = b0 + mt * 16 + r;
        if (m >= M) m = M - 1;
        xr[mt] = x + (size_t)m * K;
    }
    f32x4 acc[MT_MAX];
#pragma unroll
    for (int mt = 0; mt < MT_MAX; mt++) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_range<MT_MAX, false>(w, kg0, kg1, xr, nullptr, q, acc);
#pragma unroll
    for (int mt = 0; mt < MT_MAX; mt++)
#pragma unroll
        for (int j = 0; j < 4; j++) red[wave][mt][lane][j] = acc[mt][j];
    __syncthreads();
    const int mt = wave, m = b0 + mt * 16 + r;
    if (m < M) {
        float4 o;
        float *op = (float *)&o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = nt * 16 + q * 4 + j;
            op[j] = (((red[0][mt][lane][j] + red[1][mt][lane][j]) + red[2][mt][lane][j]) + red[3][mt][lane][j]) + bias[n];
        }
        *(float4 *)(out + (size_t)m * N + nt * 16 + q * 4) = o;
    }
}
void launch_encproj(const float *x, const float *wpk, const float *bias, float *out, int M, int K, int N, hipStream_t st) {
    hipLaunchKernelGGL(k_encproj, dim3(N / 16, (M + 16 * MT_MAX - 1) / (16 * MT_MAX)), dim3(256), 0, st, x, wpk, bias, out, M, K, N);
}

void launch_decode_begin(const DecParams &p, hipStream_t st) {
    hipLaunchKernelGGL(k_dec_begin, dim3(1), dim3(256), 0, st, p);
}
// One iteration = recompute stale prediction-network outputs, evaluate every remaining (stream, frame)
// row, commit.  Every kernel exits at once when its work list is empty, so surplus iterations of a
// blindly enqueued (graph-captured) sequence cost only their launch slots.
void launch_decode_iter(const DecParams &p, int iter, hipStream_t st) {
    (void)iter;
    hipLaunchKernelGGL(k_dec_lstm<0>, dim3(HID / 4), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_dec_lstm<1>, dim3(HID / 4), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_dec_pred, dim3(JNT / 16), dim3(256), 0, st, p);
    const int rows = p.B * p.T;
    if (rows <= 64) hipLaunchKernelGGL(k_dec_joint, dim3((VOCAB + 15) / 16), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(k_dec_joint_tiled, dim3((VOCAB + 63) / 64, (rows + 63) / 64), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_dec_commit, dim3(1), dim3(256), 0, st, p);
}
// iterations enqueued before the host looks at n_active: (symbols of the busiest stream) + 1 are needed; a shortfall
// costs one host round trip and a further round of iterations
int decode_blind_iterations(int frames) {
    if (frames <= 1) return 2;                  // one emission + the closing blank; a spare iteration would cost 1 % of the step
    return frames / 2 + 4 < 16 ? frames / 2 + 4 : 16;   // an idle iteration (~10 us) is far cheaper than a host round trip
}

}  // namespace nasr

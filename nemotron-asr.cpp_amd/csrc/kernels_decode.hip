// kernels_decode.hip -- device-resident batched RNN-T greedy decode (stage a-12..a-14).
//
// Reference: src/nemo-stream.cpp:840-930 runs, per symbol, 4 host->device copies, one graph
// launch and 1-3 device->host copies for ONE stream.  Here the whole loop lives on the device
// for all B streams of a step: a "frame-and-symbol looping" schedule in which every iteration
// evaluates LSTMx2 + joint for each still-active stream on ITS current encoder frame, then a
// commit kernel applies the reference's rules per stream (blank -> next frame, state untouched
// :908-911; non-blank -> emit, prev_token = best, commit h/c :921-926; at most 10 symbols per
// frame :849; arg-max keeps the first maximum :899-906).  Streams are independent, so the
// result is identical to the reference's sequential per-stream loop.
//
// All arithmetic is f32 (the reference keeps decoder/joint weights F32 in every GGUF flavour):
// the three mat-vec stages are batched over the streams and run on the f32-input MFMA
// (v_mfma_f32_16x16x4_f32, bit-exact fmaf chains).  Weights are pre-packed at upload into
// MFMA A-fragment order -- tile (nt, kg) = 16 rows x 16 k is 1 KiB, lane l = q*16 + r holds
// W[row(nt, r)][kg*16 + 4q .. +4) as one float4 = its operand for 4 consecutive MFMAs -- so a
// wave-load is one contiguous 1 KiB read.  The stream vectors are the B operand: lane (q, j)
// loads x[stream j][kg*16 + 4q .. +4), contiguous as well.  For the LSTM the 16 rows of a tile
// are ordered (unit u, gate g) = (r/4, r%4), which lands the four gates i,f,g,o of one hidden
// unit in the four accumulator registers of one lane: the cell update needs no data movement.
#include "nasr_internal.h"

namespace nasr {

typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int KG640 = HID / 16;     // 40 k-groups of 16
constexpr int MT_MAX = 4;           // m-tiles (16 streams each) per pass

__global__ void k_dec_begin(DecParams p) {
    if (threadIdx.x == 0) {
        int n = 0;
        for (int b = 0; b < p.B; b++) {
            const RowDesc rd = p.rows[b];
            DecCtrl *ct = &p.ctrl[rd.slot];
            ct->t = 0;
            ct->n_frames = rd.n_dec;
            ct->symbols = 0;
            ct->row = b;
            ct->active = rd.n_dec > 0 ? 1 : 0;
            n += ct->active;
            p.key[b] = 0ull;
        }
        *p.n_active = n;
    }
}

// acc[mt] += W_tile(nt, kg range) . X   for MT m-tiles; xrow[mt] = this lane's stream vector.
// The k-groups are processed in blocks of 5 with every load of the block (weights + stream vectors)
// issued before its MFMAs, so a wave has ~25 independent 16-byte loads in flight instead of one.
template <int MT>
__device__ __forceinline__ void mfma_range(const float4 *wt, int kg0, int kg1, const float *const *xrow, int q,
                                           f32x4 *acc) {
    constexpr int KB = 5;
    int kg = kg0;
    for (; kg + KB <= kg1; kg += KB) {
        float4 w[KB], x[KB][MT];
#pragma unroll
        for (int u = 0; u < KB; u++) {
            w[u] = wt[(size_t)(kg + u) * 64];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) x[u][mt] = *(const float4 *)(xrow[mt] + (kg + u) * 16 + q * 4);
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
            const float4 x = *(const float4 *)(xrow[mt] + kg * 16 + q * 4);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.z, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.w, acc[mt], 0, 0, 0);
        }
    }
}

// ---- LSTM layer L: grid = 160 (4 hidden units per workgroup), 4 waves split K ----------------------
template <int L>
__global__ __launch_bounds__(256) void k_dec_lstm(DecParams p) {
    if (*p.n_active == 0) return;
    __shared__ float red[4][2][MT_MAX][64][4];
    const int nt = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, r = lane & 15;
    const float4 *w_ih = (const float4 *)p.w_ih[L] + (size_t)nt * KG640 * 64 + lane;
    const float4 *w_hh = (const float4 *)p.w_hh[L] + (size_t)nt * KG640 * 64 + lane;
    const int kg0 = wave * (KG640 / 4), kg1 = kg0 + KG640 / 4;
    for (int b0 = 0; b0 < p.B; b0 += 16 * MT_MAX) {
        const float *xr[MT_MAX], *hr[MT_MAX];
#pragma unroll
        for (int mt = 0; mt < MT_MAX; mt++) {
            int b = b0 + mt * 16 + r;
            if (b >= p.B) b = 0;                                   // valid memory, result discarded
            const int slot = p.rows[b].slot;
            const DecCtrl ct = p.ctrl[slot];
            const float *hcom = p.h + (((size_t)slot * 2 + ct.cur) * 2) * HID;
            const float *hnew = p.h + (((size_t)slot * 2 + (ct.cur ^ 1)) * 2) * HID;
            xr[mt] = L == 0 ? p.embed + (size_t)ct.prev_token * HID : hnew;   // layer 1 input = h0'
            hr[mt] = hcom + L * HID;
        }
        f32x4 ai[MT_MAX], ah[MT_MAX];
#pragma unroll
        for (int mt = 0; mt < MT_MAX; mt++) { ai[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; ah[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        mfma_range<MT_MAX>(w_ih, kg0, kg1, xr, q, ai);             // src/nemo-ggml.cpp:595
        mfma_range<MT_MAX>(w_hh, kg0, kg1, hr, q, ah);             // :596
        __syncthreads();                                           // previous pass done with `red`
#pragma unroll
        for (int mt = 0; mt < MT_MAX; mt++)
#pragma unroll
            for (int j = 0; j < 4; j++) { red[wave][0][mt][lane][j] = ai[mt][j]; red[wave][1][mt][lane][j] = ah[mt][j]; }
        __syncthreads();
        // wave mt finishes m-tile mt: lane (q, r) holds gates i,f,g,o of unit nt*4+q for stream b
        const int mt = wave;
        const int b = b0 + mt * 16 + r;
        if (b < p.B) {
            const int slot = p.rows[b].slot;
            const DecCtrl ct = p.ctrl[slot];
            if (ct.active) {
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
    }
}

// ---- joint hidden: relu(enc_proj[frame] + W_pred . h1' + b_pred), grid = 40 (:1204-1217) ---------
__global__ __launch_bounds__(256) void k_dec_jact(DecParams p) {
    if (*p.n_active == 0) return;
    __shared__ float red[4][MT_MAX][64][4];
    const int nt = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, r = lane & 15;
    const float4 *w = (const float4 *)p.pred_w + (size_t)nt * KG640 * 64 + lane;
    const int kg0 = wave * (KG640 / 4), kg1 = kg0 + KG640 / 4;
    for (int b0 = 0; b0 < p.B; b0 += 16 * MT_MAX) {
        const float *xr[MT_MAX];
#pragma unroll
        for (int mt = 0; mt < MT_MAX; mt++) {
            int b = b0 + mt * 16 + r;
            if (b >= p.B) b = 0;
            const int slot = p.rows[b].slot;
            const int cur = p.ctrl[slot].cur;
            xr[mt] = p.h + (((size_t)slot * 2 + (cur ^ 1)) * 2 + 1) * HID;   // h1'
        }
        f32x4 acc[MT_MAX];
#pragma unroll
        for (int mt = 0; mt < MT_MAX; mt++) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_range<MT_MAX>(w, kg0, kg1, xr, q, acc);
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT_MAX; mt++)
#pragma unroll
            for (int j = 0; j < 4; j++) red[wave][mt][lane][j] = acc[mt][j];
        __syncthreads();
        const int mt = wave;
        const int b = b0 + mt * 16 + r;
        if (b < p.B) {
            const int slot = p.rows[b].slot;
            const DecCtrl ct = p.ctrl[slot];
            if (ct.active) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int n = nt * 16 + q * 4 + j;
                    const float d = (((red[0][mt][lane][j] + red[1][mt][lane][j]) + red[2][mt][lane][j]) + red[3][mt][lane][j]) + p.pred_b[n];
                    const float e = p.encproj[((size_t)b * p.T + ct.t) * JNT + n];
                    p.jact[(size_t)b * JNT + n] = fmaxf(e + d, 0.0f);
                }
            }
        }
    }
}

__device__ __forceinline__ unsigned long long pack_key(float v, int idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                 // order-preserving
    return ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (uint32_t)idx);
}

// ---- logits + arg-max, grid = 65 (1040 padded vocab rows); first maximum wins (:899-906) -----------
__global__ __launch_bounds__(256) void k_dec_logits(DecParams p) {
    if (*p.n_active == 0) return;
    __shared__ float red[4][MT_MAX][64][4];
    const int nt = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, r = lane & 15;
    const float4 *w = (const float4 *)p.out_w + (size_t)nt * KG640 * 64 + lane;
    const int kg0 = wave * (KG640 / 4), kg1 = kg0 + KG640 / 4;
    for (int b0 = 0; b0 < p.B; b0 += 16 * MT_MAX) {
        const float *xr[MT_MAX];
#pragma unroll
        for (int mt = 0; mt < MT_MAX; mt++) {
            int b = b0 + mt * 16 + r;
            if (b >= p.B) b = 0;
            xr[mt] = p.jact + (size_t)b * JNT;
        }
        f32x4 acc[MT_MAX];
#pragma unroll
        for (int mt = 0; mt < MT_MAX; mt++) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_range<MT_MAX>(w, kg0, kg1, xr, q, acc);
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT_MAX; mt++)
#pragma unroll
            for (int j = 0; j < 4; j++) red[wave][mt][lane][j] = acc[mt][j];
        __syncthreads();
        const int mt = wave;
        const int b = b0 + mt * 16 + r;
        if (b < p.B && p.ctrl[p.rows[b].slot].active) {
            unsigned long long best = 0ull;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int v = nt * 16 + q * 4 + j;
                if (v < VOCAB) {
                    const float lg = (((red[0][mt][lane][j] + red[1][mt][lane][j]) + red[2][mt][lane][j]) + red[3][mt][lane][j]) + p.out_b[v];   // :1220-1221
                    const unsigned long long k = pack_key(lg, v);
                    best = k > best ? k : best;
                }
            }
            if (best) atomicMax(&p.key[b], best);
        }
    }
}

__global__ __launch_bounds__(256) void k_dec_commit(DecParams p) {
    __shared__ int cnt;
    if (*p.n_active == 0) return;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < p.B; b += 256) {
        const int slot = p.rows[b].slot;
        DecCtrl *ct = &p.ctrl[slot];
        if (ct->active) {
            const unsigned long long k = p.key[b];
            const int best = (int)(0xffffffffu - (uint32_t)(k & 0xffffffffull));
            ct->iterations++;
            if (best == BLANK) {                       // src/nemo-stream.cpp:908-911
                ct->t++;
                ct->symbols = 0;
            } else {                                   // :921-926
                p.tok_ring[(size_t)slot * TOK_CAP + (ct->n_tok & (TOK_CAP - 1))] = best;
                ct->n_tok++;
                ct->prev_token = best;
                ct->cur ^= 1;
                if (++ct->symbols >= MAX_SYMBOLS) { ct->t++; ct->symbols = 0; }   // :849, :865
            }
            if (ct->t >= ct->n_frames) ct->active = 0;
            else atomicAdd(&cnt, 1);
        }
        p.key[b] = 0ull;
    }
    __syncthreads();
    if (threadIdx.x == 0) *p.n_active = cnt;
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
    mfma_range<MT_MAX>(w, kg0, kg1, xr, q, acc);
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
    hipLaunchKernelGGL(k_dec_begin, dim3(1), dim3(64), 0, st, p);
}
void launch_decode_iter(const DecParams &p, int iter, hipStream_t st) {
    (void)iter;
    hipLaunchKernelGGL(k_dec_lstm<0>, dim3(HID / 4), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_dec_lstm<1>, dim3(HID / 4), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_dec_jact, dim3(JNT / 16), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_dec_logits, dim3((VOCAB + 15) / 16), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_dec_commit, dim3(1), dim3(256), 0, st, p);
}

}  // namespace nasr

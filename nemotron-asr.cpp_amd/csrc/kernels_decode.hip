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
// All arithmetic is f32 (the reference keeps decoder/joint weights F32 in every GGUF flavour).
#include "nasr_internal.h"

namespace nasr {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// dot of a 640-float weight row with a 640-float vector, lanes strided by 4 floats
__device__ __forceinline__ float dot640(const float *w, const float *x, int lane) {
    float s = 0.0f;
    // 640 = 160 float4; lane handles float4 indices lane, lane+64, lane+128 (<160)
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int idx = lane + c * 64;
        if (idx < 160) {
            const float4 a = ((const float4 *)w)[idx], b = ((const float4 *)x)[idx];
            s += a.x * b.x; s += a.y * b.y; s += a.z * b.z; s += a.w * b.w;
        }
    }
    return wsum(s);
}

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
            p.key[p.B + b] = 0ull;
        }
        *p.n_active = n;
    }
}

// LSTM layer L for every active stream: one wave per hidden unit j (its 4 gate rows i,f,g,o:
// src/nemo-ggml.cpp:601-612).  grid = 160 x 4 waves.  New state goes to version 1-cur.
template <int L>
__global__ __launch_bounds__(256) void k_dec_lstm(DecParams p) {
    if (*p.n_active == 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    const float *wih = p.w_ih[L], *whh = p.w_hh[L], *bih = p.b_ih[L], *bhh = p.b_hh[L];
    for (int b = 0; b < p.B; b++) {
        const int slot = p.rows[b].slot;
        const DecCtrl ct = p.ctrl[slot];
        if (!ct.active) continue;
        const float *hc = p.h + (((size_t)slot * 2 + ct.cur) * 2) * HID;       // committed [2][640]
        const float *cc = p.c + (((size_t)slot * 2 + ct.cur) * 2) * HID;
        float *hn = p.h + (((size_t)slot * 2 + (ct.cur ^ 1)) * 2) * HID;       // candidate
        float *cn = p.c + (((size_t)slot * 2 + (ct.cur ^ 1)) * 2) * HID;
        const float *x = L == 0 ? p.embed + (size_t)ct.prev_token * HID : hn;  // layer 1 input = h0'
        const float *hp = hc + L * HID;
        float g[4];
#pragma unroll
        for (int gi = 0; gi < 4; gi++) {
            const int row = gi * HID + j;
            const float a = dot640(wih + (size_t)row * HID, x, lane);           // :595
            const float bsum = dot640(whh + (size_t)row * HID, hp, lane);       // :596
            g[gi] = ((a + bsum) + bih[row]) + bhh[row];                         // :597-599
        }
        if (lane == 0) {
            const float cprev = cc[L * HID + j];
            const float cnew = sigm(g[1]) * cprev + sigm(g[0]) * tanhf(g[2]);   // :615
            cn[L * HID + j] = cnew;
            hn[L * HID + j] = sigm(g[3]) * tanhf(cnew);                         // :618
        }
    }
}

// joint hidden: relu(enc_proj[frame] + W_pred . h1' + b_pred)  (src/nemo-ggml.cpp:1204-1217)
__global__ __launch_bounds__(256) void k_dec_jact(DecParams p) {
    if (*p.n_active == 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    for (int b = 0; b < p.B; b++) {
        const int slot = p.rows[b].slot;
        const DecCtrl ct = p.ctrl[slot];
        if (!ct.active) continue;
        const float *h1 = p.h + (((size_t)slot * 2 + (ct.cur ^ 1)) * 2 + 1) * HID;
        const float d = dot640(p.pred_w + (size_t)n * HID, h1, lane) + p.pred_b[n];
        if (lane == 0) {
            const float e = p.encproj[((size_t)b * p.T + ct.t) * JNT + n];
            p.jact[(size_t)b * JNT + n] = fmaxf(e + d, 0.0f);
        }
    }
}

__device__ __forceinline__ unsigned long long pack_key(float v, int idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                 // order-preserving
    return ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (uint32_t)idx);
}

// logits + arg-max (first maximum wins: ties resolved towards the smaller index)
__global__ __launch_bounds__(256) void k_dec_logits(DecParams p, int parity) {
    if (*p.n_active == 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + wave;
    if (v >= VOCAB) return;
    for (int b = 0; b < p.B; b++) {
        const int slot = p.rows[b].slot;
        if (!p.ctrl[slot].active) continue;
        const float lg = dot640(p.out_w + (size_t)v * JNT, p.jact + (size_t)b * JNT, lane) + p.out_b[v];  // :1220-1221
        if (lane == 0) atomicMax(&p.key[(size_t)parity * p.B + b], pack_key(lg, v));
    }
}

__global__ __launch_bounds__(256) void k_dec_commit(DecParams p, int parity) {
    __shared__ int cnt;
    if (*p.n_active == 0) return;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < p.B; b += 256) {
        const int slot = p.rows[b].slot;
        DecCtrl *ct = &p.ctrl[slot];
        if (ct->active) {
            const unsigned long long k = p.key[(size_t)parity * p.B + b];
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
        p.key[(size_t)parity * p.B + b] = 0ull;
    }
    __syncthreads();
    if (threadIdx.x == 0) *p.n_active = cnt;
}

void launch_decode_begin(const DecParams &p, hipStream_t st) {
    hipLaunchKernelGGL(k_dec_begin, dim3(1), dim3(64), 0, st, p);
}
void launch_decode_iter(const DecParams &p, int iter, hipStream_t st) {
    const int parity = iter & 1;
    hipLaunchKernelGGL(k_dec_lstm<0>, dim3(HID / 4), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_dec_lstm<1>, dim3(HID / 4), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_dec_jact, dim3(JNT / 4), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_dec_logits, dim3((VOCAB + 3) / 4), dim3(256), 0, st, p, parity);
    hipLaunchKernelGGL(k_dec_commit, dim3(1), dim3(256), 0, st, p, parity);
}

}  // namespace nasr

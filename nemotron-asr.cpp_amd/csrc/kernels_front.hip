// kernels_front.hip -- stage a-1 (PCM -> log-mel, reference src/preprocessor.cpp) and the
// direct-convolution part of stage a-2 (reference src/nemo-ggml.cpp:897-1007).
//
// The mel kernels keep the reference's operation ORDER (radix-2 DIT butterflies in the same
// sequence, sequential mel sum over the 257 bins) and use __fmul_rn/__fadd_rn/__fsub_rn so
// hipcc cannot contract them into FMAs: everything up to the final logf is then bit-identical
// to the x86-64 reference build; logf differs by <= 1-2 ulp.
#include "nasr_internal.h"

namespace nasr {

// ---- pre-emphasis + append to the stream's audio buffer (src/preprocessor.cpp:345-356) ----
__global__ __launch_bounds__(256) void k_preemph(MelParams p) {
    const PcmDesc d = p.desc[blockIdx.y];
    if (d.n <= 0) return;
    float *buf = p.abuf + ((size_t)d.slot * 2 + d.par) * ABUF_CAP;
    const float scale = 1.0f / 32768.0f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < d.n; i += gridDim.x * 256) {
        float curr = (float)d.pcm[i] * scale;
        float prev = i > 0 ? (float)d.pcm[i - 1] * scale : p.last_sample[d.slot];   // i == 0: thread 0 of block 0
        buf[d.cnt + i] = __fsub_rn(curr, __fmul_rn(0.97f, prev));
    }
    __syncthreads();      // block 0: its thread 0 has read last_sample above
    if (blockIdx.x == 0 && threadIdx.x == 0) p.last_sample[d.slot] = (float)d.pcm[d.n - 1] * scale;
}

// ---- one workgroup per (frame, stream): window, 512-point FFT in LDS, power, mel, log ------
__global__ __launch_bounds__(256) void k_melframes(MelParams p) {
    const PcmDesc d = p.desc[blockIdx.y];
    const int t = blockIdx.x;
    if (t >= d.n_frames) return;
    __shared__ float re[NFFT], im[NFFT], pw[NBINS + 3];
    const float *src = p.abuf + ((size_t)d.slot * 2 + d.par) * ABUF_CAP + (size_t)t * HOP;
    for (int i = threadIdx.x; i < NFFT; i += 256) {
        float v = __fmul_rn(src[i], p.window[i]);                      // :184-194
        int j = (int)(__brev((unsigned)i) >> 23);                      // 9-bit reversal, :96-105
        re[j] = v;
        im[j] = 0.0f;
    }
    __syncthreads();
    for (int m = 2; m <= NFFT; m <<= 1) {                              // :131-154
        const int m2 = m >> 1, step = NFFT / m;
        const int bf = threadIdx.x;
        const int k = (bf / m2) * m, j = bf % m2;
        const float wr = p.cos_t[j * step], wi = -p.sin_t[j * step];
        const int i1 = k + j, i2 = i1 + m2;
        const float r1 = re[i1], q1 = im[i1], r2 = re[i2], q2 = im[i2];
        const float tr = __fsub_rn(__fmul_rn(wr, r2), __fmul_rn(wi, q2));
        const float ti = __fadd_rn(__fmul_rn(wr, q2), __fmul_rn(wi, r2));
        re[i2] = __fsub_rn(r1, tr);
        im[i2] = __fsub_rn(q1, ti);
        re[i1] = __fadd_rn(r1, tr);
        im[i1] = __fadd_rn(q1, ti);
        // stages m <= 128: the 64 butterflies of a wave read and write only that wave's own 128 elements ([128 wave, +128)), and a wave's LDS
        // operations execute in order -- no workgroup barrier needed (round 5: 9 -> 3 barriers per frame; same butterflies, same operands, same bits)
        if (m >= 128) __syncthreads();
        else __builtin_amdgcn_wave_barrier();
    }
    for (int k = threadIdx.x; k < NBINS; k += 256) {
        float mag = sqrtf(__fadd_rn(__fmul_rn(re[k], re[k]), __fmul_rn(im[k], im[k])));  // :201
        pw[k] = __fmul_rn(mag, mag);                                                        // :363-367
    }
    __syncthreads();
    if (threadIdx.x < NMEL) {
        float sum = 0.0f;
        const int k_lo = p.fb_band[2 * threadIdx.x], k_hi = p.fb_band[2 * threadIdx.x + 1];   // see the engine: same sum
        for (int k = k_lo; k < k_hi; k++) sum = __fadd_rn(sum, __fmul_rn(p.fbT[k * NMEL + threadIdx.x], pw[k]));
        float v = logf(__fadd_rn(sum, 5.960464477539063e-8f));                             // :381
        p.mel_ring[((size_t)d.slot * MEL_RING + ((d.mel_wpos + t) & (MEL_RING - 1))) * NMEL + threadIdx.x] = v;
        if (p.tap && t < p.tap_cap) p.tap[((size_t)blockIdx.y * p.tap_cap + t) * NMEL + threadIdx.x] = v;
    }
}

// ---- keep the unconsumed tail of the audio buffer (src/preprocessor.cpp:389-393) ----------
__global__ __launch_bounds__(256) void k_abuf_shift(MelParams p) {
    const PcmDesc d = p.desc[blockIdx.x];
    if (d.n_frames <= 0) return;
    const float *src = p.abuf + ((size_t)d.slot * 2 + d.par) * ABUF_CAP + d.consumed;
    float *dst = p.abuf + ((size_t)d.slot * 2 + (d.par ^ 1)) * ABUF_CAP;
    const int left = d.cnt + d.n - d.consumed;
    for (int i = threadIdx.x; i < left; i += 256) dst[i] = src[i];
}

void launch_mel(const MelParams &p, int max_n, hipStream_t st) {
    int nblk = (max_n + 2047) / 2048;       // ~8 samples per thread
    nblk = nblk < 1 ? 1 : (nblk > 64 ? 64 : nblk);
    hipLaunchKernelGGL(k_preemph, dim3(nblk, p.B), dim3(256), 0, st, p);
    if (p.max_frames > 0) {
        hipLaunchKernelGGL(k_melframes, dim3(p.max_frames, p.B), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_abuf_shift, dim3(p.B), dim3(256), 0, st, p);
    }
}

// debug entry: mel frames supplied by the caller (staged at [B][max_frames][128])
__global__ void k_mel_put(const float *staged, const PcmDesc *desc, int max_frames, float *mel_ring) {
    const PcmDesc d = desc[blockIdx.y];
    const int t = blockIdx.x;
    if (t >= d.n_frames) return;
    mel_ring[((size_t)d.slot * MEL_RING + ((d.mel_wpos + t) & (MEL_RING - 1))) * NMEL + threadIdx.x] =
        staged[((size_t)blockIdx.y * max_frames + t) * NMEL + threadIdx.x];
}
void launch_mel_put(const float *staged, const PcmDesc *desc, int B, int max_frames, float *mel_ring, hipStream_t st) {
    if (max_frames <= 0) return;
    hipLaunchKernelGGL(k_mel_put, dim3(max_frames, B), dim3(NMEL), 0, st, staged, desc, max_frames, mel_ring);
}

// zero / pad mel frames in the ring (tail flush: src/nemo-stream.cpp:1247-1249; stream start :73-74)
__global__ void k_mel_zero(const PcmDesc *desc, float *mel_ring) {
    const PcmDesc d = desc[blockIdx.y];
    const int t = blockIdx.x;
    if (t >= d.n_frames) return;
    mel_ring[((size_t)d.slot * MEL_RING + ((d.mel_wpos + t) & (MEL_RING - 1))) * NMEL + threadIdx.x] = 0.0f;
}
void launch_mel_zero(const PcmDesc *desc, int B, int max_frames, float *mel_ring, hipStream_t st) {
    if (max_frames <= 0) return;
    hipLaunchKernelGGL(k_mel_zero, dim3(max_frames, B), dim3(NMEL), 0, st, desc, mel_ring);
}

// ---- stream start / reset in ONE launch (src/nemo-stream.cpp:36-93 ::init, :95-115 ::reset) ----------------------------------
// Rounds 1-4 issued 2 hipMemsetAsync per layer + 5 on the engine's stream for every stream that started (53 fills, 32 MB zeroed, in
// front of the next step of every live stream).  What a fresh stream needs is much less: the conv caches (64 KB per layer), the decoder
// state, the 9 literal-zero mel frames the first chunk reads, the 256 zero samples the audio buffer is pre-seeded with, last_sample
// and the decoder control block.  Of the K/V rings only the 70 WINDOW rows in front of the stream's head are zeroed (round 6; 70 x 1024 x K, V
// = 280 KB per layer in bf16, 6.9 MB per stream start): cache_valid_len = 0 hides every cached row behind the -1e9 mask (weight exactly 0,
// :1037-1043 -- the reference's own reset relies on it, :95-115), but a weight of exactly 0 times a stale NaN / Inf is NaN, and a slot's
// previous stream may have left one (float mel handed to nasr_engine_step_mel, a broken checkpoint).  The rows a masked key can ever name are
// exactly those 70 (the mask shrinks by T per chunk while the window slides by T), so with them zeroed no stream inherits a non-finite value
// from its slot (tests/test_gpu_round6.py::test_nan_in_a_recycled_slot_does_not_reach_the_next_stream); finite stale rows never mattered
// (tests/test_gpu_parity.py::test_stale_kv_rows_never_reach_a_result).
// blocks [0, n_layers): conv cache (skipped when the reference's reset semantics are asked for) + K/V window of that layer; block n_layers: the rest.
__global__ __launch_bounds__(256) void k_stream_reset(StreamResetParams p) {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((int)blockIdx.x < p.n_layers) {
        // K/V window: logical keys 0 .. LCTX-1 = ring rows (kv_head + j) % KVC, of K (plane 0) and V (plane 1)
        char *kv = (char *)p.kv_pools[blockIdx.x] + (size_t)p.slot * 2 * KVC * D * p.esz;
        const int row_f4 = D * p.esz / 16;                      // float4 per ring row
        for (int i = threadIdx.x; i < 2 * LCTX * row_f4; i += 256) {
            const int plane = i / (LCTX * row_f4), r = i % (LCTX * row_f4);
            const int ring = (p.kv_head + r / row_f4) % KVC;
            ((float4 *)(kv + ((size_t)plane * KVC + ring) * D * p.esz))[r % row_f4] = z4;
        }
        if (p.keep_reference_state) return;
        float4 *cc = (float4 *)(p.cc_pools[blockIdx.x] + (size_t)p.slot * p.cc_slot_floats);
        for (int i = threadIdx.x; i < p.cc_slot_floats / 4; i += 256) cc[i] = z4;
        return;
    }
    const size_t slot = (size_t)p.slot;
    for (int i = threadIdx.x; i < 4 * HID / 4; i += 256) {
        ((float4 *)(p.dec_h + slot * 4 * HID))[i] = z4;
        ((float4 *)(p.dec_c + slot * 4 * HID))[i] = z4;
    }
    for (int i = threadIdx.x; i < PRE_CACHE * NMEL / 4; i += 256) ((float4 *)(p.mel_ring + slot * MEL_RING * NMEL))[i] = z4;   // mel_start = 0
    if (!p.keep_reference_state) {
        for (int i = threadIdx.x; i < NFFT / 2 / 4; i += 256) ((float4 *)(p.abuf + slot * 2 * ABUF_CAP))[i] = z4;              // parity 0, src/preprocessor.cpp:220-221
        if (threadIdx.x == 0) p.last_sample[slot] = 0.0f;
    }
    if (threadIdx.x == 0) {
        DecCtrl c;
        c.t = 0; c.n_frames = 0; c.symbols = 0; c.prev_token = BLANK;      // src/nemo-stream.cpp:55-56
        c.cur = 0; c.n_tok = 0; c.active = 0; c.iterations = 0; c.row = 0;
        c.dirty = 1;                                                       // no LSTM candidate computed yet
        c.frame0 = 0; c.frame_next = 0;
        p.ctrl[slot] = c;
    }
}
void launch_stream_reset(const StreamResetParams &p, hipStream_t st) {
    hipLaunchKernelGGL(k_stream_reset, dim3(p.n_layers + 1), dim3(256), 0, st, p);
}

// ---- conv0 (3x3 stride-2, 1 -> 256 channels, + ReLU) fused into the first depthwise 3x3 stride-2 conv ----
// (src/nemo-ggml.cpp:969-978, pads (2 before, 1 after) on both axes :905-913, :936-943).  conv0's output
// [H1][65][256] f32 is 9x larger than its input and was the largest intermediate of the step (260 MB at 64
// streams x R = 13); its 9 MACs per element are cheaper to redo than to store and re-read.  Every conv0 value is
// formed exactly as the unfused kernel formed it (same order, no FMA contraction in this file), so the result is
// bit-identical.  One workgroup per (output position, chunk), one thread per channel; the mel patch addresses are
// workgroup-uniform (scalar loads).  out [B][H2][W2][256], channel fastest.
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void k_sub_conv0_dw(const RowDesc *rows, int chunk_mel, const float *mel_ring,
                                                      const float *w0t /*[9][256]*/, const float *b0,
                                                      const float *w2t /*[9][256]*/, const float *b2, void *out,
                                                      int H1, int W1, int H2, int W2) {
    // One workgroup per (output row t2, chunk): the 7 mel rows 4*t2-6 .. 4*t2 it depends on are staged in LDS,
    // zero padded (an out-of-range tap adds w*0 where the unfused kernel skipped it: same sum), 6 columns of left
    // pad so that output column f2 reads the 7x7 patch at columns 4*f2 .. 4*f2+6.
    __shared__ __attribute__((aligned(16))) float sm[7][144];
    const int c = threadIdx.x, t2 = blockIdx.x, b = blockIdx.y;
    const RowDesc rd = rows[b];
    const float *mel = mel_ring + (size_t)rd.slot * MEL_RING * NMEL;
    for (int i = threadIdx.x; i < 7 * 144; i += 256) {
        const int r = i / 144, iw = i - r * 144 - 6, ih = 4 * t2 - 6 + r;
        sm[r][i - r * 144] = (ih >= 0 && ih < chunk_mel && iw >= 0 && iw < NMEL)
                                 ? mel[(size_t)((rd.mel_start + ih) & (MEL_RING - 1)) * NMEL + iw] : 0.0f;
    }
    float w0[9], w2[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { w0[k] = w0t[k * SUBC + c]; w2[k] = w2t[k * SUBC + c]; }
    const float bias0 = b0[c], bias2 = b2[c];
    __syncthreads();
    // blockIdx.z splits the output columns when there are too few (row, chunk) pairs to fill the chip
    const int f2_lo = (int)((long)W2 * blockIdx.z / gridDim.z), f2_hi = (int)((long)W2 * (blockIdx.z + 1) / gridDim.z);
    // conv0 column 2*f2 (tap kw2 = 0) is column 2*(f2-1) + 2 (tap kw2 = 2) of the previous output: its three activated
    // values are carried over instead of recomputed (the same values: a third of the conv0 arithmetic less)
    float carry[3] = {0.0f, 0.0f, 0.0f};
    for (int f2 = f2_lo; f2 < f2_hi; f2++) {
        float p[7][8];
#pragma unroll
        for (int r = 0; r < 7; r++) {
            const float4 lo = *(const float4 *)&sm[r][4 * f2], hi = *(const float4 *)&sm[r][4 * f2 + 4];
            p[r][0] = lo.x; p[r][1] = lo.y; p[r][2] = lo.z; p[r][3] = lo.w;
            p[r][4] = hi.x; p[r][5] = hi.y; p[r][6] = hi.z; p[r][7] = hi.w;
        }
        float acc2 = 0.0f;
#pragma unroll
        for (int kh2 = 0; kh2 < 3; kh2++) {
            const int t = 2 * t2 + kh2 - 2;               // conv0 output row feeding this tap
            if (t < 0 || t >= H1) continue;
#pragma unroll
            for (int kw2 = 0; kw2 < 3; kw2++) {
                const int f = 2 * f2 + kw2 - 2;           // conv0 output column
                if (f < 0 || f >= W1) continue;
                float a0;
                if (kw2 == 0 && f2 > f2_lo) a0 = carry[kh2];
                else {
                    float acc = 0.0f;
#pragma unroll
                    for (int kh = 0; kh < 3; kh++)
#pragma unroll
                        for (int kw = 0; kw < 3; kw++) {
                            // bf16 engine (OUT_BF16): fused multiply-adds -- this kernel runs at the f32 VALU's issue rate (134 M outputs x ~130 separate multiplies and
                            // adds at 512 streams x R = 13: 500 us), the result is rounded to bf16 below and compared to the oracle with a tolerance.  The f32
                            // engine keeps the reference's separate roundings (bit-identical subsampling, tests/test_gpu_parity.py).
                            if (OUT_BF16) acc = __builtin_fmaf(w0[kh * 3 + kw], p[2 * kh2 + kh][2 * kw2 + kw], acc);
                            else acc += w0[kh * 3 + kw] * p[2 * kh2 + kh][2 * kw2 + kw];
                        }
                    a0 = fmaxf(acc + bias0, 0.0f);
                }
                if (kw2 == 2) carry[kh2] = a0;
                if (OUT_BF16) acc2 = __builtin_fmaf(w2[kh2 * 3 + kw2], a0, acc2);
                else acc2 += w2[kh2 * 3 + kw2] * a0;
            }
        }
        acc2 += bias2;
        const size_t o = (((size_t)b * H2 + t2) * W2 + f2) * SUBC + c;
        if (OUT_BF16) ((bf16_t *)out)[o] = f32_to_bf16(acc2);
        else ((float *)out)[o] = acc2;
    }
}
void launch_sub_conv0_dw(const RowDesc *rows, int B, int chunk_mel, const float *mel_ring, const float *w0t, const float *b0,
                         const float *w2t, const float *b2, void *out, int out_bf16, int H1, int W1, hipStream_t st) {
    const int H2 = H1 / 2 + 1, W2 = W1 / 2 + 1;   // W1 = 65 -> W2 = 33: columns 4*f2 + 7 <= 139 < 144
    const int fz = H2 * B <= 64 ? 11 : (H2 * B <= 512 ? 3 : 1);
    const dim3 grid(H2, B, fz);
    if (out_bf16) hipLaunchKernelGGL(k_sub_conv0_dw<true>, grid, dim3(SUBC), 0, st, rows, chunk_mel, mel_ring, w0t, b0, w2t, b2, out, H1, W1, H2, W2);
    else hipLaunchKernelGGL(k_sub_conv0_dw<false>, grid, dim3(SUBC), 0, st, rows, chunk_mel, mel_ring, w0t, b0, w2t, b2, out, H1, W1, H2, W2);
}

// ---- depthwise 3x3 stride-2 (+bias, no activation), :978, :994, :929-950 ---------------------
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void k_sub_dw(const float *in, int Hin, int Win, const float *wt /*[9][256]*/,
                                                const float *bias, void *out, int Hout, int Wout) {
    const int c = threadIdx.x;
    const int f = blockIdx.x % Wout, t = blockIdx.x / Wout, b = blockIdx.y;
    float acc = 0.0f;
#pragma unroll
    for (int kh = 0; kh < 3; kh++) {
        const int ih = 2 * t + kh - 2;
        if (ih < 0 || ih >= Hin) continue;
#pragma unroll
        for (int kw = 0; kw < 3; kw++) {
            const int iw = 2 * f + kw - 2;
            if (iw < 0 || iw >= Win) continue;
            acc += wt[(kh * 3 + kw) * SUBC + c] * in[(((size_t)b * Hin + ih) * Win + iw) * SUBC + c];
        }
    }
    acc += bias[c];
    const size_t o = (((size_t)b * Hout + t) * Wout + f) * SUBC + c;
    if (OUT_BF16) ((bf16_t *)out)[o] = f32_to_bf16(acc);
    else ((float *)out)[o] = acc;
}
// The same conv with one workgroup per OUTPUT ROW (round 5): k_sub_dw launches one 256-thread workgroup per output position -- 139 264 workgroups of nine
// 4-byte loads per thread at 512 streams x R = 13, 355 us for 536 MB: bound by the dispatch of workgroups, not by HBM.  Here a lane holds four channels
// (float4), a wave one output position, and the four waves walk the row's positions; per channel the products are added in k_sub_dw's order (kh, kw
// ascending, taps outside the image skipped; no FMA contraction in this file): same bits.
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void k_sub_dw_row(const float *in, int Hin, int Win, const float *wt /*[9][256]*/,
                                                    const float *bias, void *out, int Hout, int Wout) {
    const int t = blockIdx.x, b = blockIdx.y, wave = threadIdx.x >> 6, c4 = (threadIdx.x & 63) * 4;
    float4 w[9];
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = *(const float4 *)(wt + k * SUBC + c4);
    const float4 bs = *(const float4 *)(bias + c4);
    for (int f = wave; f < Wout; f += 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            const int ih = 2 * t + kh - 2;
            if (ih < 0 || ih >= Hin) continue;
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const int iw = 2 * f + kw - 2;
                if (iw < 0 || iw >= Win) continue;
                const float4 x = *(const float4 *)(in + (((size_t)b * Hin + ih) * Win + iw) * SUBC + c4);
                const float4 ww = w[kh * 3 + kw];
                acc.x += ww.x * x.x; acc.y += ww.y * x.y; acc.z += ww.z * x.z; acc.w += ww.w * x.w;
            }
        }
        acc.x += bs.x; acc.y += bs.y; acc.z += bs.z; acc.w += bs.w;
        const size_t o = (((size_t)b * Hout + t) * Wout + f) * SUBC + c4;
        if (OUT_BF16) {
            uint2 r;
            r.x = (uint32_t)f32_to_bf16(acc.x) | ((uint32_t)f32_to_bf16(acc.y) << 16);
            r.y = (uint32_t)f32_to_bf16(acc.z) | ((uint32_t)f32_to_bf16(acc.w) << 16);
            *(uint2 *)((bf16_t *)out + o) = r;
        } else *(float4 *)((float *)out + o) = acc;
    }
}
void launch_sub_dw(const float *in, int B, int Hin, int Win, const float *wt, const float *bias, void *out,
                   int out_bf16, hipStream_t st) {
    const int Hout = Hin / 2 + 1, Wout = Win / 2 + 1;
    if ((long)Hout * B >= 512) {        // enough rows to fill the chip (16 streams x R = 13, 128 streams x R = 0)
        if (out_bf16) hipLaunchKernelGGL(k_sub_dw_row<true>, dim3(Hout, B), dim3(256), 0, st, in, Hin, Win, wt, bias, out, Hout, Wout);
        else hipLaunchKernelGGL(k_sub_dw_row<false>, dim3(Hout, B), dim3(256), 0, st, in, Hin, Win, wt, bias, out, Hout, Wout);
        return;
    }
    if (out_bf16) hipLaunchKernelGGL(k_sub_dw<true>, dim3(Hout * Wout, B), dim3(SUBC), 0, st, in, Hin, Win, wt, bias, out, Hout, Wout);
    else hipLaunchKernelGGL(k_sub_dw<false>, dim3(Hout * Wout, B), dim3(SUBC), 0, st, in, Hin, Win, wt, bias, out, Hout, Wout);
}


// ---- diarization front end: stateless 80-mel log spectrogram of a whole window (src/diarize_audio.cpp:136-227) ----
// One workgroup per (frame, window).  Unlike the streaming ASR front end above: pre-emphasis restarts in every
// window (y[0] = x[0], :83-93), the STFT is centred with zero padding (:119-125), power = re^2 + im^2 without the
// sqrt round trip (:127-131), frames >= t_valid (the +1 frame of the centred STFT and the pad-to-16 tail) are zeros.
// Operation order as in the reference (this file is compiled without FMA contraction).
template <typename Tp>
__device__ __forceinline__ float diar_sample(const Tp *x, int i);
template <>
__device__ __forceinline__ float diar_sample<float>(const float *x, int i) { return x[i]; }
template <>
__device__ __forceinline__ float diar_sample<int16_t>(const int16_t *x, int i) { return (float)x[i] / 32768.0f; }   // src/diarize_pipeline.cpp: s16 -> [-1, 1)

template <typename Tp>
__device__ __forceinline__ void diar_frame(const DiarMelParams &p, const Tp *x, int n_win, int t, float *out, int cpitch) {
    __shared__ float re[NFFT], im[NFFT], pw[NBINS + 3];
    const int start = t * HOP - NFFT / 2;
    for (int i = threadIdx.x; i < NFFT; i += 256) {
        const int idx = start + i;
        float s = 0.0f;
        if (idx >= 0 && idx < n_win) s = idx == 0 ? diar_sample(x, 0) : __fsub_rn(diar_sample(x, idx), __fmul_rn(0.97f, diar_sample(x, idx - 1)));
        const float v = __fmul_rn(s, p.window[i]);
        const int j = (int)(__brev((unsigned)i) >> 23);
        re[j] = v;
        im[j] = 0.0f;
    }
    __syncthreads();
    for (int m = 2; m <= NFFT; m <<= 1) {                              // :56-74
        const int m2 = m >> 1, step = NFFT / m;
        const int bf = threadIdx.x;
        const int k = (bf / m2) * m, j = bf % m2;
        const float wr = p.cos_t[j * step], wi = -p.sin_t[j * step];
        const int i1 = k + j, i2 = i1 + m2;
        const float r1 = re[i1], q1 = im[i1], r2 = re[i2], q2 = im[i2];
        const float tr = __fsub_rn(__fmul_rn(wr, r2), __fmul_rn(wi, q2));
        const float ti = __fadd_rn(__fmul_rn(wr, q2), __fmul_rn(wi, r2));
        re[i2] = __fsub_rn(r1, tr);
        im[i2] = __fsub_rn(q1, ti);
        re[i1] = __fadd_rn(r1, tr);
        im[i1] = __fadd_rn(q1, ti);
        // stages m <= 128: the 64 butterflies of a wave read and write only that wave's own 128 elements ([128 wave, +128)), and a wave's LDS
        // operations execute in order -- no workgroup barrier needed (round 5: 9 -> 3 barriers per frame; same butterflies, same operands, same bits)
        if (m >= 128) __syncthreads();
        else __builtin_amdgcn_wave_barrier();
    }
    for (int k = threadIdx.x; k < NBINS; k += 256) pw[k] = __fadd_rn(__fmul_rn(re[k], re[k]), __fmul_rn(im[k], im[k]));
    __syncthreads();
    if (threadIdx.x < DIAR_NMEL) {
        float sum = 0.0f;                                              // band of the triangular filter: same sum (see k_melframes)
        const int k_lo = p.fb_band[2 * threadIdx.x], k_hi = p.fb_band[2 * threadIdx.x + 1];
        for (int k = k_lo; k < k_hi; k++) sum = __fadd_rn(sum, __fmul_rn(p.fbT[k * DIAR_NMEL + threadIdx.x], pw[k]));
        out[threadIdx.x] = logf(__fadd_rn(sum, 5.960464477539063e-8f));
    } else if ((int)threadIdx.x < cpitch) {
        out[threadIdx.x] = 0.0f;
    }
}

__global__ __launch_bounds__(256) void k_diar_logmel(DiarMelParams p) {
    const int t = blockIdx.x, w = blockIdx.y;
    float *out = p.mel + ((size_t)w * p.T_pad + t) * p.cpitch;
    if (t >= p.t_valid) {
        for (int c = threadIdx.x; c < p.cpitch; c += 256) out[c] = 0.0f;
        return;
    }
    if (p.audio_s16) diar_frame(p, p.audio_s16 + p.win_off[w], p.n_win, t, out, p.cpitch);
    else diar_frame(p, p.audio + p.win_off[w], p.n_win, t, out, p.cpitch);
}

// the sliding-window VAD: frame t of the window at sample offset i*160 covers the same samples as frame t+1 of the window
// before it, and (away from the window's edges, where the zero padding and the restart of the pre-emphasis show) is the same
// frame -- so frames are computed per descriptor, once for all the windows that share them (nasr_diar.hip)
__global__ __launch_bounds__(256) void k_diar_frames(DiarMelParams p, const DiarFrameDesc *frames, float *out) {
    const DiarFrameDesc f = frames[blockIdx.x];
    if (p.audio_s16) diar_frame(p, p.audio_s16 + f.base, f.n, f.t, out + (size_t)blockIdx.x * DIAR_NMEL, DIAR_NMEL);
    else diar_frame(p, p.audio + f.base, f.n, f.t, out + (size_t)blockIdx.x * DIAR_NMEL, DIAR_NMEL);
}
void launch_diar_frames(const DiarMelParams &p, const DiarFrameDesc *frames, int n_frames, float *out, hipStream_t st) {
    if (n_frames > 0) hipLaunchKernelGGL(k_diar_frames, dim3(n_frames), dim3(256), 0, st, p, frames, out);
}

// per-feature normalisation over the t_valid frames (:182-199): mean and Bessel-corrected std in double, as the reference
__global__ __launch_bounds__(128) void k_diar_featnorm(DiarMelParams p) {
    const int w = blockIdx.x, m = threadIdx.x;
    if (m >= DIAR_NMEL) return;
    float *base = p.mel + (size_t)w * p.T_pad * p.cpitch + m;
    const int n_eff = p.t_valid, denom = n_eff - 1 > 1 ? n_eff - 1 : 1;
    double sum = 0.0;
    for (int t = 0; t < n_eff; t++) sum += (double)base[(size_t)t * p.cpitch];
    const float mean = (float)(sum / n_eff);
    double var = 0.0;
    for (int t = 0; t < n_eff; t++) { const float d = __fsub_rn(base[(size_t)t * p.cpitch], mean); var += (double)d * (double)d; }
    const float std_v = __fadd_rn(sqrtf((float)(var / denom)), 1e-5f);
    const float inv_std = 1.0f / std_v;
    for (int t = 0; t < n_eff; t++) base[(size_t)t * p.cpitch] = __fmul_rn(__fsub_rn(base[(size_t)t * p.cpitch], mean), inv_std);
}

void launch_diar_logmel(const DiarMelParams &p, int W, bool per_feature_normalize, hipStream_t st) {
    if (W <= 0) return;
    hipLaunchKernelGGL(k_diar_logmel, dim3(p.T_pad, W), dim3(256), 0, st, p);
    if (per_feature_normalize) hipLaunchKernelGGL(k_diar_featnorm, dim3(W), dim3(128), 0, st, p);
}

}  // namespace nasr

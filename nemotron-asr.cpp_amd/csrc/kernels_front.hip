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
        __syncthreads();
    }
    for (int k = threadIdx.x; k < NBINS; k += 256) {
        float mag = sqrtf(__fadd_rn(__fmul_rn(re[k], re[k]), __fmul_rn(im[k], im[k])));  // :201
        pw[k] = __fmul_rn(mag, mag);                                                        // :363-367
    }
    __syncthreads();
    if (threadIdx.x < NMEL) {
        float sum = 0.0f;
        for (int k = 0; k < NBINS; k++) sum = __fadd_rn(sum, __fmul_rn(p.fbT[k * NMEL + threadIdx.x], pw[k]));
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

// ---- conv0: 3x3 stride-2, 1 -> 256 channels, pad (2 before, 1 after) both axes, + ReLU ----
// (src/nemo-ggml.cpp:969-973, :905-913).  out [B][H1][W1][256], channel fastest.
__global__ __launch_bounds__(256) void k_sub_conv0(const RowDesc *rows, int chunk_mel, const float *mel_ring,
                                                   const float *w0t /*[9][256]*/, const float *b0, float *out,
                                                   int H1, int W1) {
    const int c = threadIdx.x;
    const int f = blockIdx.x % W1, t = blockIdx.x / W1, b = blockIdx.y;
    const RowDesc rd = rows[b];
    float acc = 0.0f;
#pragma unroll
    for (int kh = 0; kh < 3; kh++) {
        const int ih = 2 * t + kh - 2;
        if (ih < 0 || ih >= chunk_mel) continue;
        const float *mrow = mel_ring + ((size_t)rd.slot * MEL_RING + ((rd.mel_start + ih) & (MEL_RING - 1))) * NMEL;
#pragma unroll
        for (int kw = 0; kw < 3; kw++) {
            const int iw = 2 * f + kw - 2;
            if (iw < 0 || iw >= NMEL) continue;
            acc += w0t[(kh * 3 + kw) * SUBC + c] * mrow[iw];
        }
    }
    out[(((size_t)b * H1 + t) * W1 + f) * SUBC + c] = fmaxf(acc + b0[c], 0.0f);
}
void launch_sub_conv0(const RowDesc *rows, int B, int chunk_mel, const float *mel_ring, const float *w0t,
                      const float *b0, float *out, int H1, int W1, hipStream_t st) {
    hipLaunchKernelGGL(k_sub_conv0, dim3(H1 * W1, B), dim3(SUBC), 0, st, rows, chunk_mel, mel_ring, w0t, b0, out, H1, W1);
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
void launch_sub_dw(const float *in, int B, int Hin, int Win, const float *wt, const float *bias, void *out,
                   int out_bf16, hipStream_t st) {
    const int Hout = Hin / 2 + 1, Wout = Win / 2 + 1;
    if (out_bf16) hipLaunchKernelGGL(k_sub_dw<true>, dim3(Hout * Wout, B), dim3(SUBC), 0, st, in, Hin, Win, wt, bias, out, Hout, Wout);
    else hipLaunchKernelGGL(k_sub_dw<false>, dim3(Hout * Wout, B), dim3(SUBC), 0, st, in, Hin, Win, wt, bias, out, Hout, Wout);
}

}  // namespace nasr

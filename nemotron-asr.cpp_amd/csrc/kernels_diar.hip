// kernels_diar.hip -- diarization side-car.
//
// MarbleNet VAD (reference src/diarize_vad.cpp): the reference runs one ggml graph of ~150 nodes per 0.63 s window,
// 100 windows per second of audio per stream, each on its own.  The whole network is 90 k parameters and a window is
// 64 frames x <= 128 channels, so here ONE WORKGROUP runs the whole network for one window with the activations in
// LDS (three [64][128] f32 planes = 96 KiB of the CU's 160 KiB) and the launch covers every window of every stream:
// the sliding window is a batch dimension.  Activations are [T][C], channels innermost, as the reference feeds ggml.
#include "nasr_internal.h"
#include "nasr_wave.h"

namespace nasr {

constexpr int VT = VAD_T;          // 64 frames
constexpr int VC = 128;            // widest layer

// y[t][c] = sum_i x[t + i*dil - pad][c] * w[i][c] with rows outside [0, lens) read as zero (the zero 'same' padding
// and MaskedConv1d's input mask, src/diarize_vad.cpp:232-251, :283-297); term order as the reference's graph
__device__ __forceinline__ void vad_depthwise(const float *x, float *y, const VadSub &s, int lens) {
    const int C = s.cin, pad = s.dil * (s.kernel - 1) / 2;
    for (int e = threadIdx.x; e < VT * C; e += 256) {
        const int t = e / C, c = e - t * C;
        float acc = 0.0f;
        for (int i = 0; i < s.kernel; i++) {
            const int tt = t + i * s.dil - pad;
            const float v = (tt >= 0 && tt < lens) ? x[tt * VC + c] : 0.0f;
            const float prod = v * s.dw[i * C + c];
            acc = i == 0 ? prod : acc + prod;
        }
        y[t * VC + c] = acc;
    }
}

// y[t][o] = (sum_i w[o][i] * mask(t) x[t][i]) * scale[o] + bias[o]  (+ residual) (relu)   (:226-229, :255-265)
// thread -> one output channel (its weight row lives in registers) and a contiguous group of frames; the x row is an
// LDS broadcast.  ADD: accumulate into y (the residual branch lands on top of the main branch).
template <bool ADD, bool RELU>
__device__ __forceinline__ void vad_pointwise(const float *x, float *y, const VadSub &s, int lens) {
    const int Cin = s.cin, Cout = s.cout;
    const int groups = 256 / Cout, o = threadIdx.x % Cout, g = threadIdx.x / Cout;   // Cout in {64, 128}
    const int per = VT / groups, t0 = g * per;
    float wreg[VC];
#pragma unroll
    for (int i = 0; i < VC; i++) wreg[i] = i < Cin ? s.pw[o * Cin + i] : 0.0f;
    const float sc = s.scale[o], bi = s.bias[o];
    for (int t = t0; t < t0 + per; t++) {
        float acc = 0.0f;
        if (t < lens) {
            const float *xr = x + t * VC;
#pragma unroll
            for (int i = 0; i < VC; i++)
                if (i < Cin) acc += wreg[i] * xr[i];
        }
        float v = acc * sc + bi;
        if (ADD) v += y[t * VC + o];
        if (RELU) v = fmaxf(v, 0.0f);
        y[t * VC + o] = v;
    }
}

__global__ __launch_bounds__(256) void k_vad_marblenet(VadNet net, const float *mel, const int *lens_mel, float *prob) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *A = lds, *B = lds + VT * VC, *Tm = lds + 2 * VT * VC;
    __shared__ float mean_s[VC];
    const int w = blockIdx.x;
    int lens = lens_mel[w];
    lens = lens < 0 ? 0 : (lens > VAD_TVALID ? VAD_TVALID : lens);             // :447-450
    for (int e = threadIdx.x; e < VT * DIAR_NMEL; e += 256) {
        const int t = e / DIAR_NMEL, c = e - t * DIAR_NMEL;
        A[t * VC + c] = mel[((size_t)w * VT + t) * DIAR_NMEL + c];
    }
    __syncthreads();
    // JasperBlock: sub-convs (ReLU between them), + residual, ReLU (:300-318).  Three LDS planes: `in` holds the block
    // input until the residual has used it, the depthwise result goes to f1, the pointwise result to f2 (the second
    // sub-conv of a block reads f2 through its depthwise conv first, so it may overwrite it).
    float *in = A, *f1 = B, *f2 = Tm;
    int si = 0;
    const int repeat[6] = {1, 2, 2, 2, 1, 1};
#pragma unroll 1
    for (int b = 0; b < 6; b++) {
        const float *x = in;
        const bool has_res = b >= 1 && b <= 3;
        for (int r = 0; r < repeat[b]; r++, si++) {
            const VadSub &s = net.sub[si];
            const float *pin = x;
            if (s.dw) {
                vad_depthwise(x, f1, s, lens);
                __syncthreads();
                pin = f1;
            }
            const bool last = r + 1 == repeat[b];
            if (!last || !has_res) vad_pointwise<false, true>(pin, f2, s, lens);     // ReLU between sub-convs / block without residual
            else vad_pointwise<false, false>(pin, f2, s, lens);
            __syncthreads();
            x = f2;
        }
        if (has_res) {
            vad_pointwise<true, true>(in, f2, net.res[b - 1], lens);                 // + BN(pw(mask(x_in))), ReLU (:305-311)
            __syncthreads();
        }
        float *old_in = in;
        in = f2;
        f2 = old_in;
    }
    A = in;
    // AdaptiveAvgPool1d(1) over all 64 frames, Linear(128 -> 2), softmax, P(speech) (:462-487)
    if (threadIdx.x < VC) {
        float sum = 0.0f;
        for (int t = 0; t < VT; t++) sum += A[t * VC + threadIdx.x];
        mean_s[threadIdx.x] = sum * (1.0f / (float)VT);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lg[2];
        for (int k = 0; k < 2; k++) {
            float v = net.dec_b[k];
            for (int c = 0; c < VC; c++) v += net.dec_w[k * VC + c] * mean_s[c];
            lg[k] = v;
        }
        const float mx = fmaxf(lg[0], lg[1]);
        const float e0 = expf(lg[0] - mx), e1 = expf(lg[1] - mx);
        prob[w] = e1 / (e0 + e1);
    }
}

void init_diar_kernel_attributes() {
    hipFuncSetAttribute((const void *)k_vad_marblenet, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * VT * VC * 4);
}
void launch_vad_marblenet(const VadNet &net, const float *mel, const int *lens_mel, float *prob, int W, hipStream_t st) {
    if (W <= 0) return;
    hipLaunchKernelGGL(k_vad_marblenet, dim3(W), dim3(256), 3 * VT * VC * 4, st, net, mel, lens_mel, prob);
}

}  // namespace nasr

/*
 * nasr_oracle.c -- CPU restatement of the reference streaming forward path.
 * TEST INFRASTRUCTURE ONLY (see nasr_oracle.h).  Plain C + OpenMP.
 *
 * Every function names the reference lines it follows (paths relative to the root of
 * m1el/nemotron-asr.cpp).  Layout convention: row-major, [rows][features].
 */
#include "nasr_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------ */
/* small helpers                                                                   */
/* ------------------------------------------------------------------------------ */

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* round-to-nearest-even f32 -> bf16 -> f32 (what v_cvt_pk_bf16_f32 does for finite x) */
float orc_round_bf16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) { /* NaN stays NaN */
        u |= 0x00400000u;
        u &= 0xffff0000u;
    } else {
        u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    }
    float r;
    memcpy(&r, &u, 4);
    return r;
}

static void round_bf16_inplace(float *x, int64_t n) {
    for (int64_t i = 0; i < n; i++) x[i] = orc_round_bf16(x[i]);
}

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "nasr_oracle: out of memory (%zu bytes)\n", n); abort(); }
    return p;
}
static void *xcalloc(size_t n, size_t sz) {
    void *p = calloc(n ? n : 1, sz);
    if (!p) { fprintf(stderr, "nasr_oracle: out of memory\n"); abort(); }
    return p;
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline float siluf_(float x) { return x * sigmoidf_(x); }

/* out[m][n] = (bias ? bias[n] : 0) + sum_k x[m][k] * w[n][k]
 * (ggml_mul_mat(w, x) with w stored [out][in]; src/reference/ops.cpp:28-58) */
static void gemm_nt(const float *x, int M, int K, const float *w, int N,
                    const float *bias, float *out) {
#pragma omp parallel for schedule(static) if ((size_t)N * K * M > 200000)
    for (int n = 0; n < N; n++) {
        const float *wr = w + (size_t)n * K;
        for (int m = 0; m < M; m++) {
            const float *xr = x + (size_t)m * K;
            float sum = 0.0f;
#pragma omp simd reduction(+ : sum)
            for (int k = 0; k < K; k++) sum += xr[k] * wr[k];
            out[(size_t)m * N + n] = bias ? (bias[n] + sum) : sum;
        }
    }
}

/* ---- ggml-CPU semantics of a mul_mat with a Q8_0 weight (emulation, see nasr_oracle.h "Q8_0 activations") ----------
 * ggml is an un-vendored, empty submodule of the reference (pin unrecoverable), so this follows ggml's PUBLISHED scalar
 * algorithm, not code in /root/reference: for src0 = Q8_0 weights and src1 = f32 activations, ggml_compute_forward_mul_mat
 * converts every activation row to the weight type's vec_dot_type (Q8_0) with quantize_row_q8_0 and calls
 * ggml_vec_dot_q8_0_q8_0 per (row, column):
 *   quantize_row_q8_0:  per block of 32: amax = max |x|, d = amax / 127, id = d ? 1/d : 0, q_i = roundf(x_i * id)
 *                       (round half away from zero), block scale stored as fp16(d)
 *   vec_dot_q8_0_q8_0:  sumf += (float)(sum_i qw_i * qa_i) * (fp16->f32(dw) * fp16->f32(da)), blocks in ascending order
 * (the SIMD paths accumulate the same products in 8 float lanes; differences at the 1e-7 relative level).
 * Block layout = scripts/convert_to_gguf.py:118-154 of the reference: 34 bytes = fp16 d + 32 int8. */
static float f16_bits_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu, f;
    if (exp == 0) {
        if (man == 0) f = sign;
        else {
            exp = 127 - 15 + 1;
            while (!(man & 0x400u)) { man <<= 1; exp--; }
            man &= 0x3ffu;
            f = sign | (exp << 23) | (man << 13);
        }
    } else if (exp == 31) f = sign | 0x7f800000u | (man << 13);
    else f = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float r;
    memcpy(&r, &f, 4);
    return r;
}
static uint16_t f32_to_f16_bits(float x) { /* round to nearest even, like GGML_FP32_TO_FP16 (F16C / _cvtss_sh) */
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0));
    if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);         /* >= 65520 rounds to inf */
    if (u < 0x33000001u) return (uint16_t)sign;                      /* < 2^-25 (+ ties to even) -> 0 */
    int e = (int)(u >> 23) - 127;
    uint32_t man = (u & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? 13 + (-14 - e) : 13;                       /* subnormal halves lose more bits */
    uint32_t half_man = man >> shift, rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_man & 1u))) half_man++;
    uint32_t out = e < -14 ? half_man : (((uint32_t)(e + 15) << 10) + (half_man - 0x400u));
    return (uint16_t)(sign | out);
}

static void gemm_q8_0(const float *x, int M, int K, const uint8_t *wq, int N, const float *bias, float *out) {
    const int nb = K / 32;
    int8_t *aq = (int8_t *)xmalloc((size_t)M * K);
    float *ad = (float *)xmalloc(sizeof(float) * (size_t)M * nb);
    for (int m = 0; m < M; m++)
        for (int b = 0; b < nb; b++) {
            const float *xb = x + (size_t)m * K + b * 32;
            float amax = 0.0f;
            for (int i = 0; i < 32; i++) { float v = fabsf(xb[i]); if (v > amax) amax = v; }
            const float d = amax / 127.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
            ad[(size_t)m * nb + b] = f16_bits_to_f32(f32_to_f16_bits(d));
            for (int i = 0; i < 32; i++) aq[(size_t)m * K + b * 32 + i] = (int8_t)roundf(xb[i] * id);
        }
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        const uint8_t *wr = wq + (size_t)n * nb * 34;
        for (int m = 0; m < M; m++) {
            float sumf = 0.0f;
            for (int b = 0; b < nb; b++) {
                uint16_t hd;
                memcpy(&hd, wr + (size_t)b * 34, 2);
                const int8_t *qw = (const int8_t *)(wr + (size_t)b * 34 + 2), *qa = aq + (size_t)m * K + b * 32;
                int sumi = 0;
                for (int i = 0; i < 32; i++) sumi += (int)qw[i] * (int)qa[i];
                sumf += (float)sumi * (f16_bits_to_f32(hd) * ad[(size_t)m * nb + b]);
            }
            out[(size_t)m * N + n] = bias ? (bias[n] + sumf) : sumf;
        }
    }
    free(aq); free(ad);
}

/* LayerNorm, biased variance, eps literal 1e-5 (src/nemo-stream.cpp:580-591,
 * src/reference/ops.cpp:72-108) */
static void layer_norm_rows(const float *x, int M, int D, const float *w, const float *b,
                            float *out) {
    for (int m = 0; m < M; m++) {
        const float *xr = x + (size_t)m * D;
        float *o = out + (size_t)m * D;
        float mean = 0.0f;
        for (int i = 0; i < D; i++) mean += xr[i];
        mean /= (float)D;
        float var = 0.0f;
        for (int i = 0; i < D; i++) {
            float d = xr[i] - mean;
            var += d * d;
        }
        var /= (float)D;
        float inv = 1.0f / sqrtf(var + 1e-5f);
        for (int i = 0; i < D; i++) o[i] = (xr[i] - mean) * inv * w[i] + b[i];
    }
}

/* ------------------------------------------------------------------------------ */
/* model                                                                           */
/* ------------------------------------------------------------------------------ */

typedef struct {
    const float *norm_ff1_w, *norm_ff1_b, *ff1_w1, *ff1_w2;
    const float *norm_att_w, *norm_att_b, *wq, *wk, *wv, *wpos, *wout, *bias_u, *bias_v;
    const float *norm_conv_w, *norm_conv_b, *pw1, *dw, *conv_ln_w, *conv_ln_b, *pw2;
    const float *norm_ff2_w, *norm_ff2_b, *ff2_w1, *ff2_w2;
    const float *norm_out_w, *norm_out_b;
    const float *wpos_f32; /* un-rounded linear_pos (P is computed in f32, then rounded) */
} orc_layer;

typedef struct { const float *key; const uint8_t *blocks; } q8_entry;

struct orc_model {
    int n_layers, kernel_size, num_prompts, emulate_bf16;
    int emulate_q8_act;            /* ggml-CPU Q8_0 mul_mat semantics for tensors registered with orc_model_set_tensor_q8_0 */
    q8_entry *q8; int n_q8, cap_q8;
    const float *fb, *window;
    const float *conv0_w, *conv0_b, *conv2_w, *conv2_b, *conv3_w, *conv3_b;
    const float *conv5_w, *conv5_b, *conv6_w, *conv6_b, *sub_out_w, *sub_out_b;
    orc_layer *layers;
    const float *embed, *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2];
    const float *jenc_w, *jenc_b, *jpred_w, *jpred_b, *jout_w, *jout_b;
    const float *pk1_w, *pk1_b, *pk2_w, *pk2_b;
    /* owned rounded copies */
    float **owned;
    int n_owned, cap_owned;
};

static const uint8_t *q8_of(const orc_model *m, const float *w) {
    if (!m->emulate_q8_act) return NULL;
    for (int i = 0; i < m->n_q8; i++)
        if (m->q8[i].key == w) return m->q8[i].blocks;
    return NULL;
}

/* the linear layers of the encoder: f32 GEMM, or the Q8_0 x Q8_0 product when the weight was registered as blocks */
static void gemm_w(const orc_model *m, const float *x, int M, int K, const float *w, int N, const float *bias, float *out) {
    const uint8_t *q = q8_of(m, w);
    if (q) gemm_q8_0(x, M, K, q, N, bias, out);
    else gemm_nt(x, M, K, w, N, bias, out);
}

orc_model *orc_model_create(int n_layers, int kernel_size, int num_prompts, int emulate) {
    orc_model *m = (orc_model *)xcalloc(1, sizeof(*m));
    m->n_layers = n_layers;
    m->kernel_size = kernel_size;
    m->num_prompts = num_prompts;
    m->emulate_bf16 = (emulate & ORC_EMU_BF16) ? 1 : 0;
    m->emulate_q8_act = (emulate & ORC_EMU_Q8_ACT) ? 1 : 0;
    m->layers = (orc_layer *)xcalloc((size_t)n_layers, sizeof(orc_layer));
    return m;
}

void orc_model_free(orc_model *m) {
    if (!m) return;
    for (int i = 0; i < m->n_owned; i++) free(m->owned[i]);
    free(m->owned);
    free(m->layers);
    free(m->q8);
    free(m);
}

static const float *own_rounded(orc_model *m, const float *data, int64_t n) {
    float *c = (float *)xmalloc((size_t)n * sizeof(float));
    memcpy(c, data, (size_t)n * sizeof(float));
    round_bf16_inplace(c, n);
    if (m->n_owned == m->cap_owned) {
        m->cap_owned = m->cap_owned ? 2 * m->cap_owned : 64;
        m->owned = (float **)realloc(m->owned, (size_t)m->cap_owned * sizeof(float *));
    }
    m->owned[m->n_owned++] = c;
    return c;
}

/* name -> slot table (names: src/nemo-ggml.cpp:296-398) */
typedef struct { const char *suffix; size_t off; int64_t numel; int round; } layer_slot;
#define LOFF(f) offsetof(orc_layer, f)
static const layer_slot k_layer_slots[] = {
    {"norm_feed_forward1.weight", LOFF(norm_ff1_w), 1024, 0},
    {"norm_feed_forward1.bias", LOFF(norm_ff1_b), 1024, 0},
    {"feed_forward1.linear1.weight", LOFF(ff1_w1), 4096 * 1024, 1},
    {"feed_forward1.linear2.weight", LOFF(ff1_w2), 1024 * 4096, 1},
    {"norm_self_att.weight", LOFF(norm_att_w), 1024, 0},
    {"norm_self_att.bias", LOFF(norm_att_b), 1024, 0},
    {"self_attn.linear_q.weight", LOFF(wq), 1024 * 1024, 1},
    {"self_attn.linear_k.weight", LOFF(wk), 1024 * 1024, 1},
    {"self_attn.linear_v.weight", LOFF(wv), 1024 * 1024, 1},
    {"self_attn.linear_pos.weight", LOFF(wpos), 1024 * 1024, 2},
    {"self_attn.linear_out.weight", LOFF(wout), 1024 * 1024, 1},
    {"self_attn.pos_bias_u", LOFF(bias_u), 1024, 0},
    {"self_attn.pos_bias_v", LOFF(bias_v), 1024, 0},
    {"norm_conv.weight", LOFF(norm_conv_w), 1024, 0},
    {"norm_conv.bias", LOFF(norm_conv_b), 1024, 0},
    {"conv.pointwise_conv1.weight", LOFF(pw1), 2048 * 1024, 1},
    {"conv.depthwise_conv.weight", LOFF(dw), -1, 0},
    {"conv.batch_norm.weight", LOFF(conv_ln_w), 1024, 0},
    {"conv.batch_norm.bias", LOFF(conv_ln_b), 1024, 0},
    {"conv.pointwise_conv2.weight", LOFF(pw2), 1024 * 1024, 1},
    {"norm_feed_forward2.weight", LOFF(norm_ff2_w), 1024, 0},
    {"norm_feed_forward2.bias", LOFF(norm_ff2_b), 1024, 0},
    {"feed_forward2.linear1.weight", LOFF(ff2_w1), 4096 * 1024, 1},
    {"feed_forward2.linear2.weight", LOFF(ff2_w2), 1024 * 4096, 1},
    {"norm_out.weight", LOFF(norm_out_w), 1024, 0},
    {"norm_out.bias", LOFF(norm_out_b), 1024, 0},
};

typedef struct { const char *name; size_t off; int64_t numel; int round; } model_slot;
#define MOFF(f) offsetof(orc_model, f)
static const model_slot k_model_slots[] = {
    {"preprocessor.featurizer.fb", MOFF(fb), 128 * 257, 0},
    {"preprocessor.featurizer.window", MOFF(window), 400, 0},
    {"encoder.pre_encode.conv.0.weight", MOFF(conv0_w), 256 * 9, 0},
    {"encoder.pre_encode.conv.0.bias", MOFF(conv0_b), 256, 0},
    {"encoder.pre_encode.conv.2.weight", MOFF(conv2_w), 256 * 9, 0},
    {"encoder.pre_encode.conv.2.bias", MOFF(conv2_b), 256, 0},
    {"encoder.pre_encode.conv.3.weight", MOFF(conv3_w), 256 * 256, 1},
    {"encoder.pre_encode.conv.3.bias", MOFF(conv3_b), 256, 0},
    {"encoder.pre_encode.conv.5.weight", MOFF(conv5_w), 256 * 9, 0},
    {"encoder.pre_encode.conv.5.bias", MOFF(conv5_b), 256, 0},
    {"encoder.pre_encode.conv.6.weight", MOFF(conv6_w), 256 * 256, 1},
    {"encoder.pre_encode.conv.6.bias", MOFF(conv6_b), 256, 0},
    {"encoder.pre_encode.out.weight", MOFF(sub_out_w), 1024 * 4352, 1},
    {"encoder.pre_encode.out.bias", MOFF(sub_out_b), 1024, 0},
    {"decoder.prediction.embed.weight", MOFF(embed), 1025 * 640, 0},
    {"decoder.prediction.dec_rnn.lstm.weight_ih_l0", MOFF(w_ih[0]), 2560 * 640, 0},
    {"decoder.prediction.dec_rnn.lstm.weight_hh_l0", MOFF(w_hh[0]), 2560 * 640, 0},
    {"decoder.prediction.dec_rnn.lstm.bias_ih_l0", MOFF(b_ih[0]), 2560, 0},
    {"decoder.prediction.dec_rnn.lstm.bias_hh_l0", MOFF(b_hh[0]), 2560, 0},
    {"decoder.prediction.dec_rnn.lstm.weight_ih_l1", MOFF(w_ih[1]), 2560 * 640, 0},
    {"decoder.prediction.dec_rnn.lstm.weight_hh_l1", MOFF(w_hh[1]), 2560 * 640, 0},
    {"decoder.prediction.dec_rnn.lstm.bias_ih_l1", MOFF(b_ih[1]), 2560, 0},
    {"decoder.prediction.dec_rnn.lstm.bias_hh_l1", MOFF(b_hh[1]), 2560, 0},
    {"joint.enc.weight", MOFF(jenc_w), 640 * 1024, 0},
    {"joint.enc.bias", MOFF(jenc_b), 640, 0},
    {"joint.pred.weight", MOFF(jpred_w), 640 * 640, 0},
    {"joint.pred.bias", MOFF(jpred_b), 640, 0},
    {"joint.joint_net.2.weight", MOFF(jout_w), 1025 * 640, 0},
    {"joint.joint_net.2.bias", MOFF(jout_b), 1025, 0},
    {"prompt_kernel.0.weight", MOFF(pk1_w), -2, 0},
    {"prompt_kernel.0.bias", MOFF(pk1_b), 2048, 0},
    {"prompt_kernel.2.weight", MOFF(pk2_w), 1024 * 2048, 0},
    {"prompt_kernel.2.bias", MOFF(pk2_b), 1024, 0},
};

int orc_model_set_tensor(orc_model *m, const char *name, const float *data, int64_t numel) {
    static const char *lp = "encoder.layers.";
    if (strncmp(name, lp, strlen(lp)) == 0) {
        char *end = NULL;
        long l = strtol(name + strlen(lp), &end, 10);
        if (!end || *end != '.' || l < 0) return -1;
        if (l >= m->n_layers) return 0; /* tolerated: model built with fewer layers */
        const char *suffix = end + 1;
        for (size_t i = 0; i < sizeof(k_layer_slots) / sizeof(k_layer_slots[0]); i++) {
            const layer_slot *s = &k_layer_slots[i];
            if (strcmp(suffix, s->suffix) != 0) continue;
            int64_t want = s->numel == -1 ? (int64_t)m->kernel_size * 1024 : s->numel;
            if (numel != want) {
                fprintf(stderr, "nasr_oracle: %s: numel %lld, expected %lld\n", name,
                        (long long)numel, (long long)want);
                return -1;
            }
            orc_layer *L = &m->layers[l];
            const float **slot = (const float **)((char *)L + s->off);
            if (s->round == 2) L->wpos_f32 = data;
            *slot = (m->emulate_bf16 && s->round == 1) ? own_rounded(m, data, numel) : data;
            return 0;
        }
        return -1;
    }
    for (size_t i = 0; i < sizeof(k_model_slots) / sizeof(k_model_slots[0]); i++) {
        const model_slot *s = &k_model_slots[i];
        if (strcmp(name, s->name) != 0) continue;
        int64_t want = s->numel == -2 ? (int64_t)2048 * (1024 + m->num_prompts) : s->numel;
        if (numel != want) {
            fprintf(stderr, "nasr_oracle: %s: numel %lld, expected %lld\n", name,
                    (long long)numel, (long long)want);
            return -1;
        }
        const float **slot = (const float **)((char *)m + s->off);
        *slot = (m->emulate_bf16 && s->round == 1) ? own_rounded(m, data, numel) : data;
        return 0;
    }
    return -1;
}

/* registers the Q8_0 blocks of an encoder-layer matrix that orc_model_set_tensor() already received (dequantised values) */
int orc_model_set_tensor_q8_0(orc_model *m, const char *name, const uint8_t *blocks, int64_t numel) {
    static const char *lp = "encoder.layers.";
    if (strncmp(name, lp, strlen(lp)) != 0 || numel % 32) return -1;
    char *end = NULL;
    long l = strtol(name + strlen(lp), &end, 10);
    if (!end || *end != '.' || l < 0) return -1;
    if (l >= m->n_layers) return 0;
    for (size_t i = 0; i < sizeof(k_layer_slots) / sizeof(k_layer_slots[0]); i++) {
        const layer_slot *sl = &k_layer_slots[i];
        if (strcmp(end + 1, sl->suffix) != 0) continue;
        if (sl->numel != numel) return -1;
        const float *key = *(const float **)((char *)&m->layers[l] + sl->off);
        if (!key) return -1;
        if (sl->round == 2) key = m->layers[l].wpos_f32;     /* linear_pos is used through wpos_f32 */
        if (m->n_q8 == m->cap_q8) {
            m->cap_q8 = m->cap_q8 ? 2 * m->cap_q8 : 64;
            m->q8 = (q8_entry *)realloc(m->q8, (size_t)m->cap_q8 * sizeof(q8_entry));
        }
        m->q8[m->n_q8].key = key;
        m->q8[m->n_q8].blocks = blocks;
        m->n_q8++;
        return 0;
    }
    return -1;
}

int orc_model_finalize(orc_model *m) {
    int missing = 0;
    for (size_t i = 0; i < sizeof(k_model_slots) / sizeof(k_model_slots[0]); i++) {
        const model_slot *s = &k_model_slots[i];
        if (strncmp(s->name, "prompt_kernel", 13) == 0 && m->num_prompts == 0) continue;
        if (*(const float **)((char *)m + s->off) == NULL) {
            fprintf(stderr, "nasr_oracle: missing tensor %s\n", s->name);
            missing = 1;
        }
    }
    for (int l = 0; l < m->n_layers; l++)
        for (size_t i = 0; i < sizeof(k_layer_slots) / sizeof(k_layer_slots[0]); i++) {
            const layer_slot *s = &k_layer_slots[i];
            if (*(const float **)((char *)&m->layers[l] + s->off) == NULL) {
                fprintf(stderr, "nasr_oracle: missing tensor encoder.layers.%d.%s\n", l, s->suffix);
                missing = 1;
            }
        }
    return missing ? -1 : 0;
}

/* ------------------------------------------------------------------------------ */
/* a-1: preprocessor (src/preprocessor.cpp)                                        */
/* ------------------------------------------------------------------------------ */

struct orc_preproc {
    float window[ORC_N_FFT];      /* Hann-400 centred in 512, :296-299 */
    float fb[ORC_N_MELS * ORC_N_BINS];
    float sin_t[ORC_N_FFT], cos_t[ORC_N_FFT]; /* :86-90 */
    int bit_rev[ORC_N_FFT];                   /* :96-105 */
    float last_sample;                        /* :57, :350-356 */
    float *abuf;                              /* audio_buf, pre-seeded with 256 zeros :220-221 */
    int n_abuf, cap_abuf;
};

void orc_preproc_reset(orc_preproc *pp) {
    pp->last_sample = 0.0f;
    pp->n_abuf = ORC_N_FFT / 2;
    memset(pp->abuf, 0, sizeof(float) * (size_t)pp->n_abuf);
}

orc_preproc *orc_preproc_create(const float *fb, const float *window) {
    orc_preproc *pp = (orc_preproc *)xcalloc(1, sizeof(*pp));
    const int pad = (ORC_N_FFT - ORC_WIN) / 2;
    memcpy(pp->window + pad, window, sizeof(float) * ORC_WIN);
    memcpy(pp->fb, fb, sizeof(pp->fb));
    const int n = ORC_N_FFT;
    for (int i = 0; i < n; i++) {
        float theta = (2.0f * (float)M_PI * (float)i) / (float)n;
        pp->sin_t[i] = sinf(theta);
        pp->cos_t[i] = cosf(theta);
    }
    for (int i = 0; i < n; i++) {
        int r = 0, x = i;
        for (int j = 0; j < 9; j++) { r = (r << 1) | (x & 1); x >>= 1; }
        pp->bit_rev[i] = r;
    }
    pp->cap_abuf = 4096;
    pp->abuf = (float *)xmalloc(sizeof(float) * (size_t)pp->cap_abuf);
    orc_preproc_reset(pp);
    return pp;
}

void orc_preproc_free(orc_preproc *pp) {
    if (!pp) return;
    free(pp->abuf);
    free(pp);
}

/* radix-2 DIT FFT, same butterfly order as src/preprocessor.cpp:113-161 */
static void fft512(const orc_preproc *pp, const float *frame, float *re, float *im) {
    const int n = ORC_N_FFT;
    for (int i = 0; i < n; i++) { re[pp->bit_rev[i]] = frame[i]; im[pp->bit_rev[i]] = 0.0f; }
    for (int m = 2; m <= n; m <<= 1) {
        int m2 = m >> 1, step = n / m;
        for (int k = 0; k < n; k += m)
            for (int j = 0; j < m2; j++) {
                float wr = pp->cos_t[j * step], wi = -pp->sin_t[j * step];
                int i1 = k + j, i2 = k + j + m2;
                float tr = wr * re[i2] - wi * im[i2];
                float ti = wr * im[i2] + wi * re[i2];
                re[i2] = re[i1] - tr;
                im[i2] = im[i1] - ti;
                re[i1] = re[i1] + tr;
                im[i1] = im[i1] + ti;
            }
    }
}

int orc_preproc_process(orc_preproc *pp, const int16_t *pcm, int n_samples, float *mel_out,
                        int cap_frames) {
    if (n_samples <= 0) return 0; /* :336-339 */
    int avail = pp->n_abuf + n_samples;
    int n_frames = avail < ORC_N_FFT ? 0 : (avail - ORC_N_FFT + ORC_HOP) / ORC_HOP; /* :320-328 */
    if (n_frames > cap_frames) {
        fprintf(stderr, "nasr_oracle: mel_out too small (%d > %d)\n", n_frames, cap_frames);
        return -1;
    }
    if (avail > pp->cap_abuf) {
        pp->cap_abuf = avail + 4096;
        pp->abuf = (float *)realloc(pp->abuf, sizeof(float) * (size_t)pp->cap_abuf);
    }
    /* s16 -> f32 /32768, pre-emphasis with carried last_sample, :349-356 */
    const float scale = 1.0f / 32768.0f;
    float prev = pp->last_sample;
    for (int i = 0; i < n_samples; i++) {
        float curr = (float)pcm[i] * scale;
        pp->abuf[pp->n_abuf + i] = curr - 0.97f * prev;
        prev = curr;
    }
    pp->last_sample = prev;
    pp->n_abuf = avail;

    float frame[ORC_N_FFT], re[ORC_N_FFT], im[ORC_N_FFT], power[ORC_N_BINS];
    for (int t = 0; t < n_frames; t++) {
        const float *src = pp->abuf + (size_t)t * ORC_HOP;
        for (int i = 0; i < ORC_N_FFT; i++) frame[i] = src[i] * pp->window[i]; /* :184-194 */
        fft512(pp, frame, re, im);
        for (int k = 0; k < ORC_N_BINS; k++) {
            float mag = sqrtf(re[k] * re[k] + im[k] * im[k]); /* :201 */
            power[k] = mag * mag;                             /* :363-367 */
        }
        for (int mI = 0; mI < ORC_N_MELS; mI++) { /* :374-383 */
            const float *f = pp->fb + (size_t)mI * ORC_N_BINS;
            float sum = 0.0f;
            for (int k = 0; k < ORC_N_BINS; k++) sum += f[k] * power[k];
            mel_out[(size_t)t * ORC_N_MELS + mI] = logf(sum + 5.960464477539063e-8f);
        }
    }
    /* erase consumed samples, :389-393 */
    int consumed = n_frames * ORC_HOP;
    memmove(pp->abuf, pp->abuf + consumed, sizeof(float) * (size_t)(pp->n_abuf - consumed));
    pp->n_abuf -= consumed;
    return n_frames;
}

/* ------------------------------------------------------------------------------ */
/* a-2: conv subsampling (src/nemo-ggml.cpp:897-1029)                               */
/* ------------------------------------------------------------------------------ */

static inline int sub_out_len(int n) { return n / 2 + 1; } /* (n + 3 - 3)/2 + 1 */

/* depthwise (or 1-input-channel when cin_is_one) 3x3 stride-2 conv, pad (2 before, 1 after)
 * on both axes (:905-913, :936-943).  in [Hin][Win][C or 1], out [Hout][Wout][C]. */
static void conv3x3_s2(const float *in, int Hin, int Win, int C, int cin_is_one,
                       const float *w /*[C][3][3]*/, const float *b, int relu, float *out) {
    int Hout = sub_out_len(Hin), Wout = sub_out_len(Win);
#pragma omp parallel for schedule(static)
    for (int t = 0; t < Hout; t++)
        for (int f = 0; f < Wout; f++) {
            float *o = out + ((size_t)t * Wout + f) * C;
            for (int c = 0; c < C; c++) o[c] = 0.0f;
            for (int kh = 0; kh < 3; kh++) {
                int ih = 2 * t + kh - 2;
                if (ih < 0 || ih >= Hin) continue;
                for (int kw = 0; kw < 3; kw++) {
                    int iw = 2 * f + kw - 2;
                    if (iw < 0 || iw >= Win) continue;
                    if (cin_is_one) {
                        float v = in[(size_t)ih * Win + iw];
                        for (int c = 0; c < C; c++) o[c] += w[c * 9 + kh * 3 + kw] * v;
                    } else {
                        const float *iv = in + ((size_t)ih * Win + iw) * C;
                        for (int c = 0; c < C; c++) o[c] += w[c * 9 + kh * 3 + kw] * iv[c];
                    }
                }
            }
            for (int c = 0; c < C; c++) {
                float v = o[c] + b[c];
                o[c] = (relu && v < 0.0f) ? 0.0f : v;
            }
        }
}

/* 1x1 conv + bias + ReLU over positions: in/out [P][256], w [256][256] (:983-1007) */
static void conv1x1_relu(const float *in, int P, const float *w, const float *b, float *out) {
    gemm_nt(in, P, ORC_SUB_CH, w, ORC_SUB_CH, b, out);
    for (size_t i = 0; i < (size_t)P * ORC_SUB_CH; i++)
        if (out[i] < 0.0f) out[i] = 0.0f;
}

int orc_subsampling(const orc_model *m, const float *mel, int n_frames, float *out) {
    const int C = ORC_SUB_CH;
    int H1 = sub_out_len(n_frames), W1 = sub_out_len(ORC_N_MELS); /* 65 */
    int H2 = sub_out_len(H1), W2 = sub_out_len(W1);               /* 33 */
    int H3 = sub_out_len(H2), W3 = sub_out_len(W2);               /* 17 */
    float *a = (float *)xmalloc(sizeof(float) * (size_t)H1 * W1 * C);
    float *b = (float *)xmalloc(sizeof(float) * (size_t)H1 * W1 * C);
    conv3x3_s2(mel, n_frames, ORC_N_MELS, C, 1, m->conv0_w, m->conv0_b, 1, a); /* conv0+ReLU */
    conv3x3_s2(a, H1, W1, C, 0, m->conv2_w, m->conv2_b, 0, b);                /* dw conv2 */
    if (m->emulate_bf16) round_bf16_inplace(b, (int64_t)H2 * W2 * C);          /* GEMM operand */
    conv1x1_relu(b, H2 * W2, m->conv3_w, m->conv3_b, a);                       /* pw conv3+ReLU */
    conv3x3_s2(a, H2, W2, C, 0, m->conv5_w, m->conv5_b, 0, b);                /* dw conv5 */
    if (m->emulate_bf16) round_bf16_inplace(b, (int64_t)H3 * W3 * C);
    conv1x1_relu(b, H3 * W3, m->conv6_w, m->conv6_b, a);                       /* pw conv6+ReLU */
    /* flatten flat[t][c*17 + w] (:1014-1017) */
    float *flat = (float *)xmalloc(sizeof(float) * (size_t)H3 * ORC_SUB_FLAT);
    for (int t = 0; t < H3; t++)
        for (int wv = 0; wv < W3; wv++)
            for (int c = 0; c < C; c++)
                flat[(size_t)t * ORC_SUB_FLAT + c * W3 + wv] = a[((size_t)t * W3 + wv) * C + c];
    if (m->emulate_bf16) round_bf16_inplace(flat, (int64_t)H3 * ORC_SUB_FLAT);
    gemm_nt(flat, H3, ORC_SUB_FLAT, m->sub_out_w, ORC_D_MODEL, m->sub_out_b, out); /* :1020-1023 */
    free(a); free(b); free(flat);
    return H3;
}

/* ------------------------------------------------------------------------------ */
/* a-7: sinusoid (src/nemo-ggml.cpp:17-32): [sin(p w_0), cos(p w_0), sin(p w_2), ...]   */
/* ------------------------------------------------------------------------------ */
void orc_pos_emb(int position, float *out) {
    float p = (float)position;
    for (int i = 0; i < ORC_D_MODEL; i += 2) {
        float div_term = expf(-(float)i * logf(10000.0f) / (float)ORC_D_MODEL);
        out[i] = sinf(p * div_term);
        out[i + 1] = cosf(p * div_term);
    }
}

/* ------------------------------------------------------------------------------ */
/* stream state                                                                    */
/* ------------------------------------------------------------------------------ */

struct orc_stream {
    const orc_model *m;
    int R, T, KV, chunk_mel, shift_mel, n_rel, prompt_index;
    orc_preproc *pp;
    /* caches in logical order (oldest row first) */
    float *kcache, *vcache; /* [L][70][1024]  src/nemo-stream.cpp:181-183 */
    float *convcache;       /* [L][ks-1][1024] */
    float *posproj;         /* [L][n_rel][1024]: W_pos . emb(rel), rel = 70+T-1 ... -(T-1) */
    int cache_valid_len;    /* :81, :1085 */
    float h[2 * ORC_HIDDEN], c[2 * ORC_HIDDEN];
    int prev_token;
    float *mel_buf; /* [frames][128], starts with 9 zero frames :73-74 */
    int n_mel, cap_mel;
    int total_chunks, decode_iterations;
    int frames_total;       /* encoder frames decoded so far: timed_token.frame_idx (src/nemo-ggml.h:383-395) */
    int *tok_frames; int n_tok_frames, cap_tok_frames;
    float *tap_sub, *tap_layers;
    /* decision log (orc_stream_enable_decision_log): one record per LSTM+joint evaluation */
    int log_on, n_log, cap_log;
    int *log_frame, *log_ntok, *log_best, *log_second;
    float *log_margin;
};

static void stream_build_posproj(orc_stream *s) {
    const orc_model *m = s->m;
    /* row r <-> relative position rel = (70 + T - 1) - r  (query index 70+i minus key index j;
     * src/nemo-stream.cpp:168-177 slice + :419-461 rel-shift => slice row j+T-1-i) */
    float *emb = (float *)xmalloc(sizeof(float) * (size_t)s->n_rel * ORC_D_MODEL);
    for (int r = 0; r < s->n_rel; r++) orc_pos_emb((ORC_LEFT_CTX + s->T - 1) - r, emb + (size_t)r * ORC_D_MODEL);
    for (int l = 0; l < m->n_layers; l++) {
        float *P = s->posproj + (size_t)l * s->n_rel * ORC_D_MODEL;
        gemm_w(m, emb, s->n_rel, ORC_D_MODEL, m->layers[l].wpos_f32, ORC_D_MODEL, NULL, P); /* :516 */
        if (m->emulate_bf16) round_bf16_inplace(P, (int64_t)s->n_rel * ORC_D_MODEL);
    }
    free(emb);
}

void orc_stream_reset(orc_stream *s) {
    const orc_model *m = s->m;
    int ks1 = m->kernel_size - 1;
    /* NOTE: the reference's nemo_stream_reset leaves conv/K/V cache contents in place
     * (src/nemo-stream.cpp:95-115); a *fresh* stream has them zero (:320-325).  The oracle
     * reset == fresh stream; the quirk is documented in DESIGN.md. */
    memset(s->kcache, 0, sizeof(float) * (size_t)m->n_layers * ORC_LEFT_CTX * ORC_D_MODEL);
    memset(s->vcache, 0, sizeof(float) * (size_t)m->n_layers * ORC_LEFT_CTX * ORC_D_MODEL);
    memset(s->convcache, 0, sizeof(float) * (size_t)m->n_layers * ks1 * ORC_D_MODEL);
    s->cache_valid_len = 0;
    memset(s->h, 0, sizeof(s->h));
    memset(s->c, 0, sizeof(s->c));
    s->prev_token = ORC_BLANK; /* :55-56 */
    s->n_mel = ORC_PRE_CACHE;
    memset(s->mel_buf, 0, sizeof(float) * (size_t)ORC_PRE_CACHE * ORC_N_MELS);
    s->total_chunks = 0;
    s->decode_iterations = 0;
    s->frames_total = 0;
    s->n_tok_frames = 0;
    s->n_log = 0;
    if (s->pp) orc_preproc_reset(s->pp);
}

/* nemo_stream_context::reset() AS CODED (src/nemo-stream.cpp:95-115): decoder state, mel buffer (9 zero frames), token /
 * timing counters and cache_valid_len are reset; encoder_graph.reset() only flips a flag (:31-34) that nothing reads
 * afterwards, so the K/V and conv cache tensors keep their contents, and the per-stream preprocessor (audio_buf carry,
 * last_sample) is not touched.  Stale K/V rows are invisible (valid_len = 0 masks all 70 cached keys); the stale conv
 * cache is seen by the first kernel_size-1 frames. */
void orc_stream_reset_reference(orc_stream *s) {
    s->cache_valid_len = 0;
    memset(s->h, 0, sizeof(s->h));
    memset(s->c, 0, sizeof(s->c));
    s->prev_token = ORC_BLANK;
    s->n_mel = ORC_PRE_CACHE;
    memset(s->mel_buf, 0, sizeof(float) * (size_t)ORC_PRE_CACHE * ORC_N_MELS);
    s->total_chunks = 0;
    s->decode_iterations = 0;
    s->frames_total = 0;
    s->n_tok_frames = 0;
    s->n_log = 0;
}

void orc_stream_enable_decision_log(orc_stream *s, int on) { s->log_on = on; s->n_log = 0; }
/* record i = the i-th LSTM+joint evaluation since create/reset: absolute encoder frame, tokens emitted before it, arg-max,
 * runner-up and the top-2 logit margin (best - second).  Returns the number of records. */
int orc_stream_decision_log(const orc_stream *s, int *frame, int *ntok_before, int *best, int *second, float *margin, int cap) {
    for (int i = 0; i < s->n_log && i < cap; i++) {
        if (frame) frame[i] = s->log_frame[i];
        if (ntok_before) ntok_before[i] = s->log_ntok[i];
        if (best) best[i] = s->log_best[i];
        if (second) second[i] = s->log_second[i];
        if (margin) margin[i] = s->log_margin[i];
    }
    return s->n_log;
}

orc_stream *orc_stream_create(const orc_model *m, int right_context, int prompt_index) {
    orc_stream *s = (orc_stream *)xcalloc(1, sizeof(*s));
    s->m = m;
    s->R = right_context;
    s->T = 1 + right_context;                                       /* src/nemo-stream.h:98-100 */
    s->KV = ORC_LEFT_CTX + s->T;
    s->chunk_mel = ORC_PRE_CACHE + ORC_SUBSAMPLING * (1 + right_context); /* :65-72 */
    s->shift_mel = ORC_SUBSAMPLING * (1 + right_context);                 /* :76-81 */
    s->n_rel = s->KV + s->T - 1;
    s->prompt_index = prompt_index;
    int ks1 = m->kernel_size - 1;
    s->kcache = (float *)xmalloc(sizeof(float) * (size_t)m->n_layers * ORC_LEFT_CTX * ORC_D_MODEL);
    s->vcache = (float *)xmalloc(sizeof(float) * (size_t)m->n_layers * ORC_LEFT_CTX * ORC_D_MODEL);
    s->convcache = (float *)xmalloc(sizeof(float) * (size_t)m->n_layers * ks1 * ORC_D_MODEL);
    s->posproj = (float *)xmalloc(sizeof(float) * (size_t)m->n_layers * s->n_rel * ORC_D_MODEL);
    s->cap_mel = 4 * s->chunk_mel + 64;
    s->mel_buf = (float *)xmalloc(sizeof(float) * (size_t)s->cap_mel * ORC_N_MELS);
    s->pp = m->fb && m->window ? orc_preproc_create(m->fb, m->window) : NULL;
    stream_build_posproj(s);
    orc_stream_reset(s);
    return s;
}

void orc_stream_free(orc_stream *s) {
    if (!s) return;
    free(s->kcache); free(s->vcache); free(s->convcache); free(s->posproj); free(s->mel_buf); free(s->tok_frames);
    free(s->log_frame); free(s->log_ntok); free(s->log_best); free(s->log_second); free(s->log_margin);
    orc_preproc_free(s->pp);
    free(s);
}

/* nemo_stream_set_language (src/nemo-stream.cpp:735-749): takes effect from the next chunk */
void orc_stream_set_prompt(orc_stream *s, int prompt_index) { s->prompt_index = prompt_index; }
int orc_stream_chunk_mel_frames(const orc_stream *s) { return s->chunk_mel; }
int orc_stream_chunk_len(const orc_stream *s) { return s->T; }
int orc_stream_cache_valid_len(const orc_stream *s) { return s->cache_valid_len; }
int orc_stream_total_chunks(const orc_stream *s) { return s->total_chunks; }
int orc_stream_decode_iterations(const orc_stream *s) { return s->decode_iterations; }
int orc_stream_token_frames(const orc_stream *s, int *out, int cap) {
    for (int i = 0; i < s->n_tok_frames && i < cap; i++) out[i] = s->tok_frames[i];
    return s->n_tok_frames;
}
void orc_stream_set_taps(orc_stream *s, float *sub_out, float *layer_out) {
    s->tap_sub = sub_out;
    s->tap_layers = layer_out;
}
void orc_stream_get_cache(const orc_stream *s, int which, int layer, float *out) {
    int ks1 = s->m->kernel_size - 1;
    if (which == 0) memcpy(out, s->kcache + (size_t)layer * ORC_LEFT_CTX * ORC_D_MODEL, sizeof(float) * ORC_LEFT_CTX * ORC_D_MODEL);
    else if (which == 1) memcpy(out, s->vcache + (size_t)layer * ORC_LEFT_CTX * ORC_D_MODEL, sizeof(float) * ORC_LEFT_CTX * ORC_D_MODEL);
    else memcpy(out, s->convcache + (size_t)layer * ks1 * ORC_D_MODEL, sizeof(float) * (size_t)ks1 * ORC_D_MODEL);
}
void orc_stream_get_decoder_state(const orc_stream *s, float *h, float *c, int *prev_token) {
    if (h) memcpy(h, s->h, sizeof(s->h));
    if (c) memcpy(c, s->c, sizeof(s->c));
    if (prev_token) *prev_token = s->prev_token;
}

/* ------------------------------------------------------------------------------ */
/* a-3..a-9: one cached conformer layer (src/nemo-stream.cpp:605-690)               */
/* ------------------------------------------------------------------------------ */

/* x += scale * FFN(LN(x))   (:593-603, :631-634) */
static void ffn_block(const orc_model *m, float *x, int T, const float *nw, const float *nb,
                      const float *w1, const float *w2, float *a, float *h, float *o) {
    layer_norm_rows(x, T, ORC_D_MODEL, nw, nb, a);
    if (m->emulate_bf16) round_bf16_inplace(a, (int64_t)T * ORC_D_MODEL);
    gemm_w(m, a, T, ORC_D_MODEL, w1, ORC_D_FF, NULL, h);
    for (size_t i = 0; i < (size_t)T * ORC_D_FF; i++) h[i] = siluf_(h[i]);
    if (m->emulate_bf16) round_bf16_inplace(h, (int64_t)T * ORC_D_FF);
    gemm_w(m, h, T, ORC_D_FF, w2, ORC_D_MODEL, NULL, o);
    for (size_t i = 0; i < (size_t)T * ORC_D_MODEL; i++) x[i] += 0.5f * o[i];
}

/* cached relative-position MHA (:463-573) with rel-shift folded into indexing (:419-461) */
static void mha_block(const orc_model *m, const orc_layer *L, float *x, int T, int KV,
                      float *kc, float *vc, const float *P, int valid_len, float *a, float *o) {
    const int D = ORC_D_MODEL, H = ORC_N_HEADS, dh = ORC_D_HEAD, C = ORC_LEFT_CTX;
    float *q = (float *)xmalloc(sizeof(float) * (size_t)T * D);
    float *kall = (float *)xmalloc(sizeof(float) * (size_t)KV * D);
    float *vall = (float *)xmalloc(sizeof(float) * (size_t)KV * D);
    float *ctx = (float *)xmalloc(sizeof(float) * (size_t)T * D);
    layer_norm_rows(x, T, D, L->norm_att_w, L->norm_att_b, a);
    if (m->emulate_bf16) round_bf16_inplace(a, (int64_t)T * D);
    gemm_w(m, a, T, D, L->wq, D, NULL, q);                    /* :485 */
    memcpy(kall, kc, sizeof(float) * (size_t)C * D);          /* :493-498 concat */
    memcpy(vall, vc, sizeof(float) * (size_t)C * D);
    gemm_w(m, a, T, D, L->wk, D, NULL, kall + (size_t)C * D); /* :486 */
    gemm_w(m, a, T, D, L->wv, D, NULL, vall + (size_t)C * D); /* :487 */
    if (m->emulate_bf16) {
        round_bf16_inplace(kall + (size_t)C * D, (int64_t)T * D);
        round_bf16_inplace(vall + (size_t)C * D, (int64_t)T * D);
    }
    const float scale = 1.0f / sqrtf((float)dh);             /* :545 */
    const int mask_upto = C - valid_len;                      /* :1037-1043 */
#pragma omp parallel for schedule(static) collapse(2)
    for (int hh = 0; hh < H; hh++)
        for (int i = 0; i < T; i++) {
            float sc[ORC_LEFT_CTX + 64];
            const float *qi = q + (size_t)i * D + hh * dh;
            const float *bu = L->bias_u + hh * dh, *bv = L->bias_v + hh * dh; /* :531-535 */
            float mx = -INFINITY;
            for (int j = 0; j < KV; j++) {
                const float *kj = kall + (size_t)j * D + hh * dh;
                /* rel = (70 + i) - j ; row r = (70+T-1) - rel = j + T - 1 - i */
                const float *pj = P + (size_t)(j + T - 1 - i) * D + hh * dh;
                float s1 = 0.0f, s2 = 0.0f;
                for (int d = 0; d < dh; d++) {
                    s1 += (qi[d] + bu[d]) * kj[d]; /* :538 */
                    s2 += (qi[d] + bv[d]) * pj[d]; /* :541-542 */
                }
                float v = (s1 + s2) * scale;                    /* :546-547 */
                v += (j < mask_upto) ? -1e9f : 0.0f;            /* :552-556 */
                sc[j] = v;
                if (v > mx) mx = v;
            }
            float sum = 0.0f;
            for (int j = 0; j < KV; j++) { sc[j] = expf(sc[j] - mx); sum += sc[j]; } /* :559 */
            float inv = 1.0f / sum;
            float *ci = ctx + (size_t)i * D + hh * dh;
            for (int d = 0; d < dh; d++) ci[d] = 0.0f;
            for (int j = 0; j < KV; j++) {
                float wgt = sc[j] * inv;
                const float *vj = vall + (size_t)j * D + hh * dh;
                for (int d = 0; d < dh; d++) ci[d] += wgt * vj[d]; /* :563 */
            }
        }
    if (m->emulate_bf16) round_bf16_inplace(ctx, (int64_t)T * D);
    gemm_w(m, ctx, T, D, L->wout, D, NULL, o);               /* :570 */
    for (size_t i = 0; i < (size_t)T * D; i++) x[i] += o[i]; /* :643 */
    /* new cache = last 70 rows of [cache; new] (:505-512) */
    memcpy(kc, kall + (size_t)(KV - C) * D, sizeof(float) * (size_t)C * D);
    memcpy(vc, vall + (size_t)(KV - C) * D, sizeof(float) * (size_t)C * D);
    free(q); free(kall); free(vall); free(ctx);
}

/* conv module (:646-679) with cached causal depthwise conv (:336-412) */
static void conv_block(const orc_model *m, const orc_layer *L, float *x, int T, float *cc,
                       float *a, float *o) {
    const int D = ORC_D_MODEL, ks = m->kernel_size, ks1 = ks - 1;
    float *y = (float *)xmalloc(sizeof(float) * (size_t)T * 2 * D);
    float *z = (float *)xmalloc(sizeof(float) * (size_t)(ks1 + T) * D);
    float *c = (float *)xmalloc(sizeof(float) * (size_t)T * D);
    layer_norm_rows(x, T, D, L->norm_conv_w, L->norm_conv_b, a);
    if (m->emulate_bf16) round_bf16_inplace(a, (int64_t)T * D);
    gemm_w(m, a, T, D, L->pw1, 2 * D, NULL, y);              /* :654 */
    memcpy(z, cc, sizeof(float) * (size_t)ks1 * D);          /* :351-356 */
    for (int t = 0; t < T; t++)                              /* GLU :657-664 */
        for (int ch = 0; ch < D; ch++)
            z[(size_t)(ks1 + t) * D + ch] = y[(size_t)t * 2 * D + ch] * sigmoidf_(y[(size_t)t * 2 * D + D + ch]);
    for (int t = 0; t < T; t++)                              /* :368-388, w[k*C + c] */
        for (int ch = 0; ch < D; ch++) {
            float acc = z[(size_t)t * D + ch] * L->dw[ch];
            for (int k = 1; k < ks; k++) acc += z[(size_t)(t + k) * D + ch] * L->dw[(size_t)k * D + ch];
            c[(size_t)t * D + ch] = acc;
        }
    memcpy(cc, z + (size_t)T * D, sizeof(float) * (size_t)ks1 * D); /* last ks-1 rows :396-408 */
    layer_norm_rows(c, T, D, L->conv_ln_w, L->conv_ln_b, c); /* :671-673 */
    for (size_t i = 0; i < (size_t)T * D; i++) c[i] = siluf_(c[i]); /* :674 */
    if (m->emulate_bf16) round_bf16_inplace(c, (int64_t)T * D);
    gemm_w(m, c, T, D, L->pw2, D, NULL, o);                  /* :677 */
    for (size_t i = 0; i < (size_t)T * D; i++) x[i] += o[i]; /* :679 */
    free(y); free(z); free(c);
}

static void cached_layer(const orc_model *m, int l, float *x, int T, int KV, float *kc, float *vc,
                         float *cc, const float *P, int valid_len) {
    const orc_layer *L = &m->layers[l];
    float *a = (float *)xmalloc(sizeof(float) * (size_t)T * ORC_D_MODEL);
    float *h = (float *)xmalloc(sizeof(float) * (size_t)T * ORC_D_FF);
    float *o = (float *)xmalloc(sizeof(float) * (size_t)T * ORC_D_MODEL);
    ffn_block(m, x, T, L->norm_ff1_w, L->norm_ff1_b, L->ff1_w1, L->ff1_w2, a, h, o); /* :631-634 */
    mha_block(m, L, x, T, KV, kc, vc, P, valid_len, a, o);                            /* :637-643 */
    conv_block(m, L, x, T, cc, a, o);                                                 /* :646-679 */
    ffn_block(m, x, T, L->norm_ff2_w, L->norm_ff2_b, L->ff2_w1, L->ff2_w2, a, h, o); /* :682-685 */
    layer_norm_rows(x, T, ORC_D_MODEL, L->norm_out_w, L->norm_out_b, x);              /* :687 */
    free(a); free(h); free(o);
}

void orc_layer_chunk0(const orc_model *m, int layer, const float *x, int T, float *out) {
    const int D = ORC_D_MODEL, KV = ORC_LEFT_CTX + T, n_rel = KV + T - 1, ks1 = m->kernel_size - 1;
    float *kc = (float *)xcalloc((size_t)ORC_LEFT_CTX * D, sizeof(float));
    float *vc = (float *)xcalloc((size_t)ORC_LEFT_CTX * D, sizeof(float));
    float *cc = (float *)xcalloc((size_t)ks1 * D, sizeof(float));
    float *emb = (float *)xcalloc((size_t)n_rel * D, sizeof(float));
    float *P = (float *)xmalloc(sizeof(float) * (size_t)n_rel * D);
    for (int r = 0; r < n_rel; r++) orc_pos_emb((ORC_LEFT_CTX + T - 1) - r, emb + (size_t)r * D);
    gemm_w(m, emb, n_rel, D, m->layers[layer].wpos_f32, D, NULL, P);
    if (m->emulate_bf16) round_bf16_inplace(P, (int64_t)n_rel * D);
    memcpy(out, x, sizeof(float) * (size_t)T * D);
    cached_layer(m, layer, out, T, KV, kc, vc, cc, P, 0);
    free(kc); free(vc); free(cc); free(emb); free(P);
}

/* ------------------------------------------------------------------------------ */
/* a-10: one chunk through the encoder (src/nemo-stream.cpp:132-267, :1013-1101)    */
/* ------------------------------------------------------------------------------ */
void orc_stream_encode_chunk(orc_stream *s, const float *mel_chunk, float *enc_out) {
    const orc_model *m = s->m;
    const int D = ORC_D_MODEL, T = s->T, ks1 = m->kernel_size - 1;
    float *sub = (float *)xmalloc(sizeof(float) * (size_t)(T + ORC_DROP_EXTRA + 2) * D);
    int n_out = orc_subsampling(m, mel_chunk, s->chunk_mel, sub);
    if (n_out != T + ORC_DROP_EXTRA) {
        fprintf(stderr, "nasr_oracle: subsampling gave %d frames, expected %d\n", n_out, T + ORC_DROP_EXTRA);
        abort();
    }
    /* drop the first 2 pre-encoded frames, every chunk (:154-162, :303) */
    float *x = sub + (size_t)ORC_DROP_EXTRA * D;
    if (s->tap_sub) memcpy(s->tap_sub, x, sizeof(float) * (size_t)T * D);
    for (int l = 0; l < m->n_layers; l++) {
        cached_layer(m, l, x, T, s->KV,
                     s->kcache + (size_t)l * ORC_LEFT_CTX * D, s->vcache + (size_t)l * ORC_LEFT_CTX * D,
                     s->convcache + (size_t)l * ks1 * D, s->posproj + (size_t)l * s->n_rel * D,
                     s->cache_valid_len);
        if (s->tap_layers) memcpy(s->tap_layers + (size_t)l * T * D, x, sizeof(float) * (size_t)T * D);
    }
    if (m->num_prompts > 0) { /* a-11: src/nemo-ggml.cpp:1087-1105, one-hot :1048-1059 */
        int idx = s->prompt_index;
        if (idx < 0 || idx >= m->num_prompts) idx = 0;
        const int IN = D + m->num_prompts;
        float *cat = (float *)xcalloc((size_t)T * IN, sizeof(float));
        float *hbuf = (float *)xmalloc(sizeof(float) * (size_t)T * 2048);
        for (int t = 0; t < T; t++) {
            memcpy(cat + (size_t)t * IN, x + (size_t)t * D, sizeof(float) * D);
            cat[(size_t)t * IN + D + idx] = 1.0f;
        }
        gemm_nt(cat, T, IN, m->pk1_w, 2048, m->pk1_b, hbuf);
        for (size_t i = 0; i < (size_t)T * 2048; i++) if (hbuf[i] < 0.0f) hbuf[i] = 0.0f;
        gemm_nt(hbuf, T, 2048, m->pk2_w, D, m->pk2_b, x);
        free(cat); free(hbuf);
    }
    memcpy(enc_out, x, sizeof(float) * (size_t)T * D);
    /* cache validity (:1085) */
    s->cache_valid_len += T;
    if (s->cache_valid_len > ORC_LEFT_CTX) s->cache_valid_len = ORC_LEFT_CTX;
    s->total_chunks++;
    free(sub);
}

/* ------------------------------------------------------------------------------ */
/* a-12/a-13: LSTM x2 + joint (src/nemo-ggml.cpp:580-619, :1137-1224)               */
/* ------------------------------------------------------------------------------ */
static void lstm_cell_(const float *x, const float *h, const float *c, const float *w_ih,
                       const float *w_hh, const float *b_ih, const float *b_hh, float *h_out,
                       float *c_out) {
    const int Hn = ORC_HIDDEN;
    float gi[4 * ORC_HIDDEN], gh[4 * ORC_HIDDEN];
    gemm_nt(x, 1, Hn, w_ih, 4 * Hn, NULL, gi); /* :595 */
    gemm_nt(h, 1, Hn, w_hh, 4 * Hn, NULL, gh); /* :596 */
    for (int j = 0; j < Hn; j++) {
        /* gates = ((gi + gh) + b_ih) + b_hh, order i,f,g,o (:597-612) */
        float gI = ((gi[j] + gh[j]) + b_ih[j]) + b_hh[j];
        float gF = ((gi[Hn + j] + gh[Hn + j]) + b_ih[Hn + j]) + b_hh[Hn + j];
        float gG = ((gi[2 * Hn + j] + gh[2 * Hn + j]) + b_ih[2 * Hn + j]) + b_hh[2 * Hn + j];
        float gO = ((gi[3 * Hn + j] + gh[3 * Hn + j]) + b_ih[3 * Hn + j]) + b_hh[3 * Hn + j];
        float cn = sigmoidf_(gF) * c[j] + sigmoidf_(gI) * tanhf(gG); /* :615 */
        c_out[j] = cn;
        h_out[j] = sigmoidf_(gO) * tanhf(cn);                        /* :618 */
    }
}

void orc_decoder_joint(const orc_model *m, int prev_token, const float *h, const float *c,
                       const float *enc_frame, float *logits, float *h_out, float *c_out) {
    const int Hn = ORC_HIDDEN;
    const float *emb = m->embed + (size_t)prev_token * Hn; /* src/nemo-stream.cpp:877-880 */
    lstm_cell_(emb, h, c, m->w_ih[0], m->w_hh[0], m->b_ih[0], m->b_hh[0], h_out, c_out);
    lstm_cell_(h_out, h + Hn, c + Hn, m->w_ih[1], m->w_hh[1], m->b_ih[1], m->b_hh[1], h_out + Hn, c_out + Hn);
    float ep[ORC_JOINT], dp[ORC_JOINT];
    gemm_nt(enc_frame, 1, ORC_D_MODEL, m->jenc_w, ORC_JOINT, m->jenc_b, ep);   /* :1204-1205 */
    gemm_nt(h_out + Hn, 1, Hn, m->jpred_w, ORC_JOINT, m->jpred_b, dp);         /* :1210-1211 */
    for (int i = 0; i < ORC_JOINT; i++) {
        float v = ep[i] + dp[i];
        ep[i] = v > 0.0f ? v : 0.0f;                                           /* :1216-1217 */
    }
    gemm_nt(ep, 1, ORC_JOINT, m->jout_w, ORC_VOCAB, m->jout_b, logits);        /* :1220-1221 */
}

/* a-14: greedy loop (src/nemo-stream.cpp:840-930, frame loop :1107-1118) */
int orc_stream_decode(orc_stream *s, const float *enc, int n_frames, int *tokens_out, int cap) {
    const orc_model *m = s->m;
    float logits[ORC_VOCAB], hn[2 * ORC_HIDDEN], cn[2 * ORC_HIDDEN];
    int n_tok = 0;
    for (int t = 0; t < n_frames; t++) {
        const float *frame = enc + (size_t)t * ORC_D_MODEL;
        for (int sym = 0; sym < ORC_MAX_SYMBOLS; sym++) {
            s->decode_iterations++;
            orc_decoder_joint(m, s->prev_token, s->h, s->c, frame, logits, hn, cn);
            int best = 0;
            float bs = logits[0];
            for (int v = 1; v < ORC_VOCAB; v++)
                if (logits[v] > bs) { bs = logits[v]; best = v; } /* first max, :899-906 */
            if (s->log_on) {
                if (s->n_log == s->cap_log) {
                    s->cap_log = s->cap_log ? 2 * s->cap_log : 1024;
                    s->log_frame = (int *)realloc(s->log_frame, sizeof(int) * (size_t)s->cap_log);
                    s->log_ntok = (int *)realloc(s->log_ntok, sizeof(int) * (size_t)s->cap_log);
                    s->log_best = (int *)realloc(s->log_best, sizeof(int) * (size_t)s->cap_log);
                    s->log_second = (int *)realloc(s->log_second, sizeof(int) * (size_t)s->cap_log);
                    s->log_margin = (float *)realloc(s->log_margin, sizeof(float) * (size_t)s->cap_log);
                }
                int second = best == 0 ? 1 : 0;
                for (int v = 0; v < ORC_VOCAB; v++)
                    if (v != best && logits[v] > logits[second]) second = v;
                s->log_frame[s->n_log] = s->frames_total + t;
                s->log_ntok[s->n_log] = s->n_tok_frames;
                s->log_best[s->n_log] = best;
                s->log_second[s->n_log] = second;
                s->log_margin[s->n_log] = bs - logits[second];
                s->n_log++;
            }
            if (best == ORC_BLANK) break;                          /* state untouched :908-911 */
            if (n_tok < cap) tokens_out[n_tok] = best;
            n_tok++;
            if (s->n_tok_frames == s->cap_tok_frames) {
                s->cap_tok_frames = s->cap_tok_frames ? 2 * s->cap_tok_frames : 256;
                s->tok_frames = (int *)realloc(s->tok_frames, sizeof(int) * (size_t)s->cap_tok_frames);
            }
            s->tok_frames[s->n_tok_frames++] = s->frames_total + t;
            s->prev_token = best;                                  /* :921-926 */
            memcpy(s->h, hn, sizeof(hn));
            memcpy(s->c, cn, sizeof(cn));
        }
    }
    s->frames_total += n_frames;
    return n_tok;
}

/* ------------------------------------------------------------------------------ */
/* a-15: driver                                                                    */
/* ------------------------------------------------------------------------------ */
static void mel_reserve(orc_stream *s, int extra) {
    if (s->n_mel + extra > s->cap_mel) {
        s->cap_mel = s->n_mel + extra + 256;
        s->mel_buf = (float *)realloc(s->mel_buf, sizeof(float) * (size_t)s->cap_mel * ORC_N_MELS);
    }
}

int orc_stream_push_mel(orc_stream *s, const float *mel, int n_frames, int *tokens_out, int cap) {
    mel_reserve(s, n_frames);
    memcpy(s->mel_buf + (size_t)s->n_mel * ORC_N_MELS, mel, sizeof(float) * (size_t)n_frames * ORC_N_MELS);
    s->n_mel += n_frames; /* :1162 */
    float *enc = (float *)xmalloc(sizeof(float) * (size_t)s->T * ORC_D_MODEL);
    int n_tok = 0;
    while (s->n_mel >= s->chunk_mel) { /* :1174 */
        orc_stream_encode_chunk(s, s->mel_buf, enc);
        n_tok += orc_stream_decode(s, enc, s->T, tokens_out + n_tok, cap > n_tok ? cap - n_tok : 0);
        /* erase shift frames (:1189-1195) */
        memmove(s->mel_buf, s->mel_buf + (size_t)s->shift_mel * ORC_N_MELS,
                sizeof(float) * (size_t)(s->n_mel - s->shift_mel) * ORC_N_MELS);
        s->n_mel -= s->shift_mel;
    }
    free(enc);
    return n_tok;
}

int orc_stream_process(orc_stream *s, const int16_t *pcm, int n_samples, int *tokens_out, int cap) {
    if (!pcm || n_samples <= 0 || !s->pp) return 0; /* :1150 */
    int max_frames = (n_samples + ORC_N_FFT) / ORC_HOP + 2;
    float *mel = (float *)xmalloc(sizeof(float) * (size_t)max_frames * ORC_N_MELS);
    int nf = orc_preproc_process(s->pp, pcm, n_samples, mel, max_frames); /* :1158-1160 */
    int n_tok = nf > 0 ? orc_stream_push_mel(s, mel, nf, tokens_out, cap) : 0;
    free(mel);
    return n_tok;
}

int orc_stream_finalize(orc_stream *s, int *tokens_out, int cap) {
    int n_tok = 0;
    if (s->n_mel > ORC_PRE_CACHE) {                      /* :1240-1241 */
        int real_new = s->n_mel - ORC_PRE_CACHE;
        int n_valid = real_new / ORC_SUBSAMPLING;        /* :1242-1243 */
        if (n_valid > 0) {
            if (s->n_mel < s->chunk_mel) {               /* zero-pad to graph width :1247-1249 */
                mel_reserve(s, s->chunk_mel - s->n_mel);
                memset(s->mel_buf + (size_t)s->n_mel * ORC_N_MELS, 0,
                       sizeof(float) * (size_t)(s->chunk_mel - s->n_mel) * ORC_N_MELS);
                s->n_mel = s->chunk_mel;
            }
            float *enc = (float *)xmalloc(sizeof(float) * (size_t)s->T * ORC_D_MODEL);
            orc_stream_encode_chunk(s, s->mel_buf, enc);
            int keep = n_valid < s->T ? n_valid : s->T;  /* :1094-1101 */
            n_tok = orc_stream_decode(s, enc, keep, tokens_out, cap);
            free(enc);
        }
    }
    return n_tok;
}

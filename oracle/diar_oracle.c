/* diar_oracle.c -- see diar_oracle.h.  TEST INFRASTRUCTURE ONLY. */
#include "diar_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------- */
/* front end: src/diarize_audio.cpp                                                               */
/* ------------------------------------------------------------------------------------------- */
static void fft512(const float *sin_t, const float *cos_t, const float *frame, float *re, float *im) {
    /* radix-2 DIT, same butterfly order as src/diarize_audio.cpp:47-75 */
    const int n = DORC_N_FFT;
    for (int i = 0; i < n; i++) {
        int r = 0, x = i;
        for (int j = 0; j < 9; j++) { r = (r << 1) | (x & 1); x >>= 1; }
        re[r] = frame[i];
        im[r] = 0.0f;
    }
    for (int m = 2; m <= n; m <<= 1) {
        const int m2 = m >> 1, step = n / m;
        for (int k = 0; k < n; k += m)
            for (int j = 0; j < m2; j++) {
                const float wr = cos_t[j * step], wi = -sin_t[j * step];
                const int i1 = k + j, i2 = k + j + m2;
                const float tr = wr * re[i2] - wi * im[i2];
                const float ti = wr * im[i2] + wi * re[i2];
                re[i2] = re[i1] - tr;
                im[i2] = im[i1] - ti;
                re[i1] = re[i1] + tr;
                im[i1] = im[i1] + ti;
            }
    }
}

int dorc_logmel(const float *audio_in, int n_samples, int per_feature_normalize, const float *fb, const float *window,
                float *out, int cap_frames, int *t_valid_out) {
    const int n = n_samples, half = DORC_N_FFT / 2, n_frames = 1 + n / DORC_HOP, n_bins = DORC_N_BINS;
    const int t_valid = n / DORC_HOP;                                  /* :178 */
    int t_padded = t_valid;
    if (t_valid % 16) t_padded += 16 - t_valid % 16;                   /* :209-213, pad_to = 16 */
    if (t_padded > cap_frames) return -1;
    float win_pad[DORC_N_FFT], sin_t[DORC_N_FFT], cos_t[DORC_N_FFT];
    memset(win_pad, 0, sizeof(win_pad));
    for (int i = 0; i < DORC_WIN; i++) win_pad[(DORC_N_FFT - DORC_WIN) / 2 + i] = window[i];   /* :145-147 */
    for (int i = 0; i < DORC_N_FFT; i++) {
        const float th = (2.0f * (float)M_PI * (float)i) / (float)DORC_N_FFT;                    /* :28-32 */
        sin_t[i] = sinf(th);
        cos_t[i] = cosf(th);
    }
    float *audio = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    memcpy(audio, audio_in, sizeof(float) * (size_t)n);
    if (n > 0) {                                                       /* preemphasis, y[0] = x[0], :83-93 */
        float prev = audio[0];
        for (int i = 1; i < n; i++) { const float curr = audio[i]; audio[i] = curr - 0.97f * prev; prev = curr; }
    }
    float *mel = (float *)calloc((size_t)DORC_N_MELS * (size_t)n_frames, sizeof(float));
    float frame[DORC_N_FFT], re[DORC_N_FFT], im[DORC_N_FFT], power[DORC_N_BINS];
    for (int t = 0; t < n_frames; t++) {
        const int start = t * DORC_HOP - half;                         /* center = True, zero padding, :119-125 */
        for (int k = 0; k < DORC_N_FFT; k++) {
            const int idx = start + k;
            const float s = (idx < 0 || idx >= n) ? 0.0f : audio[idx];
            frame[k] = s * win_pad[k];
        }
        fft512(sin_t, cos_t, frame, re, im);
        for (int k = 0; k < n_bins; k++) power[k] = re[k] * re[k] + im[k] * im[k];              /* :127-131 */
        for (int m = 0; m < DORC_N_MELS; m++) {                        /* :164-175 */
            float s = 0.0f;
            for (int k = 0; k < n_bins; k++) s += fb[(size_t)m * n_bins + k] * power[k];
            mel[(size_t)m * n_frames + t] = logf(s + 5.960464477539063e-8f);
        }
    }
    for (int m = 0; m < DORC_N_MELS; m++) {
        float *row = mel + (size_t)m * n_frames;
        if (per_feature_normalize) {                                   /* :182-199 */
            const int n_eff = t_valid, denom = n_eff - 1 > 1 ? n_eff - 1 : 1;
            double sum = 0.0;
            for (int t = 0; t < n_eff; t++) sum += row[t];
            const float mean = (float)(sum / n_eff);
            double var = 0.0;
            for (int t = 0; t < n_eff; t++) { const float d = row[t] - mean; var += (double)d * d; }
            const float std_v = sqrtf((float)(var / denom)) + 1e-5f;
            const float inv_std = 1.0f / std_v;
            for (int t = 0; t < n_eff; t++) row[t] = (row[t] - mean) * inv_std;
        }
        for (int t = t_valid; t < n_frames; t++) row[t] = 0.0f;        /* :197, :203-206 */
    }
    memset(out, 0, sizeof(float) * (size_t)DORC_N_MELS * (size_t)t_padded);
    for (int m = 0; m < DORC_N_MELS; m++)
        memcpy(out + (size_t)m * t_padded, mel + (size_t)m * n_frames, sizeof(float) * (size_t)t_valid);
    free(mel);
    free(audio);
    if (t_valid_out) *t_valid_out = t_valid;
    return t_padded;
}

/* ------------------------------------------------------------------------------------------- */
/* model container                                                                                */
/* ------------------------------------------------------------------------------------------- */
typedef struct { char name[160]; float *data; long long n; } dtensor;

typedef struct {
    int kernel, dilation, separable, cin, cout;
    const float *dw, *pw;      /* dw (k, ch); pw (out, in) */
    float *scale, *bias;       /* folded BN */
} subconv;

typedef struct {
    int repeat, residual, has_se, cin, cout;
    subconv subs[3], res;
    const float *se_fc1, *se_fc2;   /* (C/8, C), (C, C/8) */
} jblock;

struct dorc_model {
    dtensor *t; int n_t, cap_t;
    int finalized;
    jblock vad[6], spk[5];
    const float *vad_fb, *vad_win, *vad_dec_w, *vad_dec_b;
    const float *spk_fb, *spk_win;
    const float *a1_w, *a1_b, *a2_w, *a2_b, *emb_w, *emb_b;
    float *a_bn_scale, *a_bn_bias, *e_bn_scale, *e_bn_bias;
};

typedef struct { int kernel, dilation, repeat, cin, cout, residual, separable, has_se; } topo;
static const topo VAD_TOPO[6] = {      /* src/diarize_vad.cpp:25-32 */
    {11, 1, 1, 80, 128, 0, 1, 0}, {13, 1, 2, 128, 64, 1, 1, 0}, {15, 1, 2, 64, 64, 1, 1, 0},
    {17, 1, 2, 64, 64, 1, 1, 0},  {29, 2, 1, 64, 128, 0, 1, 0}, {1, 1, 1, 128, 128, 0, 0, 0}};
static const topo SPK_TOPO[5] = {      /* src/diarize_spk.cpp:28-34 */
    {3, 1, 1, 80, 1024, 0, 1, 1}, {7, 1, 3, 1024, 1024, 1, 1, 1}, {11, 1, 3, 1024, 1024, 1, 1, 1},
    {15, 1, 3, 1024, 1024, 1, 1, 1}, {1, 1, 1, 1024, 3072, 0, 1, 1}};

dorc_model *dorc_model_create(void) { return (dorc_model *)calloc(1, sizeof(dorc_model)); }

static void free_block(jblock *b) {
    for (int s = 0; s < 3; s++) { free(b->subs[s].scale); free(b->subs[s].bias); }
    free(b->res.scale); free(b->res.bias);
}
void dorc_model_free(dorc_model *m) {
    if (!m) return;
    for (int i = 0; i < m->n_t; i++) free(m->t[i].data);
    free(m->t);
    for (int b = 0; b < 6; b++) free_block(&m->vad[b]);
    for (int b = 0; b < 5; b++) free_block(&m->spk[b]);
    free(m->a_bn_scale); free(m->a_bn_bias); free(m->e_bn_scale); free(m->e_bn_bias);
    free(m);
}

int dorc_model_set_tensor(dorc_model *m, const char *name, const float *data, long long n) {
    if (!m || !name || !data || n <= 0) return -1;
    if (strncmp(name, "vad.", 4) && strncmp(name, "spk.", 4)) return 1;
    if (m->n_t == m->cap_t) {
        m->cap_t = m->cap_t ? 2 * m->cap_t : 256;
        m->t = (dtensor *)realloc(m->t, sizeof(dtensor) * (size_t)m->cap_t);
    }
    dtensor *d = &m->t[m->n_t++];
    snprintf(d->name, sizeof(d->name), "%s", name);
    d->n = n;
    d->data = (float *)malloc(sizeof(float) * (size_t)n);
    memcpy(d->data, data, sizeof(float) * (size_t)n);
    return 0;
}

static const float *find_t(const dorc_model *m, const char *name, long long want) {
    for (int i = 0; i < m->n_t; i++)
        if (!strcmp(m->t[i].name, name)) {
            if (want > 0 && m->t[i].n != want) { fprintf(stderr, "diar oracle: '%s' has %lld elements, want %lld\n", name, m->t[i].n, want); return NULL; }
            return m->t[i].data;
        }
    fprintf(stderr, "diar oracle: missing tensor '%s'\n", name);
    return NULL;
}

/* scale = gamma / sqrt(var + eps), bias = beta - mean * scale */
static int fold_bn(const dorc_model *m, const char *prefix, int C, float eps, float **scale, float **bias) {
    char nm[200];
    snprintf(nm, sizeof(nm), "%s.weight", prefix);       const float *g = find_t(m, nm, C);
    snprintf(nm, sizeof(nm), "%s.bias", prefix);         const float *b = find_t(m, nm, C);
    snprintf(nm, sizeof(nm), "%s.running_mean", prefix); const float *mu = find_t(m, nm, C);
    snprintf(nm, sizeof(nm), "%s.running_var", prefix);  const float *v = find_t(m, nm, C);
    if (!g || !b || !mu || !v) return -1;
    *scale = (float *)malloc(sizeof(float) * (size_t)C);
    *bias = (float *)malloc(sizeof(float) * (size_t)C);
    for (int i = 0; i < C; i++) {
        const float s = g[i] / sqrtf(v[i] + eps);
        (*scale)[i] = s;
        (*bias)[i] = b[i] - mu[i] * s;
    }
    return 0;
}

static int resolve_blocks(dorc_model *m, const char *ns, const topo *tp, int nb, jblock *out) {
    char nm[200], pre[160];
    for (int b = 0; b < nb; b++) {
        const topo *t = &tp[b];
        jblock *blk = &out[b];
        blk->repeat = t->repeat; blk->residual = t->residual; blk->has_se = t->has_se; blk->cin = t->cin; blk->cout = t->cout;
        for (int s = 0; s < t->repeat; s++) {                       /* sub-conv s: mconv.5s (dw), 5s+1 (pw), 5s+2 (bn) */
            subconv *sc = &blk->subs[s];
            const int cin = s == 0 ? t->cin : t->cout;
            int dw_i = 5 * s, pw_i = 5 * s + 1, bn_i = 5 * s + 2;
            if (!t->separable) { pw_i = 0; bn_i = 1; }              /* src/diarize_vad.cpp:160-162 */
            sc->kernel = t->kernel; sc->dilation = t->dilation; sc->separable = t->separable; sc->cin = cin; sc->cout = t->cout;
            if (t->separable) {
                snprintf(nm, sizeof(nm), "%s.encoder.encoder.%d.mconv.%d.conv.weight", ns, b, dw_i);
                if (!(sc->dw = find_t(m, nm, (long long)t->kernel * cin))) return -1;
            }
            snprintf(nm, sizeof(nm), "%s.encoder.encoder.%d.mconv.%d.conv.weight", ns, b, pw_i);
            if (!(sc->pw = find_t(m, nm, (long long)t->cout * cin))) return -1;
            snprintf(pre, sizeof(pre), "%s.encoder.encoder.%d.mconv.%d", ns, b, bn_i);
            if (fold_bn(m, pre, t->cout, 1e-3f, &sc->scale, &sc->bias)) return -1;     /* Jasper BN eps 1e-3 */
        }
        if (t->residual) {
            subconv *sc = &blk->res;
            sc->kernel = 1; sc->dilation = 1; sc->separable = 0; sc->cin = t->cin; sc->cout = t->cout;
            snprintf(nm, sizeof(nm), "%s.encoder.encoder.%d.res.0.0.conv.weight", ns, b);
            if (!(sc->pw = find_t(m, nm, (long long)t->cout * t->cin))) return -1;
            snprintf(pre, sizeof(pre), "%s.encoder.encoder.%d.res.0.1", ns, b);
            if (fold_bn(m, pre, t->cout, 1e-3f, &sc->scale, &sc->bias)) return -1;
        }
        if (t->has_se) {                                            /* src/diarize_spk.cpp:151-158 */
            const int se_i = 5 * (t->repeat - 1) + 3, C = t->cout;
            snprintf(nm, sizeof(nm), "%s.encoder.encoder.%d.mconv.%d.fc.0.weight", ns, b, se_i);
            if (!(blk->se_fc1 = find_t(m, nm, (long long)(C / 8) * C))) return -1;
            snprintf(nm, sizeof(nm), "%s.encoder.encoder.%d.mconv.%d.fc.2.weight", ns, b, se_i);
            if (!(blk->se_fc2 = find_t(m, nm, (long long)C * (C / 8)))) return -1;
        }
    }
    return 0;
}

int dorc_model_finalize(dorc_model *m) {
    if (!m) return -1;
    int have_vad = 0, have_spk = 0;
    for (int i = 0; i < m->n_t; i++) { have_vad |= !strncmp(m->t[i].name, "vad.", 4); have_spk |= !strncmp(m->t[i].name, "spk.", 4); }
    if (have_vad) {
        if (resolve_blocks(m, "vad", VAD_TOPO, 6, m->vad)) return -1;
        m->vad_fb = find_t(m, "vad.preprocessor.featurizer.fb", DORC_N_MELS * DORC_N_BINS);
        m->vad_win = find_t(m, "vad.preprocessor.featurizer.window", DORC_WIN);
        m->vad_dec_w = find_t(m, "vad.decoder.decoder_layers.0.weight", 2 * 128);
        m->vad_dec_b = find_t(m, "vad.decoder.decoder_layers.0.bias", 2);
        if (!m->vad_fb || !m->vad_win || !m->vad_dec_w || !m->vad_dec_b) return -1;
    }
    if (have_spk) {
        if (resolve_blocks(m, "spk", SPK_TOPO, 5, m->spk)) return -1;
        const int C = 3072, A = 128;
        m->spk_fb = find_t(m, "spk.preprocessor.featurizer.fb", DORC_N_MELS * DORC_N_BINS);
        m->spk_win = find_t(m, "spk.preprocessor.featurizer.window", DORC_WIN);
        m->a1_w = find_t(m, "spk.decoder._pooling.attention_layer.0.conv_layer.weight", (long long)A * 3 * C);
        m->a1_b = find_t(m, "spk.decoder._pooling.attention_layer.0.conv_layer.bias", A);
        m->a2_w = find_t(m, "spk.decoder._pooling.attention_layer.2.weight", (long long)C * A);
        m->a2_b = find_t(m, "spk.decoder._pooling.attention_layer.2.bias", C);
        m->emb_w = find_t(m, "spk.decoder.emb_layers.0.1.weight", (long long)DORC_SPK_EMB * 2 * C);
        m->emb_b = find_t(m, "spk.decoder.emb_layers.0.1.bias", DORC_SPK_EMB);
        if (!m->spk_fb || !m->spk_win || !m->a1_w || !m->a1_b || !m->a2_w || !m->a2_b || !m->emb_w || !m->emb_b) return -1;
        if (fold_bn(m, "spk.decoder._pooling.attention_layer.0.bn", A, 1e-5f, &m->a_bn_scale, &m->a_bn_bias)) return -1;   /* decoder eps 1e-5 */
        if (fold_bn(m, "spk.decoder.emb_layers.0.0", 2 * C, 1e-5f, &m->e_bn_scale, &m->e_bn_bias)) return -1;
    }
    m->finalized = 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Jasper building blocks on [T][C] activations                                                   */
/* ------------------------------------------------------------------------------------------- */
/* MaskedConv1d: positions [lens, T) are zeroed before every conv (src/diarize_vad.cpp:283-297) */
static void mask_rows(float *x, int T, int C, int lens) {
    for (int t = lens; t < T; t++) memset(x + (size_t)t * C, 0, sizeof(float) * (size_t)C);
}

/* depthwise, 'same' zero padding: acc = x[t+0*d-pad]*w[0]; acc += x[t+i*d-pad]*w[i] ... (:232-251) */
static void depthwise_same(const float *x, int T, int C, const float *w /*(k, C)*/, int kernel, int dil, float *y) {
    const int pad = dil * (kernel - 1) / 2;
#pragma omp parallel for if (T * C > 20000)
    for (int t = 0; t < T; t++)
        for (int c = 0; c < C; c++) {
            float acc = 0.0f;
            for (int i = 0; i < kernel; i++) {
                const int tt = t + i * dil - pad;
                const float v = (tt < 0 || tt >= T) ? 0.0f : x[(size_t)tt * C + c];
                const float prod = v * w[(size_t)i * C + c];
                acc = i == 0 ? prod : acc + prod;
            }
            y[(size_t)t * C + c] = acc;
        }
}

/* pointwise + folded BN: y[t][o] = (sum_i w[o][i] x[t][i]) * scale[o] + bias[o] (:226-229, :255-265) */
static void pointwise_bn(const float *x, int T, int Cin, const float *w, int Cout, const float *scale, const float *bias, float *y) {
#pragma omp parallel for if ((long long)T * Cin * Cout > 200000)
    for (int t = 0; t < T; t++)
        for (int o = 0; o < Cout; o++) {
            const float *wr = w + (size_t)o * Cin, *xr = x + (size_t)t * Cin;
            float s = 0.0f;
            for (int i = 0; i < Cin; i++) s += wr[i] * xr[i];
            y[(size_t)t * Cout + o] = s * scale[o] + bias[o];
        }
}

static void relu_inplace(float *x, size_t n) { for (size_t i = 0; i < n; i++) x[i] = x[i] > 0.0f ? x[i] : 0.0f; }

/* one Jasper block; x [T][cin] -> out [T][cout]; tmp buffers sized T * max(C) */
static void run_block(const jblock *blk, const float *x, int T, int lens, float inv_lens, float *out, float *tmp_a, float *tmp_b) {
    const float *cur = x;
    int ccur = blk->cin;
    for (int s = 0; s < blk->repeat; s++) {
        const subconv *sc = &blk->subs[s];
        float *y = tmp_a;
        memcpy(y, cur, sizeof(float) * (size_t)T * ccur);
        if (sc->separable) {
            mask_rows(y, T, ccur, lens);
            if (sc->kernel == 1) {                       /* per-channel scaling, src/diarize_spk.cpp:263-267 */
                for (int t = 0; t < T; t++) for (int c = 0; c < ccur; c++) tmp_b[(size_t)t * ccur + c] = y[(size_t)t * ccur + c] * sc->dw[c];
            } else {
                depthwise_same(y, T, ccur, sc->dw, sc->kernel, sc->dilation, tmp_b);
            }
            memcpy(y, tmp_b, sizeof(float) * (size_t)T * ccur);
        }
        mask_rows(y, T, ccur, lens);
        pointwise_bn(y, T, ccur, sc->pw, sc->cout, sc->scale, sc->bias, out);
        ccur = sc->cout;
        if (s + 1 < blk->repeat) relu_inplace(out, (size_t)T * ccur);
        cur = out;
        if (s + 1 < blk->repeat) { memcpy(tmp_b, out, sizeof(float) * (size_t)T * ccur); cur = tmp_b; }
    }
    const int C = blk->cout;
    if (blk->has_se) {                                   /* SE before the residual, src/diarize_spk.cpp:303-315, :365-368 */
        mask_rows(out, T, C, lens);
        float *mean = (float *)calloc((size_t)C, sizeof(float)), *h = (float *)malloc(sizeof(float) * (size_t)(C / 8)), *z = (float *)malloc(sizeof(float) * (size_t)C);
        for (int t = 0; t < T; t++) for (int c = 0; c < C; c++) mean[c] += out[(size_t)t * C + c];
        for (int c = 0; c < C; c++) mean[c] *= inv_lens;
        for (int j = 0; j < C / 8; j++) { float s = 0.0f; for (int c = 0; c < C; c++) s += blk->se_fc1[(size_t)j * C + c] * mean[c]; h[j] = s > 0.0f ? s : 0.0f; }
        for (int c = 0; c < C; c++) { float s = 0.0f; for (int j = 0; j < C / 8; j++) s += blk->se_fc2[(size_t)c * (C / 8) + j] * h[j]; z[c] = 1.0f / (1.0f + expf(-s)); }
        for (int t = 0; t < T; t++) for (int c = 0; c < C; c++) out[(size_t)t * C + c] *= z[c];
        free(mean); free(h); free(z);
    }
    if (blk->residual) {                                 /* :305-310 / :369-374 */
        memcpy(tmp_a, x, sizeof(float) * (size_t)T * blk->cin);
        mask_rows(tmp_a, T, blk->cin, lens);
        pointwise_bn(tmp_a, T, blk->cin, blk->res.pw, C, blk->res.scale, blk->res.bias, tmp_b);
        for (size_t i = 0; i < (size_t)T * C; i++) out[i] += tmp_b[i];
    }
    relu_inplace(out, (size_t)T * C);
}

/* ------------------------------------------------------------------------------------------- */
/* MarbleNet VAD                                                                                  */
/* ------------------------------------------------------------------------------------------- */
float dorc_vad_window(const dorc_model *m, const float *audio, int lens_samples) {
    if (!m || !m->finalized || !m->vad_fb) return -1.0f;
    enum { T = DORC_VAD_T };
    float mel[DORC_N_MELS * T];
    int tv = 0;
    if (dorc_logmel(audio, DORC_VAD_WINDOW, 0, m->vad_fb, m->vad_win, mel, T, &tv) != T) return -1.0f;
    int lens = lens_samples / DORC_HOP;                  /* :447-450 */
    if (lens > 63) lens = 63;
    if (lens < 0) lens = 0;
    static const int CMAX = 128;
    float *a = (float *)malloc(sizeof(float) * T * CMAX), *b = (float *)malloc(sizeof(float) * T * CMAX);
    float *ta = (float *)malloc(sizeof(float) * T * CMAX), *tb = (float *)malloc(sizeof(float) * T * CMAX);
    for (int c = 0; c < DORC_N_MELS; c++) for (int t = 0; t < T; t++) a[(size_t)t * DORC_N_MELS + c] = mel[(size_t)c * T + t];
    float *cur = a, *nxt = b;
    for (int blk = 0; blk < 6; blk++) {
        run_block(&m->vad[blk], cur, T, lens, 0.0f, nxt, ta, tb);
        float *sw = cur; cur = nxt; nxt = sw;
    }
    float mean[128];                                     /* AdaptiveAvgPool1d(1) over all 64 frames, :462-469 */
    memset(mean, 0, sizeof(mean));
    for (int t = 0; t < T; t++) for (int c = 0; c < 128; c++) mean[c] += cur[(size_t)t * 128 + c];
    const float inv_T = 1.0f / (float)T;
    for (int c = 0; c < 128; c++) mean[c] *= inv_T;
    float logits[2];
    for (int k = 0; k < 2; k++) {                        /* :471-478 */
        float v = m->vad_dec_b[k];
        for (int c = 0; c < 128; c++) v += m->vad_dec_w[(size_t)k * 128 + c] * mean[c];
        logits[k] = v;
    }
    const float mx = logits[0] > logits[1] ? logits[0] : logits[1];   /* :480-487 */
    const float e0 = expf(logits[0] - mx), e1 = expf(logits[1] - mx);
    free(a); free(b); free(ta); free(tb);
    return e1 / (e0 + e1);
}

int dorc_vad_batch(const dorc_model *m, const float *audio, int n_samples, float *probs, int cap) {
    if (n_samples < DORC_VAD_WINDOW) return 0;
    const int n = 1 + (n_samples - DORC_VAD_WINDOW) / DORC_HOP;
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < n; i++)
        if (i < cap) probs[i] = dorc_vad_window(m, audio + (size_t)i * DORC_HOP, DORC_VAD_WINDOW);
    return n;
}

/* ------------------------------------------------------------------------------------------- */
/* TitaNet-L speaker embedding                                                                    */
/* ------------------------------------------------------------------------------------------- */
int dorc_spk_embed(const dorc_model *m, const float *audio, int lens_samples, float *emb) {
    if (!m || !m->finalized || !m->spk_fb) return -1;
    enum { T = DORC_SPK_T, C = 3072, A = 128 };
    float *mel = (float *)malloc(sizeof(float) * DORC_N_MELS * T);
    int tv = 0;
    if (dorc_logmel(audio, DORC_SPK_SEGMENT, 1, m->spk_fb, m->spk_win, mel, T, &tv) != T) { free(mel); return -1; }
    int lens = lens_samples / DORC_HOP;                  /* src/diarize_spk.cpp:613-615, :548 */
    if (lens > 150) lens = 150;
    if (lens < 1) lens = 1;
    const float inv_lens = 1.0f / (float)lens;
    const size_t NB = (size_t)T * C;
    float *a = (float *)malloc(sizeof(float) * NB), *b = (float *)malloc(sizeof(float) * NB);
    float *ta = (float *)malloc(sizeof(float) * NB), *tb = (float *)malloc(sizeof(float) * NB);
    for (int c = 0; c < DORC_N_MELS; c++) for (int t = 0; t < T; t++) a[(size_t)t * DORC_N_MELS + c] = mel[(size_t)c * T + t];
    float *cur = a, *nxt = b;
    for (int blk = 0; blk < 5; blk++) {
        run_block(&m->spk[blk], cur, T, lens, inv_lens, nxt, ta, tb);
        float *sw = cur; cur = nxt; nxt = sw;
    }
    /* attentive statistics pooling, :384-487 */
    float *x = cur;                                       /* [T][3072] */
    mask_rows(x, T, C, lens);
    float *mean = (float *)calloc(C, sizeof(float)), *stdv = (float *)calloc(C, sizeof(float));
    for (int t = 0; t < T; t++) for (int c = 0; c < C; c++) mean[c] += x[(size_t)t * C + c];
    for (int c = 0; c < C; c++) mean[c] *= inv_lens;
    for (int t = 0; t < lens; t++)                        /* (x - mean) re-masked, squared, masked mean */
        for (int c = 0; c < C; c++) { const float d = x[(size_t)t * C + c] - mean[c]; stdv[c] += d * d; }
    for (int c = 0; c < C; c++) { float v = stdv[c] * inv_lens; if (v < 1e-10f) v = 1e-10f; if (v > 1e30f) v = 1e30f; stdv[c] = sqrtf(v); }
    float *att = (float *)malloc(sizeof(float) * (size_t)T * A);
#pragma omp parallel for
    for (int t = 0; t < T; t++)
        for (int j = 0; j < A; j++) {                     /* conv over [x_t ; mean ; std] (3C inputs) + bias, relu, BN, tanh */
            const float *w = m->a1_w + (size_t)j * 3 * C, *xr = x + (size_t)t * C;
            float s = 0.0f;
            for (int c = 0; c < C; c++) s += w[c] * xr[c];
            for (int c = 0; c < C; c++) s += w[C + c] * mean[c];
            for (int c = 0; c < C; c++) s += w[2 * C + c] * stdv[c];
            s += m->a1_b[j];
            s = s > 0.0f ? s : 0.0f;
            s = s * m->a_bn_scale[j] + m->a_bn_bias[j];
            att[(size_t)t * A + j] = tanhf(s);
        }
    float *logit = ta;                                    /* [T][C] */
#pragma omp parallel for
    for (int t = 0; t < T; t++)
        for (int c = 0; c < C; c++) {
            const float *w = m->a2_w + (size_t)c * A, *ar = att + (size_t)t * A;
            float s = 0.0f;
            for (int j = 0; j < A; j++) s += w[j] * ar[j];
            logit[(size_t)t * C + c] = s + m->a2_b[c] + (t >= lens ? -1.0e9f : 0.0f);
        }
    float *pool = (float *)malloc(sizeof(float) * 2 * C);
    for (int c = 0; c < C; c++) {                         /* softmax over T per channel, weighted mean / std */
        float mx = -INFINITY;
        for (int t = 0; t < T; t++) mx = logit[(size_t)t * C + c] > mx ? logit[(size_t)t * C + c] : mx;
        float Z = 0.0f;
        for (int t = 0; t < T; t++) Z += expf(logit[(size_t)t * C + c] - mx);
        float mu = 0.0f;
        for (int t = 0; t < T; t++) { const float al = expf(logit[(size_t)t * C + c] - mx) / Z; mu += x[(size_t)t * C + c] * al; }
        float sg = 0.0f;
        for (int t = 0; t < T; t++) { const float al = expf(logit[(size_t)t * C + c] - mx) / Z; const float d = x[(size_t)t * C + c] - mu; sg += d * d * al; }
        if (sg < 1e-10f) sg = 1e-10f;
        pool[c] = mu;
        pool[C + c] = sqrtf(sg);
    }
    for (int c = 0; c < 2 * C; c++) pool[c] = pool[c] * m->e_bn_scale[c] + m->e_bn_bias[c];
    for (int k = 0; k < DORC_SPK_EMB; k++) {
        const float *w = m->emb_w + (size_t)k * 2 * C;
        float s = 0.0f;
        for (int c = 0; c < 2 * C; c++) s += w[c] * pool[c];
        emb[k] = s + m->emb_b[k];
    }
    free(mel); free(a); free(b); free(ta); free(tb); free(mean); free(stdv); free(att); free(pool);
    return 0;
}

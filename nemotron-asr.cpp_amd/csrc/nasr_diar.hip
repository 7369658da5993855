// nasr_diar.hip -- diarization side-car behind the C ABI.
//
// Replaces the compute of the reference's vad_session / spk_session (src/diarize_vad.cpp:436-503,
// src/diarize_spk.cpp:601-626): one ggml graph per 0.63 s window / per 1.5 s sub-segment there, one launch sequence
// over ALL windows (of all streams) / all sub-segments here.  Host control flow around it (onset/offset state machine,
// sub-segment cursor, clustering, RTTM) is out of scope.
#include "nasr_internal.h"
#include "nemotron_asr_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace nasr;

void api_lock_shared();
void api_unlock_shared();
namespace nasr_eng {
void api_capture_begin();            // shared -> exclusive: stream capture is broken by what other threads do meanwhile (nasr_engine_priv.h)
void api_capture_end();
}
namespace {
struct Guard { Guard() { api_lock_shared(); } ~Guard() { api_unlock_shared(); } };

int failf(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return set_error(buf);
}
#define DCHK(x)                                                                                                  \
    do {                                                                                                         \
        hipError_t e_ = (x);                                                                                     \
        if (e_ != hipSuccess) return failf("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct Topo { int kernel, dil, repeat, cin, cout; bool residual, separable, has_se; };
const Topo VAD_TOPO[6] = {{11, 1, 1, 80, 128, false, true, false}, {13, 1, 2, 128, 64, true, true, false},
                          {15, 1, 2, 64, 64, true, true, false},   {17, 1, 2, 64, 64, true, true, false},
                          {29, 2, 1, 64, 128, false, true, false}, {1, 1, 1, 128, 128, false, false, false}};   // src/diarize_vad.cpp:25-32
}  // namespace

struct nasr_diar {
    int device = 0;
    hipStream_t st = nullptr;
    bool st_borrowed = false;        // nasr_diar_set_stream: the stream belongs to somebody else
    bool st_lent = false;            // ... to an engine of this library (nasr_engine_lend_stream): released through the borrow count
    std::vector<void *> allocs;
    std::map<std::string, std::vector<float>> host;     // tensors by name (only during create)
    bool has_vad = false, has_spk = false;
    int max_windows = 0, max_segments = 0;
    // front end constants (both namespaces use the same 80-mel preprocessor layout)
    float *window = nullptr, *cos_t = nullptr, *sin_t = nullptr;
    float *vad_fbT = nullptr; int *vad_band = nullptr;
    VadNet vad{};
    // TitaNet-L
    bool bf16 = true; int esz = 2;
    bool vad_bf16 = false;           // NASR_DIAR_VAD_BF16: MarbleNet on the bf16 MFMA with bf16 activation planes
    bool vad_f16 = false;            // NASR_DIAR_VAD_F16: ... on the f16 MFMA with IEEE-half planes (same kernel, 8x less rounding)
    struct SpkSub { float *dw = nullptr; void *pw = nullptr; float *bias = nullptr; int kernel = 1, cin = 0, cin_pad = 0, cout = 0; };
    struct SpkBlock { int repeat = 1; bool residual = false; SpkSub sub[3], res; float *fc1 = nullptr, *fc2 = nullptr; int cin = 0, cout = 0; };
    SpkBlock spk[5];
    float *spk_fbT = nullptr; int *spk_band = nullptr;
    void *a1x_w = nullptr; float *a1_w = nullptr, *a1_b = nullptr, *a_bn_s = nullptr, *a_bn_b = nullptr, *zero_bias = nullptr;
    void *a2_w = nullptr; float *a2_b = nullptr, *e_bn_s = nullptr, *e_bn_b = nullptr, *emb_w = nullptr, *emb_b = nullptr;
    float *s_mel = nullptr, *X0 = nullptr, *X1 = nullptr, *Y = nullptr, *R = nullptr, *se_z = nullptr, *st_mean = nullptr, *st_std = nullptr;
    float *att_c = nullptr, *att_g = nullptr, *pool = nullptr, *emb = nullptr, *se_h = nullptr;
    void *A = nullptr; int *s_lens = nullptr; long long *s_off = nullptr;
    // segment-tile path (bf16 engine, kernels_spk.hip): A operands ping-pong [M][1024], block outputs [M][1024] x 2, encoder output [M][3072], all bf16
    bf16_t *sA[2] = {nullptr, nullptr}, *sX[2] = {nullptr, nullptr}, *sX4 = nullptr;
    float *st_ms = nullptr, *a1ms_w = nullptr, *fc_part = nullptr;      // [S][2 C] masked mean ; std, the mean / std thirds of the attention conv (pack_mfma_f32), split-K scratch
    // one hipGraph per (segments in the call, sample type): the ~45 launches of an embedding call replayed as one (the pointers inside are the
    // engine's own persistent buffers; the staged audio buffer is part of the key's validity: spk_graph_audio)
    std::map<std::pair<int, uint32_t>, hipGraphExec_t> spk_graphs;
    const void *spk_graph_audio = nullptr;
    bool spk_use_graph = true;       // NASR_DIAR_NO_GRAPH=1: eager launches (A/B, debugging)
    hipEvent_t ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      // [vad / embed][begin / end] of the last call's launch sequence (nasr_diar_last_gpu_ms)
    float last_ms[2] = {0.f, 0.f};
    // scratch
    char *audio = nullptr; size_t audio_cap = 0;        // staged input samples (float or s16), capacities in bytes
    char *pin_audio = nullptr; size_t pin_audio_cap = 0;
    DiarFrameDesc *fr_desc = nullptr; size_t fr_cap = 0;  // VAD: frame descriptors, frames [n][80], per-window rows / lens, results
    float *fr_mel = nullptr; size_t frm_cap = 0;
    int *win_row = nullptr; size_t win_cap = 0;
    float *prob = nullptr; size_t prob_cap = 0;
    char *pin = nullptr; size_t pin_cap = 0;             // pinned staging for descriptors / results
    GatherDesc *gather_dev = nullptr, *gather_pin = nullptr; size_t gather_cap = 0;   // device-resident inputs: (source, destination, bytes) per buffer
};

namespace {
template <typename Tp>
int dalloc(nasr_diar *d, Tp **out, size_t n) {
    void *p = nullptr;
    DCHK(hipMalloc(&p, std::max<size_t>(n * sizeof(Tp), 16)));
    d->allocs.push_back(p);
    *out = (Tp *)p;
    return 0;
}
template <typename Tp>
int upload(nasr_diar *d, const std::vector<Tp> &h, Tp **dev) {
    if (dalloc(d, dev, h.size())) return -1;
    DCHK(hipMemcpy(*dev, h.data(), h.size() * sizeof(Tp), hipMemcpyHostToDevice));
    return 0;
}
const std::vector<float> *get(nasr_diar *d, const std::string &name, size_t want) {
    auto it = d->host.find(name);
    if (it == d->host.end()) { failf("diarize weights: missing tensor '%s'", name.c_str()); return nullptr; }
    if (want && it->second.size() != want) { failf("diarize weights: tensor '%s' has %zu elements, expected %zu", name.c_str(), it->second.size(), want); return nullptr; }
    return &it->second;
}
// folded batch norm (src/diarize_vad.cpp:56-79): scale = gamma / sqrt(var + eps), bias = beta - mean * scale
int fold_bn(nasr_diar *d, const std::string &prefix, int C, float eps, std::vector<float> &scale, std::vector<float> &bias) {
    const auto *g = get(d, prefix + ".weight", C), *b = get(d, prefix + ".bias", C);
    const auto *m = get(d, prefix + ".running_mean", C), *v = get(d, prefix + ".running_var", C);
    if (!g || !b || !m || !v) return -1;
    scale.resize(C); bias.resize(C);
    for (int i = 0; i < C; i++) {
        const float s = (*g)[i] / std::sqrt((*v)[i] + eps);
        scale[i] = s;
        bias[i] = (*b)[i] - (*m)[i] * s;
    }
    return 0;
}
int front_end_constants(nasr_diar *d, const std::string &ns, float **fbT, int **band) {
    const auto *fb = get(d, ns + ".preprocessor.featurizer.fb", (size_t)DIAR_NMEL * NBINS);
    const auto *win = get(d, ns + ".preprocessor.featurizer.window", WIN);
    if (!fb || !win) return -1;
    if (!d->window) {
        std::vector<float> wp(NFFT, 0.0f), ct(NFFT), sn(NFFT);
        memcpy(wp.data() + (NFFT - WIN) / 2, win->data(), WIN * sizeof(float));          // src/diarize_audio.cpp:145-147
        for (int i = 0; i < NFFT; i++) {
            const float th = (2.0f * (float)M_PI * (float)i) / (float)NFFT;                  // :28-32
            sn[i] = sinf(th);
            ct[i] = cosf(th);
        }
        if (upload(d, wp, &d->window) || upload(d, ct, &d->cos_t) || upload(d, sn, &d->sin_t)) return -1;
    }
    std::vector<float> t((size_t)NBINS * DIAR_NMEL);
    std::vector<int> bd(2 * DIAR_NMEL);
    for (int m = 0; m < DIAR_NMEL; m++) {
        int lo = NBINS, hi = 0;
        for (int k = 0; k < NBINS; k++) {
            const float v = (*fb)[(size_t)m * NBINS + k];
            t[(size_t)k * DIAR_NMEL + m] = v;
            if (v != 0.0f) { lo = std::min(lo, k); hi = k + 1; }
        }
        if (lo > hi) lo = hi = 0;
        bd[2 * m] = lo; bd[2 * m + 1] = hi;
    }
    return upload(d, t, fbT) || upload(d, bd, band) ? -1 : 0;
}

// pointwise weights [N][K] -> MFMA A-fragment order for v_mfma_f32_16x16x4_f32 (same layout as the RNN-T decoder's,
// nasr_engine.hip pack_f32_mfma): tile (nt, kg) = 16 rows x 16 k, lane q*16+r holds W[nt*16+r][kg*16+4q .. +4)
std::vector<float> pack_mfma_f32(const std::vector<float> &w, int N, int K) {
    const int NT = N / 16, KG = K / 16;
    std::vector<float> out((size_t)NT * KG * 64 * 4);
    for (int nt = 0; nt < NT; nt++)
        for (int kg = 0; kg < KG; kg++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 4; j++)
                    out[(((size_t)nt * KG + kg) * 64 + lane) * 4 + j] = w[(size_t)(nt * 16 + (lane & 15)) * K + kg * 16 + 4 * (lane >> 4) + j];
    return out;
}

// pointwise weights [N][K] -> bf16 A-fragment tiles for v_mfma_f32_16x16x32_bf16 (the encoder GEMMs' layout): tile (nt, kt) = 16 rows x
// 32 k, lane q*16+r holds W[nt*16+r][kt*32+q*8 .. +8); K zero-padded to a multiple of 32; round to nearest even
static uint16_t f32_to_half_rne(float f) {           // IEEE binary16, round to nearest even, saturating (the kernel's planes saturate too)
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                 // NaN
    if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);                // >= 65520 (rounds past the largest half) and Inf: 65504
    if (u < 0x38800000u) {                                                  // below 2^-14: subnormal half (or zero)
        if (u < 0x33000000u) return (uint16_t)sign;                         // below 2^-25
        const int shift = 113 - (int)(u >> 23);                             // 1 .. 11 extra bits to drop
        uint32_t m = (u & 0x7fffffu) | 0x800000u;
        const uint32_t drop = 13 + shift, half = 1u << (drop - 1), rest = m & ((1u << drop) - 1);
        m >>= drop;
        if (rest > half || (rest == half && (m & 1u))) m++;
        return (uint16_t)(sign | m);
    }
    uint32_t h = ((u - 0x38000000u) >> 13);
    const uint32_t rest = u & 0x1fffu;
    if (rest > 0x1000u || (rest == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

std::vector<uint16_t> pack_mfma_bf16_host(const std::vector<float> &w, int N, int K, bool half = false) {
    const int NT = N / 16, KT = (K + 31) / 32;
    std::vector<uint16_t> out((size_t)NT * KT * 64 * 8, 0);
    for (int nt = 0; nt < NT; nt++)
        for (int kt = 0; kt < KT; kt++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int k = kt * 32 + (lane >> 4) * 8 + j;
                    if (k >= K) continue;
                    uint32_t u;
                    const float f = w[(size_t)(nt * 16 + (lane & 15)) * K + k];
                    if (half) { out[(((size_t)nt * KT + kt) * 64 + lane) * 8 + j] = f32_to_half_rne(f); continue; }
                    memcpy(&u, &f, 4);
                    u += 0x7fffu + ((u >> 16) & 1u);
                    out[(((size_t)nt * KT + kt) * 64 + lane) * 8 + j] = (uint16_t)(u >> 16);
                }
    return out;
}

int load_vad(nasr_diar *d) {
    if (front_end_constants(d, "vad", &d->vad_fbT, &d->vad_band)) return -1;
    int si = 0;
    for (int b = 0; b < 6; b++) {
        const Topo &t = VAD_TOPO[b];
        const std::string pre = "vad.encoder.encoder." + std::to_string(b);
        for (int s = 0; s < t.repeat; s++, si++) {
            const int cin = s == 0 ? t.cin : t.cout;
            int dw_i = 5 * s, pw_i = 5 * s + 1, bn_i = 5 * s + 2;
            if (!t.separable) { pw_i = 0; bn_i = 1; }                                        // :160-162
            VadSub &vs = d->vad.sub[si];
            vs.kernel = t.kernel; vs.dil = t.dil; vs.cin = cin; vs.cout = t.cout; vs.dw = nullptr;
            float *p;
            if (t.separable) {
                const auto *dw = get(d, pre + ".mconv." + std::to_string(dw_i) + ".conv.weight", (size_t)t.kernel * cin);
                if (!dw || upload(d, *dw, &p)) return -1;
                vs.dw = p;
            }
            const auto *pw = get(d, pre + ".mconv." + std::to_string(pw_i) + ".conv.weight", (size_t)t.cout * cin);
            if (!pw || upload(d, pack_mfma_f32(*pw, t.cout, cin), &p)) return -1;
            vs.pw = p;
            vs.pw16 = nullptr;
            if (d->vad_bf16 || d->vad_f16) {
                uint16_t *p16;
                if (upload(d, pack_mfma_bf16_host(*pw, t.cout, cin, d->vad_f16), &p16)) return -1;
                vs.pw16 = p16;
            }
            std::vector<float> sc, bi;
            if (fold_bn(d, pre + ".mconv." + std::to_string(bn_i), t.cout, 1e-3f, sc, bi)) return -1;   // Jasper BN eps 1e-3 (:34-36)
            if (upload(d, sc, &p)) return -1;
            vs.scale = p;
            if (upload(d, bi, &p)) return -1;
            vs.bias = p;
        }
        if (t.residual) {
            VadSub &vr = d->vad.res[b - 1];
            vr.kernel = 1; vr.dil = 1; vr.cin = t.cin; vr.cout = t.cout; vr.dw = nullptr;
            float *p;
            const auto *pw = get(d, pre + ".res.0.0.conv.weight", (size_t)t.cout * t.cin);
            if (!pw || upload(d, pack_mfma_f32(*pw, t.cout, t.cin), &p)) return -1;
            vr.pw = p;
            vr.pw16 = nullptr;
            if (d->vad_bf16 || d->vad_f16) {
                uint16_t *p16;
                if (upload(d, pack_mfma_bf16_host(*pw, t.cout, t.cin, d->vad_f16), &p16)) return -1;
                vr.pw16 = p16;
            }
            std::vector<float> sc, bi;
            if (fold_bn(d, pre + ".res.0.1", t.cout, 1e-3f, sc, bi)) return -1;
            if (upload(d, sc, &p)) return -1;
            vr.scale = p;
            if (upload(d, bi, &p)) return -1;
            vr.bias = p;
        }
    }
    float *p;
    const auto *dw = get(d, "vad.decoder.decoder_layers.0.weight", 2 * 128), *db = get(d, "vad.decoder.decoder_layers.0.bias", 2);
    if (!dw || !db || upload(d, *dw, &p)) return -1;
    d->vad.dec_w = p;
    if (upload(d, *db, &p)) return -1;
    d->vad.dec_b = p;
    return 0;
}

// gathers the B input buffers (float, or s16 with NASR_FLAG_AUDIO_S16; host or device) into one device buffer;
// base[b] = element offset of buffer b
int stage_audio(nasr_diar *d, const float *const *audio, const int32_t *n, int B, uint32_t flags, std::vector<long long> &base) {
    const bool on_device = (flags & NASR_FLAG_PCM_DEVICE) != 0;
    const size_t esz = (flags & NASR_FLAG_AUDIO_S16) ? sizeof(int16_t) : sizeof(float);
    size_t total = 0;
    base.resize(B);
    for (int b = 0; b < B; b++) { base[b] = (long long)total; total += ((size_t)std::max(n[b], 0) + 7) & ~(size_t)7; }     // every buffer starts on 16 bytes
    if (total * esz > d->audio_cap) {
        DCHK(hipStreamSynchronize(d->st));
        if (d->audio) hipFree(d->audio);
        d->audio = nullptr;
        d->audio_cap = total * esz + 262144;
        DCHK(hipMalloc((void **)&d->audio, d->audio_cap));
    }
    if (on_device) {
        // one table upload + ONE gather launch (rounds 1-5: a hipMemcpyAsync per buffer)
        if ((size_t)B > d->gather_cap) {
            DCHK(hipStreamSynchronize(d->st));
            if (d->gather_dev) hipFree(d->gather_dev);
            if (d->gather_pin) hipHostFree(d->gather_pin);
            d->gather_dev = nullptr; d->gather_pin = nullptr;
            d->gather_cap = (size_t)B + 64;
            DCHK(hipMalloc((void **)&d->gather_dev, d->gather_cap * sizeof(GatherDesc)));
            DCHK(hipHostMalloc((void **)&d->gather_pin, d->gather_cap * sizeof(GatherDesc), hipHostMallocDefault));
        }
        long long max_bytes = 0;
        for (int b = 0; b < B; b++) {                   // every entry point of this file ends with a stream synchronise: the pinned table is free again
            d->gather_pin[b].src = audio[b];
            d->gather_pin[b].dst_off = base[b] * (long long)esz;
            d->gather_pin[b].bytes = (long long)std::max(n[b], 0) * (long long)esz;
            max_bytes = std::max(max_bytes, d->gather_pin[b].bytes);
        }
        DCHK(hipMemcpyAsync(d->gather_dev, d->gather_pin, (size_t)B * sizeof(GatherDesc), hipMemcpyHostToDevice, d->st));
        launch_gather_audio(d->gather_dev, B, max_bytes, d->audio, d->st);
    } else {
        // host hand-over: gather into one pinned block, one H2D copy (pageable sources would be staged piecewise by the runtime)
        if (total * esz > d->pin_audio_cap) {
            DCHK(hipStreamSynchronize(d->st));
            if (d->pin_audio) hipHostFree(d->pin_audio);
            d->pin_audio = nullptr;
            d->pin_audio_cap = total * esz + 262144;
            DCHK(hipHostMalloc((void **)&d->pin_audio, d->pin_audio_cap, hipHostMallocDefault));
        }
        DCHK(hipStreamSynchronize(d->st));          // the previous call's copy out of the pinned block has finished
        for (int b = 0; b < B; b++)
            if (n[b] > 0) memcpy(d->pin_audio + (size_t)base[b] * esz, audio[b], (size_t)n[b] * esz);
        DCHK(hipMemcpyAsync(d->audio, d->pin_audio, total * esz, hipMemcpyHostToDevice, d->st));
    }
    return 0;
}
void set_audio(const nasr_diar *d, uint32_t flags, DiarMelParams &mp) {
    if (flags & NASR_FLAG_AUDIO_S16) { mp.audio = nullptr; mp.audio_s16 = (const int16_t *)d->audio; }
    else { mp.audio = (const float *)d->audio; mp.audio_s16 = nullptr; }
}
}  // namespace

static int nasr_diar_load_spk(nasr_diar *d);

// takes no lock: the callers hold the shared API lock already (re-locking a std::shared_mutex the thread holds is undefined)
static void diar_destroy_impl(nasr_diar *d) {
    if (!d) return;
    hipSetDevice(d->device);
    if (d->st) hipStreamSynchronize(d->st);
    for (auto &kv : d->spk_graphs) hipGraphExecDestroy(kv.second);
    for (auto &pair : d->ev) for (auto &e : pair) if (e) hipEventDestroy(e);
    for (void *p : d->allocs) hipFree(p);
    if (d->audio) hipFree(d->audio);
    if (d->fr_desc) hipFree(d->fr_desc);
    if (d->fr_mel) hipFree(d->fr_mel);
    if (d->win_row) hipFree(d->win_row);
    if (d->prob) hipFree(d->prob);
    if (d->pin) hipHostFree(d->pin);
    if (d->pin_audio) hipHostFree(d->pin_audio);
    if (d->gather_dev) hipFree(d->gather_dev);
    if (d->gather_pin) hipHostFree(d->gather_pin);
    if (d->st && !d->st_borrowed) hipStreamDestroy(d->st);
    else if (d->st && d->st_lent) lent_stream_release(d->st);
    delete d;
}

extern "C" void nasr_diar_destroy(nasr_diar *d) {
    if (!d) return;
    Guard g;
    diar_destroy_impl(d);
}

// run the side-car on a stream (hipStream_t) the caller owns -- e.g. one the ASR engine has lent (nasr_engine_lend_stream), so that
// the side-car sits on a hardware queue that no encoder lane uses.  The caller keeps the stream alive until nasr_diar_destroy.
extern "C" int nasr_diar_set_stream(nasr_diar *d, void *hip_stream) {
    if (!d || !hip_stream) return failf("null argument");
    Guard g;
    hipSetDevice(d->device);
    if (d->st) {
        hipStreamSynchronize(d->st);
        if (!d->st_borrowed) hipStreamDestroy(d->st);
        else if (d->st_lent) lent_stream_release(d->st);
    }
    d->st = (hipStream_t)hip_stream;
    d->st_borrowed = true;
    d->st_lent = lent_stream_acquire(d->st);       // a stream an engine lent: counted, so that neither side leaves the other a dangling handle
    return 0;
}

extern "C" int nasr_diar_create(nasr_diar **out, int device_id, int dtype, const nasr_weight_desc *weights, int n_weights,
                                int max_windows, int max_segments) {
    Guard g;
    if (!out || !weights || n_weights <= 0) return failf("null argument");
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return failf("no HIP device visible: the diarization engine is MI355X-only, there is no CPU fallback");
    if (device_id < 0 || device_id >= n_dev) return failf("device %d out of range (%d visible)", device_id, n_dev);
    DCHK(hipSetDevice(device_id));
    nasr_diar *d = new nasr_diar();
    d->device = device_id;
    d->vad_f16 = (dtype & NASR_DIAR_VAD_F16) != 0;
    d->vad_bf16 = !d->vad_f16 && (dtype & NASR_DIAR_VAD_BF16) != 0;
    dtype &= ~(NASR_DIAR_VAD_BF16 | NASR_DIAR_VAD_F16);
    if (dtype != NASR_DTYPE_BF16 && dtype != NASR_DTYPE_F32) { delete d; return failf("dtype must be NASR_DTYPE_F32 or NASR_DTYPE_BF16 (optionally | NASR_DIAR_VAD_BF16 or NASR_DIAR_VAD_F16)"); }
    d->bf16 = dtype == NASR_DTYPE_BF16;      // TitaNet's pointwise convolutions; MarbleNet is f32 unless NASR_DIAR_VAD_BF16 is set
    d->esz = d->bf16 ? 2 : 4;
    d->max_windows = std::max(max_windows, 1);
    d->max_segments = std::max(max_segments, 1);
    for (int i = 0; i < n_weights; i++) {
        const nasr_weight_desc &w = weights[i];
        if (!w.name || !w.data) { delete d; return failf("weight %d: null name or data", i); }
        const bool vad = !strncmp(w.name, "vad.", 4), spk = !strncmp(w.name, "spk.", 4);
        if (!vad && !spk) continue;
        if (w.type != NASR_TYPE_F32) { delete d; return failf("diarize tensor '%s': only F32 is supported (scripts/convert_diarize_to_gguf.py writes F32)", w.name); }
        size_t n = 1;
        for (int k = 0; k < w.n_dims; k++) n *= (size_t)w.ne[k];
        d->host[w.name].assign((const float *)w.data, (const float *)w.data + n);
        d->has_vad |= vad;
        d->has_spk |= spk;
    }
    if (!d->has_vad && !d->has_spk) { delete d; return failf("no 'vad.*' or 'spk.*' tensors among the %d weights", n_weights); }
    if (hipStreamCreateWithFlags(&d->st, hipStreamNonBlocking) != hipSuccess) { delete d; return failf("hipStreamCreate failed"); }
    init_diar_kernel_attributes();
    init_spk_kernel_attributes();
    if (const char *ng = getenv("NASR_DIAR_NO_GRAPH")) d->spk_use_graph = !(ng[0] && ng[0] != '0');
    for (auto &pair : d->ev) for (auto &e : pair) if (hipEventCreate(&e) != hipSuccess) { diar_destroy_impl(d); return failf("hipEventCreate failed"); }
    d->pin_cap = (size_t)(d->max_windows + d->max_segments * SPK_EMB + 4096) * 16;
    if (hipHostMalloc((void **)&d->pin, d->pin_cap, hipHostMallocDefault) != hipSuccess) { diar_destroy_impl(d); return failf("hipHostMalloc failed"); }
    int rc = 0;
    if (d->has_vad) rc |= load_vad(d);
    if (!rc && d->has_spk) rc |= nasr_diar_load_spk(d);
    if (rc) { diar_destroy_impl(d); return -1; }
    d->host.clear();
    if (hipStreamSynchronize(d->st) != hipSuccess) { diar_destroy_impl(d); return failf("diar init sync failed"); }
    *out = d;
    return 0;
}

// grow-on-demand device scratch (the VAD's sizes depend on the call)
template <typename Tp>
static int ensure(nasr_diar *d, Tp **p, size_t *cap, size_t n) {
    if (n <= *cap) return 0;
    DCHK(hipStreamSynchronize(d->st));
    if (*p) hipFree(*p);
    *p = nullptr;
    *cap = n + n / 2 + 1024;
    DCHK(hipMalloc((void **)p, *cap * sizeof(Tp)));
    return 0;
}

// vad_session_run_batch (src/diarize_vad.cpp:490-503) for B buffers in one launch sequence: every 0.63 s window of
// every buffer at a 10 ms shift -> P(speech).  audio[b]: n_samples[b] float samples in [-1, 1].
//
// The reference preprocesses each window on its own (64 STFT frames per window, 100 windows per second).  Local frame t
// of the window at offset i*160 is the buffer's frame i+t; only frames 0, 1 (they contain the window's left zero
// padding and the restart of the pre-emphasis, y[0] = x[0]) and 62 (right zero padding) depend on the window, frame 63 is
// masked.  So per buffer the frames 2 .. n_windows+60 are computed ONCE and shared, plus three edge frames per window:
// n_windows + 60 + 3 n_windows frames instead of 63 n_windows, bit-identical inputs to the network.
extern "C" int nasr_diar_vad(nasr_diar *d, int B, const float *const *audio, const int32_t *n_samples, float *const *probs_out,
                             const int32_t *probs_cap, int32_t *n_windows, uint32_t flags) {
    Guard g;
    if (!d || !audio || !n_samples || B < 1) return failf("null argument or B < 1");
    if (!d->has_vad) return failf("this diarization engine was created without 'vad.*' tensors");
    DCHK(hipSetDevice(d->device));
    std::vector<long long> base;
    std::vector<int> first(B + 1, 0);
    for (int b = 0; b < B; b++) {
        if (n_samples[b] < 0 || (n_samples[b] > 0 && !audio[b])) return failf("stream %d: bad audio / n_samples", b);
        const int nw = n_samples[b] < VAD_WINDOW ? 0 : 1 + (n_samples[b] - VAD_WINDOW) / HOP;      // :492-493
        first[b + 1] = first[b] + nw;
    }
    const int W = first[B];
    if (n_windows) for (int b = 0; b < B; b++) n_windows[b] = first[b + 1] - first[b];
    if (W == 0) return 0;
    if (stage_audio(d, audio, n_samples, B, flags, base)) return -1;
    // frame descriptors: [shared frames of every buffer | 3 edge frames of every window], and per window its row in `shared`
    size_t n_shared = 0;
    for (int b = 0; b < B; b++) if (first[b + 1] > first[b]) n_shared += (size_t)(first[b + 1] - first[b]) + 60;
    const size_t n_frames = n_shared + 3 * (size_t)W;
    const size_t host_bytes = n_frames * sizeof(DiarFrameDesc) + (size_t)W * (2 * sizeof(int) + sizeof(float));
    if (host_bytes > d->pin_cap) {
        DCHK(hipStreamSynchronize(d->st));
        if (d->pin) hipHostFree(d->pin);
        d->pin = nullptr;
        d->pin_cap = host_bytes + host_bytes / 2;
        DCHK(hipHostMalloc((void **)&d->pin, d->pin_cap, hipHostMallocDefault));
    }
    DiarFrameDesc *h_fr = (DiarFrameDesc *)d->pin;
    int *h_row = (int *)(d->pin + n_frames * sizeof(DiarFrameDesc)), *h_len = h_row + W;
    float *h_prob = (float *)(h_len + W);
    size_t fs = 0;
    for (int b = 0; b < B; b++) {
        const int nw = first[b + 1] - first[b];
        if (nw == 0) continue;
        for (int gI = 2; gI <= nw + 61; gI++) { h_fr[fs + gI - 2].base = base[b]; h_fr[fs + gI - 2].n = n_samples[b]; h_fr[fs + gI - 2].t = gI; }
        for (int i = 0; i < nw; i++) {
            const int w = first[b] + i;
            h_row[w] = (int)(fs + (size_t)i) - 2;                  // shared row of local frame t = h_row + t  (t >= 2)
            h_len[w] = VAD_WINDOW / HOP;                           // full windows: 63 valid frames
            const int tl[3] = {0, 1, 62};
            for (int k = 0; k < 3; k++) {
                DiarFrameDesc &f = h_fr[n_shared + 3 * (size_t)w + k];
                f.base = base[b] + (long long)i * HOP; f.n = VAD_WINDOW; f.t = tl[k];
            }
        }
        fs += (size_t)nw + 60;
    }
    if (ensure(d, &d->fr_desc, &d->fr_cap, n_frames) || ensure(d, &d->fr_mel, &d->frm_cap, n_frames * DIAR_NMEL) ||
        ensure(d, &d->win_row, &d->win_cap, (size_t)2 * W) || ensure(d, &d->prob, &d->prob_cap, (size_t)W))
        return -1;
    DCHK(hipMemcpyAsync(d->fr_desc, h_fr, n_frames * sizeof(DiarFrameDesc), hipMemcpyHostToDevice, d->st));
    DCHK(hipMemcpyAsync(d->win_row, h_row, (size_t)2 * W * sizeof(int), hipMemcpyHostToDevice, d->st));
    DiarMelParams mp;
    memset(&mp, 0, sizeof(mp));
    set_audio(d, flags, mp);
    mp.window = d->window; mp.fbT = d->vad_fbT; mp.fb_band = d->vad_band; mp.cos_t = d->cos_t; mp.sin_t = d->sin_t;
    DCHK(hipEventRecord(d->ev[0][0], d->st));
    launch_diar_frames(mp, d->fr_desc, (int)n_frames, d->fr_mel, d->st);
    (d->vad_f16 ? launch_vad_marblenet_f16 : d->vad_bf16 ? launch_vad_marblenet_bf16 : launch_vad_marblenet)(d->vad, d->fr_mel, d->fr_mel + n_shared * DIAR_NMEL, d->win_row, d->win_row + W, d->prob, W, d->st);
    DCHK(hipEventRecord(d->ev[0][1], d->st));
    DCHK(hipMemcpyAsync(h_prob, d->prob, (size_t)W * sizeof(float), hipMemcpyDeviceToHost, d->st));
    DCHK(hipStreamSynchronize(d->st));
    if (hipEventElapsedTime(&d->last_ms[0], d->ev[0][0], d->ev[0][1]) != hipSuccess) d->last_ms[0] = 0.f;
    for (int b = 0; b < B; b++) {
        const int nw = first[b + 1] - first[b];
        const int cap = probs_out && probs_out[b] && probs_cap ? probs_cap[b] : 0;
        for (int i = 0; i < nw && i < cap; i++) probs_out[b][i] = h_prob[first[b] + i];
    }
    return 0;
}

// device time of the last call's launch sequence (HIP events on the side-car's stream around its kernels; staging copies and the read-back excluded)
extern "C" int nasr_diar_last_gpu_ms(nasr_diar *d, int which, float *ms_out) {
    if (!d || !ms_out || (which != 0 && which != 1)) return failf("nasr_diar_last_gpu_ms: which = 0 (VAD) or 1 (embeddings)");
    *ms_out = d->last_ms[which];
    return 0;
}

// parity tap: diarize_compute_logmel (src/diarize_audio.cpp:136-227) of ONE whole buffer on the device front end --
// what tests/test_diarize_preproc.cpp checks against the NeMo fixture.  Row-major [80][t_padded] like the reference.
extern "C" int nasr_diar_logmel(nasr_diar *d, int which, const float *audio, int32_t n_samples, int per_feature_normalize,
                                float *mel_out, int64_t cap, int32_t *t_valid_out) {
    Guard g;
    if (!d || !audio || n_samples < HOP) return failf("null argument or fewer than %d samples", HOP);
    const bool spk = which != 0;
    if (spk ? !d->has_spk : !d->has_vad) return failf("this diarization engine was created without '%s.*' tensors", spk ? "spk" : "vad");
    DCHK(hipSetDevice(d->device));
    const int t_valid = n_samples / HOP, t_pad = (t_valid + 15) / 16 * 16;
    if (t_valid_out) *t_valid_out = t_valid;
    if (!mel_out || cap < (int64_t)DIAR_NMEL * t_pad) return failf("mel_out holds %lld floats, need %lld", (long long)cap, (long long)DIAR_NMEL * t_pad);
    std::vector<long long> base;
    if (stage_audio(d, &audio, &n_samples, 1, 0, base)) return -1;
    float *mel = nullptr;
    long long *off = nullptr;
    DCHK(hipMalloc((void **)&mel, (size_t)t_pad * DIAR_NMEL * sizeof(float)));
    if (hipMalloc((void **)&off, sizeof(long long)) != hipSuccess) { hipFree(mel); return failf("hipMalloc failed"); }
    const long long zero = 0;
    DiarMelParams mp;
    memset(&mp, 0, sizeof(mp));
    set_audio(d, 0, mp);
    mp.win_off = off; mp.n_win = n_samples; mp.T_pad = t_pad; mp.t_valid = t_valid; mp.cpitch = DIAR_NMEL; mp.mel = mel;
    mp.window = d->window; mp.fbT = spk ? d->spk_fbT : d->vad_fbT; mp.fb_band = spk ? d->spk_band : d->vad_band;
    mp.cos_t = d->cos_t; mp.sin_t = d->sin_t;
    std::vector<float> h((size_t)t_pad * DIAR_NMEL);
    hipError_t e1 = hipMemcpyAsync(off, &zero, sizeof(zero), hipMemcpyHostToDevice, d->st);
    if (e1 == hipSuccess) {
        launch_diar_logmel(mp, 1, per_feature_normalize != 0, d->st);
        e1 = hipMemcpyAsync(h.data(), mel, h.size() * sizeof(float), hipMemcpyDeviceToHost, d->st);
    }
    const hipError_t e2 = hipStreamSynchronize(d->st);
    hipFree(mel);
    hipFree(off);
    if (e1 != hipSuccess || e2 != hipSuccess) return failf("nasr_diar_logmel: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    for (int t = 0; t < t_pad; t++)
        for (int m = 0; m < DIAR_NMEL; m++) mel_out[(size_t)m * t_pad + t] = h[(size_t)t * DIAR_NMEL + m];
    return 0;
}

// pointwise conv weights [cout][cin] with the folded BN scale multiplied in, K padded to a multiple of 32, in the GEMM's
// layout (packed bf16 tiles or f32 row-major)
static int upload_gemm_weight(nasr_diar *d, const std::vector<float> &w, const std::vector<float> *scale, int cout, int cin, int cin_pad, void **out) {
    std::vector<float> h((size_t)cout * cin_pad, 0.0f);
    for (int o = 0; o < cout; o++)
        for (int i = 0; i < cin; i++) h[(size_t)o * cin_pad + i] = w[(size_t)o * cin + i] * (scale ? (*scale)[o] : 1.0f);
    float *tmp = nullptr;
    DCHK(hipMalloc((void **)&tmp, h.size() * sizeof(float)));
    DCHK(hipMemcpy(tmp, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    if (d->bf16) {
        bf16_t *pk;
        if (dalloc(d, &pk, h.size())) { hipFree(tmp); return -1; }
        launch_pack_weight_bf16(tmp, pk, cout, cin_pad, d->st);
        DCHK(hipStreamSynchronize(d->st));
        hipFree(tmp);
        *out = pk;
    } else {
        d->allocs.push_back(tmp);
        *out = tmp;
    }
    return 0;
}

static int nasr_diar_load_spk(nasr_diar *d) {
    static const Topo SPK_TOPO[5] = {{3, 1, 1, 80, 1024, false, true, true}, {7, 1, 3, 1024, 1024, true, true, true},
                                     {11, 1, 3, 1024, 1024, true, true, true}, {15, 1, 3, 1024, 1024, true, true, true},
                                     {1, 1, 1, 1024, 3072, false, true, true}};       // src/diarize_spk.cpp:28-34
    if (front_end_constants(d, "spk", &d->spk_fbT, &d->spk_band)) return -1;
    for (int b = 0; b < 5; b++) {
        const Topo &t = SPK_TOPO[b];
        const std::string pre = "spk.encoder.encoder." + std::to_string(b);
        nasr_diar::SpkBlock &blk = d->spk[b];
        blk.repeat = t.repeat; blk.residual = t.residual; blk.cin = t.cin; blk.cout = t.cout;
        for (int s = 0; s < t.repeat; s++) {
            const int cin = s == 0 ? t.cin : t.cout, cin_pad = (cin + 63) & ~63;     // K of the GEMM: whole 64-deep chunks
            nasr_diar::SpkSub &ss = blk.sub[s];
            ss.kernel = t.kernel; ss.cin = cin; ss.cin_pad = cin_pad; ss.cout = t.cout;
            const auto *dw = get(d, pre + ".mconv." + std::to_string(5 * s) + ".conv.weight", (size_t)t.kernel * cin);
            const auto *pw = get(d, pre + ".mconv." + std::to_string(5 * s + 1) + ".conv.weight", (size_t)t.cout * cin);
            std::vector<float> sc, bi;
            if (!dw || !pw || fold_bn(d, pre + ".mconv." + std::to_string(5 * s + 2), t.cout, 1e-3f, sc, bi)) return -1;   // encoder BN eps 1e-3 (:36-41)
            // bf16 engine (segment tiles): a depthwise conv of kernel 1 is a per-channel scaling of the pointwise conv's input (:263-267) -- folded into
            // the pointwise weights, so that block 4's GEMM reads the masked bf16 output of block 3 directly
            std::vector<float> pwf;
            if (d->bf16 && t.kernel == 1) {
                pwf = *pw;
                for (int o = 0; o < t.cout; o++)
                    for (int i = 0; i < cin; i++) pwf[(size_t)o * cin + i] *= (*dw)[i];
            }
            if (upload(d, *dw, &ss.dw) || upload_gemm_weight(d, pwf.empty() ? *pw : pwf, &sc, t.cout, cin, cin_pad, &ss.pw) || upload(d, bi, &ss.bias)) return -1;
        }
        if (t.residual) {
            nasr_diar::SpkSub &rs = blk.res;
            rs.kernel = 1; rs.cin = t.cin; rs.cin_pad = t.cin; rs.cout = t.cout;
            const auto *pw = get(d, pre + ".res.0.0.conv.weight", (size_t)t.cout * t.cin);
            std::vector<float> sc, bi;
            if (!pw || fold_bn(d, pre + ".res.0.1", t.cout, 1e-3f, sc, bi)) return -1;
            if (upload_gemm_weight(d, *pw, &sc, t.cout, t.cin, t.cin, &rs.pw) || upload(d, bi, &rs.bias)) return -1;
        }
        const std::string se = pre + ".mconv." + std::to_string(5 * (t.repeat - 1) + 3);                 // :151-158
        const auto *f1 = get(d, se + ".fc.0.weight", (size_t)(t.cout / 8) * t.cout), *f2 = get(d, se + ".fc.2.weight", (size_t)t.cout * (t.cout / 8));
        // SE gate and embedding layers run on the f32 MFMA (k_encproj: packed weights, K split over the waves): through the
        // k-ascending parity GEMM their few rows (one per segment) made grids of 12-36 workgroups with K up to 6 144 --
        // 2.1 of the 4.9 ms of an embedding call
        if (!f1 || !f2 || upload(d, pack_mfma_f32(*f1, t.cout / 8, t.cout), &blk.fc1) || upload(d, pack_mfma_f32(*f2, t.cout, t.cout / 8), &blk.fc2)) return -1;
    }
    const int C = SPK_C, A = SPK_ATT;
    const std::string dp = "spk.decoder";
    const auto *a1w = get(d, dp + "._pooling.attention_layer.0.conv_layer.weight", (size_t)A * 3 * C);
    const auto *a1b = get(d, dp + "._pooling.attention_layer.0.conv_layer.bias", A);
    const auto *a2w = get(d, dp + "._pooling.attention_layer.2.weight", (size_t)C * A);
    const auto *a2b = get(d, dp + "._pooling.attention_layer.2.bias", C);
    const auto *ew = get(d, dp + ".emb_layers.0.1.weight", (size_t)SPK_EMB * 2 * C), *eb = get(d, dp + ".emb_layers.0.1.bias", SPK_EMB);
    std::vector<float> as, ab, es, ebn;
    if (!a1w || !a1b || !a2w || !a2b || !ew || !eb) return -1;
    if (fold_bn(d, dp + "._pooling.attention_layer.0.bn", A, 1e-5f, as, ab) || fold_bn(d, dp + ".emb_layers.0.0", 2 * C, 1e-5f, es, ebn)) return -1;   // decoder BN eps 1e-5
    std::vector<float> a1x((size_t)A * C);                     // the x third of the attention conv: a GEMM; the rest: k_spk_att_const
    for (int a = 0; a < A; a++) memcpy(&a1x[(size_t)a * C], &(*a1w)[(size_t)a * 3 * C], (size_t)C * sizeof(float));
    std::vector<float> a1ms((size_t)A * 2 * C);                // the mean / std thirds: one [A][2 C] linear over the segments (segment-tile path)
    for (int a = 0; a < A; a++) memcpy(&a1ms[(size_t)a * 2 * C], &(*a1w)[(size_t)a * 3 * C + C], (size_t)2 * C * sizeof(float));
    if (d->bf16 && upload(d, pack_mfma_f32(a1ms, A, 2 * C), &d->a1ms_w)) return -1;
    std::vector<float> zeros((size_t)std::max(C, 1024), 0.0f);
    if (upload_gemm_weight(d, a1x, nullptr, A, C, C, &d->a1x_w) || upload(d, *a1w, &d->a1_w) || upload(d, *a1b, &d->a1_b) ||
        upload(d, as, &d->a_bn_s) || upload(d, ab, &d->a_bn_b) || upload(d, zeros, &d->zero_bias) ||
        upload_gemm_weight(d, *a2w, nullptr, C, A, A, &d->a2_w) || upload(d, *a2b, &d->a2_b) || upload(d, es, &d->e_bn_s) ||
        upload(d, ebn, &d->e_bn_b) || upload(d, pack_mfma_f32(*ew, SPK_EMB, 2 * SPK_C), &d->emb_w) || upload(d, *eb, &d->emb_b))
        return -1;
    const size_t S = (size_t)d->max_segments, M = S * SPK_T;
    char *a = nullptr;
    if (dalloc(d, &d->s_mel, M * 96) || dalloc(d, &d->X0, M * C) || dalloc(d, &d->X1, M * C) || dalloc(d, &d->Y, M * C) ||
        dalloc(d, &d->R, M * 1024) || dalloc(d, &a, M * C * (size_t)d->esz) || dalloc(d, &d->se_z, 2 * S * C) || dalloc(d, &d->se_h, S * (C / 8)) ||
        dalloc(d, &d->st_mean, S * C) || dalloc(d, &d->st_std, S * C) || dalloc(d, &d->att_c, S * A) || dalloc(d, &d->att_g, M * A) ||
        dalloc(d, &d->pool, S * 2 * C) || dalloc(d, &d->emb, S * SPK_EMB) || dalloc(d, &d->s_lens, S) || dalloc(d, &d->s_off, S))
        return -1;
    d->A = a;
    if (d->bf16 && (dalloc(d, &d->sA[0], M * 1024) || dalloc(d, &d->sA[1], M * 1024) || dalloc(d, &d->sX[0], M * 1024) || dalloc(d, &d->sX[1], M * 1024) ||
                    dalloc(d, &d->sX4, M * C) || dalloc(d, &d->st_ms, S * 2 * C) || dalloc(d, &d->fc_part, 16 * S * 512)))
        return -1;
    return 0;
}

static void spk_gemm(nasr_diar *d, const void *A, int lda, const void *W, int M, int N, int K, const float *bias, bool relu, float *out) {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.W = W; g.M = M; g.N = N; g.K = K; g.lda = lda; g.splits = 1;
    g.epi = relu ? EPI_BIAS_RELU_F32 : EPI_BIAS_F32; g.out_f32 = out; g.ldo = N; g.bias = bias;
    g.no_persist = 1;        // the persistent tile loop does not pay with cold weights (profiles/r4_persistent_gemm.md)
    g.coresident = 1;        // M = 160 rows per sub-segment: several tiles per CU, two workgroups per CU overlap fill / multiply / store (96 segments: 3.66 -> 3.41 ms per call)
    if (d->bf16) launch_gemm_bf16(g, d->st);
    else launch_gemm_f32(g, d->st);
}

// SE gate pre-activation (:303-315): z = fc2 . relu(fc1 . mean) over the segments (the sigmoid is applied where the gate is used)
static int spk_se(nasr_diar *d, nasr_diar::SpkBlock &blk, const float *mean, int St) {
    if (launch_spk_fc(mean, blk.cout, blk.fc1, d->zero_bias, d->se_h, d->fc_part, St, blk.cout, blk.cout / 8, 1, d->st) ||
        launch_spk_fc(d->se_h, blk.cout / 8, blk.fc2, d->zero_bias, d->se_z, d->fc_part, St, blk.cout / 8, blk.cout, 0, d->st))
        return failf("TitaNet-L SE gate: bad operands");
    return 0;
}

// The bf16 engine's encoder + pooling on SEGMENT TILES (round 6, kernels_spk.hip): every pointwise conv is a GEMM whose tile is one sub-segment x 256
// channels, and the depthwise conv of the NEXT sub-block, the SE mean, the block's combine and the attentive pooling run in the epilogues.
// s_mel / s_lens are set; leaves d->pool.  ~40 launches, no f32 activation in memory except each block's Y.
static int spk_embed_segment_tiles(nasr_diar *d, int St) {
    float *mean = d->se_z + (size_t)d->max_segments * SPK_C;
    int a = 0, xb = 0;                                            // sA[a]: the A operand the next GEMM reads; sX[xb]: the last block output
    auto gemm = [&](SpkGemmParams &g) -> int {
        g.S = St; g.lens = d->s_lens;
        if (const char *why = spk_gemm_check(g)) return failf("TitaNet-L segment GEMM (mode %d, N %d, K %d): %s", g.mode, g.N, g.K, why);
        return launch_spk_gemm(g, d->st);
    };
    {   // block 0 (:28): depthwise k = 3 on the 80 mel channels (K padded to 128), pointwise -> Y, SE, relu(Y * gate) -> X + block 1's first depthwise conv
        nasr_diar::SpkBlock &blk = d->spk[0];
        nasr_diar::SpkSub &ss = blk.sub[0];
        if (ss.kernel != 3 || ss.cin != DIAR_NMEL || launch_spk_front(d->s_mel, 96, ss.dw, d->s_lens, d->sA[a], ss.cin_pad, St, d->st))
            return failf("TitaNet-L front kernel: block 0 is not the 80 -> 1024, k = 3 sub-block of src/diarize_spk.cpp:28");
        SpkGemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = d->sA[a]; g.lda = ss.cin_pad; g.W = (const bf16_t *)ss.pw; g.N = ss.cout; g.K = ss.cin_pad; g.bias = ss.bias;
        g.mode = SG_Y; g.y_out = d->Y; g.colmean = mean;
        if (gemm(g)) return -1;
        if (spk_se(d, blk, mean, St)) return -1;
        SpkTileParams t;
        memset(&t, 0, sizeof(t));
        t.y_in = d->Y; t.z = d->se_z; t.lens = d->s_lens; t.S = St; t.C = blk.cout; t.mode = ST_DW; t.x_out = d->sX[xb];
        t.dw_w = d->spk[1].sub[0].dw; t.dw_k = d->spk[1].sub[0].kernel; t.a_out = d->sA[a ^ 1]; t.lda_out = d->spk[1].sub[0].cin_pad;
        if (launch_spk_tile(t, d->st)) return failf("TitaNet-L block 0 tile kernel: bad operands");
        a ^= 1;
    }
    for (int b = 1; b <= 3; b++) {                                // Jasper blocks with residual (:351-383)
        nasr_diar::SpkBlock &blk = d->spk[b];
        for (int r = 0; r < blk.repeat; r++) {
            nasr_diar::SpkSub &ss = blk.sub[r];
            SpkGemmParams g;
            memset(&g, 0, sizeof(g));
            g.A = d->sA[a]; g.lda = ss.cin_pad; g.W = (const bf16_t *)ss.pw; g.N = ss.cout; g.K = ss.cin_pad; g.bias = ss.bias;
            if (r + 1 < blk.repeat) {
                g.mode = SG_DW; g.dw_w = blk.sub[r + 1].dw; g.dw_k = blk.sub[r + 1].kernel; g.a_out = d->sA[a ^ 1]; g.lda_out = blk.sub[r + 1].cin_pad;
            } else {
                g.mode = SG_Y; g.y_out = d->Y; g.colmean = mean;
            }
            if (gemm(g)) return -1;
            if (r + 1 < blk.repeat) a ^= 1;
        }
        if (spk_se(d, blk, mean, St)) return -1;
        SpkGemmParams g;                                          // residual 1 x 1 conv + combine (+ the next block's first depthwise conv)
        memset(&g, 0, sizeof(g));
        g.A = d->sX[xb]; g.lda = blk.cin; g.W = (const bf16_t *)blk.res.pw; g.N = blk.cout; g.K = blk.cin; g.bias = blk.res.bias;
        g.mode = SG_RES; g.y_in = d->Y; g.z = d->se_z; g.x_out = d->sX[xb ^ 1];
        nasr_diar::SpkSub &nx = d->spk[b + 1].sub[0];
        if (nx.kernel > 1) { g.dw_w = nx.dw; g.dw_k = nx.kernel; g.a_out = d->sA[a ^ 1]; g.lda_out = nx.cin_pad; }
        if (gemm(g)) return -1;
        xb ^= 1;
        if (nx.kernel > 1) a ^= 1;
    }
    {   // block 4 (:34): kernel 1 (folded into the weights at load), 1024 -> 3072, SE, relu(Y * gate) -> X4 + masked statistics (:392-410)
        nasr_diar::SpkBlock &blk = d->spk[4];
        nasr_diar::SpkSub &ss = blk.sub[0];
        SpkGemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = d->sX[xb]; g.lda = ss.cin_pad; g.W = (const bf16_t *)ss.pw; g.N = ss.cout; g.K = ss.cin_pad; g.bias = ss.bias;
        g.mode = SG_Y; g.y_out = d->Y; g.colmean = mean;
        if (gemm(g)) return -1;
        if (spk_se(d, blk, mean, St)) return -1;
        SpkTileParams t;
        memset(&t, 0, sizeof(t));
        t.y_in = d->Y; t.z = d->se_z; t.lens = d->s_lens; t.S = St; t.C = blk.cout; t.mode = ST_STATS; t.x_out = d->sX4;
        t.mean = d->st_ms; t.stdv = d->st_ms + SPK_C; t.stat_ld = 2 * SPK_C;
        if (launch_spk_tile(t, d->st)) return failf("TitaNet-L block 4 tile kernel: bad operands");
    }
    // attentive statistics pooling (:384-500): the x third of the attention conv is a GEMM on X4, the second conv's logits never leave LDS
    const int M = St * SPK_T;
    if (launch_spk_fc(d->st_ms, 2 * SPK_C, d->a1ms_w, d->a1_b, d->att_c, d->fc_part, St, 2 * SPK_C, SPK_ATT, 0, d->st)) return failf("TitaNet-L attention constant: bad operands");
    spk_gemm(d, d->sX4, SPK_C, d->a1x_w, M, SPK_ATT, SPK_C, d->zero_bias, false, d->att_g);
    launch_spk_att_post(d->att_g, d->att_c, d->a_bn_s, d->a_bn_b, d->A, 1, SPK_ATT, St, d->st);
    SpkGemmParams g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t *)d->A; g.lda = SPK_ATT; g.W = (const bf16_t *)d->a2_w; g.N = SPK_C; g.K = SPK_ATT; g.bias = d->a2_b;
    g.mode = SG_ASP; g.x_in = d->sX4; g.bn_s = d->e_bn_s; g.bn_b = d->e_bn_b; g.pool = d->pool;
    return gemm(g);
}

// spk_session_run_chunk (src/diarize_spk.cpp:601-626) for S sub-segments in one launch sequence
extern "C" int nasr_diar_embed(nasr_diar *d, int S, const float *const *audio, const int32_t *lens_samples, float *emb_out, uint32_t flags) {
    Guard g;
    if (!d || !audio || !lens_samples || !emb_out || S < 1) return failf("null argument or S < 1");
    if (!d->has_spk) return failf("this diarization engine was created without 'spk.*' tensors");
    DCHK(hipSetDevice(d->device));
    for (int s0 = 0; s0 < S; s0 += d->max_segments) {
        const int St = std::min(d->max_segments, S - s0), M = St * SPK_T;
        std::vector<int32_t> n(St, SPK_SEGMENT);
        std::vector<long long> base;
        for (int s = 0; s < St; s++) if (!audio[s0 + s]) return failf("segment %d: null audio", s0 + s);
        if (stage_audio(d, audio + s0, n.data(), St, flags, base)) return -1;
        long long *h_off = (long long *)d->pin;
        int *h_len = (int *)(d->pin + (size_t)d->max_segments * sizeof(long long));
        for (int s = 0; s < St; s++) {
            h_off[s] = base[s];
            int lm = lens_samples[s0 + s] / HOP;                        // :613-615, :548
            h_len[s] = lm > SPK_TVALID ? SPK_TVALID : (lm < 1 ? 1 : lm);
        }
        DCHK(hipMemcpyAsync(d->s_off, h_off, (size_t)St * sizeof(long long), hipMemcpyHostToDevice, d->st));
        DCHK(hipMemcpyAsync(d->s_lens, h_len, (size_t)St * sizeof(int), hipMemcpyHostToDevice, d->st));
        DiarMelParams mp;
        memset(&mp, 0, sizeof(mp));
        set_audio(d, flags, mp);
        mp.win_off = d->s_off; mp.n_win = SPK_SEGMENT; mp.T_pad = SPK_T; mp.t_valid = SPK_TVALID;
        mp.cpitch = 96; mp.mel = d->s_mel; mp.window = d->window; mp.fbT = d->spk_fbT; mp.fb_band = d->spk_band;
        mp.cos_t = d->cos_t; mp.sin_t = d->sin_t;
        DCHK(hipEventRecord(d->ev[1][0], d->st));
        if (d->bf16) {
            // front end + encoder + pooling + embedding layer: eager, or one graph per (St, sample type) replayed
            auto enqueue = [&]() -> int {
                launch_diar_logmel(mp, St, false, d->st);               // the per-feature normalisation (:578) is the first step of k_spk_front
                if (spk_embed_segment_tiles(d, St)) return -1;
                if (launch_spk_fc(d->pool, 2 * SPK_C, d->emb_w, d->emb_b, d->emb, d->fc_part, St, 2 * SPK_C, SPK_EMB, 0, d->st))      // [S][6144] x [192][6144]^T, always f32
                    return failf("TitaNet-L embedding layer: bad operands");
                return 0;
            };
            if (!d->spk_use_graph) {
                if (enqueue()) return -1;
            } else {
                if (d->spk_graph_audio != (const void *)d->audio) {     // the staging buffer was reallocated: every captured pointer to it is stale
                    for (auto &kv : d->spk_graphs) hipGraphExecDestroy(kv.second);
                    d->spk_graphs.clear();
                    d->spk_graph_audio = d->audio;
                }
                const std::pair<int, uint32_t> key(St, flags & NASR_FLAG_AUDIO_S16);
                auto it = d->spk_graphs.find(key);
                if (it == d->spk_graphs.end()) {
                    hipGraph_t graph = nullptr;
                    hipGraphExec_t exec = nullptr;
                    nasr_eng::api_capture_begin();
                    hipError_t be = hipStreamBeginCapture(d->st, hipStreamCaptureModeThreadLocal);
                    int rc = be == hipSuccess ? enqueue() : -1;
                    hipError_t ce = be == hipSuccess ? hipStreamEndCapture(d->st, &graph) : be;
                    nasr_eng::api_capture_end();
                    if (rc || ce != hipSuccess) { if (graph) hipGraphDestroy(graph); return rc ? -1 : failf("TitaNet-L graph capture failed: %s", hipGetErrorString(ce)); }
                    hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                    hipGraphDestroy(graph);
                    if (ie != hipSuccess) return failf("hipGraphInstantiate (TitaNet-L) failed: %s", hipGetErrorString(ie));
                    it = d->spk_graphs.emplace(key, exec).first;
                }
                DCHK(hipGraphLaunch(it->second, d->st));
            }
        } else {
        launch_diar_logmel(mp, St, true, d->st);                        // per-feature normalisation on (:578)
        const float *x = d->s_mel;
        int x_pitch = 96;
        float *outs[2] = {d->X0, d->X1};
        int flip = 0;
        for (int b = 0; b < 5; b++) {                                   // JasperBlock (:351-383)
            nasr_diar::SpkBlock &blk = d->spk[b];
            const float *cur = x;
            int cur_pitch = x_pitch;
            for (int r = 0; r < blk.repeat; r++) {
                nasr_diar::SpkSub &ss = blk.sub[r];
                launch_spk_depthwise(cur, cur_pitch, ss.dw, ss.kernel, ss.cin, ss.cin_pad, d->s_lens, d->A, d->bf16, St, d->st);
                spk_gemm(d, d->A, ss.cin_pad, ss.pw, M, ss.cout, ss.cin_pad, ss.bias, r + 1 < blk.repeat, d->Y);   // ReLU between sub-convs
                cur = d->Y;
                cur_pitch = ss.cout;
            }
            {   // SE gate before the residual (:303-315, :365-368): masked mean over time, two f32 GEMMs over the segments
                float *mean = d->se_z + (size_t)d->max_segments * SPK_C, *hid = d->se_h;
                launch_spk_colmean(d->Y, blk.cout, d->s_lens, mean, St, d->st);
                launch_encproj(mean, blk.fc1, d->zero_bias, hid, St, blk.cout, blk.cout / 8, d->st);
                launch_relu(hid, (int64_t)St * (blk.cout / 8), d->st);
                launch_encproj(hid, blk.fc2, d->zero_bias, d->se_z, St, blk.cout / 8, blk.cout, d->st);
            }
            const float *res = nullptr;
            if (blk.residual) {
                launch_spk_mask_cvt(x, blk.cin, d->s_lens, d->A, d->bf16, St, d->st);
                spk_gemm(d, d->A, blk.cin, blk.res.pw, M, blk.cout, blk.cin, blk.res.bias, false, d->R);
                res = d->R;
            }
            float *out = outs[flip];
            flip ^= 1;
            launch_spk_combine(d->Y, d->se_z, res, blk.cout, d->s_lens, out, St, d->st);
            x = out;
            x_pitch = blk.cout;
        }
        // attentive statistics pooling + embedding (:384-500)
        launch_spk_stats(x, SPK_C, d->s_lens, d->st_mean, d->st_std, St, d->st);
        launch_spk_att_const(d->st_mean, d->st_std, d->a1_w, d->a1_b, d->att_c, SPK_C, SPK_ATT, St, d->st);
        launch_spk_mask_cvt(x, SPK_C, d->s_lens, d->A, d->bf16, St, d->st);
        spk_gemm(d, d->A, SPK_C, d->a1x_w, M, SPK_ATT, SPK_C, d->zero_bias, false, d->att_g);
        launch_spk_att_post(d->att_g, d->att_c, d->a_bn_s, d->a_bn_b, d->A, d->bf16, SPK_ATT, St, d->st);
        spk_gemm(d, d->A, SPK_ATT, d->a2_w, M, SPK_C, SPK_ATT, d->a2_b, false, d->Y);                   // attention logits
        launch_spk_asp(x, d->Y, SPK_C, d->s_lens, d->e_bn_s, d->e_bn_b, d->pool, St, d->st);
        launch_encproj(d->pool, d->emb_w, d->emb_b, d->emb, St, 2 * SPK_C, SPK_EMB, d->st);      // [S][6144] x [192][6144]^T, always f32
        }
        float *h_emb = (float *)(d->pin + (size_t)d->max_segments * (sizeof(long long) + sizeof(int)));
        DCHK(hipEventRecord(d->ev[1][1], d->st));
        DCHK(hipMemcpyAsync(h_emb, d->emb, (size_t)St * SPK_EMB * sizeof(float), hipMemcpyDeviceToHost, d->st));
        DCHK(hipStreamSynchronize(d->st));
        float tile_ms = 0.f;
        if (hipEventElapsedTime(&tile_ms, d->ev[1][0], d->ev[1][1]) != hipSuccess) tile_ms = 0.f;
        d->last_ms[1] = (s0 == 0 ? 0.f : d->last_ms[1]) + tile_ms;
        memcpy(emb_out + (size_t)s0 * SPK_EMB, h_emb, (size_t)St * SPK_EMB * sizeof(float));
    }
    return 0;
}

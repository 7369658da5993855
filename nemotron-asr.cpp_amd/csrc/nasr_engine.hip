// nasr_engine.hip -- host side of the MI355X engine: weights, state pools, engine and stream life cycle (C ABI of
// include/nemotron_asr_amd.h; the other entry points: nasr_abi.hip).
//
// Owns device weights (re-laid-out at upload), the per-stream state pool (K/V rings, conv
// caches, LSTM state, audio/mel rings) and the per-step launch sequence.  The chunk/shift
// arithmetic of the reference's stream manager (src/nemo-stream.cpp:1145-1293,
// src/nemo-stream.h:65-100) is mirrored on the host: every count it needs is a pure
// function of the number of samples pushed, so no device read-back is needed to schedule.
#include "nasr_engine_priv.h"

static thread_local char g_err[512] = "";
namespace nasr_eng {
int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
}  // namespace nasr_eng
namespace nasr {
int set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return -1; }
}
extern "C" const char *nasr_last_error(void) { return g_err; }

static std::shared_mutex g_api_mu;
void api_lock_shared() { g_api_mu.lock_shared(); }
void api_unlock_shared() { g_api_mu.unlock_shared(); }
namespace nasr_eng {
void api_capture_begin() { g_api_mu.unlock_shared(); g_api_mu.lock(); }
void api_capture_end() { g_api_mu.unlock(); g_api_mu.lock_shared(); }
#ifdef NASR_STAMPS
// diagnostic build (make stamps; never shipped): every fused layer kernel stamps the 100 MHz real-time counter at 8 points in
// its first and last workgroup.  One region of 8 x 24 launches per pipeline slot: after a run the buffer holds the last replay
// of every slot's graphs = the time line of the last NSLOT steps on their lanes (tests/micro/stamps_timeline.py reads the dump).
unsigned long long *g_stamp_buf = nullptr;
int g_stamp_pipe = 0;
#endif
}  // namespace nasr_eng
extern "C" int nasr_abi_version(void) { return NASR_ABI_VERSION; }

namespace nasr_eng {
// ---- profiling ------------------------------------------------------------------------------
void prof_flush(nasr_engine *e) {
    Prof &pf = e->prof;
    if (pf.pending.empty()) return;
    hipStreamSynchronize(e->st);
    for (auto &r : pf.pending) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.a, r.b);
        nasr_kernel_stat &s = pf.stats[r.cat];
        s.launches++;
        s.total_ms += ms;
        s.bytes += r.bytes;
        s.flops += r.flops;
        pf.pool.push_back(r.a);
        pf.pool.push_back(r.b);
    }
    pf.pending.clear();
}
}  // namespace nasr_eng

// ---- host helpers: dequantisation of GGUF tensor types at upload ---------------------------------
namespace nasr_eng {
float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ff, f;
    if (exp == 0) {
        if (man == 0) f = sign;
        else {
            exp = 127 - 15 + 1;
            while (!(man & 0x400)) { man <<= 1; exp--; }
            man &= 0x3ff;
            f = sign | (exp << 23) | (man << 13);
        }
    } else if (exp == 31) f = sign | 0x7f800000u | (man << 13);
    else f = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float r;
    memcpy(&r, &f, 4);
    return r;
}

// numel of a desc
int64_t desc_numel(const nasr_weight_desc &d) {
    int64_t n = 1;
    for (int i = 0; i < d.n_dims && i < 4; i++) n *= d.ne[i];
    return n;
}

// returns f32 host copy (dequantised); layouts per scripts/convert_to_gguf.py:118-204
int to_f32(const nasr_weight_desc &d, std::vector<float> &out) {
    const int64_t n = desc_numel(d);
    out.resize((size_t)n);
    switch (d.type) {
    case NASR_TYPE_F32: memcpy(out.data(), d.data, (size_t)n * 4); return 0;
    case NASR_TYPE_F16: {
        const uint16_t *s = (const uint16_t *)d.data;
        for (int64_t i = 0; i < n; i++) out[(size_t)i] = f16_to_f32(s[i]);
        return 0;
    }
    case NASR_TYPE_Q8_0: {   // 34-byte blocks: f16 scale + 32 x int8
        if (n % 32) return fail("%s: Q8_0 numel not a multiple of 32", d.name);
        const uint8_t *s = (const uint8_t *)d.data;
        for (int64_t b = 0; b < n / 32; b++) {
            uint16_t hs; memcpy(&hs, s + b * 34, 2);
            const float sc = f16_to_f32(hs);
            const int8_t *qv = (const int8_t *)(s + b * 34 + 2);
            for (int i = 0; i < 32; i++) out[(size_t)(b * 32 + i)] = sc * (float)qv[i];
        }
        return 0;
    }
    case NASR_TYPE_Q4_0: {   // 18-byte blocks: f16 scale + 16 bytes; low nibbles = elems 0..15, high = 16..31
        if (n % 32) return fail("%s: Q4_0 numel not a multiple of 32", d.name);
        const uint8_t *s = (const uint8_t *)d.data;
        for (int64_t b = 0; b < n / 32; b++) {
            uint16_t hs; memcpy(&hs, s + b * 18, 2);
            const float sc = f16_to_f32(hs);
            const uint8_t *qv = s + b * 18 + 2;
            for (int i = 0; i < 16; i++) {
                out[(size_t)(b * 32 + i)] = sc * (float)((int)(qv[i] & 0xf) - 8);
                out[(size_t)(b * 32 + 16 + i)] = sc * (float)((int)(qv[i] >> 4) - 8);
            }
        }
        return 0;
    }
    }
    return fail("%s: unsupported tensor type %d", d.name, d.type);
}

// the upload-time conversion as a host utility (no device involved): lets a caller, and the CPU test suite, check what the
// engine will compute with for a given GGUF tensor
}  // namespace nasr_eng
extern "C" int64_t nasr_tensor_to_f32(const nasr_weight_desc *t, float *out, int64_t cap) {
    if (!t || !t->data || !out) return fail("null argument");
    if (t->n_dims < 1 || t->n_dims > 4) return fail("bad n_dims %d", t->n_dims);
    for (int i = 0; i < t->n_dims; i++)
        if (t->ne[i] <= 0) return fail("bad extent");
    const int64_t n = desc_numel(*t);
    if (n > cap) return fail("output buffer too small (%lld > %lld)", (long long)n, (long long)cap);
    std::vector<float> v;
    if (to_f32(*t, v)) return -1;
    memcpy(out, v.data(), (size_t)n * 4);
    return n;
}

namespace nasr_eng {
struct Loader {
    nasr_engine *e;
    std::map<std::string, const nasr_weight_desc *> by_name;
    float *scratch = nullptr; size_t scratch_cap = 0;   // device f32 staging for packing

    int get(const std::string &name, int64_t numel, std::vector<float> &out) {
        auto it = by_name.find(name);
        if (it == by_name.end()) return fail("missing tensor: %s", name.c_str());
        if (desc_numel(*it->second) != numel)
            return fail("%s: has %lld elements, expected %lld", name.c_str(), (long long)desc_numel(*it->second), (long long)numel);
        return to_f32(*it->second, out);
    }
    // plain f32 vector / small tensor
    int vec(const std::string &name, int64_t numel, float **dev) {
        std::vector<float> h;
        if (get(name, numel, h)) return -1;
        if (dalloc(e, dev, (size_t)numel)) return -1;
        HIPCHK(hipMemcpy(*dev, h.data(), (size_t)numel * 4, hipMemcpyHostToDevice));
        return 0;
    }
    template <typename Tp>
    int upload_vec(const std::vector<Tp> &h, Tp **dev) {
        if (dalloc(e, dev, h.size())) return -1;
        HIPCHK(hipMemcpy(*dev, h.data(), h.size() * sizeof(Tp), hipMemcpyHostToDevice));
        return 0;
    }
    // matrix [N][K] f32 host -> engine GEMM layout (packed bf16 tiles, or f32 row-major)
    int matrix(const std::vector<float> &h, int N, int K, void **dev) {
        const size_t n = (size_t)N * K;
        if (!e->bf16) {
            float *d;
            if (dalloc(e, &d, n)) return -1;
            HIPCHK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
            *dev = d;
            return 0;
        }
        if (n > scratch_cap) {
            if (scratch) hipFree(scratch);
            HIPCHK(hipMalloc((void **)&scratch, n * 4));
            scratch_cap = n;
        }
        HIPCHK(hipMemcpy(scratch, h.data(), n * 4, hipMemcpyHostToDevice));
        bf16_t *d;
        if (dalloc(e, &d, n)) return -1;
        launch_pack_weight_bf16(scratch, d, N, K, e->st);
        HIPCHK(hipStreamSynchronize(e->st));
        *dev = d;
        return 0;
    }
    int matrix_named(const std::string &name, int N, int K, void **dev) {
        std::vector<float> h;
        if (get(name, (int64_t)N * K, h)) return -1;
        return matrix(h, N, K, dev);
    }
};

void transpose_9x256(const std::vector<float> &w /*[256][9]*/, std::vector<float> &t /*[9][256]*/) {
    t.resize(9 * SUBC);
    for (int c = 0; c < SUBC; c++)
        for (int k = 0; k < 9; k++) t[(size_t)k * SUBC + c] = w[(size_t)c * 9 + k];
}

// f32 MFMA (16x16x4) A-fragment packing for the decoder matrices: tile (nt, kg) = 16 rows x 16 k,
// lane l = q*16 + r holds W[row(nt, r)][kg*16 + 4q .. +4).  lstm_order: tile row r = 4*u + gate
// maps to source row gate*640 + nt*4 + u, so one lane ends up with the 4 gates of one unit.
void pack_f32_mfma(const std::vector<float> &w, int N, int K, bool lstm_order, std::vector<float> &out) {
    const int NT = (N + 15) / 16, KG = K / 16;
    out.assign((size_t)NT * KG * 64 * 4, 0.0f);
    for (int nt = 0; nt < NT; nt++)
        for (int kg = 0; kg < KG; kg++)
            for (int lane = 0; lane < 64; lane++) {
                const int q = lane >> 4, r = lane & 15;
                const int row = lstm_order ? (r & 3) * HID + nt * 4 + (r >> 2) : nt * 16 + r;
                if (row >= N) continue;
                for (int sI = 0; sI < 4; sI++)
                    out[(((size_t)nt * KG + kg) * 64 + lane) * 4 + sI] = w[(size_t)row * K + kg * 16 + 4 * q + sI];
            }
}

void host_pos_emb(int position, float *out) {   // reference src/nemo-ggml.cpp:17-32
    const float p = (float)position;
    for (int i = 0; i < D; i += 2) {
        const float div_term = std::exp(-(float)i * std::log(10000.0f) / (float)D);
        out[i] = std::sin(p * div_term);
        out[i + 1] = std::cos(p * div_term);
    }
}

int load_weights(nasr_engine *e, const nasr_weight_desc *w, int n_w) {
    Loader ld;
    ld.e = e;
    for (int i = 0; i < n_w; i++) {
        if (!w[i].name || !w[i].data) return fail("weight %d: null name/data", i);
        ld.by_name[w[i].name] = &w[i];
    }
    std::vector<float> h, t;
    // ---- a-1 constants: padded window, transposed filterbank, twiddles (src/preprocessor.cpp:80-110,:296-299)
    {
        if (ld.get("preprocessor.featurizer.window", WIN, h)) return -1;
        std::vector<float> win(NFFT, 0.0f);
        memcpy(win.data() + (NFFT - WIN) / 2, h.data(), WIN * 4);
        if (ld.upload_vec(win, &e->window)) return -1;
        if (ld.get("preprocessor.featurizer.fb", (int64_t)NMEL * NBINS, h)) return -1;
        std::vector<float> fbT((size_t)NBINS * NMEL);
        for (int m = 0; m < NMEL; m++)
            for (int k = 0; k < NBINS; k++) fbT[(size_t)k * NMEL + m] = h[(size_t)m * NBINS + k];
        if (ld.upload_vec(fbT, &e->fbT)) return -1;
        // band of every (triangular) filter: the reference sums all 257 bins in order (src/preprocessor.cpp:374-383);
        // outside the band the terms are +0 * power = +0 and sum + 0 == sum, so summing the band alone is bit-identical
        std::vector<int> band(2 * NMEL);
        for (int m = 0; m < NMEL; m++) {
            int lo = NBINS, hi = 0;
            for (int k = 0; k < NBINS; k++)
                if (h[(size_t)m * NBINS + k] != 0.0f) { lo = std::min(lo, k); hi = k + 1; }
            if (lo > hi) lo = hi = 0;
            band[2 * m] = lo; band[2 * m + 1] = hi;
        }
        if (ld.upload_vec(band, &e->fb_band)) return -1;
        std::vector<float> ct(NFFT), sn(NFFT);
        for (int i = 0; i < NFFT; i++) {
            const float theta = (2.0f * (float)M_PI * (float)i) / (float)NFFT;
            sn[i] = sinf(theta);
            ct[i] = cosf(theta);
        }
        if (ld.upload_vec(ct, &e->cos_t) || ld.upload_vec(sn, &e->sin_t)) return -1;
    }
    // ---- a-2 subsampling -----------------------------------------------------------------
    const std::string pe = "encoder.pre_encode.";
    if (ld.get(pe + "conv.0.weight", SUBC * 9, h)) return -1;
    transpose_9x256(h, t);
    if (ld.upload_vec(t, &e->w0t) || ld.vec(pe + "conv.0.bias", SUBC, &e->b0)) return -1;
    if (ld.get(pe + "conv.2.weight", SUBC * 9, h)) return -1;
    transpose_9x256(h, t);
    if (ld.upload_vec(t, &e->w2t) || ld.vec(pe + "conv.2.bias", SUBC, &e->b2)) return -1;
    if (ld.get(pe + "conv.5.weight", SUBC * 9, h)) return -1;
    transpose_9x256(h, t);
    if (ld.upload_vec(t, &e->w5t) || ld.vec(pe + "conv.5.bias", SUBC, &e->b5)) return -1;
    if (ld.matrix_named(pe + "conv.3.weight", SUBC, SUBC, &e->w3) || ld.vec(pe + "conv.3.bias", SUBC, &e->b3)) return -1;
    if (ld.matrix_named(pe + "conv.6.weight", SUBC, SUBC, &e->w6) || ld.vec(pe + "conv.6.bias", SUBC, &e->b6)) return -1;
    {   // out.weight [1024][c*17+w] -> [1024][w*256+c]: our activations are channel-fastest (:1014-1017)
        if (ld.get(pe + "out.weight", (int64_t)D * SUBFLAT, h)) return -1;
        t.resize(h.size());
        for (int n = 0; n < D; n++)
            for (int c = 0; c < SUBC; c++)
                for (int wv = 0; wv < SUBF; wv++)
                    t[(size_t)n * SUBFLAT + wv * SUBC + c] = h[(size_t)n * SUBFLAT + c * SUBF + wv];
        if (ld.matrix(t, D, SUBFLAT, &e->sub_out_w) || ld.vec(pe + "out.bias", D, &e->sub_out_b)) return -1;
    }
    // ---- encoder layers --------------------------------------------------------------------
    const int ks = e->hp.kernel_size;
    e->L.resize(e->hp.n_layers);
    for (int l = 0; l < e->hp.n_layers; l++) {
        LayerW &L = e->L[l];
        memset(&L, 0, sizeof(L));
        const std::string p = "encoder.layers." + std::to_string(l) + ".";
        if (ld.vec(p + "norm_feed_forward1.weight", D, &L.ln_ff1_w) || ld.vec(p + "norm_feed_forward1.bias", D, &L.ln_ff1_b) ||
            ld.vec(p + "norm_self_att.weight", D, &L.ln_att_w) || ld.vec(p + "norm_self_att.bias", D, &L.ln_att_b) ||
            ld.vec(p + "norm_conv.weight", D, &L.ln_conv_w) || ld.vec(p + "norm_conv.bias", D, &L.ln_conv_b) ||
            ld.vec(p + "norm_feed_forward2.weight", D, &L.ln_ff2_w) || ld.vec(p + "norm_feed_forward2.bias", D, &L.ln_ff2_b) ||
            ld.vec(p + "norm_out.weight", D, &L.ln_out_w) || ld.vec(p + "norm_out.bias", D, &L.ln_out_b) ||
            ld.vec(p + "self_attn.pos_bias_u", D, &L.bias_u) || ld.vec(p + "self_attn.pos_bias_v", D, &L.bias_v) ||
            ld.vec(p + "conv.depthwise_conv.weight", (int64_t)ks * D, &L.dw) ||
            ld.vec(p + "conv.batch_norm.weight", D, &L.cln_w) || ld.vec(p + "conv.batch_norm.bias", D, &L.cln_b) ||
            ld.vec(p + "self_attn.linear_pos.weight", (int64_t)D * D, &L.wpos_f32))
            return -1;
        if (ld.matrix_named(p + "feed_forward1.linear1.weight", FF, D, &L.ff1_w1) ||
            ld.matrix_named(p + "feed_forward1.linear2.weight", D, FF, &L.ff1_w2) ||
            ld.matrix_named(p + "feed_forward2.linear1.weight", FF, D, &L.ff2_w1) ||
            ld.matrix_named(p + "feed_forward2.linear2.weight", D, FF, &L.ff2_w2) ||
            ld.matrix_named(p + "self_attn.linear_out.weight", D, D, &L.wo) ||
            ld.matrix_named(p + "conv.pointwise_conv2.weight", D, D, &L.pw2))
            return -1;
        {   // fused QKV [3072][1024]
            std::vector<float> qkv((size_t)3 * D * D), part;
            const char *nm[3] = {"self_attn.linear_q.weight", "self_attn.linear_k.weight", "self_attn.linear_v.weight"};
            for (int i = 0; i < 3; i++) {
                if (ld.get(p + nm[i], (int64_t)D * D, part)) return -1;
                memcpy(qkv.data() + (size_t)i * D * D, part.data(), (size_t)D * D * 4);
            }
            if (ld.matrix(qkv, 3 * D, D, &L.wqkv)) return -1;
        }
        {   // pointwise_conv1 rows interleaved (value c, gate c) so GLU is an epilogue (:657-664)
            if (ld.get(p + "conv.pointwise_conv1.weight", (int64_t)2 * D * D, h)) return -1;
            t.resize(h.size());
            for (int c = 0; c < D; c++) {
                memcpy(&t[(size_t)(2 * c) * D], &h[(size_t)c * D], D * 4);
                memcpy(&t[(size_t)(2 * c + 1) * D], &h[(size_t)(D + c) * D], D * 4);
            }
            if (ld.matrix(t, 2 * D, D, &L.pw1)) return -1;
        }
    }
    // ---- decoder / joint (always f32) -------------------------------------------------------
    const std::string dp = "decoder.prediction.";
    if (ld.vec(dp + "embed.weight", (int64_t)VOCAB * HID, &e->embed)) return -1;
    for (int l = 0; l < 2; l++) {
        const std::string s = std::to_string(l);
        if (ld.get(dp + "dec_rnn.lstm.weight_ih_l" + s, (int64_t)4 * HID * HID, h)) return -1;
        pack_f32_mfma(h, 4 * HID, HID, true, t);
        if (ld.upload_vec(t, &e->w_ih[l])) return -1;
        if (ld.get(dp + "dec_rnn.lstm.weight_hh_l" + s, (int64_t)4 * HID * HID, h)) return -1;
        pack_f32_mfma(h, 4 * HID, HID, true, t);
        if (ld.upload_vec(t, &e->w_hh[l])) return -1;
        if (ld.vec(dp + "dec_rnn.lstm.bias_ih_l" + s, 4 * HID, &e->b_ih[l]) ||
            ld.vec(dp + "dec_rnn.lstm.bias_hh_l" + s, 4 * HID, &e->b_hh[l]))
            return -1;
    }
    if (ld.get("joint.enc.weight", (int64_t)JNT * D, h)) return -1;
    pack_f32_mfma(h, JNT, D, false, t);
    if (ld.upload_vec(t, &e->jenc_w)) return -1;
    if (ld.vec("joint.enc.bias", JNT, &e->jenc_b) ||
        ld.vec("joint.pred.bias", JNT, &e->pred_b) || ld.vec("joint.joint_net.2.bias", VOCAB, &e->out_b))
        return -1;
    if (ld.get("joint.pred.weight", (int64_t)JNT * HID, h)) return -1;
    pack_f32_mfma(h, JNT, HID, false, t);
    if (ld.upload_vec(t, &e->pred_w)) return -1;
    if (ld.get("joint.joint_net.2.weight", (int64_t)VOCAB * JNT, h)) return -1;
    pack_f32_mfma(h, VOCAB, JNT, false, t);
    if (ld.upload_vec(t, &e->out_w)) return -1;
    // ---- prompt kernel (multilingual) -----------------------------------------------------
    const int P = e->hp.num_prompts;
    if (P > 0) {
        if (ld.get("prompt_kernel.0.weight", (int64_t)2048 * (D + P), h)) return -1;
        std::vector<float> a((size_t)2048 * D), pp((size_t)P * 2048);
        for (int n = 0; n < 2048; n++) {
            memcpy(&a[(size_t)n * D], &h[(size_t)n * (D + P)], D * 4);
            for (int k = 0; k < P; k++) pp[(size_t)k * 2048 + n] = h[(size_t)n * (D + P) + D + k];
        }
        if (ld.upload_vec(a, &e->pk1a) || ld.upload_vec(pp, &e->pk1p) || ld.vec("prompt_kernel.0.bias", 2048, &e->pk1_b) ||
            ld.vec("prompt_kernel.2.weight", (int64_t)D * 2048, &e->pk2_w) || ld.vec("prompt_kernel.2.bias", D, &e->pk2_b))
            return -1;
    } else if (ld.by_name.count("prompt_kernel.0.weight")) {
        return fail("prompt_kernel weights present but num_prompts is 0");   // src/nemo-ggml.cpp:431-434
    }
    if (ld.scratch) hipFree(ld.scratch);
    return 0;
}

// pos projection rows for chunk length T, per layer: P[r] = W_pos . emb(rel = 70+T-1-r)
// (reference recomputes this GEMM in every layer of every chunk, src/nemo-stream.cpp:514-516;
// its operands are input-independent so it is done once per (layer, T) here)
int ensure_posproj(nasr_engine *e, int T) {
    if (e->L.empty() || e->L[0].posproj[T]) return 0;
    const int n_rel = LCTX + 2 * T - 1;
    std::vector<float> emb((size_t)n_rel * D);
    for (int r = 0; r < n_rel; r++) host_pos_emb((LCTX + T - 1) - r, &emb[(size_t)r * D]);
    float *demb, *dout;
    HIPCHK(hipMalloc((void **)&demb, emb.size() * 4));
    HIPCHK(hipMalloc((void **)&dout, emb.size() * 4));
    HIPCHK(hipMemcpy(demb, emb.data(), emb.size() * 4, hipMemcpyHostToDevice));
    for (auto &L : e->L) {
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = demb; g.W = L.wpos_f32; g.M = n_rel; g.N = D; g.K = D; g.lda = D;
        g.epi = EPI_PART_F32; g.out_f32 = dout; g.ldo = D; g.splits = 1;
        g.f32_fma_tile = e->opt_f32_mfma ? 0 : 1;
        launch_gemm_f32(g, e->st);
        if (e->bf16) {
            bf16_t *pp;
            if (dalloc(e, &pp, emb.size())) return -1;
            launch_f32_to_bf16(dout, pp, (int64_t)emb.size(), e->st);
            L.posproj[T] = pp;
        } else {
            float *pp;
            if (dalloc(e, &pp, emb.size())) return -1;
            HIPCHK(hipMemcpyAsync(pp, dout, emb.size() * 4, hipMemcpyDeviceToDevice, e->st));
            L.posproj[T] = pp;
        }
        HIPCHK(hipStreamSynchronize(e->st));
    }
    hipFree(demb);
    hipFree(dout);
    return 0;
}

// one workspace set: the buffers a launch sequence passes from kernel to kernel, for w_rows rows
int alloc_ws(nasr_engine *e, nasr_engine::WS &w) {
    const size_t M = (size_t)e->w_rows;
    int rc = 0;
    rc |= dalloc(e, &w.x, M * D);
    rc |= dalloc(e, &w.x2, M * D);
    rc |= dalloc(e, &w.part, 8 * M * D);
    rc |= dalloc(e, &w.q, M * D);
    rc |= dalloc(e, &w.glu, M * D);
    rc |= dalloc(e, &w.hfuse, e->hp.num_prompts > 0 ? M * 2048 : 4);
    { char *p; rc |= dalloc(e, &p, M * D * e->esz); w.a = p; }
    { char *p; rc |= dalloc(e, &p, M * FF * e->esz); w.hbuf = p; }
    { char *p; rc |= dalloc(e, &p, M * D * e->esz); w.ctx = p; }
    { char *p; rc |= dalloc(e, &p, M * D * e->esz); w.cbuf = p; }
    // subsampling ping-pong buffers, [chunks][H2][33][256] (conv0 is fused into the first depthwise conv and never
    // stored).  Rows per encoder frame are largest for multi-chunk steps at R = 0: H2 = 2T + 3 = 5 per frame.
    rc |= dalloc(e, &w.sub_a, M * 5 * 33 * SUBC);
    { char *p; rc |= dalloc(e, &p, (M * 5 * 33 * SUBC) * 4); w.sub_b = p; }
    rc |= dalloc(e, &w.chain_flags, 128);
    if (!rc) hipMemsetAsync(w.chain_flags, 0, 128 * sizeof(unsigned), e->st);      // self re-arming afterwards (chain_wait, kernels_gemm.hip)
    if (!rc) {       // a workspace set starts from zeros (nothing reads it before it is written; an ablated measurement run does)
        hipMemsetAsync(w.x, 0, M * D * 4, e->st); hipMemsetAsync(w.x2, 0, M * D * 4, e->st); hipMemsetAsync(w.part, 0, 8 * M * D * 4, e->st);
        hipMemsetAsync(w.q, 0, M * D * 4, e->st); hipMemsetAsync(w.glu, 0, M * D * 4, e->st);
        hipMemsetAsync(w.a, 0, M * D * e->esz, e->st); hipMemsetAsync(w.hbuf, 0, M * FF * e->esz, e->st);
        hipMemsetAsync(w.ctx, 0, M * D * e->esz, e->st); hipMemsetAsync(w.cbuf, 0, M * D * e->esz, e->st);
        hipMemsetAsync(w.sub_a, 0, M * 5 * 33 * SUBC * 4, e->st); hipMemsetAsync(w.sub_b, 0, M * 5 * 33 * SUBC * 4, e->st);
    }
    return rc;
}
// the enqueue functions address the workspace through the engine's own fields: point them at a set
void use_ws(nasr_engine *e, const nasr_engine::WS &w) {
    e->x = w.x; e->x2 = w.x2; e->part = w.part; e->q = w.q; e->glu = w.glu; e->sub_a = w.sub_a; e->hfuse = w.hfuse;
    e->a = w.a; e->hbuf = w.hbuf; e->ctx = w.ctx; e->cbuf = w.cbuf; e->sub_b = w.sub_b; e->chain_flags = w.chain_flags;
}

// ---------------------------------------------------------------------------------------
}  // namespace nasr_eng
extern "C" int nasr_engine_create(nasr_engine **out, int device_id, int dtype, const nasr_hparams *hp,
                                  const nasr_weight_desc *weights, int n_weights, int max_streams) {
    return nasr_engine_create_ex(out, device_id, dtype, hp, weights, n_weights, max_streams, 0);
}
extern "C" int nasr_engine_create_ex(nasr_engine **out, int device_id, int dtype, const nasr_hparams *hp,
                                     const nasr_weight_desc *weights, int n_weights, int max_streams, int workspace_rows) {
    ApiGuard api_guard;
    if (!out || !hp || !weights) return fail("nasr_engine_create: null argument");
    if (workspace_rows < 0 || workspace_rows > (1 << 20)) return fail("workspace_rows out of range");
    *out = nullptr;
    if (dtype != NASR_DTYPE_F32 && dtype != NASR_DTYPE_BF16) return fail("unsupported dtype %d", dtype);
    if (hp->d_model != D || hp->n_heads != NH || hp->d_head != DH || hp->d_ff != FF || hp->n_mels != NMEL ||
        hp->vocab_size != VOCAB || hp->decoder_dim != HID || hp->joint_dim != JNT || hp->att_left_context != LCTX ||
        hp->subsampling_factor != 8)
        return fail("this build is specialised for d_model 1024 / 8x128 heads / d_ff 4096 / 128 mels / vocab 1025 / "
                    "LSTM 640 / left context 70 (got d_model=%d heads=%d d_ff=%d vocab=%d)",
                    hp->d_model, hp->n_heads, hp->d_ff, hp->vocab_size);
    if (hp->n_layers < 1 || hp->kernel_size < 2 || hp->kernel_size > MAX_KS) return fail("bad n_layers/kernel_size");
    if (max_streams < 1 || max_streams > 4096) return fail("max_streams out of range");
    int n_dev = 0;
    HIPCHK(hipGetDeviceCount(&n_dev));
    if (n_dev <= 0) return fail("no HIP device visible: the MI355X engine has no CPU fallback");
    if (device_id < 0 || device_id >= n_dev) return fail("device %d out of range (%d visible)", device_id, n_dev);
    HIPCHK(hipSetDevice(device_id));
    nasr_engine *e = new nasr_engine();
    e->device = device_id; e->dtype = dtype; e->max_streams = max_streams; e->hp = *hp;
    e->bf16 = dtype == NASR_DTYPE_BF16; e->esz = e->bf16 ? 2 : 4;
    if (hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking) != hipSuccess) { delete e; return fail("hipStreamCreate failed"); }
    init_gemm_kernel_attributes();
    init_fused_kernel_attributes();
#ifdef NASR_STAMPS
    if (!g_stamp_buf) { hipMalloc((void **)&g_stamp_buf, (size_t)STAMP_SLOTS * STAMP_PER_SLOT * 32 * 8); hipMemset(g_stamp_buf, 0, (size_t)STAMP_SLOTS * STAMP_PER_SLOT * 32 * 8); }
#endif
    if (load_weights(e, weights, n_weights)) { engine_destroy_impl(e); return -1; }

    const size_t S = (size_t)max_streams, Lr = (size_t)hp->n_layers, ks1 = (size_t)hp->kernel_size - 1;
    e->w_rows = std::max(std::max(max_streams * TMAX, MAXNEW), workspace_rows);      // workspace_rows: room for several chunks of every stream in ONE launch sequence (a server's backlog)
    const size_t M = (size_t)e->w_rows;
    int rc = 0;
    rc |= dalloc(e, &e->abuf, S * 2 * ABUF_CAP);
    rc |= dalloc(e, &e->last_sample, S);
    rc |= dalloc(e, &e->mel_ring, S * MEL_RING * NMEL);
    e->kv_pool.resize(Lr); e->cc_pool.resize(Lr);
    for (size_t l = 0; l < Lr && !rc; l++) {
        char *kp;
        rc |= dalloc(e, &kp, S * 2 * KVC * D * e->esz);
        e->kv_pool[l] = kp;
        rc |= dalloc(e, &e->cc_pool[l], S * 2 * ks1 * D);
        // the K/V rings are zeroed HERE and never again: a stream that starts on a used slot finds the previous stream's rows,
        // all finite, all behind the validity mask (stream_zero_state)
        if (!rc && hipMemsetAsync(kp, 0, S * 2 * KVC * D * e->esz, e->st) != hipSuccess) rc = fail("hipMemsetAsync (K/V pool) failed");
    }
    if (!rc) {
        rc |= dalloc(e, &e->cc_ptrs_dev, Lr);
        if (!rc && hipMemcpy(e->cc_ptrs_dev, e->cc_pool.data(), Lr * sizeof(float *), hipMemcpyHostToDevice) != hipSuccess) rc = fail("hipMemcpy (conv-cache table) failed");
        rc |= dalloc(e, &e->kv_ptrs_dev, Lr);
        if (!rc && hipMemcpy(e->kv_ptrs_dev, e->kv_pool.data(), Lr * sizeof(void *), hipMemcpyHostToDevice) != hipSuccess) rc = fail("hipMemcpy (K/V ring table) failed");
    }
    rc |= dalloc(e, &e->dec_h, S * 4 * HID);
    rc |= dalloc(e, &e->dec_c, S * 4 * HID);
    rc |= dalloc(e, &e->ctrl, S);
    rc |= dalloc(e, &e->tok_ring, S * TOK_CAP);
    memset(e->ws, 0, sizeof(e->ws));
    rc |= alloc_ws(e, e->ws[0]);
    use_ws(e, e->ws[0]);
    rc |= dalloc(e, &e->encproj, M * JNT);
    rc |= dalloc(e, &e->predg, S * JNT);
    rc |= dalloc(e, &e->key, M);
    rc |= dalloc(e, &e->n_active, 4);
    if (!rc) hipMemsetAsync(e->n_active, 0, 4 * sizeof(int), e->st);      // [3] = ChainParams::error
    rc |= dalloc(e, &e->dlist, S);
    rc |= dalloc(e, &e->rowmap, M);
    rc |= dalloc(e, &e->tok_frame, S * TOK_CAP);
    rc |= dalloc(e, &e->collect_dev, S * (1 + COLLECT_STRIDE) + 4);
    if (rc) { engine_destroy_impl(e); return -1; }
    e->pin_cap = 8u << 20;
    e->pin_off = 256;
    if (hipHostMalloc((void **)&e->pin, e->pin_cap, hipHostMallocDefault) != hipSuccess) { engine_destroy_impl(e); return fail("hipHostMalloc failed"); }
    e->ddesc_cap = 8u << 20;
    if (hipMalloc((void **)&e->ddesc, e->ddesc_cap) != hipSuccess) { engine_destroy_impl(e); return fail("hipMalloc desc failed"); }
    {
        const size_t desc_bytes = S * (sizeof(RowDesc) + sizeof(PcmDesc) + 2 * sizeof(int)) + M * sizeof(RowDesc) + 64;
        const size_t col_bytes = (S * (1 + COLLECT_STRIDE) + 4) * sizeof(int);
        e->desc_bytes = desc_bytes; e->col_bytes = col_bytes;
        if (hipHostMalloc((void **)&e->gh, desc_bytes + col_bytes, hipHostMallocDefault) != hipSuccess) { engine_destroy_impl(e); return fail("hipHostMalloc failed"); }
        e->gh_collect = (int *)(e->gh + desc_bytes);
        if (dalloc(e, &e->g_desc, desc_bytes)) { engine_destroy_impl(e); return -1; }
    }
    e->slots.assign(S, nullptr);
    e->tap_mel_frames.assign(S, 0);
    e->tap_mel_row.assign(S, 0);
    if (hipStreamSynchronize(e->st) != hipSuccess) { engine_destroy_impl(e); return fail("engine init sync failed"); }
    *out = e;
    return 0;
}

// ---- lent streams (nasr_internal.h) ------------------------------------------------------------------------------------------
namespace nasr {
struct LentStream { int borrowers = 0; bool engine_alive = true; };
static std::mutex g_lent_mtx;
static std::map<hipStream_t, LentStream> g_lent;
void lent_stream_register(hipStream_t s) { std::lock_guard<std::mutex> lk(g_lent_mtx); g_lent[s] = LentStream(); }
bool lent_stream_acquire(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_lent_mtx);
    auto it = g_lent.find(s);
    if (it == g_lent.end()) return false;
    it->second.borrowers++;
    return true;
}
void lent_stream_release(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_lent_mtx);
    auto it = g_lent.find(s);
    if (it == g_lent.end()) return;
    if (--it->second.borrowers <= 0 && !it->second.engine_alive) {       // the lender went first: the last borrower cleans up
        hipStreamSynchronize(s);
        hipStreamDestroy(s);
        g_lent.erase(it);
    }
}
int lent_stream_engine_gone(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_lent_mtx);
    auto it = g_lent.find(s);
    if (it == g_lent.end()) return 0;
    if (it->second.borrowers <= 0) { g_lent.erase(it); return 0; }
    it->second.engine_alive = false;
    return it->second.borrowers;
}
}  // namespace nasr

extern "C" void nasr_engine_destroy(nasr_engine *e) {
    ApiGuard api_guard;
    engine_destroy_impl(e);
}
namespace nasr_eng {
void engine_destroy_impl(nasr_engine *e) {
#ifdef NASR_STAMPS
    if (e && g_stamp_buf) {
        hipSetDevice(e->device);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)STAMP_SLOTS * STAMP_PER_SLOT * 32);
        hipMemcpy(h.data(), g_stamp_buf, h.size() * 8, hipMemcpyDeviceToHost);
        const char *path = getenv("NASR_STAMPS_OUT");
        FILE *f = path ? fopen(path, "w") : stderr;
        if (f) {                                       // one line per launch: pipeline slot, launch index (8 x layer + k), 8 + 8 stamps (10 ns ticks)
            for (int ps = 0; ps < STAMP_SLOTS; ps++)
                for (int k = 0; k < STAMP_PER_SLOT; k++) {
                    const unsigned long long *r = &h[((size_t)ps * STAMP_PER_SLOT + k) * 32];
                    if (!r[0]) continue;
                    fprintf(f, "%d %d", ps, k);
                    for (int i = 0; i < 8; i++) fprintf(f, " %llu", r[i]);
                    for (int i = 0; i < 8; i++) fprintf(f, " %llu", r[16 + i]);
                    fprintf(f, "\n");
                }
            if (f != stderr) fclose(f);
        }
    }
#endif
    if (!e) return;
    hipSetDevice(e->device);
    if (e->st) hipStreamSynchronize(e->st);
    if (getenv("NASR_STATS"))
        fprintf(stderr, "nasr: graph replays %lld (%lld pipelined), eager steps %lld, decode fallbacks %lld (%lld rounds); pipelined steps: host %.1f us in "
                "hipGraphLaunch + %.1f us waiting per step\n", (long long)e->graph_replays,
                (long long)e->pipe_steps, (long long)e->eager_steps, (long long)e->decode_fallbacks, (long long)e->decode_fallback_rounds,
                e->pipe_steps + e->gp_steps ? 1e6 * e->host_launch_s / (e->pipe_steps + e->gp_steps) : 0.0,
                e->pipe_steps + e->gp_steps ? 1e6 * e->host_wait_s / (e->pipe_steps + e->gp_steps) : 0.0);
    if (getenv("NASR_STATS") && e->gp_steps)
        fprintf(stderr, "nasr: grouped pipeline: %lld steps, chain launches %lld through graphs, %lld eager\n", (long long)e->gp_steps,
                (long long)e->gp_graph_chains, (long long)e->gp_eager_chains);
    for (auto *s : e->slots) delete s;
    for (void *p : e->allocs) hipFree(p);
    for (auto &kv : e->graphs) hipGraphExecDestroy(kv.second);
    for (auto &per_slot : e->gp_graphs) for (auto &m : per_slot) for (auto &kv : m) if (kv.second) hipGraphExecDestroy(kv.second);
    for (auto &ce : e->gp_ev) for (auto &ev : ce) if (ev) hipEventDestroy(ev);
    for (int k = 1; k < nasr_engine::MAXSEG; k++) if (e->lane[k]) hipStreamSynchronize(e->lane[k]);
    for (int p = 0; p < nasr_engine::NSLOT; p++) {
        nasr_engine::Pipe &P = e->pipe[p];
        for (auto &m : P.seg_graphs) for (auto &kv : m) hipGraphExecDestroy(kv.second);
        for (auto &kv : P.dec_graphs) hipGraphExecDestroy(kv.second);
        if (P.gh_dmeta) hipHostFree(P.gh_dmeta);
        if (p > 0 && P.gh) hipHostFree(P.gh);            // slot 0 shares the engine's own block
        for (auto ev : P.seg_done) if (ev) hipEventDestroy(ev);
        if (P.dec_done) hipEventDestroy(P.dec_done);
    }
    for (int k = 1; k < nasr_engine::MAXSEG; k++) if (e->lane[k]) hipStreamDestroy(e->lane[k]);
    for (hipStream_t ls : e->lent) {
        hipStreamSynchronize(ls);
        const int held = lent_stream_engine_gone(ls);
        if (held > 0) {          // loud, and safe: the stream stays alive for its borrower(s), the last of which destroys it
            fprintf(stderr, "nasr_engine_destroy: a stream lent by this engine (nasr_engine_lend_stream) is still held by %d client(s); "
                            "it is left to them -- destroy borrowers before the engine\n", held);
            fail("nasr_engine_destroy: a lent stream was still held by %d client(s)", held);
        } else hipStreamDestroy(ls);
    }
    if (e->gh) hipHostFree(e->gh);
    if (e->pin) hipHostFree(e->pin);
    if (e->ddesc) hipFree(e->ddesc);
    if (e->pcm_stage) hipFree(e->pcm_stage);
    for (auto &pin : e->pcm_pin) { if (pin.p) hipHostFree(pin.p); if (pin.copied) hipEventDestroy(pin.copied); }
    if (e->mel_stage) hipFree(e->mel_stage);
    if (e->tap_mel) hipFree(e->tap_mel);
    if (e->tap_sub) hipFree(e->tap_sub);
    if (e->tap_layers) hipFree(e->tap_layers);
    if (e->tap_enc) hipFree(e->tap_enc);
    for (auto &r : e->prof.pending) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (auto ev : e->prof.pool) hipEventDestroy(ev);
    if (e->st) hipStreamDestroy(e->st);
    delete e;
}

// ---- descriptor staging: pinned bump arena -> device arena (async) --------------------------


// ---- streams ------------------------------------------------------------------------------------
// keep_reference_state: what the reference's nemo_stream_context::reset() leaves behind (src/nemo-stream.cpp:95-115):
// encoder_graph.reset() only flips a flag (:31-34) and nothing re-zeroes the cache tensors, and the per-stream
// preprocessor is not touched -- so the conv cache, the K/V rows (hidden by cache_valid_len = 0: every cached key gets
// -1e9 and weight exactly 0) and the preprocessor's carry (un-framed samples, last_sample) survive.
int stream_zero_state(nasr_stream *s, bool keep_reference_state) {
    nasr_engine *e = s->e;
    // ONE launch (k_stream_reset, kernels_front.hip); of the K/V rings (zeroed once at nasr_engine_create, hidden by cache_valid_len = 0
    // afterwards) only the 70 window rows in front of the head are zeroed, so that a non-finite value left by the slot's previous stream
    // cannot meet a weight of 0 (rounds 1-4: 53 fills of 32 MB per stream start on the engine's stream)
    StreamResetParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.cc_pools = e->cc_ptrs_dev; rp.n_layers = e->hp.n_layers; rp.slot = s->slot;
    rp.kv_pools = e->kv_ptrs_dev; rp.esz = (int)e->esz; rp.kv_head = keep_reference_state ? s->kv_head : 0;
    rp.cc_slot_floats = 2 * (e->hp.kernel_size - 1) * D;
    rp.keep_reference_state = keep_reference_state ? 1 : 0;
    rp.abuf = e->abuf; rp.last_sample = e->last_sample; rp.mel_ring = e->mel_ring; rp.dec_h = e->dec_h; rp.dec_c = e->dec_c; rp.ctrl = e->ctrl;
    launch_stream_reset(rp, e->st);
    HIPCHK(hipGetLastError());
    if (!keep_reference_state) {
        s->abuf_cnt = NFFT / 2;                // 256 zero samples pre-seeded, src/preprocessor.cpp:220-221
        s->abuf_par = 0;
        s->kv_head = 0;
        s->cc_par = 0;
    }
    s->mel_start = 0;
    s->mel_count = PRE_CACHE;                  // 9 literal-zero frames, src/nemo-stream.cpp:73-74 (reset: :100-102)
    s->valid_len = 0;                          // :81 (reset: :112)
    s->chunks = 0;
    s->tok_read = 0;
    s->samples_in = 0;
    s->last_T = 0;
    s->last_ws = 0;
    return 0;
}

}  // namespace nasr_eng
extern "C" int nasr_stream_create(nasr_engine *e, int right_context, int prompt_index, nasr_stream **out) {
    ApiGuard api_guard;
    if (!e || !out) return fail("nasr_stream_create: null argument");
    *out = nullptr;
    if (right_context != 0 && right_context != 1 && right_context != 6 && right_context != 13)
        return fail("right_context must be 0, 1, 6 or 13 (src/nemo-stream.h:15-20), got %d", right_context);
    HIPCHK(hipSetDevice(e->device));
    int slot = -1;
    for (int i = 0; i < e->max_streams; i++)
        if (!e->slots[i]) { slot = i; break; }
    if (slot < 0) return fail("stream pool exhausted (max_streams=%d)", e->max_streams);
    if (ensure_posproj(e, 1 + right_context)) return -1;
    nasr_stream *s = new nasr_stream();
    s->e = e; s->slot = slot; s->R = right_context; s->T = 1 + right_context;
    s->prompt = e->hp.num_prompts > 0 ? prompt_index : -1;
    s->alive = true;
    e->slots[slot] = s;
    if (stream_zero_state(s)) { e->slots[slot] = nullptr; delete s; return -1; }
    *out = s;
    return 0;
}

extern "C" int nasr_stream_reset_ex(nasr_stream *s, int mode) {
    ApiGuard api_guard;
    if (!s) return fail("null stream");
    if (mode != NASR_RESET_FRESH && mode != NASR_RESET_REFERENCE) return fail("unknown reset mode %d", mode);
    HIPCHK(hipSetDevice(s->e->device));
    if (pipe_drain(s->e)) return -1;
    s->tok_queue.clear();
    return stream_zero_state(s, mode == NASR_RESET_REFERENCE);
}

// a reset stream == a fresh stream; the reference's own reset is NASR_RESET_REFERENCE (see stream_zero_state)
extern "C" int nasr_stream_reset(nasr_stream *s) { return nasr_stream_reset_ex(s, NASR_RESET_FRESH); }

extern "C" int nasr_stream_destroy(nasr_stream *s) {
    ApiGuard api_guard;
    if (!s) return 0;
    nasr_engine *e = s->e;
    hipSetDevice(e->device);
    pipe_drain(e);
    hipStreamSynchronize(e->st);
    e->slots[s->slot] = nullptr;
    delete s;
    return 0;
}

extern "C" int nasr_stream_set_prompt(nasr_stream *s, int prompt_index) {
    if (!s) return fail("null stream");
    if (s->e->hp.num_prompts <= 0) return fail("model is not multilingual (num_prompts=0)");   // src/nemo-stream.cpp:737-740
    if (prompt_index < 0 || prompt_index >= s->e->hp.num_prompts) return fail("prompt index %d out of range", prompt_index);
    s->prompt = prompt_index;
    return 0;
}


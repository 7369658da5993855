// nemo_amd.cpp -- see nemo_amd.h
#include "nemo_amd.h"

#include <chrono>
#include <cstdio>
#include <cstring>

#include "gguf_reader.h"
#include "nemotron_asr_amd.h"

using nasr_host::GgufFile;
using nasr_host::GgufValue;

std::string tokens_to_text(const std::vector<int> &tokens, const std::vector<std::string> &vocab) {
    std::string out;
    for (int id : tokens) {
        if (id < 0 || id >= (int)vocab.size()) continue;
        const std::string &piece = vocab[(size_t)id];
        if (piece.compare(0, 3, "\xe2\x96\x81") == 0) {   // U+2581
            out += ' ';
            out.append(piece, 3, std::string::npos);
        } else {
            out += piece;
        }
    }
    return out;
}

std::string tokens_to_text(const std::vector<timed_token> &tokens, const std::vector<std::string> &vocab, bool timestamp_words) {
    std::string out;
    for (const timed_token &t : tokens) {
        if (t.token_id < 0 || t.token_id >= (int)vocab.size()) continue;
        const std::string &piece = vocab[(size_t)t.token_id];
        if (piece.compare(0, 3, "\xe2\x96\x81") == 0) {
            out += ' ';
            if (timestamp_words) {
                char stamp[32];
                snprintf(stamp, sizeof(stamp), "{%.2f}", t.to_seconds());
                out += stamp;
            }
            out.append(piece, 3, std::string::npos);
        } else {
            out += piece;
        }
    }
    return out;
}

std::vector<timed_token> nemo_stream_get_timed_tokens(nemo_stream_context *sctx) {
    std::vector<timed_token> out;
    if (!sctx) return out;
    const size_t n = sctx->tokens.size();
    std::vector<int32_t> frames(n ? n : 1);
    const size_t first = n > 4096 ? n - 4096 : 0;       // older frames have left the device ring
    const int got = nasr_stream_get_token_frames(sctx->stream, (int64_t)first, (int32_t)(n - first), frames.data());
    if (got < 0) { fprintf(stderr, "%s: %s\n", __func__, nasr_last_error()); return out; }
    for (size_t i = 0; i < first; i++) out.emplace_back(sctx->tokens[i], -1);
    for (int i = 0; i < got; i++) out.emplace_back(sctx->tokens[first + (size_t)i], frames[(size_t)i]);
    return out;
}

nemo_context *nemo_init_with_device(const char *model_path, int device, int dtype, int max_streams) {
    return nemo_init_with_rows(model_path, device, dtype, max_streams, 0);
}
nemo_context *nemo_init_with_rows(const char *model_path, int device, int dtype, int max_streams, int workspace_rows) {
    if (!model_path) return nullptr;
    GgufFile g;
    std::string err;
    if (!g.open(model_path, err)) {
        fprintf(stderr, "%s: failed to open GGUF file: %s\n", __func__, err.c_str());
        return nullptr;
    }
    nemo_context *ctx = new nemo_context();
    nemo_hparams &hp = ctx->hparams;
    uint32_t v;
    // nemo.* keys (reference src/nemo-ggml.cpp:108-142); absent keys keep their defaults
    if (g.get_u32("nemo.n_mels", v)) hp.n_mels = (int32_t)v;
    if (g.get_u32("nemo.d_model", v)) hp.d_model = (int32_t)v;
    if (g.get_u32("nemo.n_heads", v)) hp.n_heads = (int32_t)v;
    if (g.get_u32("nemo.d_head", v)) hp.d_head = (int32_t)v;
    if (g.get_u32("nemo.d_ff", v)) hp.d_ff = (int32_t)v;
    if (g.get_u32("nemo.n_layers", v)) hp.n_layers = (int32_t)v;
    if (g.get_u32("nemo.vocab_size", v)) hp.vocab_size = (int32_t)v;
    if (g.get_u32("nemo.decoder_dim", v)) hp.decoder_dim = (int32_t)v;
    if (g.get_u32("nemo.joint_dim", v)) hp.joint_dim = (int32_t)v;
    if (g.get_u32("nemo.subsampling_factor", v)) hp.subsampling_factor = (int32_t)v;
    if (g.get_u32("nemo.att_left_context", v)) hp.att_left_context = (int32_t)v;
    if (g.get_u32("nemo.num_prompts", v)) hp.num_prompts = (int32_t)v;
    // vocabulary: string array preferred, legacy 8-byte-record blob otherwise (:149-169)
    if (const GgufValue *vl = g.find("tokenizer.vocab_list"); vl && vl->type == 9) {
        ctx->vocab = vl->arr_s;
    } else if (const GgufValue *vb = g.find("tokenizer.vocab"); vb && vb->type == 8) {
        const size_t n = (size_t)hp.vocab_size - 1;
        for (size_t i = 0; i < n && (i + 1) * 8 <= vb->s.size(); i++) {
            const char *rec = vb->s.data() + i * 8;
            ctx->vocab.emplace_back(rec, strnlen(rec, 8));
        }
    } else {
        fprintf(stderr, "%s: no vocabulary in GGUF (need tokenizer.vocab_list or tokenizer.vocab)\n", __func__);
        delete ctx;
        return nullptr;
    }
    // language prompt dictionary (:171-182)
    const GgufValue *pl = g.find("nemo.prompt_langs"), *pi = g.find("nemo.prompt_ids");
    if (pl && pi && pl->arr_s.size() == pi->arr_i.size())
        for (size_t i = 0; i < pl->arr_s.size(); i++) ctx->prompt_dict[pl->arr_s[i]] = (int)pi->arr_i[i];
    if (hp.num_prompts > 0) ctx->prompt_index = 101;   // "auto" (:459-462)
    // kernel size from the depthwise conv weight, stored (k, C) -> ne[1] = k (:357-360)
    if (const nasr_host::GgufTensor *dw = g.tensor("encoder.layers.0.conv.depthwise_conv.weight")) hp.kernel_size = (int32_t)dw->ne[1];

    std::vector<nasr_weight_desc> descs;
    descs.reserve(g.tensors().size());
    for (const auto &t : g.tensors()) {
        nasr_weight_desc d;
        d.name = t.name.c_str();
        d.type = t.type;
        d.n_dims = t.n_dims;
        for (int i = 0; i < 4; i++) d.ne[i] = t.ne[i];
        d.data = t.data;
        descs.push_back(d);
    }
    nasr_hparams nh;
    nh.n_mels = hp.n_mels; nh.d_model = hp.d_model; nh.n_heads = hp.n_heads; nh.d_head = hp.d_head; nh.d_ff = hp.d_ff;
    nh.n_layers = hp.n_layers; nh.vocab_size = hp.vocab_size; nh.decoder_dim = hp.decoder_dim; nh.joint_dim = hp.joint_dim;
    nh.subsampling_factor = hp.subsampling_factor; nh.att_left_context = hp.att_left_context; nh.kernel_size = hp.kernel_size;
    nh.num_prompts = hp.num_prompts;
    if (nasr_engine_create_ex(&ctx->engine, device, dtype, &nh, descs.data(), (int)descs.size(), max_streams, workspace_rows) < 0) {
        fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
        delete ctx;
        return nullptr;
    }
    ctx->max_streams = max_streams;
    ctx->workspace_rows = std::max(std::max(max_streams * 14, 256), workspace_rows);
    return ctx;
}

nemo_context *nemo_init(const char *model_path) { return nemo_init_with_device(model_path, 0, NASR_DTYPE_BF16, 64); }

bool nemo_set_pipeline(nemo_context *ctx, int depth) {
    if (!ctx || !ctx->engine) return false;
    if (nasr_engine_set_option(ctx->engine, "pipeline", depth) < 0) {
        fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
        return false;
    }
    return true;
}

void nemo_free(nemo_context *ctx) {
    if (!ctx) return;
    nasr_engine_destroy(ctx->engine);
    delete ctx;
}

static bool lookup_lang(nemo_context *ctx, const char *lang, const char *who, int &idx) {
    if (ctx->hparams.num_prompts <= 0) {
        fprintf(stderr, "%s: model is not multilingual (num_prompts=0)\n", who);
        return false;
    }
    auto it = ctx->prompt_dict.find(lang);
    if (it == ctx->prompt_dict.end()) {
        fprintf(stderr, "%s: unknown language code '%s'\n", who, lang);
        return false;
    }
    idx = it->second;
    return true;
}

bool nemo_set_language(nemo_context *ctx, const char *lang) {
    if (!ctx || !lang) return false;
    return lookup_lang(ctx, lang, __func__, ctx->prompt_index);
}

nemo_stream_context *nemo_stream_init(nemo_context *ctx, const nemo_cache_config *config) {
    if (!ctx) return nullptr;
    nemo_stream_context *s = new nemo_stream_context();
    s->nctx = ctx;
    if (config) s->config = *config;
    // architecture fields always come from the loaded header (reference src/nemo-stream.cpp:704-728)
    s->config.att_left_context = ctx->hparams.att_left_context;
    s->config.subsampling_factor = ctx->hparams.subsampling_factor;
    s->config.n_mels = ctx->hparams.n_mels;
    s->prompt_index = ctx->hparams.num_prompts > 0 ? ctx->prompt_index : -1;
    if (nasr_stream_create(ctx->engine, s->config.att_right_context, s->prompt_index, &s->stream) < 0) {
        fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
        delete s;
        return nullptr;
    }
    return s;
}

bool nemo_stream_set_language(nemo_stream_context *sctx, const char *lang) {
    if (!sctx || !lang) return false;
    int idx;
    if (!lookup_lang(sctx->nctx, lang, __func__, idx)) return false;
    if (nasr_stream_set_prompt(sctx->stream, idx) < 0) return false;
    sctx->prompt_index = idx;
    return true;
}

static std::string absorb(nemo_stream_context *s, const int32_t *tok, int n) {
    if (n <= 0) return "";
    std::vector<int> ids(tok, tok + n);
    s->tokens.insert(s->tokens.end(), ids.begin(), ids.end());
    std::string text = tokens_to_text(ids, s->nctx->vocab);
    s->transcript += text;
    return text;
}

bool nemo_stream_process_batch(nemo_stream_context *const *sctx, int B, const int16_t *const *audio,
                               const int *n_samples, std::string *out) {
    if (!sctx || B <= 0 || !audio || !n_samples) return false;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<nasr_stream *> st((size_t)B);
    std::vector<std::vector<int32_t>> buf((size_t)B);
    std::vector<int32_t *> ptr((size_t)B);
    std::vector<int32_t> cap((size_t)B), cnt((size_t)B), ns((size_t)B);
    for (int b = 0; b < B; b++) {
        if (!sctx[b]) return false;
        st[b] = sctx[b]->stream;
        ns[b] = n_samples[b] > 0 ? n_samples[b] : 0;
        cap[b] = (ns[b] / 1280 + 16) * 10;                   // <= 10 symbols per 80 ms frame
        buf[b].resize((size_t)cap[b]);
        ptr[b] = buf[b].data();
    }
    if (nasr_engine_step(sctx[0]->nctx->engine, st.data(), B, audio, ns.data(), ptr.data(), cap.data(), cnt.data(), 0) < 0) {
        fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
        return false;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int b = 0; b < B; b++) {
        nemo_stream_context *s = sctx[b];
        s->total_audio_seconds += (double)ns[b] / s->config.sample_rate;
        s->total_compute_seconds += dt / B;
        std::string text = absorb(s, buf[b].data(), cnt[b]);
        // a full buffer: more tokens may be queued on the stream (nothing is dropped below the ABI) -- fetch them now
        while (cnt[b] >= cap[b]) {
            if (nasr_engine_collect(s->nctx->engine, &st[b], 1, &ptr[b], &cap[b], &cnt[b]) < 0) {
                fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
                return false;
            }
            text += absorb(s, buf[b].data(), cnt[b]);
        }
        if (out) out[b] = text;
        // host-mirror counter: no device synchronisation on the per-call path (a pipelined step stays in flight)
        nasr_stream_stats stt;
        if (nasr_stream_get_progress(s->stream, &stt) == 0) s->total_chunks_processed = stt.chunks;
    }
    return true;
}

std::string nemo_stream_process_incremental(nemo_stream_context *sctx, const int16_t *audio, int n_samples) {
    if (!sctx || !audio || n_samples <= 0) return "";         // reference src/nemo-stream.cpp:1150
    std::string out;
    if (!nemo_stream_process_batch(&sctx, 1, &audio, &n_samples, &out)) return "";
    return out;
}

bool nemo_stream_collect_batch(nemo_stream_context *const *sctx, int B, std::string *out) {
    if (!sctx || B <= 0) return false;
    std::vector<nasr_stream *> st((size_t)B);
    std::vector<std::vector<int32_t>> buf((size_t)B, std::vector<int32_t>(256));
    std::vector<int32_t *> ptr((size_t)B);
    std::vector<int32_t> cap((size_t)B, 256), cnt((size_t)B);
    for (int b = 0; b < B; b++) {
        if (!sctx[b]) return false;
        st[b] = sctx[b]->stream;
        ptr[b] = buf[b].data();
    }
    for (bool more = true; more;) {
        if (nasr_engine_collect(sctx[0]->nctx->engine, st.data(), B, ptr.data(), cap.data(), cnt.data()) < 0) {
            fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
            return false;
        }
        more = false;
        for (int b = 0; b < B; b++) {
            const std::string text = absorb(sctx[b], buf[b].data(), cnt[b]);
            if (out) out[b] += text;
            more = more || cnt[b] >= cap[b];
        }
    }
    return true;
}

bool nemo_stream_finalize_batch(nemo_stream_context *const *sctx, int B, std::string *out) {
    if (!sctx || B <= 0) return false;
    std::vector<nasr_stream *> st((size_t)B);
    std::vector<std::vector<int32_t>> buf((size_t)B, std::vector<int32_t>(256));
    std::vector<int32_t *> ptr((size_t)B);
    std::vector<int32_t> cap((size_t)B, 256), cnt((size_t)B);
    for (int b = 0; b < B; b++) {
        if (!sctx[b] || sctx[b]->nctx != sctx[0]->nctx) return false;
        st[(size_t)b] = sctx[b]->stream;
        ptr[(size_t)b] = buf[(size_t)b].data();
    }
    if (nasr_engine_finalize(sctx[0]->nctx->engine, st.data(), B, ptr.data(), cap.data(), cnt.data()) < 0) {
        fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
        return false;
    }
    for (bool more = true; more;) {
        more = false;
        for (int b = 0; b < B; b++) {
            const std::string text = absorb(sctx[b], buf[(size_t)b].data(), cnt[(size_t)b]);
            if (out) out[b] += text;
            more = more || cnt[(size_t)b] >= cap[(size_t)b];
        }
        if (more && nasr_engine_collect(sctx[0]->nctx->engine, st.data(), B, ptr.data(), cap.data(), cnt.data()) < 0) {
            fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
            return false;
        }
    }
    return true;
}

std::string nemo_stream_finalize(nemo_stream_context *sctx) {
    if (!sctx) return "";
    int32_t tok[256], cap = 256, cnt = 0;
    int32_t *tp = tok;
    if (nasr_engine_finalize(sctx->nctx->engine, &sctx->stream, 1, &tp, &cap, &cnt) < 0) {
        fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
        return "";
    }
    // the text produced by the tail flush (:1292) -- with pipelined steps also the step that was still in flight
    std::string text = absorb(sctx, tok, cnt);
    while (cnt >= cap) {
        if (nasr_engine_collect(sctx->nctx->engine, &sctx->stream, 1, &tp, &cap, &cnt) < 0) {
            fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
            break;
        }
        text += absorb(sctx, tok, cnt);
    }
    return text;
}

std::string nemo_stream_get_transcript(nemo_stream_context *sctx) { return sctx ? sctx->transcript : ""; }

const std::vector<int> &nemo_stream_get_tokens(nemo_stream_context *sctx) {
    static const std::vector<int> empty;
    return sctx ? sctx->tokens : empty;
}

void nemo_stream_reset(nemo_stream_context *sctx) {
    if (!sctx) return;
    // the reference's reset as coded (src/nemo-stream.cpp:95-115): the conv cache and the preprocessor carry survive
    nasr_stream_reset_ex(sctx->stream, NASR_RESET_REFERENCE);
    sctx->tokens.clear();
    sctx->transcript.clear();
    sctx->total_audio_seconds = sctx->total_compute_seconds = 0;
    sctx->total_chunks_processed = 0;
}

void nemo_stream_free(nemo_stream_context *sctx) {
    if (!sctx) return;
    nasr_stream_destroy(sctx->stream);
    delete sctx;
}

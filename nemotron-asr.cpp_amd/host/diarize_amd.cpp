// diarize_amd.cpp -- see diarize_amd.h
#include "diarize_amd.h"

#include <cmath>
#include <cstdio>
#include <cstring>

#include "gguf_reader.h"
#include "nemotron_asr_amd.h"

using nasr_host::GgufFile;
using nasr_host::GgufTensor;

diarize_model *diarize_model_load(const char *gguf_path, int device, int dtype) {
    if (!gguf_path) return nullptr;
    GgufFile g;
    std::string err;
    if (!g.open(gguf_path, err)) {
        fprintf(stderr, "%s: failed to open GGUF file: %s\n", __func__, err.c_str());
        return nullptr;
    }
    std::vector<nasr_weight_desc> descs;
    bool vad = false, spk = false;
    for (const GgufTensor &t : g.tensors()) {
        const bool v = t.name.compare(0, 4, "vad.") == 0, s = t.name.compare(0, 4, "spk.") == 0;
        if (!v && !s) continue;
        nasr_weight_desc d;
        d.name = t.name.c_str();
        d.type = t.type;
        d.n_dims = t.n_dims;
        for (int i = 0; i < 4; i++) d.ne[i] = t.ne[i];
        d.data = t.data;
        descs.push_back(d);
        vad |= v;
        spk |= s;
    }
    if (descs.empty()) {
        fprintf(stderr, "%s: no 'vad.*' / 'spk.*' tensors in %s\n", __func__, gguf_path);
        return nullptr;
    }
    diarize_model *m = new diarize_model();
    m->has_vad = vad;
    m->has_spk = spk;
    if (nasr_diar_create(&m->engine, device, dtype ? NASR_DTYPE_BF16 : NASR_DTYPE_F32, descs.data(), (int)descs.size(), 8192, 64) < 0) {
        fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
        delete m;
        return nullptr;
    }
    return m;
}

void diarize_model_free(diarize_model *m) {
    if (!m) return;
    nasr_diar_destroy(m->engine);
    delete m;
}

size_t vad_run_batch(diarize_model *m, const float *audio, size_t n_samples, std::vector<float> &out) {
    if (!m || !audio || n_samples < 10080) return 0;
    const int32_t n = (int32_t)n_samples, nw = 1 + (n - 10080) / 160;
    const size_t before = out.size();
    out.resize(before + (size_t)nw);
    float *dst = out.data() + before;
    int32_t cap = nw, got = 0;
    if (nasr_diar_vad(m->engine, 1, &audio, &n, &dst, &cap, &got, 0) < 0) {
        fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
        out.resize(before);
        return 0;
    }
    return (size_t)got;
}

// onset / offset hysteresis, minimum duration, merge of close segments, clamp (src/diarize_vad.cpp:507-563)
std::vector<vad_segment> vad_extract_segments(const std::vector<float> &probs, const vad_post_cfg &cfg) {
    std::vector<vad_segment> segs;
    const float fp = cfg.frame_period_sec;
    const int n = (int)probs.size();
    const int min_on = (int)std::ceil(cfg.min_duration_on / fp), min_off = (int)std::ceil(cfg.min_duration_off / fp);
    int start = -1;
    auto close = [&](int end) {
        if (end - start >= min_on) segs.push_back({start * fp - cfg.pad_onset, end * fp + cfg.pad_offset});
        start = -1;
    };
    for (int t = 0; t < n; t++) {
        if (start < 0) { if (probs[t] >= cfg.onset) start = t; }
        else if (probs[t] < cfg.offset) close(t);
    }
    if (start >= 0) close(n);
    if (min_off > 0 && segs.size() >= 2) {
        std::vector<vad_segment> merged(1, segs[0]);
        for (size_t i = 1; i < segs.size(); i++) {
            if ((segs[i].start_sec - merged.back().end_sec) / fp < min_off) merged.back().end_sec = segs[i].end_sec;
            else merged.push_back(segs[i]);
        }
        segs.swap(merged);
    }
    for (vad_segment &s : segs) {
        if (s.start_sec < 0.0f) s.start_sec = 0.0f;
        if (s.end_sec < s.start_sec) s.end_sec = s.start_sec;
    }
    return segs;
}

bool spk_run_subsegments(diarize_model *m, const float *audio, size_t n_samples, const std::vector<size_t> &starts,
                         std::vector<float> &out) {
    if (!m || !audio) return false;
    const size_t S = starts.size();
    out.assign(S * 192, 0.0f);
    if (S == 0) return true;
    std::vector<std::vector<float>> seg(S, std::vector<float>(24000, 0.0f));
    std::vector<const float *> ptr(S);
    std::vector<int32_t> lens(S);
    for (size_t i = 0; i < S; i++) {
        const size_t st = starts[i] < n_samples ? starts[i] : n_samples;
        const size_t len = n_samples - st < 24000 ? n_samples - st : 24000;
        memcpy(seg[i].data(), audio + st, len * sizeof(float));
        ptr[i] = seg[i].data();
        lens[i] = (int32_t)len;
    }
    if (nasr_diar_embed(m->engine, (int)S, ptr.data(), lens.data(), out.data(), 0) < 0) {
        fprintf(stderr, "%s: %s\n", __func__, nasr_last_error());
        return false;
    }
    return true;
}

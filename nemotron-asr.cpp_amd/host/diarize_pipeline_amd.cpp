// diarize_pipeline_amd.cpp -- see diarize_pipeline_amd.h (reference: src/diarize_pipeline.cpp)
#include "diarize_pipeline_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "nemotron_asr_amd.h"

namespace {
constexpr int64_t VAD_WIN = 10080, VAD_SHIFT = 160, SPK_WIN = 24000;      // src/diarize_vad.h:86-89, src/diarize_spk.h:102
constexpr double SR = 16000.0;
constexpr int EMB = 192;
struct PendingSub { int seg_id; int64_t start, lens; };
struct Word { std::string text; double at_sec; int speaker = -1; };
struct Span { float start_sec, end_sec; int speaker; };
}  // namespace

struct diarize_pipeline {
    diarize_pipeline_cfg cfg;
    diarize_model *model = nullptr;
    bool owns_model = false;
    std::vector<float> audio;             // samples [dropped, dropped + audio.size())
    int64_t dropped = 0, total = 0;
    int64_t vad_next = 0;                 // next VAD frame (window at vad_next * 160)
    std::vector<float> probs;
    bool in_speech = false;
    int n_off_run = 0, open_seg_id = -1, next_seg_id = 0, min_off_frames = 60;
    int64_t open_start_frame = -1, open_next_sub = 0;
    std::vector<vad_segment> segments;
    std::vector<std::pair<int64_t, int64_t>> segment_frames;      // the same in VAD frames
    std::vector<diarize_subsegment> subs;
    std::vector<float> embs;              // [subs][192], L2-normalised
    std::vector<PendingSub> pending;
    std::vector<Word> words;
    size_t json_drained = 0;
    std::string word_pending;
    double word_pending_at = 0;
};

namespace {

int64_t sub_shift(const diarize_pipeline &p) { return (int64_t)std::lround(p.cfg.sub_shift_sec * SR); }
int64_t sub_window(const diarize_pipeline &p) { return (int64_t)std::lround(p.cfg.sub_window_sec * SR); }

void queue_sub(diarize_pipeline &p, int64_t start, int64_t lens) {       // emit_subseg (:168-196), deferred to the batch
    p.pending.push_back({p.open_seg_id, start, std::min<int64_t>(lens, SPK_WIN)});
    p.open_next_sub++;
}

// tail sub-segment of the open segment ending at seg_end (:224-243, :341-363)
void queue_tail(diarize_pipeline &p, int64_t seg_end, bool at_eof) {
    const int64_t seg_start = p.open_start_frame * VAD_SHIFT, min_seg = (int64_t)std::lround(p.cfg.min_seg_sec * SR);
    const int64_t covered = seg_start + (p.open_next_sub > 0 ? (p.open_next_sub - 1) * sub_shift(p) + sub_window(p) : 0);
    const int64_t left = seg_end - covered;
    if (left >= min_seg && (p.open_next_sub > 0 || at_eof)) queue_sub(p, covered, left);
    else if (p.open_next_sub == 0 && seg_end - seg_start >= min_seg) queue_sub(p, seg_start, seg_end - seg_start);
}

void close_segment(diarize_pipeline &p, int64_t end_frame) {             // :153-165
    p.segments.push_back({(float)p.open_start_frame * 0.01f, (float)end_frame * 0.01f});
    p.segment_frames.push_back({p.open_start_frame, end_frame});
    p.in_speech = false;
    p.open_seg_id = -1;
    p.open_start_frame = -1;
    p.open_next_sub = 0;
    p.n_off_run = 0;
}

// the onset / offset state machine over one new probability (:207-251), then the sub-segments whose end the frame stream
// has passed (:253-263 with "audio available" = through the end of this frame's window)
void step_frame(diarize_pipeline &p, float prob) {
    if (!p.in_speech) {
        if (prob >= p.cfg.vad_post.onset) {
            p.in_speech = true;
            p.open_seg_id = p.next_seg_id++;
            p.open_start_frame = p.vad_next;
            p.open_next_sub = 0;
            p.n_off_run = 0;
        }
    } else if (prob < p.cfg.vad_post.offset) {
        if (++p.n_off_run >= p.min_off_frames) {
            int64_t end_frame = p.vad_next + 1 - p.n_off_run;            // the segment ended min_off_frames ago
            if (end_frame < p.open_start_frame) end_frame = p.open_start_frame;
            queue_tail(p, end_frame * VAD_SHIFT, false);
            close_segment(p, end_frame);
        }
    } else {
        p.n_off_run = 0;
    }
    const int64_t have_through = p.vad_next * VAD_SHIFT + VAD_WIN;
    p.vad_next++;
    if (p.in_speech) {
        const int64_t seg_start = p.open_start_frame * VAD_SHIFT;
        while (seg_start + p.open_next_sub * sub_shift(p) + sub_window(p) <= have_through)
            queue_sub(p, seg_start + p.open_next_sub * sub_shift(p), sub_window(p));
    }
}

// ONE embedding launch sequence for everything queued (groups of 256 sub-segments bound the staging memory)
bool flush_pending(diarize_pipeline &p) {
    bool ok = true;
    for (size_t g0 = 0; g0 < p.pending.size() && ok; g0 += 256) {
        const size_t S = std::min<size_t>(256, p.pending.size() - g0);
        std::vector<float> chunk(S * (size_t)SPK_WIN, 0.0f), e(S * EMB);
        std::vector<const float *> ptr(S);
        std::vector<int32_t> lens(S);
        for (size_t i = 0; i < S; i++) {
            const PendingSub &s = p.pending[g0 + i];
            const int64_t k = s.start - p.dropped;
            if (k >= 0 && k < (int64_t)p.audio.size())
                memcpy(&chunk[i * (size_t)SPK_WIN], p.audio.data() + k, (size_t)std::min<int64_t>(s.lens, (int64_t)p.audio.size() - k) * sizeof(float));
            ptr[i] = &chunk[i * (size_t)SPK_WIN];
            lens[i] = (int32_t)s.lens;
        }
        if (nasr_diar_embed(p.model->engine, (int)S, ptr.data(), lens.data(), e.data(), 0) < 0) {
            fprintf(stderr, "diarize_pipeline: %s\n", nasr_last_error());
            ok = false;
            break;
        }
        for (size_t i = 0; i < S; i++) {
            double n2 = 0;
            for (int d = 0; d < EMB; d++) n2 += (double)e[i * EMB + d] * e[i * EMB + d];
            const float inv = 1.0f / (std::sqrt((float)n2) + 1e-8f);      // L2 normalised for the clustering (:190-193)
            for (int d = 0; d < EMB; d++) p.embs.push_back(e[i * EMB + d] * inv);
            const PendingSub &s = p.pending[g0 + i];
            p.subs.push_back({s.seg_id, (float)((double)s.start / SR), (float)((double)(s.start + s.lens) / SR), -1});
        }
    }
    p.pending.clear();
    return ok;
}

void advance(diarize_pipeline &p) {
    // every window the buffered audio completes: one batched VAD call
    const int64_t have = p.dropped + (int64_t)p.audio.size();
    const int64_t first = p.vad_next * VAD_SHIFT;
    if (have - first >= VAD_WIN) {
        const int64_t n_new = 1 + (have - first - VAD_WIN) / VAD_SHIFT;
        const float *src = p.audio.data() + (first - p.dropped);
        const int32_t n = (int32_t)(VAD_WIN + (n_new - 1) * VAD_SHIFT);
        std::vector<float> pr((size_t)n_new);
        float *dst = pr.data();
        int32_t cap = (int32_t)n_new, got = 0;
        if (nasr_diar_vad(p.model->engine, 1, &src, &n, &dst, &cap, &got, 0) < 0) {
            fprintf(stderr, "diarize_pipeline: %s\n", nasr_last_error());
            return;
        }
        for (int32_t i = 0; i < got; i++) {
            p.probs.push_back(pr[(size_t)i]);
            step_frame(p, pr[(size_t)i]);
        }
    }
    flush_pending(p);
    // drop what neither the next VAD window nor the next sub-segment of an open segment needs (:267-276)
    int64_t keep_from = p.vad_next * VAD_SHIFT;
    if (p.in_speech) keep_from = std::min(keep_from, p.open_start_frame * VAD_SHIFT + p.open_next_sub * sub_shift(p));
    if (keep_from > p.dropped) {
        const int64_t n = std::min<int64_t>(keep_from - p.dropped, (int64_t)p.audio.size());
        p.audio.erase(p.audio.begin(), p.audio.begin() + n);
        p.dropped += n;
    }
}

// labelled sub-segments -> non-overlapping speaker spans: same speaker and touching = extended, different speakers
// overlapping = cut at the midpoint of the overlap (:371-420)
std::vector<Span> speaker_timeline(const std::vector<diarize_subsegment> &subs) {
    std::vector<Span> raw;
    for (const diarize_subsegment &s : subs) raw.push_back({s.start_sec, s.end_sec, s.speaker});
    std::stable_sort(raw.begin(), raw.end(), [](const Span &a, const Span &b) { return a.start_sec < b.start_sec; });
    std::vector<Span> out;
    for (Span s : raw) {
        if (!out.empty()) {
            Span &last = out.back();
            if (last.speaker == s.speaker && s.start_sec <= last.end_sec + 1e-3f) { last.end_sec = std::max(last.end_sec, s.end_sec); continue; }
            if (s.start_sec < last.end_sec) { const float mid = 0.5f * (s.start_sec + last.end_sec); last.end_sec = mid; s.start_sec = mid; }
        }
        out.push_back(s);
    }
    return out;
}

int speaker_at(const std::vector<Span> &tl, double t) {                  // :422-435
    int best = -1;
    for (int lo = 0, hi = (int)tl.size() - 1; lo <= hi;) {
        const int mid = (lo + hi) / 2;
        if (tl[mid].start_sec <= t) { best = mid; lo = mid + 1; } else hi = mid - 1;
    }
    return best >= 0 && t <= tl[best].end_sec ? tl[best].speaker : -1;
}

void flush_word(diarize_pipeline &p) {
    if (p.word_pending.empty()) return;
    p.words.push_back({p.word_pending, p.word_pending_at, -1});
    p.word_pending.clear();
}

}  // namespace

diarize_pipeline *diarize_pipeline_init_with_model(const diarize_pipeline_cfg &cfg, diarize_model *model) {
    if (!model || !model->has_vad || !model->has_spk) {
        fprintf(stderr, "diarize_pipeline: the model needs both 'vad.*' and 'spk.*' tensors\n");
        return nullptr;
    }
    diarize_pipeline *p = new diarize_pipeline();
    p->cfg = cfg;
    p->model = model;
    p->min_off_frames = (int)std::ceil(cfg.vad_post.min_duration_off / cfg.vad_post.frame_period_sec);     // :86
    return p;
}

diarize_pipeline *diarize_pipeline_init(const diarize_pipeline_cfg &cfg) {
    diarize_model *m = diarize_model_load(cfg.diarize_gguf_path.c_str(), cfg.device, cfg.dtype);
    if (!m) {
        fprintf(stderr, "diarize_pipeline: failed to load %s\n", cfg.diarize_gguf_path.c_str());
        return nullptr;
    }
    diarize_pipeline *p = diarize_pipeline_init_with_model(cfg, m);
    if (!p) { diarize_model_free(m); return nullptr; }
    p->owns_model = true;
    return p;
}

void diarize_pipeline_free(diarize_pipeline *p) {
    if (!p) return;
    if (p->owns_model) diarize_model_free(p->model);
    delete p;
}

size_t diarize_pipeline_push_audio(diarize_pipeline *p, const float *audio, size_t n) {
    if (!p || !audio || n == 0) return 0;
    p->audio.insert(p->audio.end(), audio, audio + n);
    p->total += (int64_t)n;
    const size_t before = p->probs.size();
    advance(*p);
    return p->probs.size() - before;
}

// fragments are buffered until whitespace closes the word; a word carries the time of its last fragment (:290-318)
void diarize_pipeline_push_text(diarize_pipeline *p, const std::string &text, double at_sec) {
    if (!p) return;
    for (char c : text) {
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') flush_word(*p);
        else { p->word_pending.push_back(c); p->word_pending_at = at_sec; }
    }
}

std::string diarize_pipeline_drain_json(diarize_pipeline *p) {           // :320-336
    if (!p) return "";
    std::string out;
    for (size_t i = p->json_drained; i < p->words.size(); i++) {
        char buf[256];
        snprintf(buf, sizeof(buf), "{\"word\":\"%s\",\"at\":%.3f}\n", p->words[i].text.c_str(), p->words[i].at_sec);
        out += buf;
    }
    p->json_drained = p->words.size();
    return out;
}

std::string diarize_pipeline_finalize(diarize_pipeline *p) {             // :437-503
    if (!p) return "";
    if (p->in_speech) {                                                  // end of input = end of speech
        const int64_t end_frame = p->vad_next;
        queue_tail(*p, std::min<int64_t>(end_frame * VAD_SHIFT, p->total), true);
        close_segment(*p, end_frame);
    }
    flush_pending(*p);
    flush_word(*p);
    if (p->subs.empty()) return "";
    const nmesc_result cl = nmesc_cluster(p->embs.data(), p->subs.size(), (size_t)EMB, p->cfg.cluster);
    for (size_t i = 0; i < p->subs.size(); i++) p->subs[i].speaker = cl.labels[i];
    const std::vector<Span> timeline = speaker_timeline(p->subs);
    for (Word &w : p->words) w.speaker = speaker_at(timeline, w.at_sec);
    std::ostringstream text;
    int last = -2;
    for (const Word &w : p->words) {
        if (w.speaker != last) {
            if (last != -2) text << "\n";
            text << "[spk_" << (w.speaker < 0 ? -1 : w.speaker) << "] ";
            last = w.speaker;
        }
        text << w.text << " ";
    }
    if (!p->words.empty()) text << "\n";
    const std::string out = text.str();
    if (!p->cfg.speaker_text_path.empty() && p->cfg.speaker_text_path != "-") std::ofstream(p->cfg.speaker_text_path) << out;
    if (!p->cfg.rttm_path.empty()) {
        std::ofstream f(p->cfg.rttm_path);                               // SPEAKER <uri> <chan> <start> <dur> <NA> <NA> <label> <NA> <NA>
        for (const Span &s : timeline)
            if (s.speaker >= 0)
                f << "SPEAKER session 1 " << s.start_sec << " " << (s.end_sec - s.start_sec) << " <NA> <NA> spk_" << s.speaker << " <NA> <NA>\n";
    }
    return out;
}

extern "C" int nasr_diar_plan(const float *probs, int n_probs, long long total_samples, float onset, float offset,
                              float min_duration_off_sec, float sub_window_sec, float sub_shift_sec, float min_seg_sec,
                              long long *segs_out, int segs_cap, int *n_segs, long long *subs_out, int subs_cap, int *n_subs) {
    if (!probs || n_probs < 0 || !n_segs || !n_subs) return -1;
    diarize_pipeline p;                                   // no model: only the host state machine runs
    p.cfg = diarize_pipeline_default_cfg();
    p.cfg.vad_post.onset = onset; p.cfg.vad_post.offset = offset; p.cfg.vad_post.min_duration_off = min_duration_off_sec;
    p.cfg.sub_window_sec = sub_window_sec; p.cfg.sub_shift_sec = sub_shift_sec; p.cfg.min_seg_sec = min_seg_sec;
    p.min_off_frames = (int)std::ceil(min_duration_off_sec / p.cfg.vad_post.frame_period_sec);
    p.total = total_samples;
    for (int i = 0; i < n_probs; i++) step_frame(p, probs[i]);
    if (p.in_speech) {                                    // finalize: end of input = end of speech
        const int64_t end_frame = p.vad_next;
        queue_tail(p, std::min<int64_t>(end_frame * VAD_SHIFT, p.total), true);
        close_segment(p, end_frame);
    }
    *n_segs = (int)p.segment_frames.size();
    *n_subs = (int)p.pending.size();
    int rc = 0;
    for (int i = 0; i < *n_segs; i++) {
        if (!segs_out || i >= segs_cap) { rc = -1; break; }
        segs_out[2 * i] = p.segment_frames[(size_t)i].first; segs_out[2 * i + 1] = p.segment_frames[(size_t)i].second;
    }
    for (int i = 0; i < *n_subs; i++) {
        if (!subs_out || i >= subs_cap) { rc = -1; break; }
        subs_out[3 * i] = p.pending[(size_t)i].seg_id; subs_out[3 * i + 1] = p.pending[(size_t)i].start; subs_out[3 * i + 2] = p.pending[(size_t)i].lens;
    }
    return rc;
}

size_t diarize_pipeline_n_embeddings(const diarize_pipeline *p) { return p ? p->subs.size() + p->pending.size() : 0; }
size_t diarize_pipeline_n_segments(const diarize_pipeline *p) { return p ? p->segments.size() : 0; }
size_t diarize_pipeline_n_words(const diarize_pipeline *p) { return p ? p->words.size() : 0; }
std::vector<diarize_subsegment> diarize_pipeline_subsegments(const diarize_pipeline *p) { return p ? p->subs : std::vector<diarize_subsegment>(); }
std::vector<vad_segment> diarize_pipeline_segments(const diarize_pipeline *p) { return p ? p->segments : std::vector<vad_segment>(); }
const std::vector<float> &diarize_pipeline_vad_probs(const diarize_pipeline *p) { static const std::vector<float> none; return p ? p->probs : none; }

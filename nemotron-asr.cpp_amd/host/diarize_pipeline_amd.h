// diarize_pipeline_amd.h -- the streaming diarization pipeline of the reference (src/diarize_pipeline.h:27-88: same names,
// fields, defaults and outputs) on top of the batched side-car: audio in -> VAD probabilities -> speech segments by
// onset / offset hysteresis -> 1.5 s sub-segments at a 0.75 s shift -> TitaNet-L embeddings -> at EOF NME-SC clustering,
// speaker timeline, words tagged by time, RTTM / speaker-tagged transcript.
// MI355X-first differences (results follow the reference fed sample by sample; what the reference computes is unchanged):
//   * every VAD window a push completes goes through ONE nasr_diar_vad call (the reference runs a ggml graph per 10 ms window)
//   * the sub-segments a push completes are embedded by ONE nasr_diar_embed call
//   * a sub-segment is cut when the VAD frame stream has passed its end (audio through the end of the current VAD window),
//     so the result does not depend on how the caller slices its pushes (the reference emits as far as the buffered audio
//     reaches, src/diarize_pipeline.cpp:253-263 -- with its CLI's 89 ms reads that is the same rule)
#pragma once
#include <cstddef>
#include <string>
#include <vector>

#include "diarize_amd.h"
#include "diarize_cluster_amd.h"

struct diarize_pipeline_cfg {            // src/diarize_pipeline.h:27-50
    std::string diarize_gguf_path;
    int device = 0, dtype = 1;           // in place of diarize_backend: GPU ordinal, 1 = bf16 pointwise convolutions / 0 = f32
    float sub_window_sec = 1.5f, sub_shift_sec = 0.75f, min_seg_sec = 0.5f;
    vad_post_cfg vad_post;
    nmesc_cfg cluster;
    std::string rttm_path, speaker_text_path, json_path;
};

inline diarize_pipeline_cfg diarize_pipeline_default_cfg() {      // src/diarize_pipeline.h:52-62
    diarize_pipeline_cfg c;
    c.vad_post.onset = 0.9f; c.vad_post.offset = 0.5f; c.vad_post.min_duration_on = 0.0f; c.vad_post.min_duration_off = 0.6f;
    c.vad_post.pad_onset = 0.0f; c.vad_post.pad_offset = 0.0f; c.vad_post.frame_period_sec = 0.01f;
    return c;
}

struct diarize_pipeline;
diarize_pipeline *diarize_pipeline_init(const diarize_pipeline_cfg &cfg);
// an already loaded model (not owned): several pipelines = several calls share one set of device weights
diarize_pipeline *diarize_pipeline_init_with_model(const diarize_pipeline_cfg &cfg, diarize_model *model);
void diarize_pipeline_free(diarize_pipeline *p);
size_t diarize_pipeline_push_audio(diarize_pipeline *p, const float *audio, size_t n);      // returns the new VAD frames
void diarize_pipeline_push_text(diarize_pipeline *p, const std::string &text, double at_sec);
std::string diarize_pipeline_drain_json(diarize_pipeline *p);
std::string diarize_pipeline_finalize(diarize_pipeline *p);      // the speaker-tagged transcript; writes rttm / text files
size_t diarize_pipeline_n_embeddings(const diarize_pipeline *p);
size_t diarize_pipeline_n_segments(const diarize_pipeline *p);
size_t diarize_pipeline_n_words(const diarize_pipeline *p);

// The segment / sub-segment plan of a probability track WITHOUT a model or a GPU (host logic only, for tests and tools):
// runs the onset / offset state machine and the sub-segment cursor over probs[0..n) as push_audio + finalize would for an
// audio of total_samples samples.  segs_out: (start_frame, end_frame) pairs; subs_out: (segment id, start sample, samples)
// triples.  Returns 0, or -1 when a capacity is too small (the counts are still written).
extern "C" int nasr_diar_plan(const float *probs, int n_probs, long long total_samples, float onset, float offset,
                              float min_duration_off_sec, float sub_window_sec, float sub_shift_sec, float min_seg_sec,
                              long long *segs_out, int segs_cap, int *n_segs, long long *subs_out, int subs_cap, int *n_subs);

// introspection for tests / tools
struct diarize_subsegment { int seg_id; float start_sec, end_sec; int speaker; };   // speaker = -1 before finalize
std::vector<diarize_subsegment> diarize_pipeline_subsegments(const diarize_pipeline *p);
std::vector<vad_segment> diarize_pipeline_segments(const diarize_pipeline *p);
const std::vector<float> &diarize_pipeline_vad_probs(const diarize_pipeline *p);

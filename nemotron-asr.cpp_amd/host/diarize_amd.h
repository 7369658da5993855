// diarize_amd.h -- host side of the diarization side-car over libnemotron_asr_amd.so: the part of the reference's
// vad_session / spk_session API (src/diarize_vad.h:95-146, src/diarize_spk.h:95-120) that sits on the compute, plus the
// segment extraction of src/diarize_vad.cpp:507-563.  The pipeline above them: diarize_pipeline_amd.h, diarize_cluster_amd.h.
#pragma once
#include <string>
#include <vector>

struct nasr_diar;

struct diarize_model {                 // reference: diarize_model (src/diarize.h) = one diarize.gguf, "vad.*" + "spk.*"
    nasr_diar *engine = nullptr;
    bool has_vad = false, has_spk = false;
};

// dtype: 0 = f32, 1 = bf16 pointwise convolutions in TitaNet (MarbleNet is always f32)
diarize_model *diarize_model_load(const char *gguf_path, int device, int dtype);
void diarize_model_free(diarize_model *m);

// vad_session_run_batch (src/diarize_vad.cpp:490-503): P(speech) of every 0.63 s window at a 10 ms shift, appended to `out`
size_t vad_run_batch(diarize_model *m, const float *audio, size_t n_samples, std::vector<float> &out);

struct vad_segment { float start_sec, end_sec; };
struct vad_post_cfg {                  // src/diarize_vad.h:130-138
    float onset = 0.5f, offset = 0.5f, pad_onset = 0.0f, pad_offset = 0.0f;
    float min_duration_on = 0.0f, min_duration_off = 0.0f, frame_period_sec = 0.01f;
};
std::vector<vad_segment> vad_extract_segments(const std::vector<float> &probs, const vad_post_cfg &cfg);

// spk_session_run_chunk (src/diarize_spk.cpp:601-626) for every 1.5 s sub-segment [start, start + 24000) of `starts`
// (zero padded at the end of the audio): out = [n][192]
bool spk_run_subsegments(diarize_model *m, const float *audio, size_t n_samples, const std::vector<size_t> &starts,
                         std::vector<float> &out);

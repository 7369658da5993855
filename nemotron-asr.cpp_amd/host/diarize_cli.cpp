// diarize-amd -- VAD segments and speaker embeddings of a raw s16le 16 kHz file through diarize.gguf
//   diarize-amd <diarize.gguf> <audio.pcm> [--f32] [--device N] [--onset P] [--offset P] [--sub-shift SEC]
//               [--rttm <file> [--num-speakers K] [--push-ms MS]]   the whole pipeline: segments, sub-segments, NME-SC, RTTM
// prints "SEGMENT start end" per speech segment and "EMBED start_sec e0 e1 e2 e3 ... (192 values)" per 1.5 s sub-segment
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "diarize_amd.h"
#include "diarize_pipeline_amd.h"

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "Usage: %s <diarize.gguf> <audio.pcm> [--f32] [--device N] [--onset P] [--offset P] [--sub-shift SEC]\n", argv[0]);
        return 1;
    }
    int device = 0, dtype = 1;
    vad_post_cfg cfg;
    float sub_shift = 0.75f;
    std::string rttm;
    int num_speakers = -1, push_ms = 0;
    bool have_onset = false, have_offset = false;
    for (int i = 3; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--f32") dtype = 0;
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--onset" && i + 1 < argc) { cfg.onset = (float)atof(argv[++i]); have_onset = true; }
        else if (a == "--offset" && i + 1 < argc) { cfg.offset = (float)atof(argv[++i]); have_offset = true; }
        else if (a == "--rttm" && i + 1 < argc) rttm = argv[++i];
        else if (a == "--num-speakers" && i + 1 < argc) num_speakers = atoi(argv[++i]);
        else if (a == "--push-ms" && i + 1 < argc) push_ms = atoi(argv[++i]);
        else if (a == "--sub-shift" && i + 1 < argc) sub_shift = (float)atof(argv[++i]);
        else { fprintf(stderr, "Unknown flag: %s\n", a.c_str()); return 1; }
    }
    FILE *f = fopen(argv[2], "rb");
    if (!f) { fprintf(stderr, "Failed to open audio file: %s\n", argv[2]); return 1; }
    std::vector<int16_t> pcm;
    int16_t buf[4096];
    size_t got;
    while ((got = fread(buf, sizeof(int16_t), 4096, f)) > 0) pcm.insert(pcm.end(), buf, buf + got);
    fclose(f);
    std::vector<float> audio(pcm.size());
    for (size_t i = 0; i < pcm.size(); i++) audio[i] = (float)pcm[i] / 32768.0f;
    diarize_model *m = diarize_model_load(argv[1], device, dtype);
    if (!m) { fprintf(stderr, "Failed to load diarization model\n"); return 1; }
    if (!rttm.empty()) {
        diarize_pipeline_cfg pc = diarize_pipeline_default_cfg();
        if (have_onset) pc.vad_post.onset = cfg.onset;
        if (have_offset) pc.vad_post.offset = cfg.offset;
        pc.sub_shift_sec = sub_shift;
        pc.cluster.oracle_num_speakers = num_speakers;
        pc.cluster.min_samples_for_nmesc = 4;
        pc.rttm_path = rttm;
        diarize_pipeline *dp = diarize_pipeline_init_with_model(pc, m);
        if (!dp) { diarize_model_free(m); return 1; }
        const size_t step = push_ms > 0 ? (size_t)push_ms * 16 : audio.size();       // default: the whole file in one push
        for (size_t o = 0; o < audio.size(); o += step) diarize_pipeline_push_audio(dp, audio.data() + o, std::min(step, audio.size() - o));
        diarize_pipeline_finalize(dp);
        printf("WINDOWS %zu\n", diarize_pipeline_vad_probs(dp).size());
        for (const vad_segment &s : diarize_pipeline_segments(dp)) printf("SEGMENT %.2f %.2f\n", s.start_sec, s.end_sec);
        for (const diarize_subsegment &s : diarize_pipeline_subsegments(dp)) printf("SUBSEG %.3f %.3f seg %d spk %d\n", s.start_sec, s.end_sec, s.seg_id, s.speaker);
        diarize_pipeline_free(dp);
        diarize_model_free(m);
        return 0;
    }
    if (m->has_vad) {
        std::vector<float> probs;
        vad_run_batch(m, audio.data(), audio.size(), probs);
        printf("WINDOWS %zu\n", probs.size());
        for (const vad_segment &s : vad_extract_segments(probs, cfg)) printf("SEGMENT %.2f %.2f\n", s.start_sec, s.end_sec);
    }
    if (m->has_spk && audio.size() >= 24000) {
        std::vector<size_t> starts;
        for (size_t st = 0; st + 24000 <= audio.size(); st += (size_t)(sub_shift * 16000.0f)) starts.push_back(st);
        std::vector<float> emb;
        if (!spk_run_subsegments(m, audio.data(), audio.size(), starts, emb)) { diarize_model_free(m); return 1; }
        for (size_t i = 0; i < starts.size(); i++) {
            printf("EMBED %.2f", (double)starts[i] / 16000.0);
            for (int k = 0; k < 192; k++) printf(" %.6g", emb[i * 192 + k]);
            printf("\n");
        }
    }
    diarize_model_free(m);
    return 0;
}

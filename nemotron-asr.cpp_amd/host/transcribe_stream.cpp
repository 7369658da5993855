// nemotron-asr-amd -- streaming transcription CLI with the argv / stdout contract of the reference's
// `nemotron-asr.cpp` binary (reference src/transcribe_stream.cpp:33-297): positional
// `model.gguf audio.pcm [chunk_ms] [right_context]`, s16le 16 kHz mono from a file or stdin ("-"),
// text deltas on stdout as they are produced, configuration and the RTF summary on stderr.
// --diarize <diarize.gguf> [--rttm F] [--speaker-text F] [--json F] [--num-speakers K] [--sub-shift SEC] run the
// diarization pipeline beside the ASR stream like the reference's CLI (:146-170, :243-290).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "diarize_pipeline_amd.h"
#include "nemo_amd.h"

static void usage(const char *prog) {
    fprintf(stderr,
            "Usage: %s <model.gguf> <audio.pcm | -> [chunk_ms] [right_context] [--lang CODE] [--f32] [--device N] [--print-tokens] [--read-chunks N] [--timestamps] [--pipeline [E]]\n"
            "  audio: raw s16le, 16 kHz, mono.  right_context in {0, 1, 6, 13} (80 ms .. 1.12 s lookahead)\n"
            "  --read-chunks N: read N chunks of audio per call (default 1 = the reference's read size); a file is\n"
            "                   transcribed fastest with N = 256 and --pipeline 4: same transcript, the chunks of a read share one\n"
            "                   launch sequence and consecutive reads run side by side\n"
            "  --timestamps:    print the final transcript again with {seconds} in front of every word\n"
            "  --pipeline E:    consecutive reads overlap on the GPU, E = 0..4 (same transcript; each delta appears E reads later).\n"
            "                   1: decode of one read beside the encoder of the next; 2..4: the encoder in E pieces on E hardware queues\n"
            "                   (4 = the fastest way through a file).  --pipeline without a number = 1; --pipeline2 / --pipeline3 still work\n"
            "  --cpu | --cuda | --metal: the reference's backend selectors are accepted and ignored (this build has one backend: MI355X)\n"
            "  --diarize <diarize.gguf> [--rttm <file>] [--speaker-text <file>] [--json <file>] [--num-speakers K] [--sub-shift SEC] [--vad-onset P] [--vad-offset P]\n"
            "                   speaker diarization beside the transcript (speaker-tagged transcript on stdout at EOF)\n", prog);
}

int main(int argc, char **argv) {
    if (argc < 3) { usage(argv[0]); return 1; }
    const char *model_path = argv[1], *audio_path = argv[2];
    int chunk_ms = 80, right_context = 0, device = 0, dtype = 1, positional = 0;
    const char *lang = nullptr;
    bool print_tokens = false, timestamps = false;
    int pipeline = 0;
    int read_chunks = 1, num_speakers = -1;
    float sub_shift_sec = 0.75f, vad_onset = -1.0f, vad_offset = -1.0f;
    std::string diarize_gguf, rttm_path, speaker_text_path, json_path;
    const bool from_stdin = strcmp(audio_path, "-") == 0 || strcmp(audio_path, "--stdin") == 0;
    for (int i = 3; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--lang" && i + 1 < argc) lang = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--f32") dtype = 0;
        else if (a == "--print-tokens") print_tokens = true;
        else if (a == "--timestamps") timestamps = true;
        else if (a == "--cpu" || a == "--cuda" || a == "--metal")      // reference src/transcribe_stream.cpp:86-88
            fprintf(stderr, "note: %s ignored -- this build runs on the MI355X HIP engine only\n", a.c_str());
        else if (a == "--pipeline") {
            pipeline = 1;
            if (i + 1 < argc && strlen(argv[i + 1]) == 1 && argv[i + 1][0] >= '0' && argv[i + 1][0] <= '4') pipeline = argv[++i][0] - '0';
        }
        else if (a == "--pipeline2") pipeline = 2;
        else if (a == "--pipeline3") pipeline = 3;
        else if (a == "--pipeline4") pipeline = 4;
        else if (a == "--read-chunks" && i + 1 < argc) read_chunks = atoi(argv[++i]);
        else if (a == "--diarize" && i + 1 < argc) diarize_gguf = argv[++i];
        else if (a == "--rttm" && i + 1 < argc) rttm_path = argv[++i];
        else if (a == "--speaker-text" && i + 1 < argc) speaker_text_path = argv[++i];
        else if (a == "--json" && i + 1 < argc) json_path = argv[++i];
        else if (a == "--num-speakers" && i + 1 < argc) num_speakers = atoi(argv[++i]);
        else if (a == "--sub-shift" && i + 1 < argc) sub_shift_sec = (float)atof(argv[++i]);
        else if (a == "--vad-onset" && i + 1 < argc) vad_onset = (float)atof(argv[++i]);      // default 0.9 / 0.5 (diar_infer_meeting)
        else if (a == "--vad-offset" && i + 1 < argc) vad_offset = (float)atof(argv[++i]);
        else if (!a.empty() && a[0] == '-') { fprintf(stderr, "Unknown flag: %s\n", a.c_str()); return 1; }
        else if (positional == 0) { chunk_ms = atoi(argv[i]); positional++; }
        else if (positional == 1) { right_context = atoi(argv[i]); positional++; }
    }
    if (read_chunks < 1 || read_chunks > 4096) { fprintf(stderr, "--read-chunks must be in 1..4096 (got %d)\n", read_chunks); return 1; }
    if (chunk_ms < 10) { fprintf(stderr, "chunk_ms must be >= 10 (got %d)\n", chunk_ms); return 1; }
    fprintf(stderr, "Configuration:\n  Model:          %s\n  Audio:          %s\n  Chunk size:     %d ms\n  Right context:  %d\n\n",
            model_path, from_stdin ? "stdin" : audio_path, chunk_ms, right_context);

    nemo_context *ctx = nemo_init_with_device(model_path, device, dtype, 1);
    if (!ctx) { fprintf(stderr, "Failed to load ASR model\n"); return 1; }
    if (pipeline && !nemo_set_pipeline(ctx, pipeline)) { fprintf(stderr, "Failed to enable pipelined steps\n"); nemo_free(ctx); return 1; }
    if (lang && !nemo_set_language(ctx, lang)) { fprintf(stderr, "Failed to set language '%s'\n", lang); nemo_free(ctx); return 1; }
    nemo_cache_config cfg = nemo_cache_config::default_config();
    cfg.att_right_context = right_context;
    nemo_stream_context *sctx = nemo_stream_init(ctx, &cfg);
    if (!sctx) { fprintf(stderr, "Failed to create streaming context\n"); nemo_free(ctx); return 1; }

    diarize_pipeline *dp = nullptr;
    if (!diarize_gguf.empty()) {                       // reference :146-170
        diarize_pipeline_cfg dcfg = diarize_pipeline_default_cfg();
        dcfg.diarize_gguf_path = diarize_gguf;
        dcfg.device = device;
        dcfg.dtype = dtype;
        dcfg.sub_shift_sec = sub_shift_sec;
        if (vad_onset >= 0.0f) dcfg.vad_post.onset = vad_onset;
        if (vad_offset >= 0.0f) dcfg.vad_post.offset = vad_offset;
        dcfg.cluster.oracle_num_speakers = num_speakers;
        dcfg.cluster.min_samples_for_nmesc = 4;
        dcfg.rttm_path = rttm_path;
        dcfg.speaker_text_path = speaker_text_path.empty() ? "-" : speaker_text_path;
        dp = diarize_pipeline_init(dcfg);
        if (!dp) { fprintf(stderr, "Failed to init diarization pipeline\n"); nemo_stream_free(sctx); nemo_free(ctx); return 1; }
    }
    FILE *json_file = json_path.empty() || json_path == "-" ? nullptr : fopen(json_path.c_str(), "w");
    auto handle_text = [&](const std::string &text, size_t samples_so_far) {     // reference :196-224
        if (!text.empty()) { fputs(text.c_str(), stdout); fflush(stdout); }
        if (!dp || text.empty()) return;
        diarize_pipeline_push_text(dp, text, (double)samples_so_far / 16000.0);
        if (!json_path.empty()) {
            const std::string j = diarize_pipeline_drain_json(dp);
            if (!j.empty()) fputs(j.c_str(), json_file ? json_file : stdout);
        }
    };
    std::vector<float> f32;
    FILE *in = from_stdin ? stdin : fopen(audio_path, "rb");
    if (!in) { fprintf(stderr, "Failed to open audio file: %s\n", audio_path); nemo_stream_free(sctx); nemo_free(ctx); return 1; }
    // like the reference, the read size is the model's chunk (chunk_ms is validated and printed only)
    std::vector<int16_t> buf((size_t)cfg.get_chunk_samples() + (size_t)(read_chunks - 1) * 1280u * (size_t)(1 + right_context));
    size_t total = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const size_t got = fread(buf.data(), sizeof(int16_t), buf.size(), in);
        if (got == 0) break;
        total += got;
        handle_text(nemo_stream_process_incremental(sctx, buf.data(), (int)got), total);
        if (dp) {
            f32.resize(got);
            for (size_t k = 0; k < got; k++) f32[k] = (float)buf[k] / 32768.0f;
            diarize_pipeline_push_audio(dp, f32.data(), got);
        }
        if (got < buf.size()) break;
    }
    handle_text(nemo_stream_finalize(sctx), total);
    printf("\n");
    if (!from_stdin) fclose(in);
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double audio_s = (double)total / 16000.0;
    fprintf(stderr, "\nAudio duration:   %.2f s\nProcessing time:  %.3f s\nReal-time factor: %.4f (%.1fx real time)\nChunks: %d\n",
            audio_s, wall, audio_s > 0 ? wall / audio_s : 0.0, wall > 0 ? audio_s / wall : 0.0, sctx->total_chunks_processed);
    if (timestamps) printf("%s\n", tokens_to_text(nemo_stream_get_timed_tokens(sctx), ctx->vocab, true).c_str());
    if (print_tokens) {
        printf("TOKENS:");
        for (int t : nemo_stream_get_tokens(sctx)) printf(" %d", t);
        printf("\n");
    }
    if (dp) {                                          // reference :268-292
        fprintf(stderr, "\nFinalizing diarization (%zu sub-segments, %zu words)...\n", diarize_pipeline_n_embeddings(dp), diarize_pipeline_n_words(dp));
        const std::string spk_text = diarize_pipeline_finalize(dp);
        if (!json_path.empty()) {
            const std::string j = diarize_pipeline_drain_json(dp);
            if (!j.empty()) fputs(j.c_str(), json_file ? json_file : stdout);
        }
        if (speaker_text_path.empty() || speaker_text_path == "-") {
            fprintf(stderr, "\n=== Speaker-tagged transcript ===\n");
            fputs(spk_text.c_str(), stdout);
        }
        if (!rttm_path.empty()) fprintf(stderr, "Wrote RTTM: %s\n", rttm_path.c_str());
        diarize_pipeline_free(dp);
    }
    if (json_file) fclose(json_file);
    nemo_stream_free(sctx);
    nemo_free(ctx);
    return 0;
}

// nemotron-asr-amd -- streaming transcription CLI with the argv / stdout contract of the reference's
// `nemotron-asr.cpp` binary (reference src/transcribe_stream.cpp:33-297): positional
// `model.gguf audio.pcm [chunk_ms] [right_context]`, s16le 16 kHz mono from a file or stdin ("-"),
// text deltas on stdout as they are produced, configuration and the RTF summary on stderr.
// Diarization flags are not part of the hot path and are rejected.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "nemo_amd.h"

static void usage(const char *prog) {
    fprintf(stderr,
            "Usage: %s <model.gguf> <audio.pcm | -> [chunk_ms] [right_context] [--lang CODE] [--f32] [--device N] [--print-tokens] [--read-chunks N] [--timestamps]\n"
            "  audio: raw s16le, 16 kHz, mono.  right_context in {0, 1, 6, 13} (80 ms .. 1.12 s lookahead)\n"
            "  --read-chunks N: read N chunks of audio per call (default 1 = the reference's read size); a file is\n"
            "                   transcribed fastest with N = 256: same transcript, the chunks of a read share one launch sequence\n"
            "  --timestamps:    print the final transcript again with {seconds} in front of every word\n", prog);
}

int main(int argc, char **argv) {
    if (argc < 3) { usage(argv[0]); return 1; }
    const char *model_path = argv[1], *audio_path = argv[2];
    int chunk_ms = 80, right_context = 0, device = 0, dtype = 1, positional = 0;
    const char *lang = nullptr;
    bool print_tokens = false, timestamps = false;
    int read_chunks = 1;
    const bool from_stdin = strcmp(audio_path, "-") == 0 || strcmp(audio_path, "--stdin") == 0;
    for (int i = 3; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--lang" && i + 1 < argc) lang = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--f32") dtype = 0;
        else if (a == "--print-tokens") print_tokens = true;
        else if (a == "--timestamps") timestamps = true;
        else if (a == "--read-chunks" && i + 1 < argc) read_chunks = atoi(argv[++i]);
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
    if (lang && !nemo_set_language(ctx, lang)) { fprintf(stderr, "Failed to set language '%s'\n", lang); nemo_free(ctx); return 1; }
    nemo_cache_config cfg = nemo_cache_config::default_config();
    cfg.att_right_context = right_context;
    nemo_stream_context *sctx = nemo_stream_init(ctx, &cfg);
    if (!sctx) { fprintf(stderr, "Failed to create streaming context\n"); nemo_free(ctx); return 1; }

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
        const std::string text = nemo_stream_process_incremental(sctx, buf.data(), (int)got);
        if (!text.empty()) { fputs(text.c_str(), stdout); fflush(stdout); }
        if (got < buf.size()) break;
    }
    const std::string tail = nemo_stream_finalize(sctx);
    if (!tail.empty()) fputs(tail.c_str(), stdout);
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
    nemo_stream_free(sctx);
    nemo_free(ctx);
    return 0;
}

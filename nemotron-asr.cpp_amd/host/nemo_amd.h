// nemo_amd.h -- host-side mirror of the reference's streaming interface on top of the C ABI
// (include/nemotron_asr_amd.h).  Same names, argument meaning and error behaviour as
// reference src/nemo-stream.h:271-326 and src/nemo-ggml.h (nullptr / false / "" + a line on stderr),
// so a caller written against the reference (src/transcribe_stream.cpp, src/nemo-server.cpp)
// compiles against this header unchanged.  The GGUF file is read with host/gguf_reader (no ggml).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

struct nasr_engine;
struct nasr_stream;

struct nemo_hparams {               // reference src/nemo-ggml.h:37-55
    int32_t n_mels = 128, d_model = 1024, n_heads = 8, d_head = 128, d_ff = 4096, n_layers = 24;
    int32_t vocab_size = 1025, decoder_dim = 640, joint_dim = 640, subsampling_factor = 8;
    int32_t att_left_context = 70, kernel_size = 9, num_prompts = 0;
    int32_t blank_token() const { return vocab_size - 1; }
};

struct nemo_context {               // reference: nemo_context / nemo_model (src/nemo-ggml.h:240-252)
    nemo_hparams hparams;
    std::vector<std::string> vocab;
    std::map<std::string, int> prompt_dict;
    int prompt_index = -1;          // default language prompt (101 = "auto") for multilingual models
    nasr_engine *engine = nullptr;
    int max_streams = 0;
    int workspace_rows = 0;          // rows one engine call may carry (streams x chunks x (1 + right_context)); nasr_engine_create_ex
};

enum class nemo_latency_mode { PURE_CAUSAL = 0, ULTRA_LOW = 1, LOW = 6, DEFAULT = 13 };   // src/nemo-stream.h:15-20

struct nemo_cache_config {          // the caller-visible part of reference src/nemo-stream.h:23-128
    int32_t att_left_context = 70, att_right_context = 0;
    int32_t subsampling_factor = 8, n_mels = 128, sample_rate = 16000, hop_length = 160;
    int32_t pre_encode_cache_size = 9, drop_extra_pre_encoded = 2;
    size_t get_chunk_mel_frames() const { return (size_t)(pre_encode_cache_size + subsampling_factor * (1 + att_right_context)); }
    size_t get_shift_mel_frames() const { return (size_t)(subsampling_factor * (1 + att_right_context)); }
    int32_t get_chunk_samples() const { return (int32_t)get_chunk_mel_frames() * hop_length; }
    int32_t get_latency_ms() const { return (int32_t)get_chunk_mel_frames() * hop_length * 1000 / sample_rate; }
    int32_t get_valid_out_len() const { return 1 + att_right_context; }
    static nemo_cache_config with_latency(nemo_latency_mode m) { nemo_cache_config c; c.att_right_context = (int32_t)m; return c; }
    static nemo_cache_config default_config() { return with_latency(nemo_latency_mode::PURE_CAUSAL); }
};

struct nemo_stream_context {        // reference src/nemo-stream.h:177-262 (host-visible members)
    nemo_context *nctx = nullptr;
    nemo_cache_config config;
    nasr_stream *stream = nullptr;
    int prompt_index = -1;
    std::vector<int> tokens;
    std::string transcript;
    double total_audio_seconds = 0, total_compute_seconds = 0;
    int total_chunks_processed = 0;
    double rtf() const { return total_audio_seconds > 0 ? total_compute_seconds / total_audio_seconds : 0; }
};

// ---- model (reference src/nemo-ggml.cpp:444-540) ------------------------------------------------
// dtype: 0 = f32, 1 = bf16 (NASR_DTYPE_*).  Returns nullptr on failure (message on stderr).
nemo_context *nemo_init_with_device(const char *model_path, int device, int dtype, int max_streams);
// ... with room for several chunks of every stream in one engine call (a server's backlog): workspace_rows >= max_streams x 14, 0 = default
nemo_context *nemo_init_with_rows(const char *model_path, int device, int dtype, int max_streams, int workspace_rows);
nemo_context *nemo_init(const char *model_path);              // device 0, bf16, 64 streams
void nemo_free(nemo_context *ctx);
bool nemo_set_language(nemo_context *ctx, const char *lang);  // default prompt for new streams
// MI355X extension: pipelined steps (nasr_engine_set_option "pipeline"): the decode of one call runs beside the encoder of the
// next; nemo_stream_process_incremental then returns each text delta one call later, nemo_stream_finalize returns the rest
bool nemo_set_pipeline(nemo_context *ctx, int depth);   // 0 off, 1 decode beside the next encoder, 2 / 3 / 4: the encoder in that many pieces of consecutive steps side by side

// ---- streaming (reference src/nemo-stream.h:271-326) -------------------------------------------------
nemo_stream_context *nemo_stream_init(nemo_context *ctx, const nemo_cache_config *config = nullptr);
bool nemo_stream_set_language(nemo_stream_context *sctx, const char *lang);
std::string nemo_stream_process_incremental(nemo_stream_context *sctx, const int16_t *audio, int n_samples);
std::string nemo_stream_finalize(nemo_stream_context *sctx);
std::string nemo_stream_get_transcript(nemo_stream_context *sctx);
const std::vector<int> &nemo_stream_get_tokens(nemo_stream_context *sctx);
void nemo_stream_reset(nemo_stream_context *sctx);
void nemo_stream_free(nemo_stream_context *sctx);

// MI355X extension: one launch sequence for B streams that share right_context.  out[b] receives the
// text delta of stream b.  This is what a multi-stream server's worker calls instead of looping.
bool nemo_stream_process_batch(nemo_stream_context *const *sctx, int B, const int16_t *const *audio,
                               const int *n_samples, std::string *out);
// MI355X extension: with pipelined steps, complete the steps in flight of B streams (one engine) and return their text --
// what a batch former calls when its queue runs empty (nasr_engine_collect)
bool nemo_stream_collect_batch(nemo_stream_context *const *sctx, int B, std::string *out);

// MI355X extension: nemo_stream_finalize for B streams of one engine in ONE tail-flush launch sequence (a server ends many
// sessions at once); out[b] += the text the flush (and, with pipelined steps, what was still in flight) produced
bool nemo_stream_finalize_batch(nemo_stream_context *const *sctx, int B, std::string *out);

// token ids -> text: U+2581 starts a word (reference src/nemo-ggml.cpp:1556-1583); ids outside the vocab are skipped
std::string tokens_to_text(const std::vector<int> &tokens, const std::vector<std::string> &vocab);

// reference src/nemo-ggml.h:383-395: a token with the encoder frame it was emitted on (80 ms per frame)
struct timed_token {
    int token_id;
    int64_t frame_idx;
    timed_token(int id = 0, int64_t frame = 0) : token_id(id), frame_idx(frame) {}
    float to_seconds(int frame_samples = 1280, int sample_rate = 16000) const { return (float)frame_idx * frame_samples / sample_rate; }
};
// every token of the stream since init/reset with its frame (the engine keeps the frames of the last 4096 tokens)
std::vector<timed_token> nemo_stream_get_timed_tokens(nemo_stream_context *sctx);
// reference src/nemo-ggml.cpp:1556-1583: "{12.34}" in front of every word when timestamp_words is set
std::string tokens_to_text(const std::vector<timed_token> &tokens, const std::vector<std::string> &vocab, bool timestamp_words);

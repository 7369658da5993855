/*
 * nemotron_asr_amd.h -- C ABI of the MI355X-native streaming Conformer-ASR forward path.
 *
 * Drop-in boundary for the hot path of m1el/nemotron-asr.cpp.  The
 * reference has no plugin/FFI seam: the seam is the set of call sites where its stream
 * manager touches ggml compute.  Each entry point below names the reference interface
 * it replaces (paths relative to the reference root).  Plain pointers and sizes only;
 * no C++ / torch types cross this boundary.  INTEGRATION.md shows the reference-side
 * binding a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; nasr_last_error() gives the text
 *     (reference convention: nullptr/false/"" + fprintf(stderr); we never abort or throw)
 *   - an engine and its streams belong to ONE host thread (reference: one worker thread owns all
 *     backend state, src/nemo-server.cpp:6-10); one engine per GPU.  Several engines may live in
 *     one process, each on its own thread (the calls are serialised only while a step graph is
 *     being captured); nasr_last_error() is per thread
 *   - a push may carry any number of samples: when it completes several chunks of a stream they
 *     run as one launch sequence (up to 256 encoder frames per stream), with the results of
 *     chunk-by-chunk calls
 *   - PCM is s16le 16 kHz mono (src/transcribe_stream.cpp:13); mel is [frames][128] f32
 *     row-major (src/preprocessor.cpp:370-381); encoder out is [T][1024] f32
 *     (src/nemo-stream.cpp:1073-1075); tokens are int32 ids in [0, vocab-1)
 *   - all streams passed to one call must share right_context (static shapes per R,
 *     src/nemo-stream.h:15-20); each stream keeps its own caches and decoder state
 */
#ifndef NEMOTRON_ASR_AMD_H
#define NEMOTRON_ASR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NASR_ABI_VERSION 1

typedef struct nasr_engine nasr_engine; /* device weights + stream pool  (reference: nemo_context, src/nemo-ggml.h:240-252) */
typedef struct nasr_stream nasr_stream; /* per-stream device state       (reference: nemo_stream_context, src/nemo-stream.h:177-262) */

/* compute dtype of the encoder GEMMs / caches */
enum { NASR_DTYPE_F32 = 0, NASR_DTYPE_BF16 = 1 };

/* tensor element types arriving at the seam = GGUF/ggml type ids
 * (scripts/convert_to_gguf.py:30-57).  Q8_0/Q4_0/F16 are dequantised at upload. */
enum { NASR_TYPE_F32 = 0, NASR_TYPE_F16 = 1, NASR_TYPE_Q4_0 = 2, NASR_TYPE_Q8_0 = 8 };

/* step flags */
enum {
    NASR_FLAG_PCM_DEVICE = 1u << 0, /* pcm[] are device pointers (inputs already resident in HBM) */
    NASR_FLAG_NO_SYNC    = 1u << 1, /* do not copy tokens back / synchronise; poll with nasr_engine_collect() */
    NASR_FLAG_AUDIO_S16  = 1u << 2, /* nasr_diar_*: audio[] point at s16 PCM (sample / 32768), e.g. the ASR streams' own buffers */
};

/* model hyper-parameters = the `nemo.*` GGUF keys read at src/nemo-ggml.cpp:108-142
 * (+ kernel_size inferred from conv_dw_w->ne[1], :357-360) */
typedef struct nasr_hparams {
    int32_t n_mels;             /* 128  */
    int32_t d_model;            /* 1024 */
    int32_t n_heads;            /* 8    */
    int32_t d_head;             /* 128  */
    int32_t d_ff;               /* 4096 */
    int32_t n_layers;           /* 24   */
    int32_t vocab_size;         /* 1025 (blank = vocab_size-1) */
    int32_t decoder_dim;        /* 640  */
    int32_t joint_dim;          /* 640  */
    int32_t subsampling_factor; /* 8    */
    int32_t att_left_context;   /* 70   */
    int32_t kernel_size;        /* 9    */
    int32_t num_prompts;        /* 0 (English) or 128 (multilingual) */
} nasr_hparams;

/* one host tensor handed over by the GGUF loader: replaces the
 * fread -> ggml_backend_tensor_set upload loop of src/nemo-ggml.cpp:257-283.
 * ne[] is in ggml order (ne[0] fastest), names are the PyTorch names of :296-398. */
typedef struct nasr_weight_desc {
    const char *name;
    int32_t     type;   /* NASR_TYPE_* */
    int32_t     n_dims;
    int64_t     ne[4];
    const void *data;   /* host memory, only read during nasr_engine_create */
} nasr_weight_desc;

/* host utility, no GPU involved: the f32 values nasr_engine_create derives from one GGUF tensor at upload (F16 / Q8_0 /
 * Q4_0 are dequantised: d * q per block of 32, layouts of scripts/convert_to_gguf.py:118-204).  Returns the number of
 * elements written or < 0. */
int64_t nasr_tensor_to_f32(const nasr_weight_desc *t, float *out, int64_t cap);

typedef struct nasr_stream_stats {
    int64_t samples_in;        /* PCM samples pushed                                       */
    int32_t chunks;            /* encoder steps run      (nemo_stream_context::total_chunks_processed) */
    int32_t decode_iterations; /* LSTM+joint evaluations (::total_decode_iterations)       */
    int32_t tokens;            /* tokens emitted so far                                    */
    int32_t cache_valid_len;   /* (::cache_valid_len)                                      */
    int32_t mel_frames_buffered;
    int32_t reserved;
} nasr_stream_stats;

/* per-kernel-class timing collected with HIP events when profiling is enabled */
typedef struct nasr_kernel_stat {
    char     name[48];
    int64_t  launches;
    double   total_ms;
    double   bytes;   /* algorithmic bytes moved by those launches (weights + activations) */
    double   flops;   /* algorithmic FLOPs of those launches */
} nasr_kernel_stat;

const char *nasr_last_error(void);
int nasr_abi_version(void);

/* ---- engine: replaces nemo_init_with_backend's device side (src/nemo-ggml.cpp:448-495:
 * backend init :35-81, tensor upload :238-292 incl. compute_pos_emb :17-32, :288-292)
 * and nemo_free (:521-540).  max_streams sizes the per-stream state pool. ------------- */
int  nasr_engine_create(nasr_engine **out, int device_id, int dtype, const nasr_hparams *hp,
                        const nasr_weight_desc *weights, int n_weights, int max_streams);
/* the same with the row capacity of one launch sequence stated: workspace_rows >= max_streams x 14 lets a call hand several whole chunks of EVERY
 * stream to the engine at once (B streams x G chunks x (1 + right_context) rows <= workspace_rows): a server working off a backlog then runs
 * GEMMs of G times the rows (host/nemo_server.cpp --backlog-chunks).  0 = the default, max(max_streams x 14, 256).  ~45 KB of HBM per row and step in flight. */
int  nasr_engine_create_ex(nasr_engine **out, int device_id, int dtype, const nasr_hparams *hp,
                           const nasr_weight_desc *weights, int n_weights, int max_streams, int workspace_rows);
void nasr_engine_destroy(nasr_engine *e);

/* ---- streams: replaces nemo_stream_init (src/nemo-stream.cpp:696-733 -> ::init :36-93:
 * zeroed K/V/conv caches :320-325, decoder state zero + prev_token = blank :55-56,
 * 9 zero mel frames :73-74, cache_valid_len = 0 :81), nemo_stream_reset (:1307-1311),
 * nemo_stream_free (:1313-1317), nemo_stream_set_language (:735-749). ------------------ */
int nasr_stream_create(nasr_engine *e, int right_context, int prompt_index, nasr_stream **out);
/* reset == a fresh stream: one launch clears the conv caches, decoder state, mel / audio buffers; the K/V rows go out of
 * sight behind cache_valid_len = 0 (results bit-identical to a new engine's: tests/test_gpu_parity.py) */
int nasr_stream_reset(nasr_stream *s);
/* nemo_stream_reset AS CODED in the reference (src/nemo-stream.cpp:95-115, :31-34, :1307-1311): transcript, decoder state,
 * mel buffer (9 zero frames), cache_valid_len and the counters are reset, but the conv cache and the K/V rows are left as
 * they are (stale K/V is hidden by the validity mask, the stale conv cache is NOT: the first kernel_size-1 frames after the
 * reset see it) and the per-stream preprocessor keeps its carry (un-framed samples, last_sample).  NASR_RESET_REFERENCE
 * reproduces exactly that; the host mirror's nemo_stream_reset() uses it. */
enum { NASR_RESET_FRESH = 0, NASR_RESET_REFERENCE = 1 };
int nasr_stream_reset_ex(nasr_stream *s, int mode);
int nasr_stream_destroy(nasr_stream *s);
int nasr_stream_set_prompt(nasr_stream *s, int prompt_index);
int nasr_stream_get_stats(const nasr_stream *s, nasr_stream_stats *out);
/* the host-mirror part of the stats only (samples_in, chunks, cache_valid_len, mel_frames_buffered; reserved = tokens decoded
 * but not yet handed over; decode_iterations = tokens = -1): no device synchronisation, no copy, does not complete a
 * pipelined step in flight -- for per-call bookkeeping on a server's hot path.  Like every call on a stream it belongs to the
 * engine's ONE host thread (it reads the host mirrors nasr_engine_step mutates, without a lock) */
int nasr_stream_get_progress(const nasr_stream *s, nasr_stream_stats *out);
/* timed_token.frame_idx (src/nemo-ggml.h:383-395; time = frame * 1280 / 16000 s): absolute encoder-frame
 * index of tokens [first, first + count) of this stream, counted from create/reset.  Only the most recent
 * 4096 tokens are kept on the device.  Returns the number written, < 0 on error. */
int nasr_stream_get_token_frames(const nasr_stream *s, int64_t first, int32_t count, int32_t *frames_out);

/* ---- the step: replaces nemo_stream_process_incremental (src/nemo-stream.cpp:1145-1206)
 * for B streams at once = nemo_preprocessor_process (src/preprocessor.cpp:330-395) +
 * every full chunk through process_mel_chunk_streaming (:1013-1128: encoder graph compute
 * :1063, decode_one_step per frame :1107-1118) + the mel-buffer shift (:1189-1195).
 * pcm[b] has n_samples[b] samples (0 allowed).  New token ids of stream b are written to
 * tokens_out[b][0..n_tokens[b]), n_tokens[b] <= tokens_cap[b].  Tokens that do not fit are never dropped: they stay queued
 * on the stream and come out of the next step / collect / finalize call (n_tokens[b] == tokens_cap[b] => call
 * nasr_engine_collect until it returns fewer).  With tokens_out == NULL the tokens are discarded and counted. */
int nasr_engine_step(nasr_engine *e, nasr_stream *const *streams, int B,
                     const int16_t *const *pcm, const int32_t *n_samples,
                     int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens,
                     uint32_t flags);

/* debug/parity tap: same as nasr_engine_step but takes log-mel frames [n_frames][128] f32
 * (host memory) and skips stage a-1, i.e. enters at the mel_buffer append of :1162. */
int nasr_engine_step_mel(nasr_engine *e, nasr_stream *const *streams, int B,
                         const float *const *mel, const int32_t *n_frames,
                         int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens,
                         uint32_t flags);

/* tail flush: replaces nemo_stream_finalize (src/nemo-stream.cpp:1217-1293): if more than
 * 9 mel frames are buffered, n_valid = (frames-9)/8 outputs of one zero-padded step. */
int nasr_engine_finalize(nasr_engine *e, nasr_stream *const *streams, int B,
                         int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens);

/* with NASR_FLAG_NO_SYNC: wait for outstanding work and fetch the tokens produced since the
 * last collect (replaces nemo_stream_get_tokens, src/nemo-stream.cpp:1301-1305, as a delta) */
int nasr_engine_collect(nasr_engine *e, nasr_stream *const *streams, int B,
                        int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens);

/* ---- parity taps (reference: append_dump_tensor, src/nemo-stream.cpp:982-1010) --------- */
enum {
    NASR_TAP_MEL         = 0, /* log-mel frames produced by the last step call  [n][128]       */
    NASR_TAP_SUBSAMPLED  = 1, /* conformer input of the last chunk (after drop-2) [T][1024]     */
    NASR_TAP_LAYER_OUT   = 2, /* output of layer `index` for the last chunk      [T][1024]     */
    NASR_TAP_ENCODER_OUT = 3, /* encoder output of the last chunk                [T][1024]     */
    NASR_TAP_K_CACHE     = 4, /* K cache of layer `index`, logical order         [70][1024]    */
    NASR_TAP_V_CACHE     = 5,
    NASR_TAP_CONV_CACHE  = 6, /*                                                 [ks-1][1024]  */
    NASR_TAP_DEC_STATE   = 7, /* h[2][640], c[2][640], then prev_token as float                */
};
/* engine options: "fused" (1: small-M fused layer kernels, default) / "graph" (1: hipGraph replay of the
 * steady-state step, default) / "multichunk" (1, default) are pure performance switches; results are unchanged.
 * "pipeline" (0, default; E = 1..4): launch sequences of consecutive steps run beside each other on their own HIP streams.
 * The encoder is cut into E pieces of L / E layers and a step's piece k + 1 runs one call after its piece k, beside piece k
 * of the next step; the decode follows one call after the last piece (layer l of a step needs from the previous step only
 * what its layer l left in the K/V ring and the conv cache).  E = 1: the decode graph of step s beside the encoder graph
 * of step s + 1.  The same tokens come out, E calls later: nasr_engine_step returns what has been decoded so far;
 * nasr_engine_finalize, nasr_engine_collect and every other entry point first complete the steps in flight.  Results are
 * bit-identical to synchronous stepping.  The engine runs at most as many pieces as it finds HIP streams that truly run side
 * by side (it measures which streams share a hardware queue at the first pipelined step: normally 4).  The decode graphs run on
 * the last of these streams: a queue of their own up to E = 3, right behind the fourth piece at E = 4 (worth 1-5 % at 16-64
 * streams, nothing at one stream); fewer pieces when the process leaves the engine fewer queues, and at most three pieces whatever E from 5 600 rows ("large_step_rows"; round 4: 3 584) per
 * step (400+ streams x R = 13: more lanes are more GEMM working sets in the same L2s; tokens then come back sooner; engine option "large_step_pieces",
 * 0 = no such limit).  Throughput option for callers that push back to back
 * (a server draining a backlog, a file); a live stream keeps the default.
 * "pipeline" = 8 (round 3, experimental): for calls of one or two rows per step (one stream x R = 0 or 1, two streams x R = 0) the 8 steps in flight
 * sit at 8 stages of 3 layers and every launch of the two chains carries the same kernel of FOUR steps (grouped launches); tokens 9 calls later;
 * other shapes fall back to four lanes.  Bit-identical as well; measured +2.7 % at batch 1 (profiles/r3_grouped_pipeline.md), so 4 stays the default.
 * "lanes" (1..4): keep at most this many encoder lanes and give the other lanes' streams back -- for a process with another GPU
 * client (the diarization side-car): a stream created after this call gets a hardware queue the engine no longer uses.  Not
 * reversible for the engine's lifetime.
 * "graph_cache" (>= 1, default 16): step shapes (streams in the call, lookahead, chunks per push, pieces) whose hipGraphs are kept;
 * beyond it the least recently used shape is dropped and re-captured when it comes back -- a server whose batch size changes from
 * call to call (reference: one stream per call, src/nemo-server.cpp:192-271) holds a bounded number of graph execs.
 * Kernel-selection switches for A/B runs and bit-identity tests (round 4: these were environment variables read inside the product path;
 * set them before the first step, results never depend on them): "gemm_cores" (-1 = the engine's rule, 0 / 1 = never / always the
 * large-M GEMM kernels of which two share a CU), "persistent_gemm" (1: GEMMs with several 128 x 128 tiles per CU on the persistent tile loop; default 0),
 * "wide_tiles" (default 1: 256- / 224-row tiles from 1 792 rows where their rounds fill the chip, in pipelined steps from 1 344 rows and 32 tiles ("wide_min_rows" / "wide_min_tiles"; rounds 4: 96);
 * 256: the 256-row form only; 3: without the pipelined steps' tile-count rule; 0: off), 
 * "t64_tiles" (default 64, per engine: the split-K GEMMs with N = 1024 take 128 x 64 tiles up to this many 128 x 128 tiles; round 3: 127),
 * "tile_bands" (-1 = the rule: above 4 row chunks the tiles of a launch are handed to the XCDs in bands of column groups, so that the panels an XCD reads stay in its L2;
 * 0 / 1 = never / always), "f32_mfma" (0: f32 GEMMs above
 * four rows on the FMA tile kernel instead of the f32 MFMA), "decode_graph_iterations" (>= 1, default 12: decode iterations a
 * pipelined step's decode graph carries before the eager fallback), "resid_epilogue" (0: every residual GEMM writes split-K partial
 * slabs and k_post adds them, as in rounds 1-4; default 1: the GEMM adds to the residual stream in its epilogue where one workgroup owns a tile's
 * whole K sum -- same bits), "gemm_prio" (default 0: round 5's GEMM loops -- k_gemm_wide2 / k_gemm_tiled3: the next chunk's fragments read under this chunk's
 * MFMAs, wave-private epilogues; 20: rounds 1-4's loops; same bits), "epilogue16" (default 1: the GEMM epilogues with 16-bit outputs store eight columns = 16 bytes
 * per thread; 0: four), "wide_min_tiles" / "wide_min_rows" (defaults 32 / 1 344: pipelined steps take the 224 x 256 tiles from this many tiles and rows), "split_tasks" (default 200: residual GEMMs take
 * one K slice from this many 128 x 128 tiles), "ablate" (MEASUREMENT ONLY -- results
 * are invalid: bit mask of launches left out of a step: 1 residual + LayerNorm, 2 attention, 4 depthwise conv, 8 decode iterations, 16 front end, 32 encoder
 * GEMMs; what each costs a pipelined step: profiles/r5_ablation.md), "decode_lane" (0: the decode graphs run behind the last encoder
 * piece instead of on a stream of their own; read when the lanes are picked, so it is REJECTED after the first pipelined step or
 * nasr_engine_lend_stream). */
int nasr_engine_set_option(nasr_engine *e, const char *key, int value);
/* diagnostics: "graph_execs" (hipGraphExec objects alive), "graph_shapes" (distinct cached step shapes), "graph_evictions",
 * "graph_replays" (calls served by a hipGraph, pipelined ones included), "eager_steps" (calls that were not graph-eligible: ragged
 * pushes, streams that complete different chunk counts), "pipelined_steps", "grouped_steps", "lanes" (HIP streams the engine found
 * to overlap; 0 before the first pipelined step).  Returns 0, or -1 for an unknown name.  Like every entry point that takes an
 * engine, call it from the thread that steps that engine: it reads the graph caches without a lock. */
int nasr_engine_get_counter(const nasr_engine *e, const char *name, int64_t *value);
/* enable recording of NASR_TAP_MEL / SUBSAMPLED / LAYER_OUT (costs extra copies) */
int nasr_engine_set_debug(nasr_engine *e, int enable);
/* returns the number of floats written (<= cap) or <0.  NASR_TAP_K_CACHE / V_CACHE return the LOGICAL cache: rows that are not
 * cached yet (the first 70 - cache_valid_len) read as the zeros the reference's tensors start with (src/nemo-stream.cpp:320-325). */
int64_t nasr_stream_get_tap(nasr_stream *s, int which, int index, float *out, int64_t cap);
/* test hook: overwrites every row of the stream's K/V rings (all layers) with +-value.  A stream start / reset does not clear the
 * rings -- rows behind cache_valid_len are masked with -1e9 and weigh exactly 0, as in the reference's own reset
 * (src/nemo-stream.cpp:95-115, :1037-1043) -- and this is how the tests prove that stale rows never reach a result. */
int nasr_stream_debug_fill_kv(nasr_stream *s, float value);

/* ---- measurement: per-kernel-class HIP-event timing on the engine's own
 * stream.  Replaces the std::chrono timers of src/nemo-stream.h:236-244. ---------------- */
int nasr_engine_profile(nasr_engine *e, int enable); /* enable also resets the counters */
int nasr_engine_profile_read(nasr_engine *e, nasr_kernel_stat *out, int cap); /* returns count */
/* raw hipStream_t of the engine (for external event timing) */
void *nasr_engine_hip_stream(nasr_engine *e);
/* Hands the last of the engine's side-by-side HIP streams -- and with it a hardware queue that no encoder lane will use -- to
 * another GPU client of the process (nasr_diar_set_stream).  The engine runs one encoder piece fewer at most and still owns the
 * stream.  Borrowers inside this library are counted: destroy them first.  If the engine goes first, nasr_engine_destroy says so
 * on stderr and in nasr_last_error(), and the stream stays alive until its last borrower lets go (nothing dangles in either
 * order).  *out receives a hipStream_t. */
int nasr_engine_lend_stream(nasr_engine *e, void **out);
/* device malloc/free/copy helpers so a host written without HIP can keep PCM resident */
int nasr_device_alloc(nasr_engine *e, void **out, int64_t bytes);
int nasr_device_free(nasr_engine *e, void *p);
int nasr_device_upload(nasr_engine *e, void *dst_device, const void *src_host, int64_t bytes);
int nasr_engine_synchronize(nasr_engine *e);

/* ---- diarization side-car (BASELINE config 5): MarbleNet VAD + TitaNet-L speaker embeddings ----------------------
 * Replaces the compute of vad_session / spk_session (src/diarize_vad.h:95-135, src/diarize_spk.h:95-120).  weights =
 * the tensors of diarize.gguf ("vad.*" and/or "spk.*", F32, layouts of scripts/convert_diarize_to_gguf.py:129-158),
 * same descriptor type as nasr_engine_create.  The onset/offset state machine, sub-segment cursor, NME-SC clustering and
 * RTTM output of src/diarize_pipeline.cpp / src/diarize_cluster.cpp are host control flow ABOVE this ABI:
 * nemotron-asr.cpp_amd/host/diarize_pipeline_amd.h, diarize_cluster_amd.h. */
typedef struct nasr_diar nasr_diar;
/* dtype: NASR_DTYPE_BF16 = TitaNet's pointwise convolutions on the bf16 MFMA (f32 accumulate), NASR_DTYPE_F32 = all f32;
 * MarbleNet is f32 (P(speech) within 2e-5 of the reference arithmetic) unless NASR_DIAR_VAD_BF16 is OR'ed in: then its pointwise
 * convolutions run on the bf16 MFMA with bf16 activation planes (3 workgroups per CU instead of 1; P(speech) within a few 1e-3).
 * max_windows / max_segments size the scratch (larger calls are tiled). */
#define NASR_DIAR_VAD_BF16 0x100
/* the same kernel with IEEE-half planes and weights on the f16 MFMA (same rate, 11 significand bits instead of 8: an eighth of the
 * bf16 planes' rounding -- logit error 0.006 against 0.04, P(speech) within 1e-4 of the f32 kernel on the synthetic network;
 * segment-level comparison in profiles/r4_vad_16bit_segments.md, tests/test_gpu_diar.py); values saturate at 65 504 */
#define NASR_DIAR_VAD_F16  0x200
int  nasr_diar_create(nasr_diar **out, int device_id, int dtype, const nasr_weight_desc *weights, int n_weights,
                      int max_windows, int max_segments);
void nasr_diar_destroy(nasr_diar *d);
/* the side-car's work goes onto a stream the caller owns (hipStream_t, e.g. from nasr_engine_lend_stream) until nasr_diar_destroy */
int  nasr_diar_set_stream(nasr_diar *d, void *hip_stream);
/* vad_session_run_batch (src/diarize_vad.cpp:490-503) for B buffers in one launch sequence: P(speech) of every 0.63 s
 * window (10 080 samples) of audio[b] at a 10 ms shift (the reference runs each window as its own graph,
 * src/diarize_pipeline.cpp:204-211).  audio: float samples in [-1, 1], or s16 PCM cast to the pointer type with
 * NASR_FLAG_AUDIO_S16; host memory, or device memory with NASR_FLAG_PCM_DEVICE;
 * n_windows[b] = 1 + (n_samples[b] - 10080) / 160, or 0. */
int  nasr_diar_vad(nasr_diar *d, int B, const float *const *audio, const int32_t *n_samples, float *const *probs_out,
                   const int32_t *probs_cap, int32_t *n_windows, uint32_t flags);
/* spk_session_run_chunk (src/diarize_spk.cpp:601-626) for S sub-segments at once: audio[s] holds 24 000 samples (zero
 * padded by the caller), lens_samples[s] of them real; emb_out = [S][192]. */
int  nasr_diar_embed(nasr_diar *d, int S, const float *const *audio, const int32_t *lens_samples, float *emb_out, uint32_t flags);

/* measurement: device time (ms) of the LAST nasr_diar_vad (which = 0) / nasr_diar_embed (which = 1) call's launch sequence, from HIP events on the
 * side-car's own stream around its kernels (staging copies and the read-back excluded) -- what bench.py prices configs[4]'s roofline with.  The
 * reference has no counterpart (it times whole sessions on the host, src/diarize_pipeline.cpp). */
int  nasr_diar_last_gpu_ms(nasr_diar *d, int which, float *ms_out);

/* parity tap: diarize_compute_logmel (src/diarize_audio.cpp:136-227) of one whole host buffer on the device front end,
 * which = 0: the 'vad.*' filterbank, 1: 'spk.*'.  mel_out = [80][t_padded] row-major like the reference's output
 * (t_padded = t_valid rounded up to 16, t_valid = n_samples / 160); this is what tests/test_diarize_preproc.cpp checks
 * against the NeMo fixture (threshold 1e-3). */
int  nasr_diar_logmel(nasr_diar *d, int which, const float *audio, int32_t n_samples, int per_feature_normalize,
                      float *mel_out, int64_t cap, int32_t *t_valid_out);

#ifdef __cplusplus
}
#endif
#endif /* NEMOTRON_ASR_AMD_H */

// nasr_engine_priv.h -- what the translation units of the engine share (round 4: nasr_engine.hip was 2 900 lines holding pools, step
// driver, pipeline, graph cache and ABI).  nasr_engine.hip = weights, pools, engine / stream life cycle; nasr_encoder.hip = the
// chunk step (encoder + decode launch sequences); nasr_pipeline.hip = hipGraph steps, lanes, pipelined and grouped steps;
// nasr_abi.hip = the step driver and the remaining entry points of include/nemotron_asr_amd.h.  Internal functions live in
// namespace nasr_eng (nothing but the extern "C" ABI is meant to be bound from outside).
#pragma once
#include "nasr_internal.h"
#include "nemotron_asr_amd.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

using namespace nasr;

namespace nasr_eng {
int fail(const char *fmt, ...);                 // fills nasr_last_error() of the calling thread, returns -1
}
using namespace nasr_eng;
#define HIPCHK(x)                                                                           \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Several engines may live in one process, one host thread each (the socket server: one lane per GPU).  HIP stream
// capture, even in thread-local mode, is broken by what other threads do meanwhile ("operation failed due to a
// previous error during capture" when another thread copies or allocates).  Every entry point that talks to HIP
// therefore holds this lock shared; building a step graph (rare: once per (B, T, G)) takes it exclusively.
void api_lock_shared();                          // also used by nasr_diar.hip
void api_unlock_shared();
namespace nasr_eng {
void api_capture_begin();                        // shared -> exclusive (the caller is inside an ApiGuard)
void api_capture_end();
struct ApiGuard {
    ApiGuard() { api_lock_shared(); }
    ~ApiGuard() { api_unlock_shared(); }
};
struct CaptureExclusive {       // held by a thread that is inside an ApiGuard
    CaptureExclusive() { api_capture_begin(); }
    ~CaptureExclusive() { api_capture_end(); }
};
#ifdef NASR_STAMPS
extern unsigned long long *g_stamp_buf;
extern int g_stamp_pipe;
constexpr int STAMP_PER_SLOT = 8 * 24, STAMP_SLOTS = 5;
#endif
}  // namespace nasr_eng

// ---------------------------------------------------------------------------------------
struct nasr_engine;
namespace nasr_eng {
void prof_flush(nasr_engine *e);
}
struct LayerW {
    float *ln_ff1_w, *ln_ff1_b, *ln_att_w, *ln_att_b, *ln_conv_w, *ln_conv_b, *ln_ff2_w, *ln_ff2_b, *ln_out_w, *ln_out_b;
    void *ff1_w1, *ff1_w2, *wqkv, *wo, *pw1, *pw2, *ff2_w1, *ff2_w2;   // packed bf16 or f32 [N][K]
    float *wpos_f32;                                                    // [1024][1024] f32 (load-time pos projection)
    float *bias_u, *bias_v, *dw, *cln_w, *cln_b;
    void *posproj[TMAX + 1];                                            // per T: [70+2T-1][1024] act dtype
};

struct Prof {
    struct Rec { int cat; hipEvent_t a, b; double bytes, flops; };
    bool on = false;
    std::vector<std::string> names;
    std::vector<nasr_kernel_stat> stats;
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    int cat(const char *n) {
        for (size_t i = 0; i < names.size(); i++) if (names[i] == n) return (int)i;
        names.push_back(n);
        nasr_kernel_stat s;
        memset(&s, 0, sizeof(s));
        snprintf(s.name, sizeof(s.name), "%s", n);
        stats.push_back(s);
        return (int)names.size() - 1;
    }
};

struct nasr_stream {
    nasr_engine *e;
    int slot, R, T, prompt;
    // host mirror of the stream manager state (reference nemo_stream_context)
    int abuf_cnt, abuf_par;          // samples waiting in the audio buffer (pre-seeded 256 zeros)
    int mel_start, mel_count;        // mel ring window
    int valid_len, kv_head, cc_par;  // cache_valid_len, K/V ring head, conv-cache parity
    int chunks, tok_read;
    int64_t samples_in;
    int last_T, last_row, last_ws;   // rows of the last chunk and the workspace set they are in (for taps)
    bool alive;
    std::vector<int32_t> tok_queue;  // tokens gathered from the device, not yet handed to the caller
};

struct nasr_engine {
    int device = 0, dtype = 0, max_streams = 0;
    nasr_hparams hp;
    bool bf16 = false;
    int esz = 4;
    hipStream_t st = nullptr;
    // front-end constants
    float *window = nullptr, *fbT = nullptr, *cos_t = nullptr, *sin_t = nullptr;
    int *fb_band = nullptr;
    float *w0t, *b0, *w2t, *b2, *b3, *w5t, *b5, *b6, *sub_out_b;
    void *w3, *w6, *sub_out_w;
    std::vector<LayerW> L;
    float *embed, *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2], *jenc_w, *jenc_b, *pred_w, *pred_b, *out_w, *out_b;
    float *pk1a = nullptr, *pk1p = nullptr, *pk1_b = nullptr, *pk2_w = nullptr, *pk2_b = nullptr;
    // state pools
    float *abuf, *last_sample, *mel_ring;
    std::vector<void *> kv_pool;     // per layer [slot][2][KVC][1024] act dtype
    std::vector<float *> cc_pool;    // per layer [slot][2][ks-1][1024]
    float **cc_ptrs_dev = nullptr;   // device copy of cc_pool (k_stream_reset walks the layers in one launch)
    void **kv_ptrs_dev = nullptr;    // device copy of kv_pool (same launch: the 70 window rows of a starting stream are zeroed)
    float *dec_h, *dec_c;
    DecCtrl *ctrl;
    int *tok_ring;
    // workspace (sized for max_streams x TMAX rows)
    float *x, *x2, *part, *q, *glu, *encproj, *sub_a, *hfuse;
    bool opt_fused = true, opt_graph = true;
    int opt_graph_cache = 16;        // option "graph_cache": step shapes (B, T, G, E) whose hipGraphs are kept, per slot; least recently used goes first
    std::map<int64_t, int64_t> graph_used;   // shape key -> tick of its last use
    int64_t graph_tick = 0, graph_evictions = 0;
    // hipGraph replay of the steady-state step: fixed descriptor buffers + one exec per (B, T)
    std::map<int64_t, hipGraphExec_t> graphs;
    int w_rows = 0;                  // workspace rows = max(max_streams x TMAX, MAXNEW)
    char *g_desc = nullptr;          // device mirror of the packed descriptor block (layout: graph_desc_layout)
    bool opt_multichunk = true;
    int opt_decode_graph_iters = 12;   // blind decode iterations a pipelined step's decode graph carries at most (option "decode_graph_iterations")
    bool opt_decode_lane = true;       // the decode graphs get a lane of their own when a queue is free (option "decode_lane")
    int opt_gemm_cores = -1;           // -1: the engine's rule; 0 / 1: never / always the GEMM kernels of which two share a CU (option "gemm_cores")
    int opt_large_step_rows = 0;     // option "large_step_rows": the row count from which "large_step_pieces" applies (0 = the default of 5 600; round 4: 3 584)
    int opt_large_step_pieces = 3;   // option "large_step_pieces": most pieces of a pipelined step of 5 600 rows and more (0 = as many as "pipeline" says)
    int opt_t64_tiles = 64;          // option "t64_tiles": the split-K GEMMs with N = 1024 take 128 x 64 tiles up to this many 128 x 128 tiles (per engine, carried in GemmParams)
    int opt_tile_bands = -1;         // option "tile_bands": -1 = the rule (bands of column groups above 4 row chunks), 0 = never, 1 = always (same bits)
    int opt_wide_tiles = 1;          // option "wide_tiles": 256- / 224-row GEMM tiles from 1 792 rows where they fill the chip (k_gemm_wide; same bits); 256 = the 256-row form only, 0 = off
    bool opt_persist_gemm = false;   // option "persistent_gemm" = 1: GEMMs with >= 1.75 tiles of 128 x 128 per CU on the persistent tile loop (k_gemm_persist; same bits).  Off by default: alone on the chip with cache-resident operands it is 14-37 % faster at 7 168 rows, inside the engine (weights cold from HBM) 1 % -- profiles/r4_persistent_gemm.md
    int opt_wide_min_tiles = 0;      // option "wide_min_tiles": pipelined steps take the 224 x 256 tiles from this many tiles (0 = the default of 32)
    int opt_wide_min_rows = 0;       // option "wide_min_rows": ... and from this many rows (0 = the default of 1 344)
    int opt_gemm_prio = 0;           // option "gemm_prio": GemmParams::prio (probes of wave priority in the GEMM loops)
    bool opt_epilogue16 = true;      // option "epilogue16": SiLU rows, K / V ring rows and GLU pairs leave the GEMM epilogues as 16-byte stores (eight columns per thread; same values)
    bool opt_dwconv_stream = true;   // option "dwconv_stream": the depthwise conv with one workgroup per stream from 256 streams x 4 frames (k_dwconv_stream; same bits)
    int opt_chain = 0;               // option "chain" (OFF: measured 23 % SLOWER, profiles/r5_configs2_launch_structure.md section 3): 1 = in pipelined steps of 769 .. 1 343 rows the GEMM that reads a k_post's
                                     // rows carries that k_post as its head phase (GemmParams::chain: one launch fewer per LayerNorm, same bits); 2 = in synchronous steps too
    int opt_split_tasks = 0;         // option "split_tasks": residual GEMMs take ONE K slice from this many 128 x 128 output tiles (0 = the rule, 200; A/B runs)
    int opt_resid_epilogue = 1;       // option "resid_epilogue": residual GEMMs add to the residual stream in their own epilogue where one workgroup owns a tile's whole K sum
                                     // (k_gemm_t64w / launches without split-K): 1 = in pipelined steps (default), 2 = always, 0 = partial slabs + k_post everywhere (rounds 1-4); same bits
    int opt_ablate = 0;              // option "ablate" (MEASUREMENT ONLY, results are invalid): bit mask of launches left out of the unfused step -- 1 k_post, 2 attention,
                                     // 4 depthwise conv, 8 decode iterations, 16 front end (mel + subsampling), 32 every encoder GEMM: what each costs a pipelined step (profiles/r5_configs2_launch_structure.md, r5_ablation_b64_R13.json)
    bool opt_f32_mfma = true;        // f32 GEMMs above four rows on v_mfma_f32_32x32x2_f32 (bit-identical to the FMA tile kernel)
    char *gh = nullptr;                                                               // pinned host block
    int *gh_collect = nullptr;       // pinned landing zone of the token gather: [B][1 + COLLECT_STRIDE] + n_active
    int64_t graph_replays = 0, eager_steps = 0, decode_fallbacks = 0, decode_fallback_rounds = 0;
    // pipelined graph steps (option "pipeline" = E, 1..4): launch sequences of CONSECUTIVE steps run beside each other on
    // their own HIP streams -- see the comment at pipe_step().  Everything a step in flight owns exists once per slot:
    // workspace set, descriptor blocks, joint.enc buffer, token landing zone, graphs (their kernel arguments point into
    // the slot).  E + 1 steps are in flight; slot of a step = its sequence number mod NSLOT.
    static const int MAXSEG = 4, LSLOT = MAXSEG + 1;     // lanes mode: E + 1 steps in flight, slot = sequence number mod LSLOT
    int pipe_last_nseg = 0;          // pieces of the youngest step in flight (pipe_step drains before a step that is cut differently)
    static const int GP_C = 2, GP_Y = FUSED_GROUP, GP_S = GP_C * GP_Y;   // grouped mode ("pipeline" = 8): 2 chains x 4 problems per launch = 8 stages
    static const int NSLOT = GP_S + 3;                    // grouped mode: 8 steps in flight + the one being decoded + the one being collected + one spare
    struct WS { float *x, *x2, *part, *q, *glu, *sub_a, *hfuse; void *a, *hbuf, *ctx, *cbuf, *sub_b; unsigned *chain_flags; };
    WS ws[NSLOT];                    // ws[0] = the set the synchronous paths use (mirrored in x, x2, ... below)
    int opt_pipeline = 0;            // 0: synchronous steps; E >= 1: the encoder in E pieces + the decode, each piece one step behind the previous
    hipStream_t lane[MAXSEG] = {nullptr, nullptr, nullptr, nullptr};   // lane[k]: encoder piece k (lane[0] = st)
    int n_lanes = 1;                 // streams that run side by side (lane[0 .. n_lanes - 1], each on a hardware queue of its own): a step has at
                                     // most that many encoder pieces; its decode graph runs on the LAST of these streams -- a queue of its own while
                                     // the step has fewer pieces than there are streams, else right behind the last piece on that piece's lane
    std::vector<hipStream_t> lent;   // streams handed to another client (nasr_engine_lend_stream): still owned, destroyed with the engine
    int max_lanes = MAXSEG;          // option "lanes": the engine keeps at most this many (the others' hardware queues are left to other clients of the process)
    struct Pipe {
        bool ready = false;                               // buffers of this slot allocated
        char *g_desc = nullptr, *gh = nullptr;            // descriptor block of the encoder graphs (device / pinned)
        int *gh_collect = nullptr, *collect_dev = nullptr;
        int *g_dmeta = nullptr, *gh_dmeta = nullptr;      // k_collect meta of the decode graph [2 B] (device / pinned)
        float *encproj = nullptr;                         // [w_rows][640]: encoder graph -> decode graph
        hipEvent_t seg_done[MAXSEG] = {nullptr, nullptr, nullptr, nullptr};
        hipEvent_t dec_done = nullptr;
        bool dec_launched = false;
        std::map<int64_t, hipGraphExec_t> seg_graphs[MAXSEG], dec_graphs;     // key = (B, T, G, E)
        int stage = 0;                                    // encoder pieces launched so far (0 = slot free)
        int64_t seq = -1;                                 // sequence number of the step that occupies the slot
        std::vector<nasr_stream *> streams;
        int T = 0, G = 0, nseg = 0;
        int64_t key = 0;
    } pipe[NSLOT];
    // grouped pipeline ("pipeline" = 8, one or two rows per step on the fused path): the 8 steps in flight are at 8 stages of 3 layers;
    // chain c (HIP stream lane[c]) runs stages 4c .. 4c+3, each of its 24 launches per call carrying the same layer kind of FOUR steps
    struct GpEntry { int slot; int done; };            // a step in flight: its slot, the stages it has completed
    std::vector<GpEntry> gp_flight;                    // oldest first
    int64_t gp_calls = 0;
    int gp_next_slot = 0, gp_dec_pending = -1;          // slot whose decode graph is in flight (collected in the next call)
    hipEvent_t gp_ev[GP_C][2] = {{nullptr, nullptr}, {nullptr, nullptr}};       // chain c has finished the call of that parity
    bool gp_ev_set[GP_C][2] = {{false, false}, {false, false}};
    std::map<int64_t, hipGraphExec_t> gp_graphs[NSLOT][GP_C];                   // steady-state graphs by slot of the newest step
    int64_t gp_steps = 0, gp_graph_chains = 0, gp_eager_chains = 0;
    double host_launch_s = 0, host_wait_s = 0;     // NASR_STATS: host time inside hipGraphLaunch / waiting for the device (pipelined steps)
    int64_t pipe_seq = 0;            // steps launched through the pipeline so far
    bool pipe_ready = false;
    bool gemm_coresident = false;  // set while the graphs of a step with >= 2 launch chains are captured (run_gemm)
    size_t desc_bytes = 0, col_bytes = 0;
    int64_t pipe_steps = 0;
    void *a, *hbuf, *ctx, *cbuf, *sub_b;             // (with x, x2, part, q, glu, sub_a, hfuse: the CURRENT workspace set, see use_ws)
    unsigned *chain_flags = nullptr;                 // ... and its row-chunk counters of chained launches (a set's launches never overlap: one array per set)
    float *predg;                    // [slot][640] cached joint.pred output of the LSTM candidate
    unsigned long long *key;
    int *n_active;                   // [3] = n_active, n_dirty, n_rows
    int *dlist; unsigned *rowmap; int *tok_frame;
    int *collect_dev;                // [B][1+COLLECT_STRIDE]
    // descriptor staging
    char *pin = nullptr; size_t pin_cap = 0, pin_off = 0;
    char *ddesc = nullptr; size_t ddesc_cap = 0, ddesc_off = 0;
    int16_t *pcm_stage = nullptr; size_t pcm_stage_cap = 0;
    // host PCM hand-over: the streams' buffers are gathered into a pinned block and cross PCIe as ONE copy.  The copy is
    // asynchronous and a pipelined call returns before it has run, so the pinned blocks rotate (a block is reused four
    // calls later; up to pipeline + 1 = 5 steps are in flight, but the copy sits in piece 0 of its step, and the NEXT call launches piece 1 of that step only after the host has seen piece 0 complete).
    struct { int16_t *p = nullptr; size_t cap = 0; hipEvent_t copied = nullptr; bool pending = false; } pcm_pin[4];   // copied: recorded behind the block's H2D copy
    unsigned pcm_pin_next = 0;
    float *mel_stage = nullptr; size_t mel_stage_cap = 0;
    // debug taps
    bool debug = false;
    float *tap_mel = nullptr; int tap_mel_cap = 0;      // [max_streams][tap_mel_cap][128] by batch row
    float *tap_sub = nullptr, *tap_layers = nullptr, *tap_enc = nullptr;  // [slot][...]
    std::vector<int> tap_mel_frames;                     // per slot: frames captured in last call
    std::vector<int> tap_mel_row;
    std::vector<nasr_stream *> slots;
    std::vector<void *> allocs;
    Prof prof;
};

constexpr int COLLECT_STRIDE = 256;


struct ProfScope {
    nasr_engine *e; int cat = -1; hipEvent_t a = nullptr; double bytes, flops;
    ProfScope(nasr_engine *e_, const char *name, double bytes_ = 0, double flops_ = 0) : e(e_), bytes(bytes_), flops(flops_) {
        if (!e->prof.on) return;
        Prof &pf = e->prof;
        if (pf.pending.size() >= 4096) prof_flush(e);
        cat = pf.cat(name);
        hipEvent_t b;
        if (pf.pool.size() >= 2) { a = pf.pool.back(); pf.pool.pop_back(); b = pf.pool.back(); pf.pool.pop_back(); }
        else { hipEventCreate(&a); hipEventCreate(&b); }
        evb = b;
        hipEventRecord(a, e->st);
    }
    ~ProfScope() {
        if (cat < 0) return;
        hipEventRecord(evb, e->st);
        e->prof.pending.push_back({cat, a, evb, bytes, flops});
    }
    hipEvent_t evb = nullptr;
};

namespace nasr_eng {
struct HostTimer {          // accumulates wall time of a scope into a double (diagnostics only: NASR_STATS)
    double &acc; std::chrono::steady_clock::time_point t0;
    explicit HostTimer(double &a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
struct GraphDescLayout { size_t rows, pcm, meta, vrows, total; };
float f16_to_f32(uint16_t h);
int64_t desc_numel(const nasr_weight_desc &d);
int to_f32(const nasr_weight_desc &d, std::vector<float> &out);
void transpose_9x256(const std::vector<float> &w /*[256][9]*/, std::vector<float> &t /*[9][256]*/);
void pack_f32_mfma(const std::vector<float> &w, int N, int K, bool lstm_order, std::vector<float> &out);
void host_pos_emb(int position, float *out);
int load_weights(nasr_engine *e, const nasr_weight_desc *w, int n_w);
int ensure_posproj(nasr_engine *e, int T);
int alloc_ws(nasr_engine *e, nasr_engine::WS &w);
void use_ws(nasr_engine *e, const nasr_engine::WS &w);
void engine_destroy_impl(nasr_engine *e);
int stream_zero_state(nasr_stream *s, bool keep_reference_state = false);
double gemm_bytes(const nasr_engine *e, int M, int N, int K, int wesz);
int run_gemm(nasr_engine *e, GemmParams &g, bool f32_weights, const char *tag);
int pick_splits(const nasr_engine *e, int M, int N, int K);
int run_layers_fused(nasr_engine *e, const RowDesc *rows, int B, int T, int G, int k0, int k1, std::vector<FusedParams> *rec = nullptr);
int enqueue_encoder(nasr_engine *e, const RowDesc *rows, const RowDesc *vrows, const int *tap_slots, int B, int T, int R, int G = 1, int seg = 0,
                           int nseg = 1, int part = 0);
void make_dec_params(nasr_engine *e, const RowDesc *rows, int B, int T, DecParams &dp);
void enqueue_decode_iters(nasr_engine *e, const DecParams &dp, int B, int n, int &it, hipStream_t st = nullptr);
void chunk_bookkeeping(nasr_stream *s, int row);
void fill_row_desc(RowDesc &rd, const nasr_stream *s, int n_dec);
int run_chunk(nasr_engine *e, const std::vector<nasr_stream *> &rows_s, const std::vector<int> &n_dec);
int drain_chunks(nasr_engine *e, nasr_stream *const *streams, int B);
int validate_batch(nasr_engine *e, nasr_stream *const *streams, int B);
int consume_collect(nasr_engine *e, const int *host, nasr_stream *const *streams, int B);
void deliver(nasr_stream *const *streams, int B, int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens);
int collect_tokens(nasr_engine *e, nasr_stream *const *streams, int B, int32_t *const *tokens_out,
                          const int32_t *tokens_cap, int32_t *n_tokens);
int ensure_debug_buffers(nasr_engine *e);
GraphDescLayout graph_desc_layout(int B, int G);
int build_step_graph(nasr_engine *e, int B, int T, int R, int G, hipGraphExec_t *out);
int pipe_blind_iterations(int frames, int cap);
double spin_us(hipStream_t a, hipStream_t b);
int pick_lanes(nasr_engine *e);
void release_lanes(nasr_engine *e);
int ensure_pipe(nasr_engine *e, int p);
int build_pipe_graphs(nasr_engine *e, int p, int B, int T, int R, int G, int nseg, hipGraphExec_t *seg_out, hipGraphExec_t *dec_out);
int pipe_advance(nasr_engine *e, int p);
int pipe_finish_launch(nasr_engine *e, int p);
int pipe_finish(nasr_engine *e, int p);
int pipe_drain(nasr_engine *e);
int pipe_step(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *pcm_dev, const int32_t *n_samples, int G,
                     int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens);
bool gp_eligible(const nasr_engine *e, int B, int T, int G);
int gp_enqueue_chain(nasr_engine *e, int c, const int *slot_of_stage, int B, int T, int R, int G);
int gp_capture(nasr_engine *e, hipGraphExec_t *out, const std::function<int()> &body, hipStream_t st);
int gp_decode_graph(nasr_engine *e, int p, int B, int T, int G, hipGraphExec_t *out);
int gp_call(nasr_engine *e, int new_slot, int64_t key, int B, int T, int R, int G);
int gp_finish_decode(nasr_engine *e);
int gp_drain(nasr_engine *e);
int gp_step(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *pcm_dev, const int32_t *n_samples, int G,
                   int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens);
int try_graph_step(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *pcm_dev,
                          const int32_t *n_samples, int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens);
int push_piece(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *base, const int32_t *n_samples,
                      int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens, uint32_t flags);
void prof_flush(nasr_engine *e);
__global__ void k_collect(const int *slots, const int *tok_read, int B, const DecCtrl *ctrl, const int *tok_ring, int *out, int stride, const int *n_active);
inline int max_frames_per_push(int TS) { return 8 * TS + 16; }  // TS = frames of encoder output the push completes (+ what a first push leaves over)
inline bool streams_overlap(hipStream_t a, hipStream_t b, double alone_us) { return spin_us(a, b) < 1.5 * alone_us; }
inline bool dec_behind_last_piece(const nasr_engine *e, const nasr_engine::Pipe &P) { return P.nseg >= e->n_lanes; }
inline hipStream_t dec_stream(nasr_engine *e, const nasr_engine::Pipe &P) { return e->lane[e->n_lanes - 1]; }
template <typename Tp>
inline int stage_desc(nasr_engine *e, const std::vector<Tp> &host, const Tp **dev_out) {
    const size_t bytes = (host.size() * sizeof(Tp) + 255) & ~(size_t)255;
    if (bytes > e->pin_cap / 2) return fail("descriptor too large");
    if (e->pin_off + bytes > e->pin_cap || e->ddesc_off + bytes > e->ddesc_cap) {
        HIPCHK(hipStreamSynchronize(e->st));   // all earlier copies/kernels done: arenas reusable
        e->pin_off = 256;   // first 256 bytes: host landing zone of the decode 'n_active' read-back
        e->ddesc_off = 0;
    }
    char *hp = e->pin + e->pin_off, *dp = e->ddesc + e->ddesc_off;
    memcpy(hp, host.data(), host.size() * sizeof(Tp));
    HIPCHK(hipMemcpyAsync(dp, hp, host.size() * sizeof(Tp), hipMemcpyHostToDevice, e->st));
    e->pin_off += bytes;
    e->ddesc_off += bytes;
    *dev_out = (const Tp *)dp;
    return 0;
}
template <typename Tp>
inline int dalloc(nasr_engine *e, Tp **out, size_t n_elems) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(n_elems * sizeof(Tp), 16);
    HIPCHK(hipMalloc(&p, bytes));
    e->allocs.push_back(p);
    *out = (Tp *)p;
    return 0;
}
}  // namespace nasr_eng

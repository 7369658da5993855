// nasr_engine.hip -- host side of the MI355X engine: the C ABI of include/nemotron_asr_amd.h.
//
// Owns device weights (re-laid-out at upload), the per-stream state pool (K/V rings, conv
// caches, LSTM state, audio/mel rings) and the per-step launch sequence.  The chunk/shift
// arithmetic of the reference's stream manager (src/nemo-stream.cpp:1145-1293,
// src/nemo-stream.h:65-100) is mirrored on the host: every count it needs is a pure
// function of the number of samples pushed, so no device read-back is needed to schedule.
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
#include <shared_mutex>
#include <string>
#include <vector>

using namespace nasr;

static thread_local char g_err[512] = "";
static int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
namespace nasr {
int set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return -1; }
}
#define HIPCHK(x)                                                                           \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *nasr_last_error(void) { return g_err; }

// Several engines may live in one process, one host thread each (the socket server: one lane per GPU).  HIP stream
// capture, even in thread-local mode, is broken by what other threads do meanwhile ("operation failed due to a
// previous error during capture" when another thread copies or allocates).  Every entry point that talks to HIP
// therefore holds this lock shared; building a step graph (rare: once per (B, T, G)) takes it exclusively.
#ifdef NASR_STAMPS
// diagnostic build (make stamps; never shipped): every fused layer kernel stamps the 100 MHz real-time counter at 8 points in
// its first and last workgroup.  One region of 8 x 24 launches per pipeline slot: after a run the buffer holds the last replay
// of every slot's graphs = the time line of the last NSLOT steps on their lanes (tests/micro/stamps_timeline.py reads the dump).
static unsigned long long *g_stamp_buf = nullptr;
static int g_stamp_pipe = 0;
static const int STAMP_PER_SLOT = 8 * 24, STAMP_SLOTS = 5;
#endif
static std::shared_mutex g_api_mu;
struct ApiGuard {
    ApiGuard() { g_api_mu.lock_shared(); }
    ~ApiGuard() { g_api_mu.unlock_shared(); }
};
void api_lock_shared() { g_api_mu.lock_shared(); }       // for the other translation units (nasr_diar.hip)
void api_unlock_shared() { g_api_mu.unlock_shared(); }
struct CaptureExclusive {       // held by a thread that is inside an ApiGuard
    CaptureExclusive() { g_api_mu.unlock_shared(); g_api_mu.lock(); }
    ~CaptureExclusive() { g_api_mu.unlock(); g_api_mu.lock_shared(); }
};
extern "C" int nasr_abi_version(void) { return NASR_ABI_VERSION; }

// ---------------------------------------------------------------------------------------
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
    bool opt_persist_gemm = true;    // GEMMs with >= 1.75 tiles of 128 x 128 per CU on the persistent tile loop (k_gemm_persist; same bits)
    bool opt_f32_mfma = true;        // f32 GEMMs above four rows on v_mfma_f32_32x32x2_f32 (bit-identical to the FMA tile kernel)
    char *gh = nullptr;                                                               // pinned host block
    int *gh_collect = nullptr;       // pinned landing zone of the token gather: [B][1 + COLLECT_STRIDE] + n_active
    int64_t graph_replays = 0, eager_steps = 0, decode_fallbacks = 0, decode_fallback_rounds = 0;
    // pipelined graph steps (option "pipeline" = E, 1..4): launch sequences of CONSECUTIVE steps run beside each other on
    // their own HIP streams -- see the comment at pipe_step().  Everything a step in flight owns exists once per slot:
    // workspace set, descriptor blocks, joint.enc buffer, token landing zone, graphs (their kernel arguments point into
    // the slot).  E + 1 steps are in flight; slot of a step = its sequence number mod NSLOT.
    static const int MAXSEG = 4, LSLOT = MAXSEG + 1;     // lanes mode: E + 1 steps in flight, slot = sequence number mod LSLOT
    static const int GP_C = 2, GP_Y = FUSED_GROUP, GP_S = GP_C * GP_Y;   // grouped mode ("pipeline" = 8): 2 chains x 4 problems per launch = 8 stages
    static const int NSLOT = GP_S + 3;                    // grouped mode: 8 steps in flight + the one being decoded + the one being collected + one spare
    struct WS { float *x, *x2, *part, *q, *glu, *sub_a, *hfuse; void *a, *hbuf, *ctx, *cbuf, *sub_b; };
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
    struct { int16_t *p = nullptr; size_t cap = 0; } pcm_pin[4];
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

static const int COLLECT_STRIDE = 256;

template <typename Tp>
static int dalloc(nasr_engine *e, Tp **out, size_t n_elems) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(n_elems * sizeof(Tp), 16);
    HIPCHK(hipMalloc(&p, bytes));
    e->allocs.push_back(p);
    *out = (Tp *)p;
    return 0;
}

// ---- profiling ------------------------------------------------------------------------------
static void prof_flush(nasr_engine *e) {
    Prof &pf = e->prof;
    if (pf.pending.empty()) return;
    hipStreamSynchronize(e->st);
    for (auto &r : pf.pending) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.a, r.b);
        nasr_kernel_stat &s = pf.stats[r.cat];
        s.launches++;
        s.total_ms += ms;
        s.bytes += r.bytes;
        s.flops += r.flops;
        pf.pool.push_back(r.a);
        pf.pool.push_back(r.b);
    }
    pf.pending.clear();
}
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

// ---- host helpers: dequantisation of GGUF tensor types at upload ---------------------------------
static float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ff, f;
    if (exp == 0) {
        if (man == 0) f = sign;
        else {
            exp = 127 - 15 + 1;
            while (!(man & 0x400)) { man <<= 1; exp--; }
            man &= 0x3ff;
            f = sign | (exp << 23) | (man << 13);
        }
    } else if (exp == 31) f = sign | 0x7f800000u | (man << 13);
    else f = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float r;
    memcpy(&r, &f, 4);
    return r;
}

// numel of a desc
static int64_t desc_numel(const nasr_weight_desc &d) {
    int64_t n = 1;
    for (int i = 0; i < d.n_dims && i < 4; i++) n *= d.ne[i];
    return n;
}

// returns f32 host copy (dequantised); layouts per scripts/convert_to_gguf.py:118-204
static int to_f32(const nasr_weight_desc &d, std::vector<float> &out) {
    const int64_t n = desc_numel(d);
    out.resize((size_t)n);
    switch (d.type) {
    case NASR_TYPE_F32: memcpy(out.data(), d.data, (size_t)n * 4); return 0;
    case NASR_TYPE_F16: {
        const uint16_t *s = (const uint16_t *)d.data;
        for (int64_t i = 0; i < n; i++) out[(size_t)i] = f16_to_f32(s[i]);
        return 0;
    }
    case NASR_TYPE_Q8_0: {   // 34-byte blocks: f16 scale + 32 x int8
        if (n % 32) return fail("%s: Q8_0 numel not a multiple of 32", d.name);
        const uint8_t *s = (const uint8_t *)d.data;
        for (int64_t b = 0; b < n / 32; b++) {
            uint16_t hs; memcpy(&hs, s + b * 34, 2);
            const float sc = f16_to_f32(hs);
            const int8_t *qv = (const int8_t *)(s + b * 34 + 2);
            for (int i = 0; i < 32; i++) out[(size_t)(b * 32 + i)] = sc * (float)qv[i];
        }
        return 0;
    }
    case NASR_TYPE_Q4_0: {   // 18-byte blocks: f16 scale + 16 bytes; low nibbles = elems 0..15, high = 16..31
        if (n % 32) return fail("%s: Q4_0 numel not a multiple of 32", d.name);
        const uint8_t *s = (const uint8_t *)d.data;
        for (int64_t b = 0; b < n / 32; b++) {
            uint16_t hs; memcpy(&hs, s + b * 18, 2);
            const float sc = f16_to_f32(hs);
            const uint8_t *qv = s + b * 18 + 2;
            for (int i = 0; i < 16; i++) {
                out[(size_t)(b * 32 + i)] = sc * (float)((int)(qv[i] & 0xf) - 8);
                out[(size_t)(b * 32 + 16 + i)] = sc * (float)((int)(qv[i] >> 4) - 8);
            }
        }
        return 0;
    }
    }
    return fail("%s: unsupported tensor type %d", d.name, d.type);
}

// the upload-time conversion as a host utility (no device involved): lets a caller, and the CPU test suite, check what the
// engine will compute with for a given GGUF tensor
extern "C" int64_t nasr_tensor_to_f32(const nasr_weight_desc *t, float *out, int64_t cap) {
    if (!t || !t->data || !out) return fail("null argument");
    if (t->n_dims < 1 || t->n_dims > 4) return fail("bad n_dims %d", t->n_dims);
    for (int i = 0; i < t->n_dims; i++)
        if (t->ne[i] <= 0) return fail("bad extent");
    const int64_t n = desc_numel(*t);
    if (n > cap) return fail("output buffer too small (%lld > %lld)", (long long)n, (long long)cap);
    std::vector<float> v;
    if (to_f32(*t, v)) return -1;
    memcpy(out, v.data(), (size_t)n * 4);
    return n;
}

struct Loader {
    nasr_engine *e;
    std::map<std::string, const nasr_weight_desc *> by_name;
    float *scratch = nullptr; size_t scratch_cap = 0;   // device f32 staging for packing

    int get(const std::string &name, int64_t numel, std::vector<float> &out) {
        auto it = by_name.find(name);
        if (it == by_name.end()) return fail("missing tensor: %s", name.c_str());
        if (desc_numel(*it->second) != numel)
            return fail("%s: has %lld elements, expected %lld", name.c_str(), (long long)desc_numel(*it->second), (long long)numel);
        return to_f32(*it->second, out);
    }
    // plain f32 vector / small tensor
    int vec(const std::string &name, int64_t numel, float **dev) {
        std::vector<float> h;
        if (get(name, numel, h)) return -1;
        if (dalloc(e, dev, (size_t)numel)) return -1;
        HIPCHK(hipMemcpy(*dev, h.data(), (size_t)numel * 4, hipMemcpyHostToDevice));
        return 0;
    }
    template <typename Tp>
    int upload_vec(const std::vector<Tp> &h, Tp **dev) {
        if (dalloc(e, dev, h.size())) return -1;
        HIPCHK(hipMemcpy(*dev, h.data(), h.size() * sizeof(Tp), hipMemcpyHostToDevice));
        return 0;
    }
    // matrix [N][K] f32 host -> engine GEMM layout (packed bf16 tiles, or f32 row-major)
    int matrix(const std::vector<float> &h, int N, int K, void **dev) {
        const size_t n = (size_t)N * K;
        if (!e->bf16) {
            float *d;
            if (dalloc(e, &d, n)) return -1;
            HIPCHK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
            *dev = d;
            return 0;
        }
        if (n > scratch_cap) {
            if (scratch) hipFree(scratch);
            HIPCHK(hipMalloc((void **)&scratch, n * 4));
            scratch_cap = n;
        }
        HIPCHK(hipMemcpy(scratch, h.data(), n * 4, hipMemcpyHostToDevice));
        bf16_t *d;
        if (dalloc(e, &d, n)) return -1;
        launch_pack_weight_bf16(scratch, d, N, K, e->st);
        HIPCHK(hipStreamSynchronize(e->st));
        *dev = d;
        return 0;
    }
    int matrix_named(const std::string &name, int N, int K, void **dev) {
        std::vector<float> h;
        if (get(name, (int64_t)N * K, h)) return -1;
        return matrix(h, N, K, dev);
    }
};

static void transpose_9x256(const std::vector<float> &w /*[256][9]*/, std::vector<float> &t /*[9][256]*/) {
    t.resize(9 * SUBC);
    for (int c = 0; c < SUBC; c++)
        for (int k = 0; k < 9; k++) t[(size_t)k * SUBC + c] = w[(size_t)c * 9 + k];
}

// f32 MFMA (16x16x4) A-fragment packing for the decoder matrices: tile (nt, kg) = 16 rows x 16 k,
// lane l = q*16 + r holds W[row(nt, r)][kg*16 + 4q .. +4).  lstm_order: tile row r = 4*u + gate
// maps to source row gate*640 + nt*4 + u, so one lane ends up with the 4 gates of one unit.
static void pack_f32_mfma(const std::vector<float> &w, int N, int K, bool lstm_order, std::vector<float> &out) {
    const int NT = (N + 15) / 16, KG = K / 16;
    out.assign((size_t)NT * KG * 64 * 4, 0.0f);
    for (int nt = 0; nt < NT; nt++)
        for (int kg = 0; kg < KG; kg++)
            for (int lane = 0; lane < 64; lane++) {
                const int q = lane >> 4, r = lane & 15;
                const int row = lstm_order ? (r & 3) * HID + nt * 4 + (r >> 2) : nt * 16 + r;
                if (row >= N) continue;
                for (int sI = 0; sI < 4; sI++)
                    out[(((size_t)nt * KG + kg) * 64 + lane) * 4 + sI] = w[(size_t)row * K + kg * 16 + 4 * q + sI];
            }
}

static void host_pos_emb(int position, float *out) {   // reference src/nemo-ggml.cpp:17-32
    const float p = (float)position;
    for (int i = 0; i < D; i += 2) {
        const float div_term = std::exp(-(float)i * std::log(10000.0f) / (float)D);
        out[i] = std::sin(p * div_term);
        out[i + 1] = std::cos(p * div_term);
    }
}

static int load_weights(nasr_engine *e, const nasr_weight_desc *w, int n_w) {
    Loader ld;
    ld.e = e;
    for (int i = 0; i < n_w; i++) {
        if (!w[i].name || !w[i].data) return fail("weight %d: null name/data", i);
        ld.by_name[w[i].name] = &w[i];
    }
    std::vector<float> h, t;
    // ---- a-1 constants: padded window, transposed filterbank, twiddles (src/preprocessor.cpp:80-110,:296-299)
    {
        if (ld.get("preprocessor.featurizer.window", WIN, h)) return -1;
        std::vector<float> win(NFFT, 0.0f);
        memcpy(win.data() + (NFFT - WIN) / 2, h.data(), WIN * 4);
        if (ld.upload_vec(win, &e->window)) return -1;
        if (ld.get("preprocessor.featurizer.fb", (int64_t)NMEL * NBINS, h)) return -1;
        std::vector<float> fbT((size_t)NBINS * NMEL);
        for (int m = 0; m < NMEL; m++)
            for (int k = 0; k < NBINS; k++) fbT[(size_t)k * NMEL + m] = h[(size_t)m * NBINS + k];
        if (ld.upload_vec(fbT, &e->fbT)) return -1;
        // band of every (triangular) filter: the reference sums all 257 bins in order (src/preprocessor.cpp:374-383);
        // outside the band the terms are +0 * power = +0 and sum + 0 == sum, so summing the band alone is bit-identical
        std::vector<int> band(2 * NMEL);
        for (int m = 0; m < NMEL; m++) {
            int lo = NBINS, hi = 0;
            for (int k = 0; k < NBINS; k++)
                if (h[(size_t)m * NBINS + k] != 0.0f) { lo = std::min(lo, k); hi = k + 1; }
            if (lo > hi) lo = hi = 0;
            band[2 * m] = lo; band[2 * m + 1] = hi;
        }
        if (ld.upload_vec(band, &e->fb_band)) return -1;
        std::vector<float> ct(NFFT), sn(NFFT);
        for (int i = 0; i < NFFT; i++) {
            const float theta = (2.0f * (float)M_PI * (float)i) / (float)NFFT;
            sn[i] = sinf(theta);
            ct[i] = cosf(theta);
        }
        if (ld.upload_vec(ct, &e->cos_t) || ld.upload_vec(sn, &e->sin_t)) return -1;
    }
    // ---- a-2 subsampling -----------------------------------------------------------------
    const std::string pe = "encoder.pre_encode.";
    if (ld.get(pe + "conv.0.weight", SUBC * 9, h)) return -1;
    transpose_9x256(h, t);
    if (ld.upload_vec(t, &e->w0t) || ld.vec(pe + "conv.0.bias", SUBC, &e->b0)) return -1;
    if (ld.get(pe + "conv.2.weight", SUBC * 9, h)) return -1;
    transpose_9x256(h, t);
    if (ld.upload_vec(t, &e->w2t) || ld.vec(pe + "conv.2.bias", SUBC, &e->b2)) return -1;
    if (ld.get(pe + "conv.5.weight", SUBC * 9, h)) return -1;
    transpose_9x256(h, t);
    if (ld.upload_vec(t, &e->w5t) || ld.vec(pe + "conv.5.bias", SUBC, &e->b5)) return -1;
    if (ld.matrix_named(pe + "conv.3.weight", SUBC, SUBC, &e->w3) || ld.vec(pe + "conv.3.bias", SUBC, &e->b3)) return -1;
    if (ld.matrix_named(pe + "conv.6.weight", SUBC, SUBC, &e->w6) || ld.vec(pe + "conv.6.bias", SUBC, &e->b6)) return -1;
    {   // out.weight [1024][c*17+w] -> [1024][w*256+c]: our activations are channel-fastest (:1014-1017)
        if (ld.get(pe + "out.weight", (int64_t)D * SUBFLAT, h)) return -1;
        t.resize(h.size());
        for (int n = 0; n < D; n++)
            for (int c = 0; c < SUBC; c++)
                for (int wv = 0; wv < SUBF; wv++)
                    t[(size_t)n * SUBFLAT + wv * SUBC + c] = h[(size_t)n * SUBFLAT + c * SUBF + wv];
        if (ld.matrix(t, D, SUBFLAT, &e->sub_out_w) || ld.vec(pe + "out.bias", D, &e->sub_out_b)) return -1;
    }
    // ---- encoder layers --------------------------------------------------------------------
    const int ks = e->hp.kernel_size;
    e->L.resize(e->hp.n_layers);
    for (int l = 0; l < e->hp.n_layers; l++) {
        LayerW &L = e->L[l];
        memset(&L, 0, sizeof(L));
        const std::string p = "encoder.layers." + std::to_string(l) + ".";
        if (ld.vec(p + "norm_feed_forward1.weight", D, &L.ln_ff1_w) || ld.vec(p + "norm_feed_forward1.bias", D, &L.ln_ff1_b) ||
            ld.vec(p + "norm_self_att.weight", D, &L.ln_att_w) || ld.vec(p + "norm_self_att.bias", D, &L.ln_att_b) ||
            ld.vec(p + "norm_conv.weight", D, &L.ln_conv_w) || ld.vec(p + "norm_conv.bias", D, &L.ln_conv_b) ||
            ld.vec(p + "norm_feed_forward2.weight", D, &L.ln_ff2_w) || ld.vec(p + "norm_feed_forward2.bias", D, &L.ln_ff2_b) ||
            ld.vec(p + "norm_out.weight", D, &L.ln_out_w) || ld.vec(p + "norm_out.bias", D, &L.ln_out_b) ||
            ld.vec(p + "self_attn.pos_bias_u", D, &L.bias_u) || ld.vec(p + "self_attn.pos_bias_v", D, &L.bias_v) ||
            ld.vec(p + "conv.depthwise_conv.weight", (int64_t)ks * D, &L.dw) ||
            ld.vec(p + "conv.batch_norm.weight", D, &L.cln_w) || ld.vec(p + "conv.batch_norm.bias", D, &L.cln_b) ||
            ld.vec(p + "self_attn.linear_pos.weight", (int64_t)D * D, &L.wpos_f32))
            return -1;
        if (ld.matrix_named(p + "feed_forward1.linear1.weight", FF, D, &L.ff1_w1) ||
            ld.matrix_named(p + "feed_forward1.linear2.weight", D, FF, &L.ff1_w2) ||
            ld.matrix_named(p + "feed_forward2.linear1.weight", FF, D, &L.ff2_w1) ||
            ld.matrix_named(p + "feed_forward2.linear2.weight", D, FF, &L.ff2_w2) ||
            ld.matrix_named(p + "self_attn.linear_out.weight", D, D, &L.wo) ||
            ld.matrix_named(p + "conv.pointwise_conv2.weight", D, D, &L.pw2))
            return -1;
        {   // fused QKV [3072][1024]
            std::vector<float> qkv((size_t)3 * D * D), part;
            const char *nm[3] = {"self_attn.linear_q.weight", "self_attn.linear_k.weight", "self_attn.linear_v.weight"};
            for (int i = 0; i < 3; i++) {
                if (ld.get(p + nm[i], (int64_t)D * D, part)) return -1;
                memcpy(qkv.data() + (size_t)i * D * D, part.data(), (size_t)D * D * 4);
            }
            if (ld.matrix(qkv, 3 * D, D, &L.wqkv)) return -1;
        }
        {   // pointwise_conv1 rows interleaved (value c, gate c) so GLU is an epilogue (:657-664)
            if (ld.get(p + "conv.pointwise_conv1.weight", (int64_t)2 * D * D, h)) return -1;
            t.resize(h.size());
            for (int c = 0; c < D; c++) {
                memcpy(&t[(size_t)(2 * c) * D], &h[(size_t)c * D], D * 4);
                memcpy(&t[(size_t)(2 * c + 1) * D], &h[(size_t)(D + c) * D], D * 4);
            }
            if (ld.matrix(t, 2 * D, D, &L.pw1)) return -1;
        }
    }
    // ---- decoder / joint (always f32) -------------------------------------------------------
    const std::string dp = "decoder.prediction.";
    if (ld.vec(dp + "embed.weight", (int64_t)VOCAB * HID, &e->embed)) return -1;
    for (int l = 0; l < 2; l++) {
        const std::string s = std::to_string(l);
        if (ld.get(dp + "dec_rnn.lstm.weight_ih_l" + s, (int64_t)4 * HID * HID, h)) return -1;
        pack_f32_mfma(h, 4 * HID, HID, true, t);
        if (ld.upload_vec(t, &e->w_ih[l])) return -1;
        if (ld.get(dp + "dec_rnn.lstm.weight_hh_l" + s, (int64_t)4 * HID * HID, h)) return -1;
        pack_f32_mfma(h, 4 * HID, HID, true, t);
        if (ld.upload_vec(t, &e->w_hh[l])) return -1;
        if (ld.vec(dp + "dec_rnn.lstm.bias_ih_l" + s, 4 * HID, &e->b_ih[l]) ||
            ld.vec(dp + "dec_rnn.lstm.bias_hh_l" + s, 4 * HID, &e->b_hh[l]))
            return -1;
    }
    if (ld.get("joint.enc.weight", (int64_t)JNT * D, h)) return -1;
    pack_f32_mfma(h, JNT, D, false, t);
    if (ld.upload_vec(t, &e->jenc_w)) return -1;
    if (ld.vec("joint.enc.bias", JNT, &e->jenc_b) ||
        ld.vec("joint.pred.bias", JNT, &e->pred_b) || ld.vec("joint.joint_net.2.bias", VOCAB, &e->out_b))
        return -1;
    if (ld.get("joint.pred.weight", (int64_t)JNT * HID, h)) return -1;
    pack_f32_mfma(h, JNT, HID, false, t);
    if (ld.upload_vec(t, &e->pred_w)) return -1;
    if (ld.get("joint.joint_net.2.weight", (int64_t)VOCAB * JNT, h)) return -1;
    pack_f32_mfma(h, VOCAB, JNT, false, t);
    if (ld.upload_vec(t, &e->out_w)) return -1;
    // ---- prompt kernel (multilingual) -----------------------------------------------------
    const int P = e->hp.num_prompts;
    if (P > 0) {
        if (ld.get("prompt_kernel.0.weight", (int64_t)2048 * (D + P), h)) return -1;
        std::vector<float> a((size_t)2048 * D), pp((size_t)P * 2048);
        for (int n = 0; n < 2048; n++) {
            memcpy(&a[(size_t)n * D], &h[(size_t)n * (D + P)], D * 4);
            for (int k = 0; k < P; k++) pp[(size_t)k * 2048 + n] = h[(size_t)n * (D + P) + D + k];
        }
        if (ld.upload_vec(a, &e->pk1a) || ld.upload_vec(pp, &e->pk1p) || ld.vec("prompt_kernel.0.bias", 2048, &e->pk1_b) ||
            ld.vec("prompt_kernel.2.weight", (int64_t)D * 2048, &e->pk2_w) || ld.vec("prompt_kernel.2.bias", D, &e->pk2_b))
            return -1;
    } else if (ld.by_name.count("prompt_kernel.0.weight")) {
        return fail("prompt_kernel weights present but num_prompts is 0");   // src/nemo-ggml.cpp:431-434
    }
    if (ld.scratch) hipFree(ld.scratch);
    return 0;
}

// pos projection rows for chunk length T, per layer: P[r] = W_pos . emb(rel = 70+T-1-r)
// (reference recomputes this GEMM in every layer of every chunk, src/nemo-stream.cpp:514-516;
// its operands are input-independent so it is done once per (layer, T) here)
static int ensure_posproj(nasr_engine *e, int T) {
    if (e->L.empty() || e->L[0].posproj[T]) return 0;
    const int n_rel = LCTX + 2 * T - 1;
    std::vector<float> emb((size_t)n_rel * D);
    for (int r = 0; r < n_rel; r++) host_pos_emb((LCTX + T - 1) - r, &emb[(size_t)r * D]);
    float *demb, *dout;
    HIPCHK(hipMalloc((void **)&demb, emb.size() * 4));
    HIPCHK(hipMalloc((void **)&dout, emb.size() * 4));
    HIPCHK(hipMemcpy(demb, emb.data(), emb.size() * 4, hipMemcpyHostToDevice));
    for (auto &L : e->L) {
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = demb; g.W = L.wpos_f32; g.M = n_rel; g.N = D; g.K = D; g.lda = D;
        g.epi = EPI_PART_F32; g.out_f32 = dout; g.ldo = D; g.splits = 1;
        g.f32_fma_tile = e->opt_f32_mfma ? 0 : 1;
        launch_gemm_f32(g, e->st);
        if (e->bf16) {
            bf16_t *pp;
            if (dalloc(e, &pp, emb.size())) return -1;
            launch_f32_to_bf16(dout, pp, (int64_t)emb.size(), e->st);
            L.posproj[T] = pp;
        } else {
            float *pp;
            if (dalloc(e, &pp, emb.size())) return -1;
            HIPCHK(hipMemcpyAsync(pp, dout, emb.size() * 4, hipMemcpyDeviceToDevice, e->st));
            L.posproj[T] = pp;
        }
        HIPCHK(hipStreamSynchronize(e->st));
    }
    hipFree(demb);
    hipFree(dout);
    return 0;
}

// one workspace set: the buffers a launch sequence passes from kernel to kernel, for w_rows rows
static int alloc_ws(nasr_engine *e, nasr_engine::WS &w) {
    const size_t M = (size_t)e->w_rows;
    int rc = 0;
    rc |= dalloc(e, &w.x, M * D);
    rc |= dalloc(e, &w.x2, M * D);
    rc |= dalloc(e, &w.part, 8 * M * D);
    rc |= dalloc(e, &w.q, M * D);
    rc |= dalloc(e, &w.glu, M * D);
    rc |= dalloc(e, &w.hfuse, e->hp.num_prompts > 0 ? M * 2048 : 4);
    { char *p; rc |= dalloc(e, &p, M * D * e->esz); w.a = p; }
    { char *p; rc |= dalloc(e, &p, M * FF * e->esz); w.hbuf = p; }
    { char *p; rc |= dalloc(e, &p, M * D * e->esz); w.ctx = p; }
    { char *p; rc |= dalloc(e, &p, M * D * e->esz); w.cbuf = p; }
    // subsampling ping-pong buffers, [chunks][H2][33][256] (conv0 is fused into the first depthwise conv and never
    // stored).  Rows per encoder frame are largest for multi-chunk steps at R = 0: H2 = 2T + 3 = 5 per frame.
    rc |= dalloc(e, &w.sub_a, M * 5 * 33 * SUBC);
    { char *p; rc |= dalloc(e, &p, (M * 5 * 33 * SUBC) * 4); w.sub_b = p; }
    return rc;
}
// the enqueue functions address the workspace through the engine's own fields: point them at a set
static void use_ws(nasr_engine *e, const nasr_engine::WS &w) {
    e->x = w.x; e->x2 = w.x2; e->part = w.part; e->q = w.q; e->glu = w.glu; e->sub_a = w.sub_a; e->hfuse = w.hfuse;
    e->a = w.a; e->hbuf = w.hbuf; e->ctx = w.ctx; e->cbuf = w.cbuf; e->sub_b = w.sub_b;
}

// ---------------------------------------------------------------------------------------
static void engine_destroy_impl(nasr_engine *e);
extern "C" int nasr_engine_create(nasr_engine **out, int device_id, int dtype, const nasr_hparams *hp,
                                  const nasr_weight_desc *weights, int n_weights, int max_streams) {
    ApiGuard api_guard;
    if (!out || !hp || !weights) return fail("nasr_engine_create: null argument");
    *out = nullptr;
    if (dtype != NASR_DTYPE_F32 && dtype != NASR_DTYPE_BF16) return fail("unsupported dtype %d", dtype);
    if (hp->d_model != D || hp->n_heads != NH || hp->d_head != DH || hp->d_ff != FF || hp->n_mels != NMEL ||
        hp->vocab_size != VOCAB || hp->decoder_dim != HID || hp->joint_dim != JNT || hp->att_left_context != LCTX ||
        hp->subsampling_factor != 8)
        return fail("this build is specialised for d_model 1024 / 8x128 heads / d_ff 4096 / 128 mels / vocab 1025 / "
                    "LSTM 640 / left context 70 (got d_model=%d heads=%d d_ff=%d vocab=%d)",
                    hp->d_model, hp->n_heads, hp->d_ff, hp->vocab_size);
    if (hp->n_layers < 1 || hp->kernel_size < 2 || hp->kernel_size > MAX_KS) return fail("bad n_layers/kernel_size");
    if (max_streams < 1 || max_streams > 4096) return fail("max_streams out of range");
    int n_dev = 0;
    HIPCHK(hipGetDeviceCount(&n_dev));
    if (n_dev <= 0) return fail("no HIP device visible: the MI355X engine has no CPU fallback");
    if (device_id < 0 || device_id >= n_dev) return fail("device %d out of range (%d visible)", device_id, n_dev);
    HIPCHK(hipSetDevice(device_id));
    nasr_engine *e = new nasr_engine();
    e->device = device_id; e->dtype = dtype; e->max_streams = max_streams; e->hp = *hp;
    e->bf16 = dtype == NASR_DTYPE_BF16; e->esz = e->bf16 ? 2 : 4;
    if (hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking) != hipSuccess) { delete e; return fail("hipStreamCreate failed"); }
    init_gemm_kernel_attributes();
    init_fused_kernel_attributes();
#ifdef NASR_STAMPS
    if (!g_stamp_buf) { hipMalloc((void **)&g_stamp_buf, (size_t)STAMP_SLOTS * STAMP_PER_SLOT * 32 * 8); hipMemset(g_stamp_buf, 0, (size_t)STAMP_SLOTS * STAMP_PER_SLOT * 32 * 8); }
#endif
    if (load_weights(e, weights, n_weights)) { engine_destroy_impl(e); return -1; }

    const size_t S = (size_t)max_streams, Lr = (size_t)hp->n_layers, ks1 = (size_t)hp->kernel_size - 1;
    e->w_rows = std::max(max_streams * TMAX, MAXNEW);
    const size_t M = (size_t)e->w_rows;
    int rc = 0;
    rc |= dalloc(e, &e->abuf, S * 2 * ABUF_CAP);
    rc |= dalloc(e, &e->last_sample, S);
    rc |= dalloc(e, &e->mel_ring, S * MEL_RING * NMEL);
    e->kv_pool.resize(Lr); e->cc_pool.resize(Lr);
    for (size_t l = 0; l < Lr && !rc; l++) {
        char *kp;
        rc |= dalloc(e, &kp, S * 2 * KVC * D * e->esz);
        e->kv_pool[l] = kp;
        rc |= dalloc(e, &e->cc_pool[l], S * 2 * ks1 * D);
    }
    rc |= dalloc(e, &e->dec_h, S * 4 * HID);
    rc |= dalloc(e, &e->dec_c, S * 4 * HID);
    rc |= dalloc(e, &e->ctrl, S);
    rc |= dalloc(e, &e->tok_ring, S * TOK_CAP);
    memset(e->ws, 0, sizeof(e->ws));
    rc |= alloc_ws(e, e->ws[0]);
    use_ws(e, e->ws[0]);
    rc |= dalloc(e, &e->encproj, M * JNT);
    rc |= dalloc(e, &e->predg, S * JNT);
    rc |= dalloc(e, &e->key, M);
    rc |= dalloc(e, &e->n_active, 4);
    rc |= dalloc(e, &e->dlist, S);
    rc |= dalloc(e, &e->rowmap, M);
    rc |= dalloc(e, &e->tok_frame, S * TOK_CAP);
    rc |= dalloc(e, &e->collect_dev, S * (1 + COLLECT_STRIDE) + 4);
    if (rc) { engine_destroy_impl(e); return -1; }
    e->pin_cap = 8u << 20;
    e->pin_off = 256;
    if (hipHostMalloc((void **)&e->pin, e->pin_cap, hipHostMallocDefault) != hipSuccess) { engine_destroy_impl(e); return fail("hipHostMalloc failed"); }
    e->ddesc_cap = 8u << 20;
    if (hipMalloc((void **)&e->ddesc, e->ddesc_cap) != hipSuccess) { engine_destroy_impl(e); return fail("hipMalloc desc failed"); }
    {
        const size_t desc_bytes = S * (sizeof(RowDesc) + sizeof(PcmDesc) + 2 * sizeof(int)) + M * sizeof(RowDesc) + 64;
        const size_t col_bytes = (S * (1 + COLLECT_STRIDE) + 4) * sizeof(int);
        e->desc_bytes = desc_bytes; e->col_bytes = col_bytes;
        if (hipHostMalloc((void **)&e->gh, desc_bytes + col_bytes, hipHostMallocDefault) != hipSuccess) { engine_destroy_impl(e); return fail("hipHostMalloc failed"); }
        e->gh_collect = (int *)(e->gh + desc_bytes);
        if (dalloc(e, &e->g_desc, desc_bytes)) { engine_destroy_impl(e); return -1; }
    }
    e->slots.assign(S, nullptr);
    e->tap_mel_frames.assign(S, 0);
    e->tap_mel_row.assign(S, 0);
    if (hipStreamSynchronize(e->st) != hipSuccess) { engine_destroy_impl(e); return fail("engine init sync failed"); }
    *out = e;
    return 0;
}

extern "C" void nasr_engine_destroy(nasr_engine *e) {
    ApiGuard api_guard;
    engine_destroy_impl(e);
}
static void engine_destroy_impl(nasr_engine *e) {
#ifdef NASR_STAMPS
    if (e && g_stamp_buf) {
        hipSetDevice(e->device);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)STAMP_SLOTS * STAMP_PER_SLOT * 32);
        hipMemcpy(h.data(), g_stamp_buf, h.size() * 8, hipMemcpyDeviceToHost);
        const char *path = getenv("NASR_STAMPS_OUT");
        FILE *f = path ? fopen(path, "w") : stderr;
        if (f) {                                       // one line per launch: pipeline slot, launch index (8 x layer + k), 8 + 8 stamps (10 ns ticks)
            for (int ps = 0; ps < STAMP_SLOTS; ps++)
                for (int k = 0; k < STAMP_PER_SLOT; k++) {
                    const unsigned long long *r = &h[((size_t)ps * STAMP_PER_SLOT + k) * 32];
                    if (!r[0]) continue;
                    fprintf(f, "%d %d", ps, k);
                    for (int i = 0; i < 8; i++) fprintf(f, " %llu", r[i]);
                    for (int i = 0; i < 8; i++) fprintf(f, " %llu", r[16 + i]);
                    fprintf(f, "\n");
                }
            if (f != stderr) fclose(f);
        }
    }
#endif
    if (!e) return;
    hipSetDevice(e->device);
    if (e->st) hipStreamSynchronize(e->st);
    if (getenv("NASR_STATS"))
        fprintf(stderr, "nasr: graph replays %lld (%lld pipelined), eager steps %lld, decode fallbacks %lld (%lld rounds); pipelined steps: host %.1f us in "
                "hipGraphLaunch + %.1f us waiting per step\n", (long long)e->graph_replays,
                (long long)e->pipe_steps, (long long)e->eager_steps, (long long)e->decode_fallbacks, (long long)e->decode_fallback_rounds,
                e->pipe_steps + e->gp_steps ? 1e6 * e->host_launch_s / (e->pipe_steps + e->gp_steps) : 0.0,
                e->pipe_steps + e->gp_steps ? 1e6 * e->host_wait_s / (e->pipe_steps + e->gp_steps) : 0.0);
    if (getenv("NASR_STATS") && e->gp_steps)
        fprintf(stderr, "nasr: grouped pipeline: %lld steps, chain launches %lld through graphs, %lld eager\n", (long long)e->gp_steps,
                (long long)e->gp_graph_chains, (long long)e->gp_eager_chains);
    for (auto *s : e->slots) delete s;
    for (void *p : e->allocs) hipFree(p);
    for (auto &kv : e->graphs) hipGraphExecDestroy(kv.second);
    for (auto &per_slot : e->gp_graphs) for (auto &m : per_slot) for (auto &kv : m) if (kv.second) hipGraphExecDestroy(kv.second);
    for (auto &ce : e->gp_ev) for (auto &ev : ce) if (ev) hipEventDestroy(ev);
    for (int k = 1; k < nasr_engine::MAXSEG; k++) if (e->lane[k]) hipStreamSynchronize(e->lane[k]);
    for (int p = 0; p < nasr_engine::NSLOT; p++) {
        nasr_engine::Pipe &P = e->pipe[p];
        for (auto &m : P.seg_graphs) for (auto &kv : m) hipGraphExecDestroy(kv.second);
        for (auto &kv : P.dec_graphs) hipGraphExecDestroy(kv.second);
        if (P.gh_dmeta) hipHostFree(P.gh_dmeta);
        if (p > 0 && P.gh) hipHostFree(P.gh);            // slot 0 shares the engine's own block
        for (auto ev : P.seg_done) if (ev) hipEventDestroy(ev);
        if (P.dec_done) hipEventDestroy(P.dec_done);
    }
    for (int k = 1; k < nasr_engine::MAXSEG; k++) if (e->lane[k]) hipStreamDestroy(e->lane[k]);
    for (hipStream_t ls : e->lent) { hipStreamSynchronize(ls); hipStreamDestroy(ls); }
    if (e->gh) hipHostFree(e->gh);
    if (e->pin) hipHostFree(e->pin);
    if (e->ddesc) hipFree(e->ddesc);
    if (e->pcm_stage) hipFree(e->pcm_stage);
    for (auto &pin : e->pcm_pin) if (pin.p) hipHostFree(pin.p);
    if (e->mel_stage) hipFree(e->mel_stage);
    if (e->tap_mel) hipFree(e->tap_mel);
    if (e->tap_sub) hipFree(e->tap_sub);
    if (e->tap_layers) hipFree(e->tap_layers);
    if (e->tap_enc) hipFree(e->tap_enc);
    for (auto &r : e->prof.pending) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (auto ev : e->prof.pool) hipEventDestroy(ev);
    if (e->st) hipStreamDestroy(e->st);
    delete e;
}

// ---- descriptor staging: pinned bump arena -> device arena (async) --------------------------
template <typename Tp>
static int stage_desc(nasr_engine *e, const std::vector<Tp> &host, const Tp **dev_out) {
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

static int pipe_drain(nasr_engine *e);     // completes a pipelined step that is still in flight (defined with the graph steps)
static void release_lanes(nasr_engine *e);  // destroys the lane streams beyond max_lanes

// ---- streams ------------------------------------------------------------------------------------
// keep_reference_state: what the reference's nemo_stream_context::reset() leaves behind (src/nemo-stream.cpp:95-115):
// encoder_graph.reset() only flips a flag (:31-34) and nothing re-zeroes the cache tensors, and the per-stream
// preprocessor is not touched -- so the conv cache, the K/V rows (hidden by cache_valid_len = 0: every cached key gets
// -1e9 and weight exactly 0) and the preprocessor's carry (un-framed samples, last_sample) survive.
static int stream_zero_state(nasr_stream *s, bool keep_reference_state = false) {
    nasr_engine *e = s->e;
    const size_t slot = (size_t)s->slot, ks1 = (size_t)e->hp.kernel_size - 1;
    if (!keep_reference_state) {
        for (int l = 0; l < e->hp.n_layers; l++) {
            HIPCHK(hipMemsetAsync((char *)e->kv_pool[l] + slot * 2 * KVC * D * e->esz, 0, (size_t)2 * KVC * D * e->esz, e->st));
            HIPCHK(hipMemsetAsync(e->cc_pool[l] + slot * 2 * ks1 * D, 0, 2 * ks1 * D * 4, e->st));
        }
        HIPCHK(hipMemsetAsync(e->abuf + slot * 2 * ABUF_CAP, 0, (size_t)2 * ABUF_CAP * 4, e->st));
        HIPCHK(hipMemsetAsync(e->last_sample + slot, 0, 4, e->st));
    }
    HIPCHK(hipMemsetAsync(e->dec_h + slot * 4 * HID, 0, 4 * HID * 4, e->st));
    HIPCHK(hipMemsetAsync(e->dec_c + slot * 4 * HID, 0, 4 * HID * 4, e->st));
    HIPCHK(hipMemsetAsync(e->mel_ring + slot * MEL_RING * NMEL, 0, (size_t)MEL_RING * NMEL * 4, e->st));
    DecCtrl c;
    memset(&c, 0, sizeof(c));
    c.prev_token = BLANK;                      // src/nemo-stream.cpp:55-56
    c.dirty = 1;                               // no LSTM candidate computed yet
    std::vector<DecCtrl> cv(1, c);
    const DecCtrl *dsrc;
    if (stage_desc(e, cv, &dsrc)) return -1;
    HIPCHK(hipMemcpyAsync(e->ctrl + slot, dsrc, sizeof(DecCtrl), hipMemcpyDeviceToDevice, e->st));
    if (!keep_reference_state) {
        s->abuf_cnt = NFFT / 2;                // 256 zero samples pre-seeded, src/preprocessor.cpp:220-221
        s->abuf_par = 0;
        s->kv_head = 0;
        s->cc_par = 0;
    }
    s->mel_start = 0;
    s->mel_count = PRE_CACHE;                  // 9 literal-zero frames, src/nemo-stream.cpp:73-74 (reset: :100-102)
    s->valid_len = 0;                          // :81 (reset: :112)
    s->chunks = 0;
    s->tok_read = 0;
    s->samples_in = 0;
    s->last_T = 0;
    s->last_ws = 0;
    return 0;
}

extern "C" int nasr_stream_create(nasr_engine *e, int right_context, int prompt_index, nasr_stream **out) {
    ApiGuard api_guard;
    if (!e || !out) return fail("nasr_stream_create: null argument");
    *out = nullptr;
    if (right_context != 0 && right_context != 1 && right_context != 6 && right_context != 13)
        return fail("right_context must be 0, 1, 6 or 13 (src/nemo-stream.h:15-20), got %d", right_context);
    HIPCHK(hipSetDevice(e->device));
    int slot = -1;
    for (int i = 0; i < e->max_streams; i++)
        if (!e->slots[i]) { slot = i; break; }
    if (slot < 0) return fail("stream pool exhausted (max_streams=%d)", e->max_streams);
    if (ensure_posproj(e, 1 + right_context)) return -1;
    nasr_stream *s = new nasr_stream();
    s->e = e; s->slot = slot; s->R = right_context; s->T = 1 + right_context;
    s->prompt = e->hp.num_prompts > 0 ? prompt_index : -1;
    s->alive = true;
    e->slots[slot] = s;
    if (stream_zero_state(s)) { e->slots[slot] = nullptr; delete s; return -1; }
    *out = s;
    return 0;
}

extern "C" int nasr_stream_reset_ex(nasr_stream *s, int mode) {
    ApiGuard api_guard;
    if (!s) return fail("null stream");
    if (mode != NASR_RESET_FRESH && mode != NASR_RESET_REFERENCE) return fail("unknown reset mode %d", mode);
    HIPCHK(hipSetDevice(s->e->device));
    if (pipe_drain(s->e)) return -1;
    s->tok_queue.clear();
    return stream_zero_state(s, mode == NASR_RESET_REFERENCE);
}

// a reset stream == a fresh stream; the reference's own reset is NASR_RESET_REFERENCE (see stream_zero_state)
extern "C" int nasr_stream_reset(nasr_stream *s) { return nasr_stream_reset_ex(s, NASR_RESET_FRESH); }

extern "C" int nasr_stream_destroy(nasr_stream *s) {
    ApiGuard api_guard;
    if (!s) return 0;
    nasr_engine *e = s->e;
    hipSetDevice(e->device);
    pipe_drain(e);
    hipStreamSynchronize(e->st);
    e->slots[s->slot] = nullptr;
    delete s;
    return 0;
}

extern "C" int nasr_stream_set_prompt(nasr_stream *s, int prompt_index) {
    if (!s) return fail("null stream");
    if (s->e->hp.num_prompts <= 0) return fail("model is not multilingual (num_prompts=0)");   // src/nemo-stream.cpp:737-740
    if (prompt_index < 0 || prompt_index >= s->e->hp.num_prompts) return fail("prompt index %d out of range", prompt_index);
    s->prompt = prompt_index;
    return 0;
}

// ---- the chunk step: encoder + decode for the rows that have a full chunk buffered ----------------
static double gemm_bytes(const nasr_engine *e, int M, int N, int K, int wesz) {
    return (double)N * K * wesz + (double)M * K * e->esz + (double)M * N * 4;
}

static int run_gemm(nasr_engine *e, GemmParams &g, bool f32_weights, const char *tag) {
    (void)tag;
    const bool use_bf16 = e->bf16 && !f32_weights;
    const char *name = !use_bf16 ? "k_gemm_f32" : (g.M <= gemm_skinny_max_m() ? "k_gemm_skinny" : "k_gemm_tiled");
    ProfScope ps(e, name, gemm_bytes(e, g.M, g.N, g.K, use_bf16 ? 2 : 4), 2.0 * g.M * g.N * g.K);
    g.coresident = e->opt_gemm_cores == 0 ? 3 : e->opt_gemm_cores == 1 ? 2 : (e->gemm_coresident ? 1 : 0);
    g.f32_fma_tile = e->opt_f32_mfma ? 0 : 1;
    g.no_persist = e->opt_persist_gemm ? 0 : 1;
    if (use_bf16) launch_gemm_bf16(g, e->st);
    else launch_gemm_f32(g, e->st);
    return 0;
}

// residual GEMM: part = A.W^T (split-K), followed by k_post
static int pick_splits(const nasr_engine *e, int M, int N, int K) {
    if (!e->bf16) return 1;
    const bool skinny = M <= gemm_skinny_max_m();
    int tasks = skinny ? (N / 16) * ((M + 63) / 64) : (N / gemm_tile_n(M, N, EPI_PART_F32)) * ((M + 127) / 128);
    // partial traffic grows with the split factor, and with pipelined steps the CUs a launch leaves idle run another chain's
    // kernels: four splits only up to 40 tiles (three lanes, R = 13: 12 / 16 streams = 32 tiles 1.15 / 1.23 ms with 4 splits
    // against 1.23 / 1.30 with 2; 24 streams = 48 tiles 1.53 vs 1.50; 32 streams = 64 tiles 1.82 vs 1.68; 64 streams = 112 tiles:
    // 2 splits 2.76, 1 split 2.75, 4 splits 3.03)
    if (!skinny) return tasks <= 40 ? 4 : (tasks < 256 ? 2 : 1);
    constexpr int skinny_cap = 8;
    int s = 1;
    while (s < skinny_cap && tasks * s < 256 && (K / 32) / (s * 2) >= 4) s *= 2;
    return s;
}

// ---- small-M form of the 24 layers: 8 launches per layer (kernels_fused.hip) ------------------------
// launches [k0, k1) of the 8 x n_layers launches of the fused layers: a piece boundary may sit inside a layer (every
// intermediate lives in the step's workspace set), so the pieces of a pipelined step can be balanced to a launch
// rec != null: the launches are RECORDED (their parameters appended to *rec) instead of launched, and the closing k_post of the
// last layer is left out (the grouped pipeline launches several steps' records together, then calls this with k0 == k1 == 8 nL for it)
static int run_layers_fused(nasr_engine *e, const RowDesc *rows, int B, int T, int G, int k0, int k1, std::vector<FusedParams> *rec = nullptr) {
    const int TS = G * T;                      // rows per stream in this launch (G chunks batched)
    const int M = B * TS, nL = e->hp.n_layers, ks = e->hp.kernel_size;
    hipStream_t st = e->st;
    float *X[2] = {e->x, e->x2};
    int cur = 0;
    int prev_splits = 0;                // split-K partials pending from the previous layer's FFN2
    int kidx = 0;                       // index of the launch being described
#ifdef NASR_STAMPS
    unsigned long long *stamp_buf = g_stamp_buf + (size_t)g_stamp_pipe * STAMP_PER_SLOT * 32;
#endif
    auto launch = [&](FusedParams &f, const char *name, double bytes, double flops) {
        const int k = kidx++;
        if (k < k0 || k >= k1) return;                          // another piece's launch
        if (rec) { rec->push_back(f); return; }
        ProfScope ps(e, name, bytes, flops);
#ifdef NASR_STAMPS
        f.stamps = stamp_buf + (size_t)k * 32;
#endif
        launch_fused_skinny(f, st);
    };
    auto wbytes = [&](int N, int K) { return (double)N * K * 2 + (double)M * (K + N) * 4; };
    for (int l = k0 / 8; l < (k1 + 7) / 8 && l < nL; l++) {
        LayerW &L = e->L[l];
        FusedParams f;
        kidx = 8 * l; cur = 0;
        prev_splits = l > 0 ? 4 : 0;
        // K1: [norm_out of layer l-1] + LN_ff1 -> W1 -> SiLU
        memset(&f, 0, sizeof(f));
        f.pro = PRO_LN; f.x_in = X[cur]; f.x_out = X[cur ^ 1]; f.part = e->part; f.part_splits = prev_splits; f.scale = 0.5f;
        if (l > 0) { f.lno_w = e->L[l - 1].ln_out_w; f.lno_b = e->L[l - 1].ln_out_b; }
        f.ln_w = L.ln_ff1_w; f.ln_b = L.ln_ff1_b;
        f.g.W = L.ff1_w1; f.g.M = M; f.g.N = FF; f.g.K = D; f.g.splits = 1; f.g.epi = EPI_SILU_ACT; f.g.out_act = e->hbuf; f.g.ldo_act = FF;
        launch(f, "k_fused_ln_gemm", wbytes(FF, D), 2.0 * M * FF * D);
        cur ^= 1;
        // K2: W2 (split-K 4) -> partials
        memset(&f, 0, sizeof(f));
        f.pro = PRO_PLAIN; f.g.A = e->hbuf; f.g.lda = FF; f.g.W = L.ff1_w2; f.g.M = M; f.g.N = D; f.g.K = FF; f.g.splits = 4;
        f.g.epi = EPI_PART_F32; f.g.out_f32 = e->part; f.g.ldo = D;
        launch(f, "k_fused_plain_gemm", wbytes(D, FF), 2.0 * M * D * FF);
        // K3: x += 0.5 * FFN1 ; LN_att -> QKV (K/V straight into the rings)
        memset(&f, 0, sizeof(f));
        f.pro = PRO_LN; f.x_in = X[cur]; f.x_out = X[cur ^ 1]; f.part = e->part; f.part_splits = 4; f.scale = 0.5f;
        f.ln_w = L.ln_att_w; f.ln_b = L.ln_att_b;
        f.g.W = L.wqkv; f.g.M = M; f.g.N = 3 * D; f.g.K = D; f.g.splits = 1; f.g.epi = EPI_QKV; f.g.q_out = e->q;
        f.g.kv_pool = e->kv_pool[l]; f.g.kv_slot_stride = (int64_t)2 * KVC * D; f.g.rows = rows; f.g.T = TS;
        launch(f, "k_fused_ln_gemm", wbytes(3 * D, D), 2.0 * M * 3 * D * D);
        cur ^= 1;
        // K4: attention -> out projection.  M <= 2: fused (one head per blockIdx.y recomputes the tiny attention,
        // split-K over the 8 heads).  Larger M: the redundancy (64 workgroups per head) stops paying, so attention is
        // its own launch (one workgroup per (head, stream)) followed by the plain weight-streaming GEMM.
        constexpr int fuse_max_m = FUSE_MAX_M;
        int wo_splits = NH;
        if (M <= fuse_max_m) {
            memset(&f, 0, sizeof(f));
            f.pro = PRO_ATTN; f.at.q = e->q; f.at.kv_pool = e->kv_pool[l]; f.at.kv_slot_stride = (int64_t)2 * KVC * D; f.at.act_bf16 = 1;
            f.at.posproj = L.posproj[T]; f.at.bias_u = L.bias_u; f.at.bias_v = L.bias_v; f.at.rows = rows; f.at.B = B; f.at.T = T; f.at.TS = TS;
            f.g.W = L.wo; f.g.M = M; f.g.N = D; f.g.K = D; f.g.splits = NH; f.g.epi = EPI_PART_F32; f.g.out_f32 = e->part; f.g.ldo = D;
            launch(f, "k_fused_attn_gemm", wbytes(D, D) + (double)B * (3.0 * (LCTX + T)) * D * 2, 2.0 * M * D * D);
        } else {
            AttnParams ap;
            memset(&ap, 0, sizeof(ap));
            ap.q = e->q; ap.kv_pool = e->kv_pool[l]; ap.kv_slot_stride = (int64_t)2 * KVC * D; ap.act_bf16 = 1;
            ap.posproj = L.posproj[T]; ap.bias_u = L.bias_u; ap.bias_v = L.bias_v; ap.rows = rows; ap.B = B; ap.T = T; ap.TS = TS;
            ap.ctx_out = e->ctx;
            if (kidx >= k0 && kidx < k1) { ProfScope ps(e, "k_attention", (double)B * (3.0 * (LCTX + T)) * D * 2, 2.0 * M * (LCTX + T) * D * 3); launch_attention(ap, st); }   // rides with the launch that consumes it
            wo_splits = 4;
            memset(&f, 0, sizeof(f));
            f.pro = PRO_PLAIN; f.g.A = e->ctx; f.g.lda = D; f.g.W = L.wo; f.g.M = M; f.g.N = D; f.g.K = D; f.g.splits = 4;
            f.g.epi = EPI_PART_F32; f.g.out_f32 = e->part; f.g.ldo = D;
            launch(f, "k_fused_plain_gemm", wbytes(D, D), 2.0 * M * D * D);
        }
        // K5: x += attn ; LN_conv -> pointwise conv 1 -> GLU
        memset(&f, 0, sizeof(f));
        f.pro = PRO_LN; f.x_in = X[cur]; f.x_out = X[cur ^ 1]; f.part = e->part; f.part_splits = wo_splits; f.scale = 1.0f;
        f.ln_w = L.ln_conv_w; f.ln_b = L.ln_conv_b;
        f.g.W = L.pw1; f.g.M = M; f.g.N = 2 * D; f.g.K = D; f.g.splits = 1; f.g.epi = EPI_GLU; f.g.out_f32 = e->glu; f.g.ldo = D;
        launch(f, "k_fused_ln_gemm", wbytes(2 * D, D), 2.0 * M * 2 * D * D);
        cur ^= 1;
        // K6: cached depthwise conv + LN + SiLU -> pointwise conv 2; same rule as K4
        const int pw2_splits = 4;
        if (M <= fuse_max_m) {
            memset(&f, 0, sizeof(f));
            f.pro = PRO_DWCONV; f.cv.glu = e->glu; f.cv.cc_pool = e->cc_pool[l]; f.cv.cc_slot_stride = (int64_t)2 * (ks - 1) * D;
            f.cv.dw = L.dw; f.cv.ln_w = L.cln_w; f.cv.ln_b = L.cln_b; f.cv.rows = rows; f.cv.B = B; f.cv.T = TS; f.cv.ks = ks;
            f.g.W = L.pw2; f.g.M = M; f.g.N = D; f.g.K = D; f.g.splits = pw2_splits; f.g.epi = EPI_PART_F32; f.g.out_f32 = e->part; f.g.ldo = D;
            launch(f, "k_fused_dwconv_gemm", wbytes(D, D), 2.0 * M * D * D);
        } else {
            ConvParams cp;
            memset(&cp, 0, sizeof(cp));
            cp.glu = e->glu; cp.cc_pool = e->cc_pool[l]; cp.cc_slot_stride = (int64_t)2 * (ks - 1) * D;
            cp.dw = L.dw; cp.ln_w = L.cln_w; cp.ln_b = L.cln_b; cp.rows = rows; cp.B = B; cp.T = TS; cp.ks = ks;
            cp.c_out = e->cbuf; cp.act_bf16 = 1;
            if (kidx >= k0 && kidx < k1) { ProfScope ps(e, "k_dwconv", (double)M * D * 6 + (double)B * 2 * (ks - 1) * D * 4, 2.0 * M * D * ks); launch_dwconv(cp, st); }
            memset(&f, 0, sizeof(f));
            f.pro = PRO_PLAIN; f.g.A = e->cbuf; f.g.lda = D; f.g.W = L.pw2; f.g.M = M; f.g.N = D; f.g.K = D; f.g.splits = 4;
            f.g.epi = EPI_PART_F32; f.g.out_f32 = e->part; f.g.ldo = D;
            launch(f, "k_fused_plain_gemm", wbytes(D, D), 2.0 * M * D * D);
        }
        // K7: x += conv ; LN_ff2 -> W1 -> SiLU
        memset(&f, 0, sizeof(f));
        f.pro = PRO_LN; f.x_in = X[cur]; f.x_out = X[cur ^ 1]; f.part = e->part; f.part_splits = pw2_splits; f.scale = 1.0f;
        f.ln_w = L.ln_ff2_w; f.ln_b = L.ln_ff2_b;
        f.g.W = L.ff2_w1; f.g.M = M; f.g.N = FF; f.g.K = D; f.g.splits = 1; f.g.epi = EPI_SILU_ACT; f.g.out_act = e->hbuf; f.g.ldo_act = FF;
        launch(f, "k_fused_ln_gemm", wbytes(FF, D), 2.0 * M * FF * D);
        cur ^= 1;
        // K8: W2 (split-K 4) -> partials, consumed by the next layer's K1 (or the final k_post)
        memset(&f, 0, sizeof(f));
        f.pro = PRO_PLAIN; f.g.A = e->hbuf; f.g.lda = FF; f.g.W = L.ff2_w2; f.g.M = M; f.g.N = D; f.g.K = FF; f.g.splits = 4;
        f.g.epi = EPI_PART_F32; f.g.out_f32 = e->part; f.g.ldo = D;
        launch(f, "k_fused_plain_gemm", wbytes(D, FF), 2.0 * M * D * FF);
        prev_splits = 4;
    }
    if (k1 < 8 * nL || rec) return 0;          // the next piece's first kernel picks the intermediates up
    // x = norm_out(x + 0.5 * FFN2) of the last layer (cur is back at X[0] = e->x: 4 flips per layer)
    cur = 0;
    PostParams q;
    memset(&q, 0, sizeof(q));
    q.x = X[cur]; q.M = M; q.part = e->part; q.splits = 4; q.scale = 0.5f; q.ln_out = 1;
    q.ln1_w = e->L[nL - 1].ln_out_w; q.ln1_b = e->L[nL - 1].ln_out_b;
    if (X[cur] != e->x) q.copy_out = e->x;
    ProfScope ps(e, "k_post", (double)M * D * 24);
    launch_post(q, st);
    return 0;
}

// enqueue one chunk step up to (and including) the joint's encoder projection: no host syncs, no
// host state changes -- capturable into a hipGraph.  tap_slots != null only in debug mode.
// G > 1: G consecutive chunks of every stream in one launch sequence (rows of a stream are (chunk, frame)-major;
// vrows has one descriptor per (stream, chunk) for the subsampling stage).  Only the fused small-M path does this.
// seg / nseg: piece `seg` of `nseg` of the encoder (pipelined steps capture every piece into its own graph): piece k covers
// layers [L k / nseg, L (k + 1) / nseg); piece 0 starts with the subsampling, the last piece ends with prompt fusion and
// joint.enc.  nseg = 1: the whole encoder.
// part: 0 = the piece as described; 1 = the front end only (subsampling: no layers, no tail); 2 = the tail only (prompt fusion, joint.enc)
static int enqueue_encoder(nasr_engine *e, const RowDesc *rows, const RowDesc *vrows, const int *tap_slots, int B, int T, int R, int G = 1, int seg = 0,
                           int nseg = 1, int part = 0) {
    const int Bs = B * G;                      // subsampling batch: one entry per (stream, chunk)
    const int M = Bs * T;
    const int chunk_mel = PRE_CACHE + 8 * (1 + R);
    hipStream_t st = e->st;
    const int act = e->bf16 ? 1 : 0;
    const int nLayers = e->hp.n_layers;
    // the first piece also carries the front end and the subsampling: with two pieces the boundary sits one layer early
    // (11 + 13 layers; measured against 12 + 12 and 10 + 14: batch 1 0.642 / 0.653 / 0.669 ms, 64 streams x R = 13 2.948 / 2.961 / 3.019);
    // with three it is 7 + 9 + 8 (batch 1: 0.500 ms; 8 + 8 + 8 0.522, 7 + 8 + 9 0.511, 7 + 10 + 7 0.526, 6 + 10 + 8 0.527)
    const int shift = nseg == 2 && nLayers >= 8 ? 1 : 0;
    auto bound = [&](int k) {
        if (k <= 0) return 0;
        if (k >= nseg) return nLayers;
        if (nseg == 3 && nLayers >= 6) return k == 1 ? nLayers * 7 / 24 : nLayers * 16 / 24;
        // four pieces: 6 + 7 + 7 + 4 -- the last lane also runs the decode graphs (64 streams x 80 ms: 0.92 ms per step; 6 + 6 + 7 + 5
        // 0.95, 6 + 6 + 6 + 6 1.02, 7 + 6 + 6 + 5 1.00, 6 + 7 + 8 + 3 1.00; 64 x 1.12 s 2.67 / 2.68 / - / 2.77 / 2.72)
        if (nseg == 4 && nLayers >= 8) return k == 1 ? nLayers * 6 / 24 : k == 2 ? nLayers * 13 / 24 : nLayers * 20 / 24;
        return std::max(1, nLayers * k / nseg - shift);
    };
    const int l0 = bound(seg), l1 = bound(seg + 1);
    const bool front = part == 0 ? seg == 0 : part == 1, tail = part == 0 ? seg == nseg - 1 : part == 2;
    GemmParams g;

    // debug taps are indexed by slot: [slot][TMAX][1024] (+ layers)
    auto tap_copy = [&](float *tap_base, size_t per_slot, size_t layer_off) -> int {
        for (int b = 0; b < B; b++)
            HIPCHK(hipMemcpyAsync(tap_base + (size_t)tap_slots[b] * per_slot + layer_off, e->x + (size_t)b * T * D,
                                  (size_t)T * D * 4, hipMemcpyDeviceToDevice, st));
        return 0;
    };
    // ---- a-2 subsampling ------------------------------------------------------------------
    const int H1 = chunk_mel / 2 + 1, W1 = 65, H2 = H1 / 2 + 1, W2 = 33, H3 = H2 / 2 + 1, W3 = 17;
    if (front) {
    {
        ProfScope ps(e, "k_sub_conv0_dw", (double)Bs * (chunk_mel * NMEL * 4 + H2 * W2 * SUBC * (act ? 2 : 4)), 2.0 * Bs * H2 * W2 * SUBC * 90);
        launch_sub_conv0_dw(vrows, Bs, chunk_mel, e->mel_ring, e->w0t, e->b0, e->w2t, e->b2, e->sub_b, act, H1, W1, st);
    }
    memset(&g, 0, sizeof(g));
    g.A = e->sub_b; g.W = e->w3; g.M = Bs * H2 * W2; g.N = SUBC; g.K = SUBC; g.lda = SUBC; g.splits = 1;
    g.epi = EPI_BIAS_RELU_F32; g.out_f32 = e->sub_a; g.ldo = SUBC; g.bias = e->b3;
    run_gemm(e, g, false, "sub_pw3");
    {
        ProfScope ps(e, "k_sub_dw", (double)B * H2 * W2 * SUBC * 4, 2.0 * B * H3 * W3 * SUBC * 9);
        launch_sub_dw(e->sub_a, Bs, H2, W2, e->w5t, e->b5, e->sub_b, act, st);
    }
    memset(&g, 0, sizeof(g));
    g.A = e->sub_b; g.W = e->w6; g.M = Bs * H3 * W3; g.N = SUBC; g.K = SUBC; g.lda = SUBC; g.splits = 1;
    g.epi = EPI_BIAS_RELU_ACT; g.out_act = e->sub_a; g.ldo_act = SUBC; g.bias = e->b6;
    run_gemm(e, g, false, "sub_pw6");
    // out projection on the last T of the T+2 frames (drop 2: src/nemo-stream.cpp:154-162,:303)
    memset(&g, 0, sizeof(g));
    g.A = e->sub_a; g.W = e->sub_out_w; g.M = M; g.N = D; g.K = SUBFLAT; g.lda = SUBFLAT; g.splits = 1;
    g.rows_per_batch = T; g.batch_stride = H3 * SUBFLAT; g.row_offset = DROP_EXTRA;
    g.epi = EPI_BIAS_F32; g.out_f32 = e->x; g.ldo = D; g.bias = e->sub_out_b;
    run_gemm(e, g, false, "sub_out");
    if (e->debug && tap_copy(e->tap_sub, (size_t)TMAX * D, 0)) return -1;
    }   // front

    // Up to 4 rows the 8-launch fused layer wins; above, its per-workgroup prologues (every workgroup redoes the
    // LayerNorm of all rows) cost more than the 6 extra launches of the unfused layer (measured at R = 0:
    // 8 rows 2.13 vs 1.89 ms, 16 rows 2.78 vs 1.96 ms per step).
    constexpr int fused_rows = 4;
    const bool fused = e->bf16 && e->opt_fused && !e->debug && M <= fused_rows;
    const int TS = G * T;                      // rows per stream in this launch
    if (G > 1 && e->debug) return fail("internal: multi-chunk steps are not available in debug mode");
    if (part != 0) {
        // front end or tail only: the layers are launched by the caller (grouped pipeline)
    } else if (fused) {
        // pieces of the fused path can be cut at any launch: the first one also carries the front end (about 8 launches' worth
        // of time), the last one the joint's encoder projection.  Three pieces: 57 + 69 + 66 launches of 192.  Worth little:
        // batch 1 0.492-0.497 ms per step against 0.498-0.500 at 7 + 9 + 8 layers (56 + 72 + 64) -- with three lanes the step is
        // no longer bound by its longest lane (tests/micro/stamps_timeline.py: the kernels of the three chains mostly alternate
        // instead of overlapping: 0 / 1 / 2 / 3 kernels in flight 29 / 40 / 20 / 10 % of the time).
        auto bound8 = [&](int k) {
            if (k <= 0) return 0;
            if (k >= nseg) return 8 * nLayers;
            if (nseg == 3 && nLayers >= 6) return (k == 1 ? 57 : 126) * nLayers / 24;
            // four pieces: 48 + 56 + 60 + 28 launches (batch 1: 0.426 ms per step; 48 + 56 + 56 + 32 0.437, 50 + 56 + 56 + 30 0.433,
            // 48 + 54 + 62 + 28 0.434, 48 + 56 + 64 + 24 0.440; three lanes 0.454)
            if (nseg == 4 && nLayers >= 8) return (k == 1 ? 48 : k == 2 ? 104 : 164) * nLayers / 24;
            return 8 * bound(k);
        };
        // The launches of a layer that touch PER-STREAM state shared by all steps -- K3 / K4 (the layer's K/V ring) and K6 (its conv
        // cache) -- must run on the SAME lane whatever the step's shape: steps of one stream follow each other through a layer in lane
        // order only.  The unfused path (more than four rows) cuts at whole layers, bound(k); a fused cut may therefore only move
        // launches that touch nothing but the step's own workspace across that boundary: K7 / K8 of the layer before it (FFN2) or
        // K1 / K2 of the layer after it (FFN1), i.e. it must lie in [8 bound(k) - 2, 8 bound(k) + 2].  Round 2 shipped 164 for the third cut
        // of the 24-layer model (layer 20's K3 / K4 on lane 2 for one-to-four-row steps, on lane 3 for larger ones): a stream whose steps
        // alternate between the two forms while both are in flight could read or write that layer's ring out of order -- found in
        // round 3 by the soak test on 8 layers (cuts 16 | 34 | 54), never seen at 24; the cut is 162 now.
        auto snap8 = [&](int k, int b) {
            if (k <= 0 || k >= nseg) return b;
            const int lb = 8 * bound(k);
            return std::min(std::max(b, lb - 2), lb + 2);
        };
        if (run_layers_fused(e, rows, B, T, G, snap8(seg, bound8(seg)), snap8(seg + 1, bound8(seg + 1)))) return -1;
    } else {
    // ---- 24 cached conformer layers -----------------------------------------------------------
        if (front) {
            PostParams pp;
            memset(&pp, 0, sizeof(pp));
            pp.x = e->x; pp.M = M; pp.ln2_w = e->L[0].ln_ff1_w; pp.ln2_b = e->L[0].ln_ff1_b; pp.a_out = e->a; pp.act_bf16 = act;
            { ProfScope ps(e, "k_post", (double)M * D * (4 + e->esz)); launch_post(pp, st); }
        }

        const int nL = e->hp.n_layers, ks = e->hp.kernel_size;
        for (int l = l0; l < l1; l++) {
            LayerW &L = e->L[l];
            auto ffn = [&](void *w1, void *w2, const float *nln_w, const float *nln_b, bool last) {
                GemmParams a;
                memset(&a, 0, sizeof(a));
                a.A = e->a; a.W = w1; a.M = M; a.N = FF; a.K = D; a.lda = D; a.splits = 1;
                a.epi = EPI_SILU_ACT; a.out_act = e->hbuf; a.ldo_act = FF;
                run_gemm(e, a, false, "ffn_w1");
                memset(&a, 0, sizeof(a));
                a.A = e->hbuf; a.W = w2; a.M = M; a.N = D; a.K = FF; a.lda = FF; a.splits = pick_splits(e, M, D, FF);
                a.epi = EPI_PART_F32; a.out_f32 = e->part; a.ldo = D;
                run_gemm(e, a, false, "ffn_w2");
                PostParams q;
                memset(&q, 0, sizeof(q));
                q.x = e->x; q.M = M; q.part = e->part; q.splits = a.splits; q.scale = 0.5f;   // :633-634
                q.a_out = e->a; q.act_bf16 = act;
                if (last) { q.ln_out = 1; q.ln1_w = L.ln_out_w; q.ln1_b = L.ln_out_b; }       // :687
                q.ln2_w = nln_w; q.ln2_b = nln_b;
                ProfScope ps(e, "k_post", (double)M * D * (8 + 4 * a.splits + e->esz));
                launch_post(q, st);
            };
            // 1. FFN1 (:631-634) -> a = LN_att(x)
            ffn(L.ff1_w1, L.ff1_w2, L.ln_att_w, L.ln_att_b, false);
            // 2. attention (:637-643)
            memset(&g, 0, sizeof(g));
            g.A = e->a; g.W = L.wqkv; g.M = M; g.N = 3 * D; g.K = D; g.lda = D; g.splits = 1;
            g.epi = EPI_QKV; g.q_out = e->q; g.kv_pool = e->kv_pool[l]; g.kv_slot_stride = (int64_t)2 * KVC * D;
            g.rows = rows; g.T = TS;
            run_gemm(e, g, false, "qkv");
            {
                AttnParams ap;
                memset(&ap, 0, sizeof(ap));
                ap.q = e->q; ap.kv_pool = e->kv_pool[l]; ap.kv_slot_stride = (int64_t)2 * KVC * D; ap.act_bf16 = act;
                ap.posproj = L.posproj[T]; ap.bias_u = L.bias_u; ap.bias_v = L.bias_v; ap.rows = rows; ap.B = B; ap.T = T; ap.TS = TS;
                ap.ctx_out = e->ctx;
                const int KV = LCTX + T;
                ProfScope ps(e, "k_attention", (double)B * (2.0 * KV + KV + T - 1) * D * e->esz, 2.0 * B * T * KV * D * 3);
                launch_attention(ap, st);
            }
            memset(&g, 0, sizeof(g));
            g.A = e->ctx; g.W = L.wo; g.M = M; g.N = D; g.K = D; g.lda = D; g.splits = pick_splits(e, M, D, D);
            g.epi = EPI_PART_F32; g.out_f32 = e->part; g.ldo = D;
            run_gemm(e, g, false, "attn_out");
            {
                PostParams q;
                memset(&q, 0, sizeof(q));
                q.x = e->x; q.M = M; q.part = e->part; q.splits = g.splits; q.scale = 1.0f;
                q.ln2_w = L.ln_conv_w; q.ln2_b = L.ln_conv_b; q.a_out = e->a; q.act_bf16 = act;
                ProfScope ps(e, "k_post", (double)M * D * (8 + 4 * g.splits + e->esz));
                launch_post(q, st);
            }
            // 3. conv module (:646-679)
            memset(&g, 0, sizeof(g));
            g.A = e->a; g.W = L.pw1; g.M = M; g.N = 2 * D; g.K = D; g.lda = D; g.splits = 1;
            g.epi = EPI_GLU; g.out_f32 = e->glu; g.ldo = D;
            run_gemm(e, g, false, "pw1");
            {
                ConvParams cp;
                memset(&cp, 0, sizeof(cp));
                cp.glu = e->glu; cp.cc_pool = e->cc_pool[l]; cp.cc_slot_stride = (int64_t)2 * (ks - 1) * D;
                cp.dw = L.dw; cp.ln_w = L.cln_w; cp.ln_b = L.cln_b; cp.rows = rows; cp.B = B; cp.T = TS; cp.ks = ks;
                cp.c_out = e->cbuf; cp.act_bf16 = act;
                ProfScope ps(e, "k_dwconv", (double)M * D * (4 + e->esz) + (double)B * 2 * (ks - 1) * D * 4, 2.0 * M * D * ks);
                launch_dwconv(cp, st);
            }
            memset(&g, 0, sizeof(g));
            g.A = e->cbuf; g.W = L.pw2; g.M = M; g.N = D; g.K = D; g.lda = D; g.splits = pick_splits(e, M, D, D);
            g.epi = EPI_PART_F32; g.out_f32 = e->part; g.ldo = D;
            run_gemm(e, g, false, "pw2");
            {
                PostParams q;
                memset(&q, 0, sizeof(q));
                q.x = e->x; q.M = M; q.part = e->part; q.splits = g.splits; q.scale = 1.0f;
                q.ln2_w = L.ln_ff2_w; q.ln2_b = L.ln_ff2_b; q.a_out = e->a; q.act_bf16 = act;
                ProfScope ps(e, "k_post", (double)M * D * (8 + 4 * g.splits + e->esz));
                launch_post(q, st);
            }
            // 4. FFN2 (:682-685) + norm_out (:687); then the next layer's first LayerNorm
            const bool has_next = l + 1 < nL;
            ffn(L.ff2_w1, L.ff2_w2, has_next ? e->L[l + 1].ln_ff1_w : nullptr, has_next ? e->L[l + 1].ln_ff1_b : nullptr, true);
            if (e->debug && tap_copy(e->tap_layers, (size_t)nL * TMAX * D, (size_t)l * TMAX * D)) return -1;
        }
    }
    if (!tail) return 0;
    // ---- a-11 prompt fusion (multilingual only, src/nemo-ggml.cpp:1087-1105) ---------------------
    if (e->hp.num_prompts > 0) {
        memset(&g, 0, sizeof(g));
        g.A = e->x; g.W = e->pk1a; g.M = M; g.N = 2048; g.K = D; g.lda = D; g.splits = 1;
        g.epi = EPI_BIAS_F32; g.out_f32 = e->hfuse; g.ldo = 2048; g.bias = e->pk1_b; g.f32_fma_tile = e->opt_f32_mfma ? 0 : 1;
        { ProfScope ps(e, "k_gemm_f32", gemm_bytes(e, M, 2048, D, 4), 2.0 * M * 2048 * D); launch_gemm_f32(g, st); }
        launch_prompt_add_relu(e->hfuse, e->pk1p, rows, M, G * T, e->hp.num_prompts, st);
        memset(&g, 0, sizeof(g));
        g.A = e->hfuse; g.W = e->pk2_w; g.M = M; g.N = D; g.K = 2048; g.lda = 2048; g.splits = 1;
        g.epi = EPI_BIAS_F32; g.out_f32 = e->x; g.ldo = D; g.bias = e->pk2_b; g.f32_fma_tile = e->opt_f32_mfma ? 0 : 1;
        { ProfScope ps(e, "k_gemm_f32", gemm_bytes(e, M, D, 2048, 4), 2.0 * M * D * 2048); launch_gemm_f32(g, st); }
    }
    if (e->debug && tap_copy(e->tap_enc, (size_t)TMAX * D, 0)) return -1;

    // ---- a-13 encoder projection of the joint, hoisted out of the symbol loop --------------------
    {
        ProfScope ps(e, "k_encproj", (double)JNT * D * 4 + (double)M * (D + JNT) * 4, 2.0 * M * JNT * D);
        launch_encproj(e->x, e->jenc_w, e->jenc_b, e->encproj, M, D, JNT, st);
    }
    return 0;
}

static void make_dec_params(nasr_engine *e, const RowDesc *rows, int B, int T, DecParams &dp) {
    memset(&dp, 0, sizeof(dp));
    dp.rows = rows; dp.B = B; dp.T = T; dp.ctrl = e->ctrl; dp.h = e->dec_h; dp.c = e->dec_c; dp.encproj = e->encproj;
    dp.embed = e->embed;
    for (int i = 0; i < 2; i++) { dp.w_ih[i] = e->w_ih[i]; dp.w_hh[i] = e->w_hh[i]; dp.b_ih[i] = e->b_ih[i]; dp.b_hh[i] = e->b_hh[i]; }
    dp.pred_w = e->pred_w; dp.pred_b = e->pred_b; dp.out_w = e->out_w; dp.out_b = e->out_b;
    dp.predg = e->predg; dp.key = e->key; dp.n_active = e->n_active; dp.n_dirty = e->n_active + 1; dp.n_rows = e->n_active + 2;
    dp.dlist = e->dlist; dp.rowmap = e->rowmap; dp.tok_ring = e->tok_ring; dp.tok_frame = e->tok_frame;
}

static void enqueue_decode_iters(nasr_engine *e, const DecParams &dp, int B, int n, int &it, hipStream_t st = nullptr) {
    ProfScope ps(e, "k_dec_iter", (double)n * (4.0 * 4 * HID * HID * 4 + (double)JNT * HID * 4 + (double)VOCAB * JNT * 4),
                 (double)n * 2.0 * B * (4.0 * 4 * HID * HID + JNT * HID + VOCAB * JNT));
    for (int k = 0; k < n; k++) launch_decode_iter(dp, it++, st ? st : e->st);
}

// host mirror of the stream manager bookkeeping after a chunk (:1085, :1189-1195)
static void chunk_bookkeeping(nasr_stream *s, int row) {
    const int T = s->T, shift = 8 * T;
    s->valid_len = std::min(s->valid_len + T, LCTX);
    s->kv_head = (s->kv_head + T) % KVC;
    s->cc_par ^= 1;
    s->mel_start = (s->mel_start + shift) & (MEL_RING - 1);
    s->mel_count -= shift;
    s->chunks++;
    s->last_T = T;
    s->last_row = row;
    s->last_ws = 0;                            // pipelined steps overwrite this with their slot
}

static void fill_row_desc(RowDesc &rd, const nasr_stream *s, int n_dec) {
    rd.slot = s->slot; rd.valid_len = s->valid_len; rd.kv_head = s->kv_head;
    rd.mel_start = s->mel_start; rd.cc_par = s->cc_par; rd.n_dec = n_dec;
    rd.prompt = s->prompt; rd.pad = 0;
}

static int run_chunk(nasr_engine *e, const std::vector<nasr_stream *> &rows_s, const std::vector<int> &n_dec) {
    const int B = (int)rows_s.size();
    const int T = rows_s[0]->T, R = rows_s[0]->R;
    std::vector<RowDesc> rd(B);
    std::vector<int> slots(B);
    for (int b = 0; b < B; b++) { fill_row_desc(rd[b], rows_s[b], n_dec[b]); slots[b] = rows_s[b]->slot; }
    const RowDesc *rows;
    if (stage_desc(e, rd, &rows)) return -1;
    hipStream_t st = e->st;
    if (enqueue_encoder(e, rows, rows, e->debug ? slots.data() : nullptr, B, T, R)) return -1;
    // ---- a-12..a-14 greedy decode, device resident -----------------------------------------------
    DecParams dp;
    make_dec_params(e, rows, B, T, dp);
    launch_decode_begin(dp, st);
    int max_dec = 0;
    for (int b = 0; b < B; b++) max_dec = std::max(max_dec, n_dec[b]);
    int it = 0, budget = decode_blind_iterations(max_dec);
    int *h_active = (int *)e->pin;   // first 256 bytes of the pinned arena are reserved for this
    while (max_dec > 0) {
        enqueue_decode_iters(e, dp, B, budget, it);
        HIPCHK(hipMemcpyAsync(h_active, e->n_active, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (*h_active == 0) break;
        if (it > max_dec * MAX_SYMBOLS + 64) return fail("decode did not terminate");
        budget = std::min(2 * budget, 32);      // a burst (up to 10 symbols per frame): double the round, few syncs
    }
    for (int b = 0; b < B; b++) chunk_bookkeeping(rows_s[b], b);
    return 0;
}

// run chunk steps while any of the given streams has a full chunk buffered (:1174)
static int drain_chunks(nasr_engine *e, nasr_stream *const *streams, int B) {
    for (;;) {
        std::vector<nasr_stream *> ready;
        for (int b = 0; b < B; b++)
            if (streams[b]->mel_count >= PRE_CACHE + 8 * streams[b]->T) ready.push_back(streams[b]);
        if (ready.empty()) return 0;
        std::vector<int> nd(ready.size(), ready[0]->T);
        if (run_chunk(e, ready, nd)) return -1;
    }
}

static int validate_batch(nasr_engine *e, nasr_stream *const *streams, int B) {
    if (!e || !streams || B < 1) return fail("null engine/streams or B < 1");
    if (B > e->max_streams) return fail("B=%d exceeds max_streams=%d", B, e->max_streams);
    for (int b = 0; b < B; b++) {
        if (!streams[b] || streams[b]->e != e) return fail("stream %d does not belong to this engine", b);
        if (streams[b]->R != streams[0]->R) return fail("all streams of one call must share right_context");
        for (int c = 0; c < b; c++)
            if (streams[c] == streams[b]) return fail("stream %d passed twice", b);
    }
    return 0;
}

// gather new tokens of the B streams
__global__ void k_collect(const int *slots, const int *tok_read, int B, const DecCtrl *ctrl, const int *tok_ring, int *out, int stride,
                          const int *n_active) {
    const int b = blockIdx.x;
    if (b == 0 && threadIdx.x == 0 && n_active) out[(size_t)B * (1 + stride)] = *n_active;   // rides along in the same D2H copy
    const int slot = slots[b];
    const int n_tok = ctrl[slot].n_tok, rd = tok_read[b];
    const int n_new = n_tok - rd;
    if (threadIdx.x == 0) out[(size_t)b * (1 + stride)] = n_new;
    for (int i = threadIdx.x; i < n_new && i < stride; i += blockDim.x)
        out[(size_t)b * (1 + stride) + 1 + i] = tok_ring[(size_t)slot * TOK_CAP + ((rd + i) & (TOK_CAP - 1))];
}

// Tokens gathered from the device go to the stream's host queue; every token-returning entry point ends with deliver().
static int consume_collect(nasr_engine *e, const int *host, nasr_stream *const *streams, int B) {
    for (int b = 0; b < B; b++) {
        const int *rec = &host[(size_t)b * (1 + COLLECT_STRIDE)];
        const int n_new = rec[0];
        if (n_new < 0 || n_new > TOK_CAP) return fail("token ring overrun on stream %d (%d new tokens)", b, n_new);
        std::vector<int32_t> &q = streams[b]->tok_queue;
        if (n_new <= COLLECT_STRIDE) {
            q.insert(q.end(), rec + 1, rec + 1 + n_new);
        } else {   // rare long push: fetch straight from the ring
            std::vector<int> ring(TOK_CAP);
            HIPCHK(hipMemcpy(ring.data(), e->tok_ring + (size_t)streams[b]->slot * TOK_CAP, TOK_CAP * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < n_new; i++) q.push_back(ring[(streams[b]->tok_read + i) & (TOK_CAP - 1)]);
        }
        streams[b]->tok_read += n_new;
    }
    return 0;
}

// hands the queued tokens of the B streams to the caller: at most tokens_cap[b] of them, n_tokens[b] = the number written.
// What does not fit STAYS queued and comes out of the next step / collect / finalize call (nothing is ever dropped); a
// caller that passes no buffer at all (null tokens_out) discards its tokens by contract and gets the count it discarded.
static void deliver(nasr_stream *const *streams, int B, int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens) {
    for (int b = 0; b < B; b++) {
        std::vector<int32_t> &q = streams[b]->tok_queue;
        if (!tokens_out || !tokens_out[b] || !tokens_cap) {
            if (n_tokens) n_tokens[b] = (int32_t)q.size();
            q.clear();
            continue;
        }
        const int n_copy = std::min((int)q.size(), std::max(tokens_cap[b], 0));
        for (int i = 0; i < n_copy; i++) tokens_out[b][i] = q[(size_t)i];
        if (n_tokens) n_tokens[b] = n_copy;
        q.erase(q.begin(), q.begin() + n_copy);
    }
}

static int pipe_drain(nasr_engine *e);

static int collect_tokens(nasr_engine *e, nasr_stream *const *streams, int B, int32_t *const *tokens_out,
                          const int32_t *tokens_cap, int32_t *n_tokens) {
    if (pipe_drain(e)) return -1;
    std::vector<int> meta(2 * (size_t)B);
    for (int b = 0; b < B; b++) { meta[b] = streams[b]->slot; meta[B + b] = streams[b]->tok_read; }
    const int *dmeta;
    if (stage_desc(e, meta, &dmeta)) return -1;
    hipLaunchKernelGGL(k_collect, dim3(B), dim3(64), 0, e->st, dmeta, dmeta + B, B, e->ctrl, e->tok_ring, e->collect_dev, COLLECT_STRIDE, (const int *)nullptr);
    std::vector<int> host((size_t)B * (1 + COLLECT_STRIDE));
    HIPCHK(hipMemcpyAsync(host.data(), e->collect_dev, host.size() * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipStreamSynchronize(e->st));
    if (consume_collect(e, host.data(), streams, B)) return -1;
    deliver(streams, B, tokens_out, tokens_cap, n_tokens);
    return 0;
}

static int ensure_debug_buffers(nasr_engine *e) {
    if (e->tap_sub) return 0;
    const size_t S = (size_t)e->max_streams;
    e->tap_mel_cap = 128;
    HIPCHK(hipMalloc((void **)&e->tap_mel, S * e->tap_mel_cap * NMEL * 4));
    HIPCHK(hipMalloc((void **)&e->tap_sub, S * TMAX * D * 4));
    HIPCHK(hipMalloc((void **)&e->tap_layers, S * (size_t)e->hp.n_layers * TMAX * D * 4));
    HIPCHK(hipMalloc((void **)&e->tap_enc, S * TMAX * D * 4));
    return 0;
}

extern "C" int nasr_engine_set_option(nasr_engine *e, const char *key, int value) {
    if (!e || !key) return fail("null argument");
    if (!strcmp(key, "fused")) e->opt_fused = value != 0;
    else if (!strcmp(key, "graph")) e->opt_graph = value != 0;
    else if (!strcmp(key, "graph_cache")) { if (value < 1) return fail("graph_cache must be >= 1"); e->opt_graph_cache = value; }
    else if (!strcmp(key, "multichunk")) e->opt_multichunk = value != 0;
    else if (!strcmp(key, "persistent_gemm")) e->opt_persist_gemm = value != 0;      // like "fused": set before the first step
    else if (!strcmp(key, "gemm_cores")) { if (value < -1 || value > 1) return fail("gemm_cores must be -1, 0 or 1"); e->opt_gemm_cores = value; }
    else if (!strcmp(key, "decode_graph_iterations")) { if (value < 1) return fail("decode_graph_iterations must be >= 1"); e->opt_decode_graph_iters = value; }
    else if (!strcmp(key, "decode_lane")) {
        ApiGuard api_guard;
        HIPCHK(hipSetDevice(e->device));
        if (pipe_drain(e)) return -1;
        e->opt_decode_lane = value != 0;
        release_lanes(e);                       // the lanes are picked again on the next pipelined step
    }
    else if (!strcmp(key, "f32_mfma")) e->opt_f32_mfma = value != 0;      // 0: f32 GEMMs above four rows on the FMA tile kernel (round 3's path); like "fused", set before the first step
    else if (!strcmp(key, "pipeline")) {
        ApiGuard api_guard;
        HIPCHK(hipSetDevice(e->device));
        if (pipe_drain(e)) return -1;
        if ((value < 0 || value > nasr_engine::MAXSEG) && value != nasr_engine::GP_S) return fail("pipeline must be 0 .. %d, or %d (grouped)", (int)nasr_engine::MAXSEG, (int)nasr_engine::GP_S);
        e->opt_pipeline = value;
    }
    else if (!strcmp(key, "lanes")) {
        // give hardware queues back: another GPU client of the process (the diarization side-car) whose stream is created AFTER
        // this call lands on a queue this engine no longer uses (the runtime hands a new stream the least-used queue)
        ApiGuard api_guard;
        HIPCHK(hipSetDevice(e->device));
        if (value < 1 || value > nasr_engine::MAXSEG) return fail("lanes must be 1 .. %d", (int)nasr_engine::MAXSEG);
        if (pipe_drain(e)) return -1;
        e->max_lanes = value;
        release_lanes(e);
    }
    else return fail("unknown option '%s'", key);
    return 0;
}

extern "C" int nasr_engine_set_debug(nasr_engine *e, int enable) {
    ApiGuard api_guard;
    if (!e) return fail("null engine");
    HIPCHK(hipSetDevice(e->device));
    if (pipe_drain(e)) return -1;
    if (enable && ensure_debug_buffers(e)) return -1;
    e->debug = enable != 0;
    return 0;
}

// ---- hipGraph replay of the steady-state step ----------------------------------------------------
// Eligible when every stream of the call receives one sub-push that completes exactly one chunk
// (the normal streaming cadence: 1280*(1+R) samples per push).  The launch sequence is then fixed
// for a given (B, T): descriptors live at fixed addresses and are refreshed by memcpy nodes.
static int max_frames_per_push(int TS) { return 8 * TS + 16; }  // TS = frames of encoder output the push completes (+ what a first push leaves over)

// Descriptors of a graph step, packed so that ONE memcpy node refreshes them: [RowDesc B][PcmDesc B][meta 2B][RowDesc B*G]
struct GraphDescLayout { size_t rows, pcm, meta, vrows, total; };
static GraphDescLayout graph_desc_layout(int B, int G) {
    GraphDescLayout l;
    l.rows = 0;
    l.pcm = l.rows + (size_t)B * sizeof(RowDesc);
    l.meta = l.pcm + (size_t)B * sizeof(PcmDesc);
    l.vrows = (l.meta + (size_t)2 * B * sizeof(int) + 15) & ~(size_t)15;
    l.total = l.vrows + (G > 1 ? (size_t)B * G * sizeof(RowDesc) : 0);
    return l;
}

static int build_step_graph(nasr_engine *e, int B, int T, int R, int G, hipGraphExec_t *out) {
    hipStream_t st = e->st;
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    auto body = [&]() -> int {
        const GraphDescLayout L = graph_desc_layout(B, G);
        const RowDesc *g_rows = (const RowDesc *)(e->g_desc + L.rows), *g_vrows = (const RowDesc *)(e->g_desc + L.vrows);
        const PcmDesc *g_pcm = (const PcmDesc *)(e->g_desc + L.pcm);
        const int *g_meta = (const int *)(e->g_desc + L.meta);
        HIPCHK(hipMemcpyAsync(e->g_desc, e->gh, L.total, hipMemcpyHostToDevice, st));
        MelParams mp;
        memset(&mp, 0, sizeof(mp));
        mp.desc = g_pcm; mp.B = B; mp.max_frames = max_frames_per_push(T * G); mp.abuf = e->abuf; mp.last_sample = e->last_sample;
        mp.mel_ring = e->mel_ring; mp.window = e->window; mp.fbT = e->fbT; mp.fb_band = e->fb_band; mp.cos_t = e->cos_t; mp.sin_t = e->sin_t;
        launch_mel(mp, mp.max_frames * HOP + NFFT, st);
        if (enqueue_encoder(e, g_rows, G > 1 ? g_vrows : g_rows, nullptr, B, T, R, G)) return -1;
        DecParams dp;
        make_dec_params(e, g_rows, B, T * G, dp);
        launch_decode_begin(dp, st);
        int it = 0;
        enqueue_decode_iters(e, dp, B, decode_blind_iterations(T * G), it);
        hipLaunchKernelGGL(k_collect, dim3(B), dim3(64), 0, st, g_meta, g_meta + B, B, e->ctrl, e->tok_ring, e->collect_dev, COLLECT_STRIDE, e->n_active);
        HIPCHK(hipMemcpyAsync(e->gh_collect, e->collect_dev, ((size_t)B * (1 + COLLECT_STRIDE) + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
        return 0;
    };
    rc = body();
    hipError_t ce = hipStreamEndCapture(st, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return -1; }
    if (ce != hipSuccess) return fail("hipStreamEndCapture failed: %s", hipGetErrorString(ce));
    hipError_t ie = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (ie != hipSuccess) return fail("hipGraphInstantiate failed: %s", hipGetErrorString(ie));
    return 0;
}

// ---- pipelined graph steps (engine option "pipeline" = E, 1..4) ----------------------------------------------------------
// A step is a chain of dependent launches (209 at batch 1, ~350 at 64 streams x R = 13) and the chip idles at every link
// (boundary, arrival of the previous kernel's output, pipeline fill, epilogue tail: DESIGN.md section 5).  What fills those
// gaps is another, independent chain -- and consecutive steps of the SAME streams provide one: layer l of step s + 1 needs
// from step s only what its layer l left in the K/V ring and the conv cache.  So the step is cut into E encoder pieces
// (piece k = layers [L k / E, L (k + 1) / E); piece 0 starts with the front end and the subsampling, the last one ends with
// joint.enc) plus the decode, each on its own HIP stream and each one step behind the piece before it:
//   call s:   piece 0 of step s | piece 1 of step s-1 | ... | piece E-1 of step s-E+1 | decode of step s-E
// Stream order keeps a piece behind the same piece of the previous step (layer l of step s + 1 after layer l of step s);
// an event keeps it behind the previous piece of its own step.  E = 1 is "decode beside the next encoder".
// Measured with independent engines on one GPU (tests/micro/lanes_probe.py, round 2): two chains side by side move 1.69x
// (batch 1), 1.60x (64 streams x 80 ms) and 1.28x (64 streams x 1.12 s) the audio of one.
// The price is token latency at this synchronous interface: the call of step s returns the tokens of step s - E;
// finalize / collect / any other entry point first completes what is in flight.  Results are bit-identical to synchronous
// stepping (same kernels, same inputs, same order per stream).  Everything a step in flight owns exists once per slot.
// The decode graph is launched only once its input is ready: parked behind an event wait for the ~1 ms the encoder takes
// it made every boundary of the encoder chain slower (round 1: 1.21 vs 1.07 ms per step at batch 1).
struct HostTimer {          // accumulates wall time of a scope into a double (diagnostics only: NASR_STATS)
    double &acc; std::chrono::steady_clock::time_point t0;
    explicit HostTimer(double &a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

static int pipe_blind_iterations(int frames, int cap) {     // cap: engine option "decode_graph_iterations" (12)
    // off the critical path an idle iteration is free: give one frame its worst case (10 symbols + the closing blank)
    const int worst = frames * MAX_SYMBOLS + 1;
    return std::max(decode_blind_iterations(frames), std::min(worst, cap));
}

// ---- which HIP streams run side by side -------------------------------------------------------------------------------------
// The runtime multiplexes the streams of a process onto a few hardware queues (4 by default, GPU_MAX_HW_QUEUES): two streams on
// one queue are ONE launch chain.  Which queue a new stream gets depends on what the process created before, so the lanes are
// chosen by measurement: candidate streams are created until three more are found that overlap with the engine's stream and
// with each other (two 150 us spin kernels launched back to back take the time of one on different queues, of two on one).  Measured
// (batch 1, MI355X): three encoder lanes + the decode on four queues 0.52 ms per step, the same option with two lanes landing
// on one queue 0.73 ms -- slower than two lanes (0.63 ms).  More than 4 queues is no way out: hardware queues beyond the four
// compute pipes are time-sliced (GPU_MAX_HW_QUEUES=8: 2 ms per step).
__global__ void k_spin(unsigned long long ticks) {           // 100 MHz real-time counter
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
static double spin_us(hipStream_t a, hipStream_t b) {          // wall time of one spin kernel on a (b == nullptr) or one on each
    const unsigned long long ticks = 15000;                   // 150 us
    double best = 1e9;
    for (int rep = 0; rep < 2; rep++) {
        hipStreamSynchronize(a);
        if (b) hipStreamSynchronize(b);
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks);
        if (b) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, ticks);
        hipStreamSynchronize(a);
        if (b) hipStreamSynchronize(b);
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    return best;
}
// side by side: about the time of one spin (~165 us); one queue: two (~315 us).  The threshold is relative to what ONE spin
// takes from this host thread right now, so a slow or busy host does not turn into "everything shares a queue".
static bool streams_overlap(hipStream_t a, hipStream_t b, double alone_us) { return spin_us(a, b) < 1.5 * alone_us; }
static int pick_lanes(nasr_engine *e) {
    std::vector<hipStream_t> chosen{e->st}, rejected;
    const double alone_us = spin_us(e->st, nullptr);
    for (int tries = 0; tries < 12 && (int)chosen.size() < 4; tries++) {
        hipStream_t c = nullptr;
        HIPCHK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        bool ok = true;
        for (hipStream_t s : chosen) ok = ok && streams_overlap(s, c, alone_us);
        (ok ? chosen : rejected).push_back(c);
    }
    for (hipStream_t c : rejected) hipStreamDestroy(c);
    // 4 queues: up to 3 pieces + the decode graphs on the fourth, or 4 pieces with the decode behind the last one; 3 queues: 2 + decode
    // or 3 with the decode behind; ... 1 (GPU_MAX_HW_QUEUES=1): everything on the engine's stream
    if (!e->opt_decode_lane && chosen.size() > 1) { hipStreamDestroy(chosen.back()); chosen.pop_back(); }   // option "decode_lane" = 0: one queue fewer, the decode graphs run behind the last encoder piece
    e->n_lanes = std::max(1, std::min((int)chosen.size(), (int)nasr_engine::MAXSEG));
    e->lane[0] = e->st;
    for (int k = 1; k < e->n_lanes; k++) e->lane[k] = chosen[(size_t)k];
    if (getenv("NASR_STATS")) fprintf(stderr, "nasr: pipelined steps: %d stream(s) side by side (encoder pieces + decode)\n", e->n_lanes);
    return 0;
}

static void release_lanes(nasr_engine *e) {
    if (!e->pipe_ready) return;                                // applied when the lanes are picked
    const int keep = std::min((int)nasr_engine::MAXSEG, std::max(1, e->max_lanes) + 1);      // max_lanes pieces + the decode stream
    for (int k = keep; k < nasr_engine::MAXSEG; k++)
        if (e->lane[k]) { hipStreamSynchronize(e->lane[k]); hipStreamDestroy(e->lane[k]); e->lane[k] = nullptr; }
    e->n_lanes = std::min(e->n_lanes, keep);
}

// streams, events and the buffers of slot p (allocated when first used: E + 1 slots for E encoder pieces)
static int ensure_pipe(nasr_engine *e, int p) {
    if (!e->pipe_ready) {
        if (pick_lanes(e)) return -1;
        e->pipe_ready = true;
        release_lanes(e);
    }
    nasr_engine::Pipe &P = e->pipe[p];
    if (P.ready) return 0;
    const size_t S = (size_t)e->max_streams, M = (size_t)e->w_rows;
    for (auto &ev : P.seg_done) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&P.dec_done, hipEventDisableTiming));
    if (p == 0) {
        P.g_desc = e->g_desc; P.gh = e->gh; P.gh_collect = e->gh_collect; P.collect_dev = e->collect_dev; P.encproj = e->encproj;
    } else {
        HIPCHK(hipHostMalloc((void **)&P.gh, e->desc_bytes + e->col_bytes, hipHostMallocDefault));
        P.gh_collect = (int *)(P.gh + e->desc_bytes);
        if (dalloc(e, &P.g_desc, e->desc_bytes) || dalloc(e, &P.collect_dev, S * (1 + COLLECT_STRIDE) + 4) || dalloc(e, &P.encproj, M * JNT)) return -1;
        if (alloc_ws(e, e->ws[p])) return -1;
    }
    HIPCHK(hipHostMalloc((void **)&P.gh_dmeta, 2 * S * sizeof(int), hipHostMallocDefault));
    if (dalloc(e, &P.g_dmeta, 2 * S)) return -1;
    P.ready = true;
    return 0;
}

// the graphs of one (B, T, G, E) on one slot: the E encoder pieces and the decode.  Their kernel arguments point into the
// slot's workspace set, descriptor block and joint.enc buffer.
static int build_pipe_graphs(nasr_engine *e, int p, int B, int T, int R, int G, int nseg, hipGraphExec_t *seg_out, hipGraphExec_t *dec_out) {
    nasr_engine::Pipe &P = e->pipe[p];
    const GraphDescLayout L = graph_desc_layout(B, G);
    const RowDesc *g_rows = (const RowDesc *)(P.g_desc + L.rows), *g_vrows = (const RowDesc *)(P.g_desc + L.vrows);
    const PcmDesc *g_pcm = (const PcmDesc *)(P.g_desc + L.pcm);
    float *const encproj_saved = e->encproj;
    auto capture = [&](hipStream_t st, const char *what, hipGraphExec_t *out, const std::function<int()> &body) -> int {
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        e->encproj = P.encproj;
        use_ws(e, e->ws[p]);
#ifdef NASR_STAMPS
        g_stamp_pipe = p;
#endif
        const int rc = body();
        e->encproj = encproj_saved;
        use_ws(e, e->ws[0]);
        hipError_t ce = hipStreamEndCapture(st, &graph);
        if (rc) { if (graph) hipGraphDestroy(graph); return -1; }
        if (ce != hipSuccess) return fail("hipStreamEndCapture (%s graph) failed: %s", what, hipGetErrorString(ce));
        hipError_t ie = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (ie != hipSuccess) return fail("hipGraphInstantiate (%s graph) failed: %s", what, hipGetErrorString(ie));
        return 0;
    };
    for (int k = 0; k < nseg; k++) {
        // captured on the engine's stream (the enqueue functions launch there), replayed on lane k
        if (capture(e->st, "encoder piece", &seg_out[k], [&]() -> int {
                if (k == 0) {       // descriptors, front end
                    HIPCHK(hipMemcpyAsync(P.g_desc, P.gh, L.total, hipMemcpyHostToDevice, e->st));
                    MelParams mp;
                    memset(&mp, 0, sizeof(mp));
                    mp.desc = g_pcm; mp.B = B; mp.max_frames = max_frames_per_push(T * G); mp.abuf = e->abuf; mp.last_sample = e->last_sample;
                    mp.mel_ring = e->mel_ring; mp.window = e->window; mp.fbT = e->fbT; mp.fb_band = e->fb_band; mp.cos_t = e->cos_t; mp.sin_t = e->sin_t;
                    launch_mel(mp, mp.max_frames * HOP + NFFT, e->st);
                }
                return enqueue_encoder(e, g_rows, G > 1 ? g_vrows : g_rows, nullptr, B, T, R, G, k, nseg) ? -1 : 0;
            })) return -1;
    }
    // decode graph on the decode stream
    hipStream_t cs = e->lane[e->n_lanes - 1];       // where the decode graph is captured (it is replayed on dec_stream())
    if (capture(cs, "decode", dec_out, [&]() -> int {
            HIPCHK(hipMemcpyAsync(P.g_dmeta, P.gh_dmeta, (size_t)2 * B * sizeof(int), hipMemcpyHostToDevice, cs));
            DecParams dp;
            make_dec_params(e, g_rows, B, T * G, dp);
            dp.encproj = P.encproj;
            launch_decode_begin(dp, cs);
            int it = 0;
            for (int k = 0, n = pipe_blind_iterations(T * G, e->opt_decode_graph_iters); k < n; k++) launch_decode_iter(dp, it++, cs);
            hipLaunchKernelGGL(k_collect, dim3(B), dim3(64), 0, cs, P.g_dmeta, P.g_dmeta + B, B, e->ctrl, e->tok_ring, P.collect_dev, COLLECT_STRIDE, e->n_active);
            HIPCHK(hipMemcpyAsync(P.gh_collect, P.collect_dev, ((size_t)B * (1 + COLLECT_STRIDE) + 1) * sizeof(int), hipMemcpyDeviceToHost, cs));
            return 0;
        })) return -1;
    return 0;
}

// next encoder piece of the step in slot p: queued on its lane behind the previous piece's event.  That wait is short in
// steady state (the previous piece was launched a whole call earlier) -- unlike the decode graph it is not parked for long.
static int pipe_advance(nasr_engine *e, int p) {
    nasr_engine::Pipe &P = e->pipe[p];
    if (P.stage == 0 || P.stage >= P.nseg) return 0;
    const int k = P.stage;
    // the previous piece was launched a whole call earlier: normally it is done.  If not, the HOST waits: a stream wait would
    // put a barrier packet that finds its event pending into the lane's queue, and pending cross-queue barriers slow every
    // queue's dispatch down (tests/micro/pipe_probe.hip: 2.0 -> 2.6 us per kernel at 2 lanes, far worse with more queues)
    if (hipEventQuery(P.seg_done[k - 1]) != hipSuccess) { HostTimer ht(e->host_wait_s); HIPCHK(hipEventSynchronize(P.seg_done[k - 1])); }
    { HostTimer ht(e->host_launch_s); HIPCHK(hipGraphLaunch(P.seg_graphs[k][P.key], e->lane[k])); }
    HIPCHK(hipEventRecord(P.seg_done[k], e->lane[k]));
    P.stage = k + 1;
    return 0;
}

// Completes the step in slot p: launches the encoder pieces it still lacks, waits for its encoder, launches its decode
// graph on the decode stream (by now the younger steps' encoder pieces are queued on their lanes), waits for that,
// finishes the decode eagerly if the graph's iteration budget fell short, queues the tokens.
// the decode graph runs on the decode stream, or -- when no hardware queue is left for one -- on the lane of the last encoder
// piece, stream-ordered behind it
static bool dec_behind_last_piece(const nasr_engine *e, const nasr_engine::Pipe &P) { return P.nseg >= e->n_lanes; }
static hipStream_t dec_stream(nasr_engine *e, const nasr_engine::Pipe &P) { return e->lane[e->n_lanes - 1]; }

static int pipe_finish_launch(nasr_engine *e, int p) {
    nasr_engine::Pipe &P = e->pipe[p];
    if (P.stage == 0 || P.dec_launched) return 0;
    while (P.stage < P.nseg)
        if (pipe_advance(e, p)) return -1;
    const int B = (int)P.streams.size();
    if (!dec_behind_last_piece(e, P)) { HostTimer ht(e->host_wait_s); HIPCHK(hipEventSynchronize(P.seg_done[P.nseg - 1])); }
    for (int b = 0; b < B; b++) { P.gh_dmeta[b] = P.streams[b]->slot; P.gh_dmeta[B + b] = P.streams[b]->tok_read; }
    { HostTimer ht(e->host_launch_s); HIPCHK(hipGraphLaunch(P.dec_graphs[P.key], dec_stream(e, P))); }
    HIPCHK(hipEventRecord(P.dec_done, dec_stream(e, P)));
    P.dec_launched = true;
    return 0;
}

static int pipe_finish(nasr_engine *e, int p) {
    nasr_engine::Pipe &P = e->pipe[p];
    if (P.stage == 0) return 0;
    if (pipe_finish_launch(e, p)) return -1;
    P.dec_launched = false;
    const int B = (int)P.streams.size(), TS = P.T * P.G;
    hipStream_t ds = dec_stream(e, P);
    { HostTimer ht(e->host_wait_s); HIPCHK(hipEventSynchronize(P.dec_done)); }
    int *gh_active = P.gh_collect + (size_t)B * (1 + COLLECT_STRIDE);      // k_collect appends n_active to its records
    if (*gh_active != 0) {
        const GraphDescLayout L = graph_desc_layout(B, P.G);
        DecParams dp;
        make_dec_params(e, (const RowDesc *)(P.g_desc + L.rows), B, TS, dp);
        dp.encproj = P.encproj;
        int itn = pipe_blind_iterations(TS, e->opt_decode_graph_iters), round = 8;
        e->decode_fallbacks++;
        for (;;) {
            e->decode_fallback_rounds++;
            enqueue_decode_iters(e, dp, B, round, itn, ds);
            HIPCHK(hipMemcpyAsync(gh_active, e->n_active, sizeof(int), hipMemcpyDeviceToHost, ds));
            HIPCHK(hipStreamSynchronize(ds));
            if (*gh_active == 0) break;
            if (itn > TS * MAX_SYMBOLS + 64) return fail("decode did not terminate");
            round = std::min(2 * round, 32);
        }
        hipLaunchKernelGGL(k_collect, dim3(B), dim3(64), 0, ds, P.g_dmeta, P.g_dmeta + B, B, e->ctrl, e->tok_ring, P.collect_dev, COLLECT_STRIDE, e->n_active);
        HIPCHK(hipMemcpyAsync(P.gh_collect, P.collect_dev, ((size_t)B * (1 + COLLECT_STRIDE) + 1) * sizeof(int), hipMemcpyDeviceToHost, ds));
        HIPCHK(hipStreamSynchronize(ds));
    }
    P.stage = 0;
    return consume_collect(e, P.gh_collect, P.streams.data(), B);
}

// completes whatever the pipeline has in flight, oldest step first (tokens stay queued on their streams); cheap when nothing is
static int gp_drain(nasr_engine *e);
static int pipe_drain(nasr_engine *e) {
    if (!e->pipe_ready) return 0;
    if (gp_drain(e)) return -1;
    for (int64_t q = e->pipe_seq - nasr_engine::LSLOT; q < e->pipe_seq; q++) {
        if (q < 0) continue;
        const int p = (int)(q % nasr_engine::LSLOT);
        if (e->pipe[p].stage != 0 && e->pipe[p].seq == q && pipe_finish(e, p)) return -1;
    }
    return 0;
}

static int pipe_step(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *pcm_dev, const int32_t *n_samples, int G,
                     int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens) {
    const int T = streams[0]->T, R = streams[0]->R, shift = 8 * T;
    if (ensure_pipe(e, (int)(e->pipe_seq % nasr_engine::LSLOT))) return -1;     // also picks the lanes
    const int nseg = std::max(1, std::min({e->opt_pipeline, e->n_lanes, e->max_lanes, (int)nasr_engine::MAXSEG, (int)e->hp.n_layers}));
    const int64_t seq = e->pipe_seq;
    const int p = (int)(seq % nasr_engine::LSLOT);
    if (ensure_pipe(e, p)) return -1;
    nasr_engine::Pipe &P = e->pipe[p];
    if (pipe_finish(e, p)) return -1;                      // the slot's previous occupant (LSLOT steps ago): done in steady state
    const int64_t key = ((int64_t)B << 40) | ((int64_t)T << 24) | ((int64_t)G << 8) | (int64_t)nseg;
    auto ge = P.seg_graphs[0].find(key);
    e->graph_used[key | ((int64_t)1 << 62)] = ++e->graph_tick;
    if (ge == P.seg_graphs[0].end()) {
        if (pipe_drain(e)) return -1;
        HIPCHK(hipStreamSynchronize(e->st));
        // A new step shape: capture it for EVERY slot now, in one drained, exclusive section (a step uses the slots in turn: captured
        // lazily, the next NSLOT - 1 calls would each drain the pipeline and hold the API lock again).
        // Bounded cache: a server whose batch size changes from call to call would otherwise keep NSLOT x (E + 1) graph execs per
        // shape it has ever seen.  Nothing is in flight here: the least recently used shape of a full slot goes.
        for (int q = 0; q < nasr_engine::LSLOT; q++) {
            if (ensure_pipe(e, q)) return -1;
            nasr_engine::Pipe &Q = e->pipe[q];
            if (Q.seg_graphs[0].count(key)) continue;
            while ((int)Q.seg_graphs[0].size() >= e->opt_graph_cache) {
                int64_t victim = 0, oldest = INT64_MAX;
                for (auto &kv : Q.seg_graphs[0]) {
                    auto u = e->graph_used.find(kv.first | ((int64_t)1 << 62));
                    const int64_t t = u == e->graph_used.end() ? 0 : u->second;
                    if (t < oldest) { oldest = t; victim = kv.first; }
                }
                for (auto &m : Q.seg_graphs) { auto f = m.find(victim); if (f != m.end()) { if (f->second) hipGraphExecDestroy(f->second); m.erase(f); } }
                auto f = Q.dec_graphs.find(victim);
                if (f != Q.dec_graphs.end()) { if (f->second) hipGraphExecDestroy(f->second); Q.dec_graphs.erase(f); }
                if (q == p) e->graph_evictions++;
            }
            hipGraphExec_t gs[nasr_engine::MAXSEG] = {nullptr, nullptr, nullptr, nullptr}, dec = nullptr;
            {
                CaptureExclusive alone;
                e->gemm_coresident = nseg >= 2;          // several launch chains side by side: co-resident GEMM variants
                const int rc = build_pipe_graphs(e, q, B, T, R, G, nseg, gs, &dec);
                e->gemm_coresident = false;
                if (rc) return -1;
            }
            for (int k = 0; k < nseg; k++) Q.seg_graphs[k][key] = gs[k];
            Q.dec_graphs[key] = dec;
        }
        ge = P.seg_graphs[0].find(key);
    }
    const GraphDescLayout L = graph_desc_layout(B, G);
    RowDesc *gh_rows = (RowDesc *)(P.gh + L.rows), *gh_vrows = (RowDesc *)(P.gh + L.vrows);
    PcmDesc *gh_pcm = (PcmDesc *)(P.gh + L.pcm);
    for (int b = 0; b < B; b++) {
        nasr_stream *s = streams[b];
        PcmDesc &d = gh_pcm[b];
        memset(&d, 0, sizeof(d));
        d.pcm = pcm_dev[b]; d.slot = s->slot; d.n = n_samples[b]; d.cnt = s->abuf_cnt; d.par = s->abuf_par;
        const int avail = d.cnt + d.n;
        d.n_frames = avail < NFFT ? 0 : (avail - NFFT + HOP) / HOP;
        d.mel_wpos = (s->mel_start + s->mel_count) & (MEL_RING - 1);
        d.consumed = d.n_frames * HOP;
        fill_row_desc(gh_rows[b], s, T * G);
        for (int g = 0; g < G; g++) {
            RowDesc &v = gh_vrows[b * G + g];
            v = gh_rows[b];
            v.mel_start = (s->mel_start + g * shift) & (MEL_RING - 1);
        }
    }
    { HostTimer ht(e->host_launch_s); HIPCHK(hipGraphLaunch(ge->second, e->st)); }
    HIPCHK(hipEventRecord(P.seg_done[0], e->st));
    for (int b = 0; b < B; b++) {                          // every count is a pure function of the samples pushed
        nasr_stream *s = streams[b];
        const PcmDesc &d = gh_pcm[b];
        s->abuf_cnt = d.cnt + d.n - d.consumed;
        if (d.n_frames > 0) s->abuf_par ^= 1;
        s->mel_count += d.n_frames;
        const int par = s->cc_par;
        for (int g = 0; g < G; g++) chunk_bookkeeping(s, b);
        s->cc_par = par ^ 1;
        s->last_T = T * G; s->last_row = b; s->last_ws = p;
    }
    P.stage = 1;
    P.seq = seq;
    P.nseg = nseg;
    P.streams.assign(streams, streams + B);
    P.T = T; P.G = G; P.key = key;
    e->pipe_seq = seq + 1;
    // the steps before this one move on by one piece each, the one that has had all its pieces is decoded (a step whose
    // number of pieces differs -- the option was changed in between -- simply completes when its turn comes)
    int fin[nasr_engine::MAXSEG], nfin = 0;                // steps that have had all their pieces: decoded in this call, oldest first
    for (int k = nasr_engine::MAXSEG; k >= 1; k--) {
        if (k > seq) continue;
        const int q = (int)((seq - k) % nasr_engine::LSLOT);
        const nasr_engine::Pipe &Q = e->pipe[q];
        if (Q.stage != 0 && Q.seq == seq - k && Q.stage >= Q.nseg) fin[nfin++] = q;
    }
    // decode on the last piece's lane: queue it there before the next step's last piece goes onto that lane
    for (int i = 0; i < nfin; i++)
        if (dec_behind_last_piece(e, e->pipe[fin[i]]) && pipe_finish_launch(e, fin[i])) return -1;
    for (int k = 1; k <= nasr_engine::MAXSEG && k <= seq; k++) {
        const int q = (int)((seq - k) % nasr_engine::LSLOT);
        nasr_engine::Pipe &Q = e->pipe[q];
        if (Q.stage == 0 || Q.seq != seq - k) continue;
        if (Q.stage < Q.nseg) { if (pipe_advance(e, q)) return -1; }
    }
    for (int i = 0; i < nfin; i++)                         // the decodes of two steps share the decoder state: one at a time
        if (pipe_finish(e, fin[i])) return -1;
    e->graph_replays++;
    e->pipe_steps++;
    deliver(streams, B, tokens_out, tokens_cap, n_tokens);
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Grouped pipeline ("pipeline" = 8; fused path, one or two rows per step).  tests/micro/dual_probe.hip: four chains of one-problem
// launches (the lanes above) stream 2.45 TB/s of weights, two chains of FOUR-problem launches 4.1 TB/s -- twice the bytes in flight
// per launch boundary.  The 24 layers are 8 stages of 3; a step advances one stage per call, so 8 steps are in flight; chain c (HIP
// stream lane[c]) runs stages 4c .. 4c+3 and each of its 24 launches per call carries the same kernel of FOUR steps at four different
// layers (k_fused_skinny_grp: blockIdx.z = problem).  Chain 0 also carries the newest step's front end, chain 1 the oldest step's
// closing k_post + joint.enc and, behind them, its decode graph.  Per problem the code and the order of operations are those of the
// one-problem kernels: tokens, caches and decoder state are bit-identical to synchronous stepping (tests).  Tokens come out 8 calls
// later; every entry point that needs finished steps drains (bubbles run through the remaining stages).
// ---------------------------------------------------------------------------------------------------------------------------
static bool gp_eligible(const nasr_engine *e, int B, int T, int G) {
    constexpr int fuse_max_m = FUSE_MAX_M;
    return e->opt_pipeline == nasr_engine::GP_S && e->bf16 && e->opt_fused && !e->debug && B * T * G <= std::min(2, fuse_max_m) &&
           e->hp.n_layers % nasr_engine::GP_S == 0 && e->hp.num_prompts == 0 && e->n_lanes >= nasr_engine::GP_C && e->max_lanes >= nasr_engine::GP_C;
}

// everything chain c does in one call; slot_of_stage[j] = slot of the step at stage j, or -1.  Launches go to e->st (the caller
// captures them or has pointed e->st at the chain's stream).
static int gp_enqueue_chain(nasr_engine *e, int c, const int *slot_of_stage, int B, int T, int R, int G) {
    const int nL = e->hp.n_layers, per_stage = 8 * nL / nasr_engine::GP_S;
    float *const encproj_saved = e->encproj;
    const GraphDescLayout L = graph_desc_layout(B, G);
    int rc = 0;
    if (c == 0 && slot_of_stage[0] >= 0) {                       // the newest step: descriptors, mel, subsampling
        nasr_engine::Pipe &P = e->pipe[slot_of_stage[0]];
        const RowDesc *g_rows = (const RowDesc *)(P.g_desc + L.rows), *g_vrows = (const RowDesc *)(P.g_desc + L.vrows);
        use_ws(e, e->ws[slot_of_stage[0]]);
        HIPCHK(hipMemcpyAsync(P.g_desc, P.gh, L.total, hipMemcpyHostToDevice, e->st));
        MelParams mp;
        memset(&mp, 0, sizeof(mp));
        mp.desc = (const PcmDesc *)(P.g_desc + L.pcm); mp.B = B; mp.max_frames = max_frames_per_push(T * G); mp.abuf = e->abuf; mp.last_sample = e->last_sample;
        mp.mel_ring = e->mel_ring; mp.window = e->window; mp.fbT = e->fbT; mp.fb_band = e->fb_band; mp.cos_t = e->cos_t; mp.sin_t = e->sin_t;
        launch_mel(mp, mp.max_frames * HOP + NFFT, e->st);
        rc = enqueue_encoder(e, g_rows, G > 1 ? g_vrows : g_rows, nullptr, B, T, R, G, 0, 1, 1);
    }
    std::vector<FusedParams> rec[nasr_engine::GP_Y];
    for (int y = 0; y < nasr_engine::GP_Y && !rc; y++) {
        const int j = c * nasr_engine::GP_Y + y, slot = slot_of_stage[j];
        if (slot < 0) continue;
        nasr_engine::Pipe &P = e->pipe[slot];
        use_ws(e, e->ws[slot]);
        rc = run_layers_fused(e, (const RowDesc *)(P.g_desc + L.rows), B, T, G, per_stage * j, per_stage * (j + 1), &rec[y]);
        if (!rc && (int)rec[y].size() != per_stage) rc = fail("internal: grouped pipeline expects %d launches per stage, got %d", per_stage, (int)rec[y].size());
    }
    for (int i = 0; i < per_stage && !rc; i++) {
        FusedParamsGroup grp;
        memset(&grp, 0, sizeof(grp));
        for (int y = 0; y < nasr_engine::GP_Y; y++)
            if (!rec[y].empty()) grp.p[y] = rec[y][(size_t)i];           // an empty stage keeps g.M == 0: skipped by the kernel
        launch_fused_skinny_group(grp, nasr_engine::GP_Y, e->st);
    }
    if (!rc && c == nasr_engine::GP_C - 1 && slot_of_stage[nasr_engine::GP_S - 1] >= 0) {       // the oldest step: norm_out of layer 24, joint.enc
        const int slot = slot_of_stage[nasr_engine::GP_S - 1];
        nasr_engine::Pipe &P = e->pipe[slot];
        const RowDesc *g_rows = (const RowDesc *)(P.g_desc + L.rows);
        use_ws(e, e->ws[slot]);
        e->encproj = P.encproj;
        rc = run_layers_fused(e, g_rows, B, T, G, 8 * nL, 8 * nL);
        if (!rc) rc = enqueue_encoder(e, g_rows, g_rows, nullptr, B, T, R, G, 0, 1, 2);
    }
    e->encproj = encproj_saved;
    use_ws(e, e->ws[0]);
    return rc ? -1 : 0;
}

static int gp_capture(nasr_engine *e, hipGraphExec_t *out, const std::function<int()> &body, hipStream_t st) {
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = body();
    hipError_t ce = hipStreamEndCapture(st, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return -1; }
    if (ce != hipSuccess) return fail("hipStreamEndCapture (grouped pipeline) failed: %s", hipGetErrorString(ce));
    hipError_t ie = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (ie != hipSuccess) return fail("hipGraphInstantiate (grouped pipeline) failed: %s", hipGetErrorString(ie));
    return 0;
}

// the decode graph of slot p (same content as the lanes mode's)
static int gp_decode_graph(nasr_engine *e, int p, int B, int T, int G, hipGraphExec_t *out) {
    nasr_engine::Pipe &P = e->pipe[p];
    const GraphDescLayout L = graph_desc_layout(B, G);
    const RowDesc *g_rows = (const RowDesc *)(P.g_desc + L.rows);
    hipStream_t cs = e->lane[nasr_engine::GP_C - 1];
    return gp_capture(e, out, [&]() -> int {
        HIPCHK(hipMemcpyAsync(P.g_dmeta, P.gh_dmeta, (size_t)2 * B * sizeof(int), hipMemcpyHostToDevice, cs));
        DecParams dp;
        make_dec_params(e, g_rows, B, T * G, dp);
        dp.encproj = P.encproj;
        launch_decode_begin(dp, cs);
        int it = 0;
        for (int k = 0, n = pipe_blind_iterations(T * G, e->opt_decode_graph_iters); k < n; k++) launch_decode_iter(dp, it++, cs);
        hipLaunchKernelGGL(k_collect, dim3(B), dim3(64), 0, cs, P.g_dmeta, P.g_dmeta + B, B, e->ctrl, e->tok_ring, P.collect_dev, COLLECT_STRIDE, e->n_active);
        HIPCHK(hipMemcpyAsync(P.gh_collect, P.collect_dev, ((size_t)B * (1 + COLLECT_STRIDE) + 1) * sizeof(int), hipMemcpyDeviceToHost, cs));
        return 0;
    }, cs);
}

// one call of the grouped pipeline: every step in flight advances one stage (new_slot >= 0: a new step enters at stage 0); the step
// that leaves the last stage is decoded and its tokens are queued on its streams
static int gp_finish_decode(nasr_engine *e);
static int gp_call(nasr_engine *e, int new_slot, int64_t key, int B, int T, int R, int G) {
    constexpr int S = nasr_engine::GP_S, C = nasr_engine::GP_C;
    if (new_slot >= 0) e->gp_flight.push_back({new_slot, 0});
    if (e->gp_flight.empty()) return 0;
    int slot_of_stage[S];
    for (int j = 0; j < S; j++) slot_of_stage[j] = -1;
    for (const auto &en : e->gp_flight) slot_of_stage[en.done] = en.slot;
    bool full = new_slot >= 0;
    for (int j = 0; j < S && full; j++) full = slot_of_stage[j] == (new_slot - j + 2 * nasr_engine::NSLOT) % nasr_engine::NSLOT;
    const int par = (int)(e->gp_calls & 1);
    for (int c = 0; c < C; c++) {
        bool any = c == 0 ? slot_of_stage[0] >= 0 : false;
        for (int y = 0; y < nasr_engine::GP_Y; y++) any |= slot_of_stage[c * nasr_engine::GP_Y + y] >= 0;
        if (!any) { e->gp_ev_set[c][par] = false; continue; }
        if (c > 0 && e->gp_ev_set[c - 1][par ^ 1] && hipEventQuery(e->gp_ev[c - 1][par ^ 1]) != hipSuccess) {
            // stage 4c of this call reads what stage 4c - 1 wrote in the previous call on the other chain: normally long done; the HOST
            // waits if not (a pending cross-queue barrier packet slows every queue's dispatch: lanes mode, tests/micro/pipe_probe.hip)
            HostTimer ht(e->host_wait_s);
            HIPCHK(hipEventSynchronize(e->gp_ev[c - 1][par ^ 1]));
        }
        hipGraphExec_t ex = nullptr;
        if (full) {
            auto &m = e->gp_graphs[new_slot][c];
            auto it = m.find(key);
            if (it != m.end()) ex = it->second;
        }
        if (ex) {
            HostTimer ht(e->host_launch_s);
            HIPCHK(hipGraphLaunch(ex, e->lane[c]));
            e->gp_graph_chains++;
        } else {
            e->gp_eager_chains++;                                                     // fill, drain, or a shape not captured yet: the same launches, eagerly
            hipStream_t keep = e->st;
            e->st = e->lane[c];
            const int rc = gp_enqueue_chain(e, c, slot_of_stage, B, T, R, G);
            e->st = keep;
            if (rc) return -1;
        }
        HIPCHK(hipEventRecord(e->gp_ev[c][par], e->lane[c]));
        e->gp_ev_set[c][par] = true;
    }
    e->gp_calls++;
    for (auto &en : e->gp_flight) en.done++;
    // The decode launched in the PREVIOUS call is completed now -- after this call's chains have been queued, so the device is never
    // idle while the host waits -- and only then the decode of the step that has just left the encoder is launched (its token
    // gather needs the read position the previous decode's tokens have moved).  It runs behind chain C - 1 and is collected next call.
    if (gp_finish_decode(e)) return -1;
    if (!e->gp_flight.empty() && e->gp_flight.front().done >= S) {
        const int p = e->gp_flight.front().slot;
        e->gp_flight.erase(e->gp_flight.begin());
        nasr_engine::Pipe &P = e->pipe[p];
        const int nB = (int)P.streams.size();
        hipStream_t ds = e->lane[C - 1];
        for (int b = 0; b < nB; b++) { P.gh_dmeta[b] = P.streams[b]->slot; P.gh_dmeta[nB + b] = P.streams[b]->tok_read; }
        { HostTimer ht(e->host_launch_s); HIPCHK(hipGraphLaunch(P.dec_graphs[P.key], ds)); }
        HIPCHK(hipEventRecord(P.dec_done, ds));
        e->gp_dec_pending = p;
    }
    return 0;
}

// completes the decode that is in flight (launched one call earlier), queues its tokens on its streams
static int gp_finish_decode(nasr_engine *e) {
    if (e->gp_dec_pending < 0) return 0;
    nasr_engine::Pipe &P = e->pipe[e->gp_dec_pending];
    e->gp_dec_pending = -1;
    const int nB = (int)P.streams.size();
    hipStream_t ds = e->lane[nasr_engine::GP_C - 1];
    { HostTimer ht(e->host_wait_s); HIPCHK(hipEventSynchronize(P.dec_done)); }
    int *gh_active = P.gh_collect + (size_t)nB * (1 + COLLECT_STRIDE);
    if (*gh_active != 0) {                                      // a burst beyond the graph's iteration budget: finish eagerly
        const GraphDescLayout L = graph_desc_layout(nB, P.G);
        DecParams dp;
        make_dec_params(e, (const RowDesc *)(P.g_desc + L.rows), nB, P.T * P.G, dp);
        dp.encproj = P.encproj;
        int itn = pipe_blind_iterations(P.T * P.G, e->opt_decode_graph_iters), round = 8;
        e->decode_fallbacks++;
        for (;;) {
            e->decode_fallback_rounds++;
            enqueue_decode_iters(e, dp, nB, round, itn, ds);
            HIPCHK(hipMemcpyAsync(gh_active, e->n_active, sizeof(int), hipMemcpyDeviceToHost, ds));
            HIPCHK(hipStreamSynchronize(ds));
            if (*gh_active == 0) break;
            if (itn > P.T * P.G * MAX_SYMBOLS + 64) return fail("decode did not terminate");
            round = std::min(2 * round, 32);
        }
        hipLaunchKernelGGL(k_collect, dim3(nB), dim3(64), 0, ds, P.g_dmeta, P.g_dmeta + nB, nB, e->ctrl, e->tok_ring, P.collect_dev, COLLECT_STRIDE, e->n_active);
        HIPCHK(hipMemcpyAsync(P.gh_collect, P.collect_dev, ((size_t)nB * (1 + COLLECT_STRIDE) + 1) * sizeof(int), hipMemcpyDeviceToHost, ds));
        HIPCHK(hipStreamSynchronize(ds));
    }
    return consume_collect(e, P.gh_collect, P.streams.data(), nB);
}

static int gp_drain(nasr_engine *e) {
    while (!e->gp_flight.empty()) {
        nasr_engine::Pipe &P = e->pipe[e->gp_flight.front().slot];
        if (gp_call(e, -1, P.key, (int)P.streams.size(), P.T, P.streams[0]->R, P.G)) return -1;
    }
    return gp_finish_decode(e);
}

static int gp_step(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *pcm_dev, const int32_t *n_samples, int G,
                   int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens) {
    const int T = streams[0]->T, R = streams[0]->R, shift = 8 * T;
    const int64_t key = ((int64_t)B << 40) | ((int64_t)T << 24) | ((int64_t)G << 8) | (int64_t)nasr_engine::GP_S;
    // lanes-mode steps in flight, or grouped steps of another shape or other streams: complete them first
    for (int q = 0; q < nasr_engine::LSLOT; q++)
        if (e->pipe[q].stage != 0) { if (pipe_drain(e)) return -1; break; }
    if (!e->gp_flight.empty()) {
        nasr_engine::Pipe &O = e->pipe[e->gp_flight.back().slot];
        bool same = O.key == key && (int)O.streams.size() == B;
        for (int b = 0; b < B && same; b++) same = O.streams[(size_t)b] == streams[b];
        if (!same && gp_drain(e)) return -1;
    }
    const int p = e->gp_next_slot;
    e->gp_next_slot = (p + 1) % nasr_engine::NSLOT;
    if (ensure_pipe(e, p)) return -1;
    for (auto &ce : e->gp_ev)
        for (auto &ev : ce)
            if (!ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    nasr_engine::Pipe &P = e->pipe[p];
    e->graph_used[key | ((int64_t)1 << 62)] = ++e->graph_tick;
    if (!P.dec_graphs.count(key)) {                                  // a new shape: every slot's decode graph and chain graphs, once
        if (gp_drain(e)) return -1;
        for (int c = 0; c < nasr_engine::GP_C; c++) HIPCHK(hipStreamSynchronize(e->lane[c]));
        CaptureExclusive alone;
        for (int q = 0; q < nasr_engine::NSLOT; q++) {
            if (ensure_pipe(e, q)) return -1;
            if (e->pipe[q].dec_graphs.count(key)) continue;
            hipGraphExec_t dec = nullptr;
            if (gp_decode_graph(e, q, B, T, G, &dec)) return -1;
            e->pipe[q].dec_graphs[key] = dec;
        }
        for (int q = 0; q < nasr_engine::NSLOT; q++) {
            int sos[nasr_engine::GP_S];
            for (int j = 0; j < nasr_engine::GP_S; j++) sos[j] = (q - j + 2 * nasr_engine::NSLOT) % nasr_engine::NSLOT;
            for (int c = 0; c < nasr_engine::GP_C; c++) {
                hipGraphExec_t ex = nullptr;
                if (gp_capture(e, &ex, [&]() -> int { return gp_enqueue_chain(e, c, sos, B, T, R, G); }, e->st)) return -1;
                e->gp_graphs[q][c][key] = ex;
            }
        }
    }
    const GraphDescLayout L = graph_desc_layout(B, G);
    RowDesc *gh_rows = (RowDesc *)(P.gh + L.rows), *gh_vrows = (RowDesc *)(P.gh + L.vrows);
    PcmDesc *gh_pcm = (PcmDesc *)(P.gh + L.pcm);
    for (int b = 0; b < B; b++) {
        nasr_stream *s = streams[b];
        PcmDesc &d = gh_pcm[b];
        memset(&d, 0, sizeof(d));
        d.pcm = pcm_dev[b]; d.slot = s->slot; d.n = n_samples[b]; d.cnt = s->abuf_cnt; d.par = s->abuf_par;
        const int avail = d.cnt + d.n;
        d.n_frames = avail < NFFT ? 0 : (avail - NFFT + HOP) / HOP;
        d.mel_wpos = (s->mel_start + s->mel_count) & (MEL_RING - 1);
        d.consumed = d.n_frames * HOP;
        fill_row_desc(gh_rows[b], s, T * G);
        for (int g = 0; g < G; g++) {
            RowDesc &v = gh_vrows[b * G + g];
            v = gh_rows[b];
            v.mel_start = (s->mel_start + g * shift) & (MEL_RING - 1);
        }
    }
    for (int b = 0; b < B; b++) {                          // every count is a pure function of the samples pushed
        nasr_stream *s = streams[b];
        const PcmDesc &d = gh_pcm[b];
        s->abuf_cnt = d.cnt + d.n - d.consumed;
        if (d.n_frames > 0) s->abuf_par ^= 1;
        s->mel_count += d.n_frames;
        const int par = s->cc_par;
        for (int g = 0; g < G; g++) chunk_bookkeeping(s, b);
        s->cc_par = par ^ 1;
        s->last_T = T * G; s->last_row = b; s->last_ws = p;
    }
    P.streams.assign(streams, streams + B);
    P.T = T; P.G = G; P.key = key; P.seq = -1; P.stage = 0;
    if (gp_call(e, p, key, B, T, R, G)) return -1;
    e->graph_replays++;
    e->gp_steps++;
    deliver(streams, B, tokens_out, tokens_cap, n_tokens);
    return 1;
}

// returns 1 if the step was executed through the graph, 0 if not eligible, <0 on error
static int try_graph_step(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *pcm_dev,
                          const int32_t *n_samples, int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens) {
    const int T = streams[0]->T, R = streams[0]->R;
    const int chunk_mel = PRE_CACHE + 8 * T, shift = 8 * T;
    int G = -1;
    for (int b = 0; b < B; b++) {
        const nasr_stream *s = streams[b];
        const int n = n_samples[b];
        if (n <= 0 || n > MAX_PUSH) return 0;
        const int avail = s->abuf_cnt + n;
        const int nf = avail < NFFT ? 0 : (avail - NFFT + HOP) / HOP;
        const int mc = s->mel_count + nf;
        if (mc < chunk_mel) return 0;
        const int g = (mc - chunk_mel) / shift + 1;               // chunks this push completes
        if (G < 0) G = g;
        if (g != G) return 0;                                     // every stream must complete the same number
        if (nf > max_frames_per_push(T * G)) return 0;
    }
    if (G > 1) {
        // G consecutive chunks of a stream are one launch sequence (same results: a chunk's layer-l
        // inputs do not depend on the previous chunk's layer-l outputs, only on its K/V and conv state).
        // Needs the fused small-M path and the new rows to fit in the K/V ring next to the 70-row window.
        if (!e->opt_multichunk || B * G * T > e->w_rows || G * T > MAXNEW) return 0;
    }
    if (e->opt_pipeline) {
        if (!e->pipe_ready && ensure_pipe(e, 0)) return -1;          // picks the lanes
        if (gp_eligible(e, B, T, G)) return gp_step(e, streams, B, pcm_dev, n_samples, G, tokens_out, tokens_cap, n_tokens);
        if (gp_drain(e)) return -1;
        return pipe_step(e, streams, B, pcm_dev, n_samples, G, tokens_out, tokens_cap, n_tokens);
    }
    if (pipe_drain(e)) return -1;
    const int64_t key = ((int64_t)B << 32) | ((int64_t)T << 16) | (int64_t)G;
    auto it = e->graphs.find(key);
    e->graph_used[key] = ++e->graph_tick;
    if (it == e->graphs.end()) {
        HIPCHK(hipStreamSynchronize(e->st));
        while ((int)e->graphs.size() >= e->opt_graph_cache) {          // bounded cache, least recently used shape first
            int64_t victim = 0, oldest = INT64_MAX;
            for (auto &kv : e->graphs) {
                auto u = e->graph_used.find(kv.first);
                const int64_t t = u == e->graph_used.end() ? 0 : u->second;
                if (t < oldest) { oldest = t; victim = kv.first; }
            }
            hipGraphExecDestroy(e->graphs[victim]);
            e->graphs.erase(victim);
            e->graph_used.erase(victim);
            e->graph_evictions++;
        }
        hipGraphExec_t ex = nullptr;
        {
            CaptureExclusive alone;
            if (build_step_graph(e, B, T, R, G, &ex)) return -1;
        }
        it = e->graphs.emplace(key, ex).first;
    }
    const GraphDescLayout L = graph_desc_layout(B, G);
    RowDesc *gh_rows = (RowDesc *)(e->gh + L.rows), *gh_vrows = (RowDesc *)(e->gh + L.vrows);
    PcmDesc *gh_pcm = (PcmDesc *)(e->gh + L.pcm);
    int *gh_meta = (int *)(e->gh + L.meta);
    int *gh_active = e->gh_collect + (size_t)B * (1 + COLLECT_STRIDE);      // k_collect appends n_active to its records
    for (int b = 0; b < B; b++) {
        nasr_stream *s = streams[b];
        PcmDesc &d = gh_pcm[b];
        memset(&d, 0, sizeof(d));
        d.pcm = pcm_dev[b]; d.slot = s->slot; d.n = n_samples[b]; d.cnt = s->abuf_cnt; d.par = s->abuf_par;
        const int avail = d.cnt + d.n;
        d.n_frames = avail < NFFT ? 0 : (avail - NFFT + HOP) / HOP;
        d.mel_wpos = (s->mel_start + s->mel_count) & (MEL_RING - 1);
        d.consumed = d.n_frames * HOP;
        fill_row_desc(gh_rows[b], s, T * G);
        for (int g = 0; g < G; g++) {
            RowDesc &v = gh_vrows[b * G + g];
            v = gh_rows[b];
            v.mel_start = (s->mel_start + g * shift) & (MEL_RING - 1);
        }
        gh_meta[b] = s->slot;
        gh_meta[B + b] = s->tok_read;
    }
    HIPCHK(hipGraphLaunch(it->second, e->st));
    HIPCHK(hipStreamSynchronize(e->st));
    e->graph_replays++;
    for (int b = 0; b < B; b++) {
        nasr_stream *s = streams[b];
        const PcmDesc &d = gh_pcm[b];
        s->abuf_cnt = d.cnt + d.n - d.consumed;
        if (d.n_frames > 0) s->abuf_par ^= 1;
        s->mel_count += d.n_frames;
    }
    if (*gh_active != 0) {   // some stream emitted more symbols than the graph's iteration budget: finish eagerly
        DecParams dp;
        make_dec_params(e, (const RowDesc *)(e->g_desc + L.rows), B, T * G, dp);
        int itn = decode_blind_iterations(T * G), round = T * G > 1 ? 8 : 4;   // idle iterations ~10 us each, a round trip ~40 us
        e->decode_fallbacks++;
        for (;;) {
            e->decode_fallback_rounds++;
            enqueue_decode_iters(e, dp, B, round, itn);
            HIPCHK(hipMemcpyAsync(gh_active, e->n_active, sizeof(int), hipMemcpyDeviceToHost, e->st));
            HIPCHK(hipStreamSynchronize(e->st));
            if (*gh_active == 0) break;
            if (itn > T * G * MAX_SYMBOLS + 64) return fail("decode did not terminate");
            round = std::min(2 * round, 32);
        }
        for (int b = 0; b < B; b++) {
            const int par = streams[b]->cc_par;
            for (int g = 0; g < G; g++) chunk_bookkeeping(streams[b], b);
            streams[b]->cc_par = par ^ 1;          // one launch = one conv-cache buffer flip, whatever G is
            streams[b]->last_T = T * G; streams[b]->last_row = b;
        }
        return collect_tokens(e, streams, B, tokens_out, tokens_cap, n_tokens) ? -1 : 1;
    }
    for (int b = 0; b < B; b++) {
        const int par = streams[b]->cc_par;
        for (int g = 0; g < G; g++) chunk_bookkeeping(streams[b], b);
        streams[b]->cc_par = par ^ 1;
        streams[b]->last_T = T * G; streams[b]->last_row = b;
    }
    if (consume_collect(e, e->gh_collect, streams, B)) return -1;
    deliver(streams, B, tokens_out, tokens_cap, n_tokens);
    return 1;
}

// one piece of a push (device-resident PCM): the graph-replayed launch sequence when eligible, else the eager
// sub-push loop (mel -> chunk by chunk) -- then the new tokens of every stream
static int push_piece(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *base, const int32_t *n_samples,
                      int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens, uint32_t flags) {
    std::vector<int64_t> off(B, 0);
    if (e->opt_graph && !e->debug && !e->prof.on && !(flags & NASR_FLAG_NO_SYNC)) {
        const int gr = try_graph_step(e, streams, B, base, n_samples, tokens_out, tokens_cap, n_tokens);
        if (gr < 0) return -1;
        if (gr == 1) return 0;
    }
    if (pipe_drain(e)) return -1;
    e->eager_steps++;
    // sub-pushes of at most MAX_PUSH samples keep the audio buffer and the mel ring bounded
    for (;;) {
        std::vector<PcmDesc> pd;
        int max_frames = 0, max_n = 0;
        std::vector<int> who;
        for (int b = 0; b < B; b++) {
            const int64_t rem = n_samples[b] - off[b];
            if (rem <= 0) continue;
            nasr_stream *s = streams[b];
            PcmDesc d;
            memset(&d, 0, sizeof(d));
            d.pcm = base[b] + off[b];
            d.slot = s->slot;
            d.n = (int)std::min<int64_t>(rem, MAX_PUSH);
            d.cnt = s->abuf_cnt;
            d.par = s->abuf_par;
            const int avail = d.cnt + d.n;
            d.n_frames = avail < NFFT ? 0 : (avail - NFFT + HOP) / HOP;   // src/preprocessor.cpp:320-328
            d.mel_wpos = (s->mel_start + s->mel_count) & (MEL_RING - 1);
            d.consumed = d.n_frames * HOP;
            pd.push_back(d);
            who.push_back(b);
            max_frames = std::max(max_frames, d.n_frames);
            max_n = std::max(max_n, d.n);
        }
        if (pd.empty()) break;
        const PcmDesc *dpd;
        if (stage_desc(e, pd, &dpd)) return -1;
        MelParams mp;
        memset(&mp, 0, sizeof(mp));
        mp.desc = dpd; mp.B = (int)pd.size(); mp.max_frames = max_frames; mp.abuf = e->abuf; mp.last_sample = e->last_sample;
        mp.mel_ring = e->mel_ring; mp.window = e->window; mp.fbT = e->fbT; mp.fb_band = e->fb_band; mp.cos_t = e->cos_t; mp.sin_t = e->sin_t;
        if (e->debug) { mp.tap = e->tap_mel; mp.tap_cap = e->tap_mel_cap; }
        {
            ProfScope ps(e, "k_mel", 0, 0);
            launch_mel(mp, max_n, e->st);
        }
        for (size_t i = 0; i < pd.size(); i++) {
            nasr_stream *s = streams[who[i]];
            off[who[i]] += pd[i].n;
            s->abuf_cnt = pd[i].cnt + pd[i].n - pd[i].consumed;
            if (pd[i].n_frames > 0) s->abuf_par ^= 1;
            s->mel_count += pd[i].n_frames;
            if (e->debug) { e->tap_mel_frames[s->slot] = pd[i].n_frames; e->tap_mel_row[s->slot] = (int)i; }
        }
        if (drain_chunks(e, streams, B)) return -1;
    }
    if (flags & NASR_FLAG_NO_SYNC) {
        if (n_tokens) for (int b = 0; b < B; b++) n_tokens[b] = 0;
        return 0;
    }
    return collect_tokens(e, streams, B, tokens_out, tokens_cap, n_tokens);
}

extern "C" int nasr_engine_step(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *pcm,
                                const int32_t *n_samples, int32_t *const *tokens_out, const int32_t *tokens_cap,
                                int32_t *n_tokens, uint32_t flags) {
    ApiGuard api_guard;
    if (validate_batch(e, streams, B)) return -1;
    if (!pcm || !n_samples) return fail("null pcm / n_samples");
    HIPCHK(hipSetDevice(e->device));
    std::vector<const int16_t *> base(B, nullptr);
    size_t total = 0;
    for (int b = 0; b < B; b++) {
        if (n_samples[b] < 0) return fail("negative n_samples");
        if (n_samples[b] > 0 && !pcm[b]) return fail("null pcm for stream %d", b);
        total += (size_t)n_samples[b];
    }
    if (!(flags & NASR_FLAG_PCM_DEVICE)) {
        // hand-over of host buffers: one gather into the device staging area
        if (total > e->pcm_stage_cap) {
            HIPCHK(hipStreamSynchronize(e->st));
            if (e->pcm_stage) hipFree(e->pcm_stage);
            e->pcm_stage_cap = total + 65536;
            HIPCHK(hipMalloc((void **)&e->pcm_stage, e->pcm_stage_cap * 2));
        }
        auto &pin = e->pcm_pin[e->pcm_pin_next++ & 3];
        if (total > pin.cap) {
            HIPCHK(hipStreamSynchronize(e->st));
            if (pin.p) hipHostFree(pin.p);
            pin.p = nullptr;
            pin.cap = total + 65536;
            HIPCHK(hipHostMalloc((void **)&pin.p, pin.cap * 2, hipHostMallocDefault));
        }
        size_t o = 0;
        for (int b = 0; b < B; b++) {
            if (n_samples[b] > 0) memcpy(pin.p + o, pcm[b], (size_t)n_samples[b] * 2);
            base[b] = e->pcm_stage + o;
            o += (size_t)n_samples[b];
        }
        ProfScope ps(e, "h2d_pcm", (double)total * 2);
        if (total > 0) HIPCHK(hipMemcpyAsync(e->pcm_stage, pin.p, total * 2, hipMemcpyHostToDevice, e->st));
    } else {
        for (int b = 0; b < B; b++) base[b] = pcm[b];
    }
    if (e->debug) for (int b = 0; b < B; b++) { e->tap_mel_frames[streams[b]->slot] = 0; e->tap_mel_row[streams[b]->slot] = b; }
    for (int b = 0; b < B; b++) streams[b]->samples_in += n_samples[b];
    // A push longer than one launch sequence can take (MAXNEW encoder frames per stream, w_rows rows in all) is
    // cut into pieces of whole chunks; each piece is a multi-chunk step when the streams are aligned.
    const int T = streams[0]->T;
    int gcap = std::min(MAXNEW / T, e->w_rows / (B * T));
    if (gcap < 1) gcap = 1;
    const int64_t piece = (int64_t)gcap * 8 * T * HOP;
    bool multi = false;
    for (int b = 0; b < B; b++) multi = multi || n_samples[b] > piece;
    if (!multi) return push_piece(e, streams, B, base.data(), n_samples, tokens_out, tokens_cap, n_tokens, flags);
    std::vector<int64_t> off(B, 0);
    std::vector<int32_t> acc(B, 0), np(B), cap_left(B), got(B);
    std::vector<const int16_t *> ptr(B);
    std::vector<int32_t *> outp(B);
    for (;;) {
        bool any = false;
        for (int b = 0; b < B; b++) {
            const int64_t rem = n_samples[b] - off[b];
            np[b] = (int32_t)std::min<int64_t>(rem, piece);
            any = any || np[b] > 0;
            ptr[b] = base[b] + off[b];
            const int32_t cap = tokens_out && tokens_out[b] && tokens_cap ? tokens_cap[b] : 0;
            const int32_t used = std::min(acc[b], cap);
            outp[b] = cap > 0 ? tokens_out[b] + used : nullptr;
            cap_left[b] = cap - used;
        }
        if (!any) break;
        if (push_piece(e, streams, B, ptr.data(), np.data(), outp.data(), cap_left.data(), got.data(), flags)) return -1;
        for (int b = 0; b < B; b++) { off[b] += np[b]; acc[b] += got[b]; }
    }
    if (n_tokens) for (int b = 0; b < B; b++) n_tokens[b] = acc[b];
    return 0;
}

extern "C" int nasr_engine_step_mel(nasr_engine *e, nasr_stream *const *streams, int B, const float *const *mel,
                                    const int32_t *n_frames, int32_t *const *tokens_out, const int32_t *tokens_cap,
                                    int32_t *n_tokens, uint32_t flags) {
    ApiGuard api_guard;
    if (validate_batch(e, streams, B)) return -1;
    if (!mel || !n_frames) return fail("null mel / n_frames");
    HIPCHK(hipSetDevice(e->device));
    if (pipe_drain(e)) return -1;
    std::vector<int> off(B, 0);
    const int piece = 8 * streams[0]->T;   // one shift at a time keeps the ring bounded
    for (;;) {
        std::vector<PcmDesc> pd;
        std::vector<int> who;
        for (int b = 0; b < B; b++) {
            if (n_frames[b] < 0 || (n_frames[b] > 0 && !mel[b])) return fail("bad mel input for stream %d", b);
            const int rem = n_frames[b] - off[b];
            if (rem <= 0) continue;
            PcmDesc d;
            memset(&d, 0, sizeof(d));
            d.slot = streams[b]->slot;
            d.n_frames = std::min(rem, piece);
            d.mel_wpos = (streams[b]->mel_start + streams[b]->mel_count) & (MEL_RING - 1);
            pd.push_back(d);
            who.push_back(b);
        }
        if (pd.empty()) break;
        const size_t need = pd.size() * (size_t)piece * NMEL;
        if (need > e->mel_stage_cap) {
            HIPCHK(hipStreamSynchronize(e->st));
            if (e->mel_stage) hipFree(e->mel_stage);
            e->mel_stage_cap = need;
            HIPCHK(hipMalloc((void **)&e->mel_stage, need * 4));
        }
        for (size_t i = 0; i < pd.size(); i++)
            HIPCHK(hipMemcpyAsync(e->mel_stage + i * (size_t)piece * NMEL, mel[who[i]] + (size_t)off[who[i]] * NMEL,
                                  (size_t)pd[i].n_frames * NMEL * 4, hipMemcpyHostToDevice, e->st));
        const PcmDesc *dpd;
        if (stage_desc(e, pd, &dpd)) return -1;
        launch_mel_put(e->mel_stage, dpd, (int)pd.size(), piece, e->mel_ring, e->st);
        for (size_t i = 0; i < pd.size(); i++) {
            off[who[i]] += pd[i].n_frames;
            streams[who[i]]->mel_count += pd[i].n_frames;
        }
        if (drain_chunks(e, streams, B)) return -1;
        HIPCHK(hipStreamSynchronize(e->st));   // host mel staging is reused next round
    }
    if (flags & NASR_FLAG_NO_SYNC) {
        if (n_tokens) for (int b = 0; b < B; b++) n_tokens[b] = 0;
        return 0;
    }
    return collect_tokens(e, streams, B, tokens_out, tokens_cap, n_tokens);
}

extern "C" int nasr_engine_finalize(nasr_engine *e, nasr_stream *const *streams, int B, int32_t *const *tokens_out,
                                    const int32_t *tokens_cap, int32_t *n_tokens) {
    ApiGuard api_guard;
    if (validate_batch(e, streams, B)) return -1;
    HIPCHK(hipSetDevice(e->device));
    if (pipe_drain(e)) return -1;
    // src/nemo-stream.cpp:1234-1258: frames > 9 -> n_valid = (frames-9)/8 outputs of one zero-padded step
    std::vector<nasr_stream *> rows;
    std::vector<int> nd;
    std::vector<PcmDesc> pd;
    int max_pad = 0;
    for (int b = 0; b < B; b++) {
        nasr_stream *s = streams[b];
        const int chunk_mel = PRE_CACHE + 8 * s->T;
        if (s->mel_count <= PRE_CACHE) continue;
        const int n_valid = (s->mel_count - PRE_CACHE) / 8;
        if (n_valid <= 0) continue;
        if (s->mel_count < chunk_mel) {
            PcmDesc d;
            memset(&d, 0, sizeof(d));
            d.slot = s->slot;
            d.n_frames = chunk_mel - s->mel_count;
            d.mel_wpos = (s->mel_start + s->mel_count) & (MEL_RING - 1);
            pd.push_back(d);
            max_pad = std::max(max_pad, d.n_frames);
            s->mel_count = chunk_mel;
        }
        rows.push_back(s);
        nd.push_back(std::min(n_valid, s->T));
    }
    if (!pd.empty()) {
        const PcmDesc *dpd;
        if (stage_desc(e, pd, &dpd)) return -1;
        launch_mel_zero(dpd, (int)pd.size(), max_pad, e->mel_ring, e->st);
    }
    if (!rows.empty() && run_chunk(e, rows, nd)) return -1;
    return collect_tokens(e, streams, B, tokens_out, tokens_cap, n_tokens);
}

extern "C" int nasr_engine_collect(nasr_engine *e, nasr_stream *const *streams, int B, int32_t *const *tokens_out,
                                   const int32_t *tokens_cap, int32_t *n_tokens) {
    ApiGuard api_guard;
    if (validate_batch(e, streams, B)) return -1;
    HIPCHK(hipSetDevice(e->device));
    return collect_tokens(e, streams, B, tokens_out, tokens_cap, n_tokens);
}

extern "C" int nasr_stream_get_token_frames(const nasr_stream *s, int64_t first, int32_t count, int32_t *frames_out) {
    ApiGuard api_guard;
    if (!s || (count > 0 && !frames_out)) return fail("null argument");
    if (first < 0 || count < 0) return fail("negative token range");
    nasr_engine *e = s->e;
    HIPCHK(hipSetDevice(e->device));
    if (pipe_drain(e)) return -1;
    HIPCHK(hipStreamSynchronize(e->st));
    DecCtrl c;
    HIPCHK(hipMemcpy(&c, e->ctrl + s->slot, sizeof(c), hipMemcpyDeviceToHost));
    if (first + count > c.n_tok) count = first < c.n_tok ? (int32_t)(c.n_tok - first) : 0;
    if (count > 0 && c.n_tok - first > TOK_CAP) return fail("token %lld is older than the %d-token device ring", (long long)first, TOK_CAP);
    if (count <= 0) return 0;
    std::vector<int> ring(TOK_CAP);
    HIPCHK(hipMemcpy(ring.data(), e->tok_frame + (size_t)s->slot * TOK_CAP, TOK_CAP * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < count; i++) frames_out[i] = ring[(size_t)((first + i) & (TOK_CAP - 1))];
    return count;
}

// host mirror only: no pipeline drain, no stream synchronisation, no copy (the per-call path of a server)
extern "C" int nasr_engine_get_counter(const nasr_engine *e, const char *name, int64_t *value) {
    if (!e || !name || !value) return fail("null argument");
    int64_t execs = (int64_t)e->graphs.size(), shapes = (int64_t)e->graphs.size();
    std::map<int64_t, int> keys;
    for (int p = 0; p < nasr_engine::NSLOT; p++) {
        for (auto &m : e->pipe[p].seg_graphs) for (auto &kv : m) { execs += kv.second != nullptr; keys[kv.first] = 1; }
        for (auto &kv : e->pipe[p].dec_graphs) execs += kv.second != nullptr;
    }
    shapes += (int64_t)keys.size();
    for (auto &per_slot : e->gp_graphs) for (auto &m : per_slot) execs += (int64_t)m.size();
    if (!strcmp(name, "graph_execs")) *value = execs;
    else if (!strcmp(name, "graph_shapes")) *value = shapes;
    else if (!strcmp(name, "graph_evictions")) *value = e->graph_evictions;
    else if (!strcmp(name, "graph_replays")) *value = e->graph_replays;
    else if (!strcmp(name, "eager_steps")) *value = e->eager_steps;
    else if (!strcmp(name, "pipelined_steps")) *value = e->pipe_steps;
    else if (!strcmp(name, "grouped_steps")) *value = e->gp_steps;
    else if (!strcmp(name, "lanes")) *value = e->pipe_ready ? e->n_lanes : 0;      // HIP streams found to overlap (0: not picked yet)
    else return fail("unknown counter '%s'", name);
    return 0;
}

extern "C" int nasr_stream_get_progress(const nasr_stream *s, nasr_stream_stats *out) {
    if (!s || !out) return fail("null argument");
    memset(out, 0, sizeof(*out));
    out->samples_in = s->samples_in;
    out->chunks = s->chunks;
    out->decode_iterations = -1;               // device counters: nasr_stream_get_stats
    out->tokens = -1;
    out->cache_valid_len = s->valid_len;
    out->mel_frames_buffered = s->mel_count;
    out->reserved = (int32_t)s->tok_queue.size();   // tokens decoded but not yet handed to the caller
    return 0;
}

extern "C" int nasr_stream_get_stats(const nasr_stream *s, nasr_stream_stats *out) {
    ApiGuard api_guard;
    if (!s || !out) return fail("null argument");
    nasr_engine *e = s->e;
    HIPCHK(hipSetDevice(e->device));
    if (pipe_drain(e)) return -1;
    DecCtrl c;
    HIPCHK(hipStreamSynchronize(e->st));
    HIPCHK(hipMemcpy(&c, e->ctrl + s->slot, sizeof(c), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    out->samples_in = s->samples_in;
    out->chunks = s->chunks;
    out->decode_iterations = c.iterations;
    out->tokens = c.n_tok;
    out->cache_valid_len = s->valid_len;
    out->mel_frames_buffered = s->mel_count;
    return 0;
}

extern "C" int64_t nasr_stream_get_tap(nasr_stream *s, int which, int index, float *out, int64_t cap) {
    ApiGuard api_guard;
    if (!s || !out) return fail("null argument");
    nasr_engine *e = s->e;
    HIPCHK(hipSetDevice(e->device));
    if (pipe_drain(e)) return -1;
    HIPCHK(hipStreamSynchronize(e->st));
    const size_t slot = (size_t)s->slot;
    const int T = s->last_T;
    auto need_debug = [&]() { return e->tap_sub != nullptr; };
    switch (which) {
    case NASR_TAP_MEL: {
        if (!need_debug()) return fail("debug taps not enabled");
        const int n = std::min(e->tap_mel_frames[slot], e->tap_mel_cap);
        if ((int64_t)n * NMEL > cap) return fail("tap buffer too small");
        HIPCHK(hipMemcpy(out, e->tap_mel + (size_t)e->tap_mel_row[slot] * e->tap_mel_cap * NMEL, (size_t)n * NMEL * 4, hipMemcpyDeviceToHost));
        return (int64_t)n * NMEL;
    }
    case NASR_TAP_SUBSAMPLED:
    case NASR_TAP_ENCODER_OUT:
    case NASR_TAP_LAYER_OUT: {
        if ((int64_t)T * D > cap) return fail("tap buffer too small");
        if (which == NASR_TAP_ENCODER_OUT && !(e->debug && need_debug())) {
            // without debug buffers: valid until the next chunk step of this engine
            if (e->hp.num_prompts > 0 || T == 0) return fail("encoder-out tap needs debug mode here");
            HIPCHK(hipMemcpy(out, e->ws[s->last_ws].x + (size_t)s->last_row * T * D, (size_t)T * D * 4, hipMemcpyDeviceToHost));
            return (int64_t)T * D;
        }
        if (!need_debug()) return fail("debug taps not enabled");
        const float *src = which == NASR_TAP_SUBSAMPLED ? e->tap_sub + slot * TMAX * D
                         : which == NASR_TAP_ENCODER_OUT ? e->tap_enc + slot * TMAX * D
                         : e->tap_layers + (slot * e->hp.n_layers + (size_t)index) * TMAX * D;
        if (which == NASR_TAP_LAYER_OUT && (index < 0 || index >= e->hp.n_layers)) return fail("layer index out of range");
        HIPCHK(hipMemcpy(out, src, (size_t)T * D * 4, hipMemcpyDeviceToHost));
        return (int64_t)T * D;
    }
    case NASR_TAP_K_CACHE:
    case NASR_TAP_V_CACHE: {
        if (index < 0 || index >= e->hp.n_layers) return fail("layer index out of range");
        if ((int64_t)LCTX * D > cap) return fail("tap buffer too small");
        const int v = which == NASR_TAP_V_CACHE ? 1 : 0;
        std::vector<char> raw((size_t)KVC * D * e->esz);
        HIPCHK(hipMemcpy(raw.data(), (char *)e->kv_pool[index] + (slot * 2 + v) * KVC * D * e->esz, raw.size(), hipMemcpyDeviceToHost));
        for (int j = 0; j < LCTX; j++) {   // logical order: ring[(kv_head + j) % KVC]
            const int ring = (s->kv_head + j) % KVC;
            for (int d = 0; d < D; d++) {
                if (e->bf16) {
                    uint32_t u = (uint32_t)((const uint16_t *)raw.data())[(size_t)ring * D + d] << 16;
                    memcpy(&out[(size_t)j * D + d], &u, 4);
                } else out[(size_t)j * D + d] = ((const float *)raw.data())[(size_t)ring * D + d];
            }
        }
        return (int64_t)LCTX * D;
    }
    case NASR_TAP_CONV_CACHE: {
        if (index < 0 || index >= e->hp.n_layers) return fail("layer index out of range");
        const size_t ks1 = (size_t)e->hp.kernel_size - 1;
        if ((int64_t)(ks1 * D) > cap) return fail("tap buffer too small");
        HIPCHK(hipMemcpy(out, e->cc_pool[index] + (slot * 2 + s->cc_par) * ks1 * D, ks1 * D * 4, hipMemcpyDeviceToHost));
        return (int64_t)(ks1 * D);
    }
    case NASR_TAP_DEC_STATE: {
        if (cap < 4 * HID + 1) return fail("tap buffer too small");
        DecCtrl c;
        HIPCHK(hipMemcpy(&c, e->ctrl + slot, sizeof(c), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(out, e->dec_h + (slot * 2 + c.cur) * 2 * HID, 2 * HID * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(out + 2 * HID, e->dec_c + (slot * 2 + c.cur) * 2 * HID, 2 * HID * 4, hipMemcpyDeviceToHost));
        out[4 * HID] = (float)c.prev_token;
        return 4 * HID + 1;
    }
    }
    return fail("unknown tap %d", which);
}

extern "C" int nasr_engine_profile(nasr_engine *e, int enable) {
    ApiGuard api_guard;
    if (!e) return fail("null engine");
    HIPCHK(hipSetDevice(e->device));
    if (pipe_drain(e)) return -1;
    prof_flush(e);
    if (enable) for (auto &s : e->prof.stats) { s.launches = 0; s.total_ms = 0; s.bytes = 0; s.flops = 0; }
    e->prof.on = enable != 0;
    return 0;
}

extern "C" int nasr_engine_profile_read(nasr_engine *e, nasr_kernel_stat *out, int cap) {
    ApiGuard api_guard;
    if (!e) return fail("null engine");
    HIPCHK(hipSetDevice(e->device));
    prof_flush(e);
    int n = 0;
    for (auto &s : e->prof.stats) {
        if (s.launches == 0) continue;
        if (out && n < cap) out[n] = s;
        n++;
    }
    return n;
}

extern "C" void *nasr_engine_hip_stream(nasr_engine *e) { return e ? (void *)e->st : nullptr; }

// hands the LAST of the engine's side-by-side streams (its hardware queue) to another GPU client of the process, e.g. the
// diarization side-car (nasr_diar_set_stream): the engine keeps one stream fewer (one encoder piece fewer at most) and still
// owns the stream -- the borrower must be done with it before nasr_engine_destroy
extern "C" int nasr_engine_lend_stream(nasr_engine *e, void **out) {
    ApiGuard api_guard;
    if (!e || !out) return fail("null argument");
    HIPCHK(hipSetDevice(e->device));
    if (pipe_drain(e)) return -1;
    if (!e->pipe_ready) {
        if (pick_lanes(e)) return -1;
        e->pipe_ready = true;
        release_lanes(e);
    }
    if (e->n_lanes < 2) return fail("no side-by-side stream to lend (the engine found %d)", e->n_lanes);
    e->n_lanes--;
    e->lent.push_back(e->lane[e->n_lanes]);          // destroyed with the engine
    *out = (void *)e->lane[e->n_lanes];
    e->lane[e->n_lanes] = nullptr;
    return 0;
}

extern "C" int nasr_device_alloc(nasr_engine *e, void **out, int64_t bytes) {
    ApiGuard api_guard;
    if (!e || !out || bytes <= 0) return fail("bad argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMalloc(out, (size_t)bytes));
    return 0;
}
extern "C" int nasr_device_free(nasr_engine *e, void *p) {
    ApiGuard api_guard;
    if (!e) return fail("null engine");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipFree(p));
    return 0;
}
extern "C" int nasr_device_upload(nasr_engine *e, void *dst, const void *src, int64_t bytes) {
    ApiGuard api_guard;
    if (!e || !dst || !src) return fail("bad argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int nasr_engine_synchronize(nasr_engine *e) {
    ApiGuard api_guard;
    if (!e) return fail("null engine");
    HIPCHK(hipSetDevice(e->device));
    if (pipe_drain(e)) return -1;                       // the decode graph in flight, if any (its tokens stay queued)
    HIPCHK(hipStreamSynchronize(e->st));
    return 0;
}

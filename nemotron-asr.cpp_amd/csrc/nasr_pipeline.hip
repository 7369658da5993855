// nasr_pipeline.hip -- hipGraph replay of the steady-state step, the lanes (HIP streams measured to overlap), pipelined steps (engine
// option "pipeline" = 1..4) and the grouped pipeline (= 8), the bounded graph cache.
#include "nasr_engine_priv.h"

// ---- hipGraph replay of the steady-state step ----------------------------------------------------
// Eligible when every stream of the call receives one sub-push that completes exactly one chunk
// (the normal streaming cadence: 1280*(1+R) samples per push).  The launch sequence is then fixed
// for a given (B, T): descriptors live at fixed addresses and are refreshed by memcpy nodes.

// Descriptors of a graph step, packed so that ONE memcpy node refreshes them: [RowDesc B][PcmDesc B][meta 2B][RowDesc B*G]
namespace nasr_eng {
GraphDescLayout graph_desc_layout(int B, int G) {
    GraphDescLayout l;
    l.rows = 0;
    l.pcm = l.rows + (size_t)B * sizeof(RowDesc);
    l.meta = l.pcm + (size_t)B * sizeof(PcmDesc);
    l.vrows = (l.meta + (size_t)2 * B * sizeof(int) + 15) & ~(size_t)15;
    l.total = l.vrows + (G > 1 ? (size_t)B * G * sizeof(RowDesc) : 0);
    return l;
}

int build_step_graph(nasr_engine *e, int B, int T, int R, int G, hipGraphExec_t *out) {
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

int pipe_blind_iterations(int frames, int cap) {     // cap: engine option "decode_graph_iterations" (12)
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
double spin_us(hipStream_t a, hipStream_t b) {          // wall time of one spin kernel on a (b == nullptr) or one on each
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
int pick_lanes(nasr_engine *e) {
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

void release_lanes(nasr_engine *e) {
    if (!e->pipe_ready) return;                                // applied when the lanes are picked
    const int keep = std::min((int)nasr_engine::MAXSEG, std::max(1, e->max_lanes) + 1);      // max_lanes pieces + the decode stream
    for (int k = keep; k < nasr_engine::MAXSEG; k++)
        if (e->lane[k]) { hipStreamSynchronize(e->lane[k]); hipStreamDestroy(e->lane[k]); e->lane[k] = nullptr; }
    e->n_lanes = std::min(e->n_lanes, keep);
}

// streams, events and the buffers of slot p (allocated when first used: E + 1 slots for E encoder pieces)
int ensure_pipe(nasr_engine *e, int p) {
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
int build_pipe_graphs(nasr_engine *e, int p, int B, int T, int R, int G, int nseg, hipGraphExec_t *seg_out, hipGraphExec_t *dec_out) {
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
                    if (!(e->opt_ablate & 16)) launch_mel(mp, mp.max_frames * HOP + NFFT, e->st);
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
            for (int k = 0, n = (e->opt_ablate & 8) ? 0 : pipe_blind_iterations(T * G, e->opt_decode_graph_iters); k < n; k++) launch_decode_iter(dp, it++, cs);
            hipLaunchKernelGGL(k_collect, dim3(B), dim3(64), 0, cs, P.g_dmeta, P.g_dmeta + B, B, e->ctrl, e->tok_ring, P.collect_dev, COLLECT_STRIDE, e->n_active);
            HIPCHK(hipMemcpyAsync(P.gh_collect, P.collect_dev, ((size_t)B * (1 + COLLECT_STRIDE) + 1) * sizeof(int), hipMemcpyDeviceToHost, cs));
            return 0;
        })) return -1;
    return 0;
}

// next encoder piece of the step in slot p: queued on its lane behind the previous piece's event.  That wait is short in
// steady state (the previous piece was launched a whole call earlier) -- unlike the decode graph it is not parked for long.
int pipe_advance(nasr_engine *e, int p) {
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

int pipe_finish_launch(nasr_engine *e, int p) {
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

int pipe_finish(nasr_engine *e, int p) {
    nasr_engine::Pipe &P = e->pipe[p];
    if (P.stage == 0) return 0;
    if (pipe_finish_launch(e, p)) return -1;
    P.dec_launched = false;
    const int B = (int)P.streams.size(), TS = P.T * P.G;
    hipStream_t ds = dec_stream(e, P);
    { HostTimer ht(e->host_wait_s); HIPCHK(hipEventSynchronize(P.dec_done)); }
    int *gh_active = P.gh_collect + (size_t)B * (1 + COLLECT_STRIDE);      // k_collect appends n_active to its records
    if (*gh_active < 0) {          // reported once: the flag is cleared so that the steps after this one are judged on their own (advisor, round 5: it was sticky)
        hipMemsetAsync(e->n_active + 3, 0, sizeof(int), e->st);
        return fail("a chained GEMM launch gave up waiting for its head workgroups (GemmParams::chain): results of this step are invalid");
    }
    if (*gh_active != 0 && !(e->opt_ablate & 8)) {
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
int pipe_drain(nasr_engine *e) {
    if (!e->pipe_ready) return 0;
    if (gp_drain(e)) return -1;
    for (int64_t q = e->pipe_seq - nasr_engine::LSLOT; q < e->pipe_seq; q++) {
        if (q < 0) continue;
        const int p = (int)(q % nasr_engine::LSLOT);
        if (e->pipe[p].stage != 0 && e->pipe[p].seq == q && pipe_finish(e, p)) return -1;
    }
    return 0;
}

int pipe_step(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *pcm_dev, const int32_t *n_samples, int G,
                     int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens) {
    const int T = streams[0]->T, R = streams[0]->R, shift = 8 * T;
    if (ensure_pipe(e, (int)(e->pipe_seq % nasr_engine::LSLOT))) return -1;     // also picks the lanes
    int nseg = std::max(1, std::min({e->opt_pipeline, e->n_lanes, e->max_lanes, (int)nasr_engine::MAXSEG, (int)e->hp.n_layers}));
    // From 3 584 rows a step's GEMMs fill the chip by themselves, and every further lane is another GEMM's working set in the same L2s: fewer
    // pieces are faster than four.  ms per step at 2 / 3 / 4 pieces, R = 13 (with the pipelined steps' 224-row tiles): 512 streams 14.57 / 14.32 /
    // 14.61, 384 streams 11.51 / 10.95 / 11.06, 256 streams 7.99 / 7.93 / 7.97 (before those tiles two pieces were best: 15.29 / 15.60 / 15.97);
    // below 3 584 rows four win (192 streams 6.38 / 6.14 / 6.22 before the tiles, 128 streams 4.55 / 4.33 / 4.31, 64 streams 2.95 / 2.57 / 2.40) --
    // Round 5 (GEMM launches of less CU-time: k_gemm_wide2, the 224 x 256 tiles from 32 of them): from 5 600 rows instead of 3 584 -- four pieces against three, same box, ms per step:
    // 256 streams x R = 13 6.92-6.98 / 7.19-7.22, 320 streams 8.77-8.79 / 8.86-8.87, 384 streams 10.37-10.41 / 10.41-10.43, 512 streams 13.84-13.88 / 13.61 (option "large_step_rows").
    // profiles/r4_tile_order.md.  "pipeline" = E stays the upper bound, engine option "large_step_pieces" (default 3) is this one; a step's
    // tokens come back at most E calls later whatever the piece count.
    if ((long)B * T * G >= (e->opt_large_step_rows > 0 ? e->opt_large_step_rows : 5600) && e->opt_large_step_pieces > 0) nseg = std::min(nseg, e->opt_large_step_pieces);
    // steps in flight order their layers through the lanes (piece k of every step on lane k): a step cut differently from the ones before it
    // must not overtake them -- complete those first (a call population that crosses 3 584 rows, or an option change; never the steady state)
    if (e->pipe_last_nseg != nseg) {
        if (e->pipe_last_nseg && pipe_drain(e)) return -1;
        e->pipe_last_nseg = nseg;
    }
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
bool gp_eligible(const nasr_engine *e, int B, int T, int G) {
    constexpr int fuse_max_m = FUSE_MAX_M;
    return e->opt_pipeline == nasr_engine::GP_S && e->bf16 && e->opt_fused && !e->debug && B * T * G <= std::min(2, fuse_max_m) &&
           e->hp.n_layers % nasr_engine::GP_S == 0 && e->hp.num_prompts == 0 && e->n_lanes >= nasr_engine::GP_C && e->max_lanes >= nasr_engine::GP_C;
}

// everything chain c does in one call; slot_of_stage[j] = slot of the step at stage j, or -1.  Launches go to e->st (the caller
// captures them or has pointed e->st at the chain's stream).
int gp_enqueue_chain(nasr_engine *e, int c, const int *slot_of_stage, int B, int T, int R, int G) {
    const int nL = e->hp.n_layers, per_stage = 8 * nL / nasr_engine::GP_S;
    float *const encproj_saved = e->encproj;
    const GraphDescLayout L = graph_desc_layout(B, G);
    int rc = 0;
    if (c == 0 && slot_of_stage[0] >= 0) {                       // the newest step: descriptors, mel, subsampling
        nasr_engine::Pipe &P = e->pipe[slot_of_stage[0]];
        const RowDesc *g_rows = (const RowDesc *)(P.g_desc + L.rows), *g_vrows = (const RowDesc *)(P.g_desc + L.vrows);
        use_ws(e, e->ws[slot_of_stage[0]]);
        // no early return in this function: the workspace, encproj (and, in the callers, e->st and an open capture) are restored below
        const hipError_t he = hipMemcpyAsync(P.g_desc, P.gh, L.total, hipMemcpyHostToDevice, e->st);
        if (he != hipSuccess) rc = fail("hipMemcpyAsync (grouped pipeline descriptors) failed: %s", hipGetErrorString(he));
        if (!rc) {
            MelParams mp;
            memset(&mp, 0, sizeof(mp));
            mp.desc = (const PcmDesc *)(P.g_desc + L.pcm); mp.B = B; mp.max_frames = max_frames_per_push(T * G); mp.abuf = e->abuf; mp.last_sample = e->last_sample;
            mp.mel_ring = e->mel_ring; mp.window = e->window; mp.fbT = e->fbT; mp.fb_band = e->fb_band; mp.cos_t = e->cos_t; mp.sin_t = e->sin_t;
            launch_mel(mp, mp.max_frames * HOP + NFFT, e->st);
            rc = enqueue_encoder(e, g_rows, G > 1 ? g_vrows : g_rows, nullptr, B, T, R, G, 0, 1, 1);
        }
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

int gp_capture(nasr_engine *e, hipGraphExec_t *out, const std::function<int()> &body, hipStream_t st) {
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
int gp_decode_graph(nasr_engine *e, int p, int B, int T, int G, hipGraphExec_t *out) {
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
int gp_call(nasr_engine *e, int new_slot, int64_t key, int B, int T, int R, int G) {
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
int gp_finish_decode(nasr_engine *e) {
    if (e->gp_dec_pending < 0) return 0;
    nasr_engine::Pipe &P = e->pipe[e->gp_dec_pending];
    e->gp_dec_pending = -1;
    const int nB = (int)P.streams.size();
    hipStream_t ds = e->lane[nasr_engine::GP_C - 1];
    { HostTimer ht(e->host_wait_s); HIPCHK(hipEventSynchronize(P.dec_done)); }
    int *gh_active = P.gh_collect + (size_t)nB * (1 + COLLECT_STRIDE);
    if (*gh_active < 0) {          // reported once: the flag is cleared so that the steps after this one are judged on their own (advisor, round 5: it was sticky)
        hipMemsetAsync(e->n_active + 3, 0, sizeof(int), e->st);
        return fail("a chained GEMM launch gave up waiting for its head workgroups (GemmParams::chain): results of this step are invalid");
    }
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

int gp_drain(nasr_engine *e) {
    while (!e->gp_flight.empty()) {
        nasr_engine::Pipe &P = e->pipe[e->gp_flight.front().slot];
        if (gp_call(e, -1, P.key, (int)P.streams.size(), P.T, P.streams[0]->R, P.G)) return -1;
    }
    return gp_finish_decode(e);
}

int gp_step(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *pcm_dev, const int32_t *n_samples, int G,
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

}  // namespace nasr_eng

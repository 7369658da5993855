// nasr_abi.hip -- the step driver (graph step when eligible, eager sub-push loop otherwise) and the entry points of
// include/nemotron_asr_amd.h that are not life cycle: options, step / finalize / collect, counters, taps, profiling, device helpers.
#include "nasr_engine_priv.h"

extern "C" int nasr_engine_set_option(nasr_engine *e, const char *key, int value) {
    if (!e || !key) return fail("null argument");
    if (!strcmp(key, "fused")) e->opt_fused = value != 0;
    else if (!strcmp(key, "graph")) e->opt_graph = value != 0;
    else if (!strcmp(key, "graph_cache")) { if (value < 1) return fail("graph_cache must be >= 1"); e->opt_graph_cache = value; }
    else if (!strcmp(key, "multichunk")) e->opt_multichunk = value != 0;
    else if (!strcmp(key, "large_step_pieces")) { if (value < 0 || value > nasr_engine::MAXSEG) return fail("large_step_pieces must be 0 .. %d", (int)nasr_engine::MAXSEG); e->opt_large_step_pieces = value; }
    else if (!strcmp(key, "t64_tiles")) { if (value < 0) return fail("t64_tiles must be >= 0"); e->opt_t64_tiles = value; }      // like "fused": set before the first step
    else if (!strcmp(key, "tile_bands")) e->opt_tile_bands = value;
    else if (!strcmp(key, "wide_tiles")) e->opt_wide_tiles = value;              // like "fused": set before the first step
    else if (!strcmp(key, "persistent_gemm")) e->opt_persist_gemm = value != 0;      // like "fused": set before the first step
    else if (!strcmp(key, "gemm_cores")) { if (value < -1 || value > 1) return fail("gemm_cores must be -1, 0 or 1"); e->opt_gemm_cores = value; }
    else if (!strcmp(key, "decode_graph_iterations")) { if (value < 1) return fail("decode_graph_iterations must be >= 1"); e->opt_decode_graph_iters = value; }
    else if (!strcmp(key, "decode_lane")) {
        // read by pick_lanes() only, and the lanes are picked once: later the option would be a silent no-op (round-4 advisor)
        if (e->pipe_ready) return fail("decode_lane must be set before the first pipelined step (the lanes are already picked)");
        e->opt_decode_lane = value != 0;
    }
    else if (!strcmp(key, "wide_min_tiles")) e->opt_wide_min_tiles = value;
    else if (!strcmp(key, "large_step_rows")) e->opt_large_step_rows = value;
    else if (!strcmp(key, "wide_min_rows")) e->opt_wide_min_rows = value;
    else if (!strcmp(key, "gemm_prio")) e->opt_gemm_prio = value;                    // probe: GemmParams::prio (measurement only)
    else if (!strcmp(key, "epilogue16")) e->opt_epilogue16 = value != 0;            // like "fused": set before the first step
    else if (!strcmp(key, "dwconv_stream")) e->opt_dwconv_stream = value != 0;      // like "fused": set before the first step
    else if (!strcmp(key, "chain")) { if (value < 0 || value > 2) return fail("chain must be 0, 1 or 2"); e->opt_chain = value; }      // like "fused": set before the first step
    else if (!strcmp(key, "split_tasks")) e->opt_split_tasks = value;                // like "fused": set before the first step
    else if (!strcmp(key, "resid_epilogue")) { if (value < 0 || value > 2) return fail("resid_epilogue must be 0, 1 or 2"); e->opt_resid_epilogue = value; }      // like "fused": set before the first step
    else if (!strcmp(key, "ablate")) e->opt_ablate = value;          // measurement only (see the header); before the first step
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

// returns 1 if the step was executed through the graph, 0 if not eligible, <0 on error
namespace nasr_eng {
int try_graph_step(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *pcm_dev,
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
    if (*gh_active < 0) {          // reported once: the flag is cleared so that the steps after this one are judged on their own (advisor, round 5: it was sticky)
        hipMemsetAsync(e->n_active + 3, 0, sizeof(int), e->st);
        return fail("a chained GEMM launch gave up waiting for its head workgroups (GemmParams::chain): results of this step are invalid");
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
int push_piece(nasr_engine *e, nasr_stream *const *streams, int B, const int16_t *const *base, const int32_t *n_samples,
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

}  // namespace nasr_eng
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
        // the block is reused every fourth call: its last copy must have been executed (round-3 advisor: the grouped pipeline keeps
        // eight calls in flight and nothing ordered the host's memcpy behind copy n - 4; in steady state the event is long complete)
        if (pin.pending) { HIPCHK(hipEventSynchronize(pin.copied)); pin.pending = false; }
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
        if (total > 0) {
            HIPCHK(hipMemcpyAsync(e->pcm_stage, pin.p, total * 2, hipMemcpyHostToDevice, e->st));
            if (!pin.copied) HIPCHK(hipEventCreateWithFlags(&pin.copied, hipEventDisableTiming));
            HIPCHK(hipEventRecord(pin.copied, e->st));
            pin.pending = true;
        }
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

// host-side counters.  Walks the graph caches without a lock: call it from the thread that steps the engine (as every entry point that
// takes an engine or a stream: one thread per engine; the server's worker prints them at exit)
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

// host mirror only: no pipeline drain, no stream synchronisation, no copy (the per-call path of a server)
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
            if (j < LCTX - s->valid_len) {     // not cached yet: the reference's tensor holds its initial zeros there (src/nemo-stream.cpp:320-325); here the
                for (int d = 0; d < D; d++) out[(size_t)j * D + d] = 0.0f;      // ring may hold an earlier stream's rows, which no kernel ever weighs (the mask)
                continue;
            }
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

// test hook: every K/V ring row of the stream's slot, in every layer, := value (see the header)
extern "C" int nasr_stream_debug_fill_kv(nasr_stream *s, float value) {
    ApiGuard api_guard;
    if (!s) return fail("null stream");
    nasr_engine *e = s->e;
    HIPCHK(hipSetDevice(e->device));
    if (pipe_drain(e)) return -1;
    const size_t n = (size_t)2 * KVC * D;
    std::vector<char> host(n * e->esz);
    for (size_t i = 0; i < n; i++) {
        const float v = (i & 1) ? -value : value;
        if (e->bf16) { uint32_t u; memcpy(&u, &v, 4); ((uint16_t *)host.data())[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
        else ((float *)host.data())[i] = v;
    }
    HIPCHK(hipStreamSynchronize(e->st));
    for (int l = 0; l < e->hp.n_layers; l++)
        HIPCHK(hipMemcpy((char *)e->kv_pool[l] + (size_t)s->slot * n * e->esz, host.data(), host.size(), hipMemcpyHostToDevice));
    return 0;
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
    e->lent.push_back(e->lane[e->n_lanes]);          // destroyed with the engine, unless a borrower still holds it then
    lent_stream_register(e->lane[e->n_lanes]);
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


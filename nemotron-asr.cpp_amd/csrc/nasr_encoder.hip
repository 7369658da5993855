// nasr_encoder.hip -- the chunk step: launch sequences of the encoder (subsampling, 24 cached conformer layers in their small-M and
// large-M forms, prompt fusion, joint.enc) and of the decode, token collection (reference src/nemo-stream.cpp:336-690, :840-930).
#include "nasr_engine_priv.h"

// ---- the chunk step: encoder + decode for the rows that have a full chunk buffered ----------------
namespace nasr_eng {
double gemm_bytes(const nasr_engine *e, int M, int N, int K, int wesz) {
    return (double)N * K * wesz + (double)M * K * e->esz + (double)M * N * 4;
}

int run_gemm(nasr_engine *e, GemmParams &g, bool f32_weights, const char *tag) {
    (void)tag;
    const bool use_bf16 = e->bf16 && !f32_weights;
    const char *name = !use_bf16 ? "k_gemm_f32" : (g.M <= gemm_skinny_max_m() ? "k_gemm_skinny" : "k_gemm_tiled");
    ProfScope ps(e, name, gemm_bytes(e, g.M, g.N, g.K, use_bf16 ? 2 : 4), 2.0 * g.M * g.N * g.K);
    g.coresident = e->opt_gemm_cores == 0 ? 3 : e->opt_gemm_cores == 1 ? 2 : (e->gemm_coresident ? 1 : 0);
    g.f32_fma_tile = e->opt_f32_mfma ? 0 : 1;
    g.no_persist = e->opt_persist_gemm ? 0 : 1;
    g.no_wide = e->opt_wide_tiles ? 0 : 1;
    g.wide_rows = e->opt_wide_tiles == 256 ? 256 : e->opt_wide_tiles == 3 ? 2 : 0;
    g.tile_bands = e->opt_tile_bands < 0 ? 0 : e->opt_tile_bands == 0 ? 2 : 1;
    g.t64_tiles_p1 = e->opt_t64_tiles + 1;
    g.narrow_stores = e->opt_epilogue16 ? 0 : 1;
    g.prio = e->opt_gemm_prio;
    g.wide_min_tiles = e->opt_wide_min_tiles;
    g.wide_min_rows = e->opt_wide_min_rows;
    if (e->opt_ablate & 32) return 0;
    if (use_bf16) launch_gemm_bf16(g, e->st);
    else launch_gemm_f32(g, e->st);
    return 0;
}

// residual GEMM: part = A.W^T (split-K), followed by k_post
int pick_splits(const nasr_engine *e, int M, int N, int K) {
    if (!e->bf16) return 1;
    const bool skinny = M <= gemm_skinny_max_m();
    int tasks = skinny ? (N / 16) * ((M + 63) / 64) : (N / gemm_tile_n(M, N, EPI_PART_F32, e->opt_t64_tiles + 1)) * ((M + 127) / 128);
    // partial traffic grows with the split factor, and with pipelined steps the CUs a launch leaves idle run another chain's
    // kernels: four splits only up to 40 tiles (three lanes, R = 13: 12 / 16 streams = 32 tiles 1.15 / 1.23 ms with 4 splits
    // against 1.23 / 1.30 with 2; 24 streams = 48 tiles 1.53 vs 1.50; 32 streams = 64 tiles 1.82 vs 1.68; 64 streams = 112 tiles:
    // 2 splits 2.76, 1 split 2.75, 4 splits 3.03)
    // Round 5: one slice from 200 tiles of 128 x 128 (256 streams x R = 13: 224 tiles, 7.76 -> 7.48 ms per pipelined step): that many workgroups fill the chip
    // by themselves, and a GEMM that owns its tiles' whole K sums adds to the residual stream in its own epilogue -- no partial slabs, k_post is the
    // LayerNorm alone.  Counted in 128 x 128 tiles whatever the tile the launcher takes, so that every kernel variant sums in the same order.
    if (!skinny && (N / 128) * ((M + 127) / 128) >= (e->opt_split_tasks > 0 ? e->opt_split_tasks : 200)) return 1;
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
int run_layers_fused(nasr_engine *e, const RowDesc *rows, int B, int T, int G, int k0, int k1, std::vector<FusedParams> *rec) {
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
int enqueue_encoder(nasr_engine *e, const RowDesc *rows, const RowDesc *vrows, const int *tap_slots, int B, int T, int R, int G, int seg,
                           int nseg, int part) {
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
    if (front && !(e->opt_ablate & 16)) {
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
        // Round 5, chained launches: a k_post that is due is held back (`pend`) and rides as the HEAD PHASE of the GEMM that reads its rows
        // (GemmParams::chain; W1, QKV, pw1 -- same arithmetic, same bits, one launch and one dependent boundary fewer); if the next launch is
        // not such a GEMM, or the piece ends, it is launched on its own as before.
        const bool chain = e->bf16 && !e->debug && !e->prof.on && e->opt_chain && (e->opt_chain == 2 || e->gemm_coresident) && M >= 769 && M < 1344;
        PostParams pend;
        bool has_pend = false;
        double pend_bytes = 0;
        auto flush_post = [&]() {
            if (!has_pend) return;
            has_pend = false;
            if (e->opt_ablate & 1) return;
            ProfScope ps(e, "k_post", pend_bytes);
            launch_post(pend, st);
        };
        auto post = [&](const PostParams &q, double bytes) {
            pend = q; pend_bytes = bytes; has_pend = true;
            if (!chain) flush_post();
        };
        auto gemm_a = [&](GemmParams &gp, const char *tag) {          // a GEMM whose A operand is e->a, the rows of the k_post before it
            if (has_pend && !(e->opt_ablate & 1) && gemm_chain_ok(gp.M, gp.N, gp.K, gp.splits)) {
                gp.chain.post = pend;
                gp.chain.head_rows = 4;
                gp.chain.head_wgs = (((M + 3) / 4) + 7) & ~7;          // a multiple of 8: the tiles behind keep their XCDs
                gp.chain.flags = e->chain_flags;
                gp.chain.error = (unsigned *)(e->n_active + 3);
                has_pend = false;
            } else flush_post();
            run_gemm(e, gp, false, tag);
        };
        if (front) {
            PostParams pp;
            memset(&pp, 0, sizeof(pp));
            pp.x = e->x; pp.M = M; pp.ln2_w = e->L[0].ln_ff1_w; pp.ln2_b = e->L[0].ln_ff1_b; pp.a_out = e->a; pp.act_bf16 = act;
            post(pp, (double)M * D * (4 + e->esz));
        }

        const int nL = e->hp.n_layers, ks = e->hp.kernel_size;
        for (int l = l0; l < l1; l++) {
            LayerW &L = e->L[l];
            // a residual GEMM (N = 1024): x += scale * A.W^T.  Round 5: where one workgroup owns the complete K sum of a tile (no split-K, or the
            // welded two-slice form k_gemm_t64w) the GEMM adds to x in its own epilogue and the k_post behind it is left with the LayerNorm
            // (returns 0 splits for it); otherwise split-K partial slabs + k_post as in rounds 1-4.  Same bits either way.
            auto resid_gemm = [&](const void *A, int lda, void *w, int K, float scale, const char *tag) -> int {
                GemmParams a;
                memset(&a, 0, sizeof(a));
                a.A = A; a.W = w; a.M = M; a.N = D; a.K = K; a.lda = lda; a.splits = pick_splits(e, M, D, K);
                // without split-K the fold costs nothing (same kernel, no slab): always; the welded two-slice kernel is two co-resident workgroups in one --
                // what pipelined steps run anyway, 12 % slower than k_gemm_t64<4> on 224 CUs when the step is alone on the chip: pipelined steps only
                const bool fold = e->bf16 && e->opt_resid_epilogue && (a.splits == 1 || e->opt_resid_epilogue == 2 || e->gemm_coresident) && gemm_resid_foldable(M, D, K, a.splits, e->opt_t64_tiles + 1);
                if (fold) { a.epi = EPI_RESID_F32; a.out_f32 = e->x; a.resid = e->x; a.resid_scale = scale; a.ldo = D; }
                else { a.epi = EPI_PART_F32; a.out_f32 = e->part; a.ldo = D; }
                run_gemm(e, a, false, tag);
                return fold ? 0 : a.splits;
            };
            auto ffn = [&](void *w1, void *w2, const float *nln_w, const float *nln_b, bool last) {
                GemmParams a;
                memset(&a, 0, sizeof(a));
                a.A = e->a; a.W = w1; a.M = M; a.N = FF; a.K = D; a.lda = D; a.splits = 1;
                a.epi = EPI_SILU_ACT; a.out_act = e->hbuf; a.ldo_act = FF;
                gemm_a(a, "ffn_w1");
                a.splits = resid_gemm(e->hbuf, FF, w2, FF, 0.5f, "ffn_w2");
                PostParams q;
                memset(&q, 0, sizeof(q));
                q.x = e->x; q.M = M; q.part = e->part; q.splits = a.splits; q.scale = 0.5f;   // :633-634
                q.a_out = e->a; q.act_bf16 = act;
                if (last) { q.ln_out = 1; q.ln1_w = L.ln_out_w; q.ln1_b = L.ln_out_b; }       // :687
                q.ln2_w = nln_w; q.ln2_b = nln_b;
                post(q, (double)M * D * (8 + 4 * a.splits + e->esz));
            };
            // 1. FFN1 (:631-634) -> a = LN_att(x)
            ffn(L.ff1_w1, L.ff1_w2, L.ln_att_w, L.ln_att_b, false);
            // 2. attention (:637-643)
            memset(&g, 0, sizeof(g));
            g.A = e->a; g.W = L.wqkv; g.M = M; g.N = 3 * D; g.K = D; g.lda = D; g.splits = 1;
            g.epi = EPI_QKV; g.q_out = e->q; g.kv_pool = e->kv_pool[l]; g.kv_slot_stride = (int64_t)2 * KVC * D;
            g.rows = rows; g.T = TS;
            gemm_a(g, "qkv");
            {
                AttnParams ap;
                memset(&ap, 0, sizeof(ap));
                ap.q = e->q; ap.kv_pool = e->kv_pool[l]; ap.kv_slot_stride = (int64_t)2 * KVC * D; ap.act_bf16 = act;
                ap.posproj = L.posproj[T]; ap.bias_u = L.bias_u; ap.bias_v = L.bias_v; ap.rows = rows; ap.B = B; ap.T = T; ap.TS = TS;
                ap.ctx_out = e->ctx; ap.ablate = (e->opt_ablate >> 6) & 3;
                const int KV = LCTX + T;
                ProfScope ps(e, "k_attention", (double)B * (2.0 * KV + KV + T - 1) * D * e->esz, 2.0 * B * T * KV * D * 3);
                if (!(e->opt_ablate & 2)) launch_attention(ap, st);
            }
            g.splits = resid_gemm(e->ctx, D, L.wo, D, 1.0f, "attn_out");
            {
                PostParams q;
                memset(&q, 0, sizeof(q));
                q.x = e->x; q.M = M; q.part = e->part; q.splits = g.splits; q.scale = 1.0f;
                q.ln2_w = L.ln_conv_w; q.ln2_b = L.ln_conv_b; q.a_out = e->a; q.act_bf16 = act;
                post(q, (double)M * D * (8 + 4 * g.splits + e->esz));
            }
            // 3. conv module (:646-679)
            memset(&g, 0, sizeof(g));
            g.A = e->a; g.W = L.pw1; g.M = M; g.N = 2 * D; g.K = D; g.lda = D; g.splits = 1;
            g.epi = EPI_GLU; g.out_f32 = e->glu; g.ldo = D;
            gemm_a(g, "pw1");
            {
                ConvParams cp;
                memset(&cp, 0, sizeof(cp));
                cp.glu = e->glu; cp.cc_pool = e->cc_pool[l]; cp.cc_slot_stride = (int64_t)2 * (ks - 1) * D;
                cp.dw = L.dw; cp.ln_w = L.cln_w; cp.ln_b = L.cln_b; cp.rows = rows; cp.B = B; cp.T = TS; cp.ks = ks;
                cp.c_out = e->cbuf; cp.act_bf16 = act; cp.stream_form = e->opt_dwconv_stream ? 1 : 0;
                ProfScope ps(e, "k_dwconv", (double)M * D * (4 + e->esz) + (double)B * 2 * (ks - 1) * D * 4, 2.0 * M * D * ks);
                if (!(e->opt_ablate & 4)) launch_dwconv(cp, st);
            }
            g.splits = resid_gemm(e->cbuf, D, L.pw2, D, 1.0f, "pw2");
            {
                PostParams q;
                memset(&q, 0, sizeof(q));
                q.x = e->x; q.M = M; q.part = e->part; q.splits = g.splits; q.scale = 1.0f;
                q.ln2_w = L.ln_ff2_w; q.ln2_b = L.ln_ff2_b; q.a_out = e->a; q.act_bf16 = act;
                post(q, (double)M * D * (8 + 4 * g.splits + e->esz));
            }
            // 4. FFN2 (:682-685) + norm_out (:687); then the next layer's first LayerNorm
            const bool has_next = l + 1 < nL;
            ffn(L.ff2_w1, L.ff2_w2, has_next ? e->L[l + 1].ln_ff1_w : nullptr, has_next ? e->L[l + 1].ln_ff1_b : nullptr, true);
            if (e->debug) { flush_post(); if (tap_copy(e->tap_layers, (size_t)nL * TMAX * D, (size_t)l * TMAX * D)) return -1; }
        }
        flush_post();          // the piece ends: the last k_post has no GEMM of this piece behind it
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

void make_dec_params(nasr_engine *e, const RowDesc *rows, int B, int T, DecParams &dp) {
    memset(&dp, 0, sizeof(dp));
    dp.rows = rows; dp.B = B; dp.T = T; dp.ctrl = e->ctrl; dp.h = e->dec_h; dp.c = e->dec_c; dp.encproj = e->encproj;
    dp.embed = e->embed;
    for (int i = 0; i < 2; i++) { dp.w_ih[i] = e->w_ih[i]; dp.w_hh[i] = e->w_hh[i]; dp.b_ih[i] = e->b_ih[i]; dp.b_hh[i] = e->b_hh[i]; }
    dp.pred_w = e->pred_w; dp.pred_b = e->pred_b; dp.out_w = e->out_w; dp.out_b = e->out_b;
    dp.predg = e->predg; dp.key = e->key; dp.n_active = e->n_active; dp.n_dirty = e->n_active + 1; dp.n_rows = e->n_active + 2;
    dp.dlist = e->dlist; dp.rowmap = e->rowmap; dp.tok_ring = e->tok_ring; dp.tok_frame = e->tok_frame;
}

void enqueue_decode_iters(nasr_engine *e, const DecParams &dp, int B, int n, int &it, hipStream_t st) {
    ProfScope ps(e, "k_dec_iter", (double)n * (4.0 * 4 * HID * HID * 4 + (double)JNT * HID * 4 + (double)VOCAB * JNT * 4),
                 (double)n * 2.0 * B * (4.0 * 4 * HID * HID + JNT * HID + VOCAB * JNT));
    for (int k = 0; k < n; k++) launch_decode_iter(dp, it++, st ? st : e->st);
}

// host mirror of the stream manager bookkeeping after a chunk (:1085, :1189-1195)
void chunk_bookkeeping(nasr_stream *s, int row) {
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

void fill_row_desc(RowDesc &rd, const nasr_stream *s, int n_dec) {
    rd.slot = s->slot; rd.valid_len = s->valid_len; rd.kv_head = s->kv_head;
    rd.mel_start = s->mel_start; rd.cc_par = s->cc_par; rd.n_dec = n_dec;
    rd.prompt = s->prompt; rd.pad = 0;
}

int run_chunk(nasr_engine *e, const std::vector<nasr_stream *> &rows_s, const std::vector<int> &n_dec) {
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
int drain_chunks(nasr_engine *e, nasr_stream *const *streams, int B) {
    for (;;) {
        std::vector<nasr_stream *> ready;
        for (int b = 0; b < B; b++)
            if (streams[b]->mel_count >= PRE_CACHE + 8 * streams[b]->T) ready.push_back(streams[b]);
        if (ready.empty()) return 0;
        std::vector<int> nd(ready.size(), ready[0]->T);
        if (run_chunk(e, ready, nd)) return -1;
    }
}

int validate_batch(nasr_engine *e, nasr_stream *const *streams, int B) {
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
    if (b == 0 && threadIdx.x == 0 && n_active) out[(size_t)B * (1 + stride)] = n_active[3] ? -1 : *n_active;   // rides along in the same D2H copy; -1: a chained GEMM launch gave up waiting (ChainParams::error = n_active + 3)
    const int slot = slots[b];
    const int n_tok = ctrl[slot].n_tok, rd = tok_read[b];
    const int n_new = n_tok - rd;
    if (threadIdx.x == 0) out[(size_t)b * (1 + stride)] = n_new;
    for (int i = threadIdx.x; i < n_new && i < stride; i += blockDim.x)
        out[(size_t)b * (1 + stride) + 1 + i] = tok_ring[(size_t)slot * TOK_CAP + ((rd + i) & (TOK_CAP - 1))];
}

// Tokens gathered from the device go to the stream's host queue; every token-returning entry point ends with deliver().
int consume_collect(nasr_engine *e, const int *host, nasr_stream *const *streams, int B) {
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
void deliver(nasr_stream *const *streams, int B, int32_t *const *tokens_out, const int32_t *tokens_cap, int32_t *n_tokens) {
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


int collect_tokens(nasr_engine *e, nasr_stream *const *streams, int B, int32_t *const *tokens_out,
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

int ensure_debug_buffers(nasr_engine *e) {
    if (e->tap_sub) return 0;
    const size_t S = (size_t)e->max_streams;
    e->tap_mel_cap = 128;
    HIPCHK(hipMalloc((void **)&e->tap_mel, S * e->tap_mel_cap * NMEL * 4));
    HIPCHK(hipMalloc((void **)&e->tap_sub, S * TMAX * D * 4));
    HIPCHK(hipMalloc((void **)&e->tap_layers, S * (size_t)e->hp.n_layers * TMAX * D * 4));
    HIPCHK(hipMalloc((void **)&e->tap_enc, S * TMAX * D * 4));
    return 0;
}

}  // namespace nasr_eng

// ref_shim.cpp -- C entry points over the UNMODIFIED reference sources.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with the
// reference's own files where they lie under $(REF) = /root/reference
//   src/preprocessor.cpp                (hot-path stage a-1, std-only)
//   src/reference/*.cpp                 (the reference's scalar parity oracle)
// into oracle/_ref/libnemo_ref.so by oracle/Makefile.  No reference source is copied
// into this repository; the .so is git-ignored and only travels to the GPU box as a
// built artefact.  It is used (a) by tests/golden/gen_golden.py to produce the golden
// vectors and (b) by tests that pin oracle/nasr_oracle.c when the .so is present.
//
// The reference modules expose their weights as public `const float*` members
// (src/reference/include/*.h), so the shim points them at caller memory directly.
#include <cstdint>
#include <cstring>
#include <vector>

#include "preprocessor.h"                 // $(REF)/src
// RelPositionMultiHeadAttention::rel_shift is a private member of the (unmodified) reference class; the shim's own
// translation unit sees it as public so that the golden generator can call the compiled function itself (access
// specifiers do not change the class layout; the reference's .cpp files are compiled as they are).
#define private public
#include "include/conformer_modules.h"
#undef private
#include "include/conformer_encoder.h"    // $(REF)/src/reference
#include "include/conformer_modules.h"
#include "include/conv_subsampling.h"
#include "include/greedy_decode.h"
#include "include/ops.h"
#include "include/rnnt_decoder.h"
#include "include/rnnt_joint.h"

extern "C" {

// ---- a-1: src/preprocessor.cpp --------------------------------------------------
void *ref_preproc_create(const float *fb, const float *window) {
    return nemo_preprocessor_init_from_data(fb, 128 * 257, window, 400);
}
void ref_preproc_free(void *pp) { nemo_preprocessor_free((nemo_preprocessor *)pp); }
int ref_preproc_process(void *pp, const int16_t *pcm, int n, float *out, int cap_frames) {
    std::vector<float> mel;
    size_t nf = nemo_preprocessor_process((nemo_preprocessor *)pp, pcm, (size_t)n, mel);
    if ((int)nf > cap_frames) return -1;
    if (nf) std::memcpy(out, mel.data(), nf * 128 * sizeof(float));
    return (int)nf;
}

// ---- a-2: src/reference/conv_subsampling.cpp ---------------------------------------
// w: conv0_w, conv0_b, conv2_w, conv2_b, conv3_w, conv3_b, conv5_w, conv5_b, conv6_w,
//    conv6_b, out_w, out_b
int ref_subsampling(const float *const *w, const float *mel, int n_frames, float *out) {
    nemo::ConvSubsampling s;
    s.conv0_weight = w[0]; s.conv0_bias = w[1];
    s.conv2_weight = w[2]; s.conv2_bias = w[3];
    s.conv3_weight = w[4]; s.conv3_bias = w[5];
    s.conv5_weight = w[6]; s.conv5_bias = w[7];
    s.conv6_weight = w[8]; s.conv6_bias = w[9];
    s.out_weight = w[10];  s.out_bias = w[11];
    nemo::TensorF in({1, (size_t)n_frames, 128}), o;
    std::memcpy(in.ptr(), mel, sizeof(float) * (size_t)n_frames * 128);
    s.forward(in, o);
    std::memcpy(out, o.ptr(), sizeof(float) * o.numel());
    return (int)o.shape[1];
}

// ---- a-7: RelPositionalEncoding (src/reference/conformer_modules.cpp:128-172) -------
void ref_pos_emb(int seq_len, float *out /*[2*seq_len-1][1024]*/) {
    nemo::RelPositionalEncoding pe;
    nemo::TensorF p;
    pe.get_pos_emb((size_t)seq_len, p);
    std::memcpy(out, p.ptr(), sizeof(float) * p.numel());
}

// ---- a-6: RelPositionMultiHeadAttention::rel_shift (src/reference/conformer_modules.cpp:188-240) ------
// x [heads][qlen][2*qlen-1] -> out [heads][qlen][qlen]; the input pattern of tests/test_compute.cpp:1028-1052
void ref_rel_shift(const float *x, int heads, int qlen, float *out) {
    nemo::RelPositionMultiHeadAttention a;
    const size_t pl = 2 * (size_t)qlen - 1;
    nemo::TensorF in({1, (size_t)heads, (size_t)qlen, pl}), o;
    std::memcpy(in.ptr(), x, sizeof(float) * (size_t)heads * qlen * pl);
    a.rel_shift(in, o);
    std::memcpy(out, o.ptr(), sizeof(float) * o.numel());
}

// ---- a-3..a-9: ConformerLayer::forward (src/reference/conformer_encoder.cpp:29-69) ---
// w order: norm_ff1 w,b; ff1 l1,l2; norm_att w,b; q,k,v,pos,out; bias_u,bias_v;
//          norm_conv w,b; pw1; dw (PyTorch [1024][1][9]); bn w,b; pw2;
//          norm_ff2 w,b; ff2 l1,l2; norm_out w,b        (26 pointers)
static void bind_layer(nemo::ConformerLayer &L, const float *const *w) {
    int i = 0;
    L.norm_ff1_weight = w[i++]; L.norm_ff1_bias = w[i++];
    L.ffn1.linear1_weight = w[i++]; L.ffn1.linear2_weight = w[i++];
    L.norm_attn_weight = w[i++]; L.norm_attn_bias = w[i++];
    L.self_attn.linear_q_weight = w[i++]; L.self_attn.linear_k_weight = w[i++];
    L.self_attn.linear_v_weight = w[i++]; L.self_attn.linear_pos_weight = w[i++];
    L.self_attn.linear_out_weight = w[i++];
    L.self_attn.pos_bias_u = w[i++]; L.self_attn.pos_bias_v = w[i++];
    L.norm_conv_weight = w[i++]; L.norm_conv_bias = w[i++];
    L.conv.pointwise_conv1_weight = w[i++]; L.conv.depthwise_conv_weight = w[i++];
    L.conv.batch_norm_weight = w[i++]; L.conv.batch_norm_bias = w[i++];
    L.conv.pointwise_conv2_weight = w[i++];
    L.norm_ff2_weight = w[i++]; L.norm_ff2_bias = w[i++];
    L.ffn2.linear1_weight = w[i++]; L.ffn2.linear2_weight = w[i++];
    L.norm_out_weight = w[i++]; L.norm_out_bias = w[i++];
}

void ref_conformer_layer(const float *const *w, const float *x, int T, float *out) {
    nemo::ConformerLayer L;
    bind_layer(L, w);
    nemo::RelPositionalEncoding pe;
    nemo::TensorF pos, in({1, (size_t)T, 1024}), o;
    pe.get_pos_emb((size_t)T, pos);
    std::memcpy(in.ptr(), x, sizeof(float) * (size_t)T * 1024);
    L.forward(in, pos, o);
    std::memcpy(out, o.ptr(), sizeof(float) * o.numel());
}

// module-level taps of the same layer (for finer pins)
void ref_ffn(const float *w1, const float *w2, const float *x, int T, float *out) {
    nemo::ConformerFeedForward f;
    f.linear1_weight = w1; f.linear2_weight = w2;
    nemo::TensorF in({1, (size_t)T, 1024}), o;
    std::memcpy(in.ptr(), x, sizeof(float) * (size_t)T * 1024);
    f.forward(in, o);
    std::memcpy(out, o.ptr(), sizeof(float) * o.numel());
}

void ref_layer_norm(const float *w, const float *b, const float *x, int T, float *out) {
    nemo::TensorF in({1, (size_t)T, 1024}), o;
    std::memcpy(in.ptr(), x, sizeof(float) * (size_t)T * 1024);
    nemo::layer_norm(in, w, b, 1024, 1e-5f, o);
    std::memcpy(out, o.ptr(), sizeof(float) * o.numel());
}

// ---- a-12/a-13: RNNTDecoder / RNNTJoint ----------------------------------------------
// w order: embed; w_ih0,w_hh0,b_ih0,b_hh0; w_ih1,w_hh1,b_ih1,b_hh1;
//          enc_w,enc_b,pred_w,pred_b,out_w,out_b                      (15 pointers)
static void bind_dec(nemo::RNNTDecoder &d, nemo::RNNTJoint &j, const float *const *w) {
    d.embed_weight = w[0];
    d.lstm_weight_ih[0] = w[1]; d.lstm_weight_hh[0] = w[2];
    d.lstm_bias_ih[0] = w[3];   d.lstm_bias_hh[0] = w[4];
    d.lstm_weight_ih[1] = w[5]; d.lstm_weight_hh[1] = w[6];
    d.lstm_bias_ih[1] = w[7];   d.lstm_bias_hh[1] = w[8];
    j.enc_weight = w[9];  j.enc_bias = w[10];
    j.pred_weight = w[11]; j.pred_bias = w[12];
    j.out_weight = w[13]; j.out_bias = w[14];
}

// logits for a token sequence fed from zero state against one encoder frame each:
// step i: decoder consumes tokens[i], joint(enc[i], dec_out) -> logits[i][1025]
void ref_decoder_joint_seq(const float *const *w, const int *tokens, int n, const float *enc,
                           float *logits, float *h_out, float *c_out) {
    nemo::RNNTDecoder d;
    nemo::RNNTJoint j;
    bind_dec(d, j, w);
    d.init_state(1);
    nemo::TensorF dec, e({1, 1024}), lg;
    for (int i = 0; i < n; i++) {
        d.forward_step(tokens[i], dec);
        std::memcpy(e.ptr(), enc + (size_t)i * 1024, sizeof(float) * 1024);
        j.forward(e, dec, lg);
        std::memcpy(logits + (size_t)i * 1025, lg.ptr(), sizeof(float) * 1025);
    }
    std::memcpy(h_out, d.state.h.ptr(), sizeof(float) * 2 * 640);
    std::memcpy(c_out, d.state.c.ptr(), sizeof(float) * 2 * 640);
}

// ---- a-14: GreedyDecoder::decode (src/reference/greedy_decode.cpp:5-59) ---------------
int ref_greedy(const float *const *w, const float *enc, int T, int *tokens, int cap) {
    nemo::RNNTDecoder d;
    nemo::RNNTJoint j;
    bind_dec(d, j, w);
    nemo::GreedyDecoder g;
    g.init(&d, &j);
    nemo::TensorF e({1, (size_t)T, 1024});
    std::memcpy(e.ptr(), enc, sizeof(float) * (size_t)T * 1024);
    std::vector<int> t = g.decode(e);
    int n = (int)t.size();
    for (int i = 0; i < n && i < cap; i++) tokens[i] = t[i];
    return n;
}

}  // extern "C"

// ---- f-4 front end: src/diarize_audio.cpp (ggml-free, compiled unmodified) ---------------------
#include "diarize_audio.h"
extern "C" int ref_diar_logmel(const float *audio, int n_samples, int per_feature_normalize, const float *fb,
                               const float *window, float *out /*[80][t_padded]*/, int cap_frames, int *t_valid) {
    diarize_audio_cfg cfg;
    cfg.per_feature_normalize = per_feature_normalize != 0;
    std::vector<float> mel;
    size_t tv = 0;
    const size_t tp = diarize_compute_logmel(audio, (size_t)n_samples, cfg, fb, window, mel, &tv);
    if ((int)tp > cap_frames) return -1;
    std::memcpy(out, mel.data(), mel.size() * sizeof(float));
    if (t_valid) *t_valid = (int)tv;
    return (int)tp;
}

/* diar_oracle.h -- CPU restatement of the reference's diarization forward paths.
 *
 * TEST INFRASTRUCTURE ONLY (see nasr_oracle.h): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * may use it.
 *
 * Pin status
 *   - dorc_logmel: pinned.  The reference's src/diarize_audio.cpp is ggml-free and is compiled unmodified into
 *     oracle/_ref (ref_diar_logmel in oracle/ref_shim.cpp); tests/golden holds its outputs.
 *   - dorc_vad_window / dorc_spk_embed (MarbleNet, TitaNet-L): PARITY UNPINNED.  src/diarize_vad.cpp and
 *     src/diarize_spk.cpp build ggml graphs and cannot be compiled here (ggml is an empty submodule), and the
 *     committed NeMo fixtures under tests/diarize/ need a diarize.gguf that is not in the tree.  The restatement
 *     follows the graph builders line by line (cited at every step).
 *
 * Layout: activations are [T][C] (channels innermost), as the reference feeds ggml (ne = (C, T)).
 * Weights as converted by scripts/convert_diarize_to_gguf.py:129-158: depthwise (k, ch), pointwise (out, in).
 */
#ifndef DIAR_ORACLE_H
#define DIAR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define DORC_N_MELS 80
#define DORC_N_FFT 512
#define DORC_N_BINS 257
#define DORC_WIN 400
#define DORC_HOP 160
#define DORC_VAD_WINDOW 10080   /* src/diarize_vad.h:86 */
#define DORC_VAD_T 64           /* :89 */
#define DORC_SPK_SEGMENT 24000  /* src/diarize_spk.h:102 */
#define DORC_SPK_T 160          /* :104 */
#define DORC_SPK_EMB 192        /* :106 */

typedef struct dorc_model dorc_model;

/* src/diarize_audio.cpp:136-227.  out = [80][t_padded] (mel-major, as the reference returns it); returns t_padded */
int dorc_logmel(const float *audio, int n_samples, int per_feature_normalize, const float *fb, const float *window,
                float *out, int cap_frames, int *t_valid);

dorc_model *dorc_model_create(void);
void dorc_model_free(dorc_model *m);
/* tensor names as in diarize.gguf ("vad.*" / "spk.*"); data is copied.  0 = ok, 1 = unknown name (ignored), <0 = error */
int dorc_model_set_tensor(dorc_model *m, const char *name, const float *data, long long n_elems);
/* folds the batch norms (src/diarize_vad.cpp:56-79, src/diarize_spk.cpp:60-77); 0 = ok, <0 = a tensor is missing */
int dorc_model_finalize(dorc_model *m);

/* src/diarize_vad.cpp:436-488: P(speech) of one 0.63 s window; audio holds DORC_VAD_WINDOW samples */
float dorc_vad_window(const dorc_model *m, const float *audio, int lens_samples);
/* src/diarize_vad.cpp:490-503: every window of a buffer, shift 160 samples; returns the window count */
int dorc_vad_batch(const dorc_model *m, const float *audio, int n_samples, float *probs, int cap);
/* src/diarize_spk.cpp:601-626: 192-d embedding of one 1.5 s sub-segment; audio holds DORC_SPK_SEGMENT samples */
int dorc_spk_embed(const dorc_model *m, const float *audio, int lens_samples, float *emb);

#ifdef __cplusplus
}
#endif
#endif

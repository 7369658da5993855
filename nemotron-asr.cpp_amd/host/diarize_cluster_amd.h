// diarize_cluster_amd.h -- NME-SC speaker clustering of the diarization side-car ("clustering stays on
// the CPU").  Same names, fields and defaults as the reference's src/diarize_cluster.h:14-51, so its callers and its test
// (tests/test_diarize_cluster.cpp) read unchanged; the reference links Eigen, this file carries its own symmetric
// eigen-solver (Householder reduction + implicit-shift QL).  Pinned by the NeMo fixture of the reference's tests
// (tests/golden/nemo_diar_v1.npz: 60 embeddings -> affinity -> labels, est_num_spk 2, p_hat 15).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

struct nmesc_cfg {                       // src/diarize_cluster.h:14-24
    int   max_num_speakers      = 8;
    float max_rp_threshold      = 0.25f;
    int   sparse_search_volume  = 30;
    int   nme_mat_size          = 512;
    int   min_samples_for_nmesc = 6;
    int   oracle_num_speakers   = -1;    // < 0: estimate
    float fixed_thres           = -1.0f; // > 0: skip the NME analysis
    int   kmeans_random_trials  = 1;
    uint64_t kmeans_seed        = 0;
};

struct nmesc_result {                    // src/diarize_cluster.h:26-30
    int est_num_speakers = 1;
    int p_hat = 1;
    std::vector<int> labels;             // length N, values in [0, est_num_speakers)
};

// embeddings: N x D row-major
nmesc_result nmesc_cluster(const float *embeddings, size_t N, size_t D, const nmesc_cfg &cfg = {});

// NeMo getCosAffinityMatrix: cosine similarity (norm + 3.5e-4), diagonal 1, min-max scaled to [0, 1]; N x N row-major
std::vector<float> nmesc_cosine_affinity(const float *embeddings, size_t N, size_t D);

// eigenvalues (ascending) and, if `vectors` is not null, the eigenvectors (column k of the n x n row-major matrix belongs to
// eigenvalue k) of the symmetric matrix `a` (n x n row-major, destroyed)
void nmesc_sym_eigen(std::vector<double> &a, int n, std::vector<double> &values, std::vector<double> *vectors);

extern "C" {
// plain-C face for bindings / tests: returns 0, or -1 on bad arguments
int nasr_nmesc_affinity(const float *embeddings, int N, int D, float *out /* N*N */);
int nasr_nmesc_cluster(const float *embeddings, int N, int D, int max_num_speakers, float max_rp_threshold,
                       int sparse_search_volume, int nme_mat_size, int oracle_num_speakers, uint64_t kmeans_seed,
                       int32_t *labels_out /* N */, int32_t *est_num_speakers, int32_t *p_hat);
int nasr_sym_eigen(const double *a, int n, double *values /* n */, double *vectors /* n*n or null */);
}

// diarize_cluster_amd.cpp -- see diarize_cluster_amd.h.  NME-SC as NeMo's offline_clustering.py defines it (the reference's
// port: src/diarize_cluster.cpp): cosine affinity -> for every candidate p a p-nearest-neighbour graph, its Laplacian's
// eigen-gaps and the ratio g_p = (p / N) / (largest normalised gap) -> the p with the smallest ratio gives the number of
// speakers and the graph for the spectral embedding -> k-means++.
#include "diarize_cluster_amd.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>

namespace {

constexpr float  AFF_EPS = 3.5e-4f;      // src/diarize_cluster.cpp:31
constexpr double EIG_EPS = 1e-10;        // :32
constexpr int    MIN_P   = 2;            // :33

// ---- symmetric eigen-solver ---------------------------------------------------------------------------------------
// Householder reflections bring A to tridiagonal form T = Q^T A Q (Q accumulated as its transpose: row i of qt = column i
// of Q, so both the accumulation and the QL rotations below run over contiguous rows); implicit-shift QL then diagonalises T.
void tridiagonalise(std::vector<double> &a, int n, std::vector<double> &d, std::vector<double> &e, std::vector<double> *qt) {
    d.assign(n, 0.0);
    e.assign(n, 0.0);
    if (qt) {
        qt->assign((size_t)n * n, 0.0);
        for (int i = 0; i < n; i++) (*qt)[(size_t)i * n + i] = 1.0;
    }
    std::vector<double> v(n), pv(n);
    for (int k = 0; k + 2 < n; k++) {
        const int m = n - k - 1;                              // order of the trailing block
        double below = 0.0;
        for (int i = 1; i < m; i++) below += a[(size_t)(k + 1 + i) * n + k] * a[(size_t)(k + 1 + i) * n + k];
        if (below == 0.0) continue;                           // column already tridiagonal
        const double x0 = a[(size_t)(k + 1) * n + k];
        const double alpha = x0 > 0.0 ? -std::sqrt(x0 * x0 + below) : std::sqrt(x0 * x0 + below);
        v[0] = x0 - alpha;
        for (int i = 1; i < m; i++) v[i] = a[(size_t)(k + 1 + i) * n + k];
        const double beta = 2.0 / (v[0] * v[0] + below);      // H = I - beta v v^T
        double vp = 0.0;
        for (int i = 0; i < m; i++) {
            const double *row = &a[(size_t)(k + 1 + i) * n + k + 1];
            double s = 0.0;
            for (int j = 0; j < m; j++) s += row[j] * v[j];
            pv[i] = beta * s;
            vp += v[i] * pv[i];
        }
        const double half = 0.5 * beta * vp;
        for (int i = 0; i < m; i++) pv[i] -= half * v[i];     // w = p - (beta/2)(v.p) v ;  A22 -= v w^T + w v^T
        for (int i = 0; i < m; i++) {
            double *row = &a[(size_t)(k + 1 + i) * n + k + 1];
            const double vi = v[i], wi = pv[i];
            for (int j = 0; j < m; j++) row[j] -= vi * pv[j] + wi * v[j];
        }
        a[(size_t)(k + 1) * n + k] = a[(size_t)k * n + k + 1] = alpha;
        for (int i = 1; i < m; i++) a[(size_t)(k + 1 + i) * n + k] = a[(size_t)k * n + k + 1 + i] = 0.0;
        if (qt) {                                             // Q <- Q H on columns k+1.. = rows k+1.. of Q^T
            std::vector<double> &s = pv;                      // reuse as the n projections
            std::fill(s.begin(), s.end(), 0.0);
            for (int j = 0; j < m; j++) {
                const double *qr = &(*qt)[(size_t)(k + 1 + j) * n];
                const double vj = v[j];
                for (int r = 0; r < n; r++) s[r] += qr[r] * vj;
            }
            for (int j = 0; j < m; j++) {
                double *qr = &(*qt)[(size_t)(k + 1 + j) * n];
                const double bv = beta * v[j];
                for (int r = 0; r < n; r++) qr[r] -= s[r] * bv;
            }
        }
    }
    for (int i = 0; i < n; i++) d[i] = a[(size_t)i * n + i];
    for (int i = 0; i + 1 < n; i++) e[i] = a[(size_t)(i + 1) * n + i];
}

bool ql_implicit(std::vector<double> &d, std::vector<double> &e, int n, std::vector<double> *qt) {
    for (int l = 0; l < n; l++) {
        for (int iter = 0;; iter++) {
            int m = l;
            for (; m + 1 < n; m++) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= std::numeric_limits<double>::epsilon() * dd) break;
            }
            if (m == l) break;
            if (iter >= 200) return false;
            double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
            double r = std::hypot(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? r : -r));
            double s = 1.0, c = 1.0, p = 0.0;
            int i = m - 1;
            for (; i >= l; i--) {
                double f = s * e[i];
                const double b = c * e[i];
                r = std::hypot(f, g);
                e[i + 1] = r;
                if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
                s = f / r;
                c = g / r;
                g = d[i + 1] - p;
                r = (d[i] - g) * s + 2.0 * c * b;
                p = s * r;
                d[i + 1] = g + p;
                g = c * r - b;
                if (qt) {
                    double *z0 = &(*qt)[(size_t)i * n], *z1 = &(*qt)[(size_t)(i + 1) * n];
                    for (int k = 0; k < n; k++) {
                        f = z1[k];
                        z1[k] = s * z0[k] + c * f;
                        z0[k] = c * z0[k] - s * f;
                    }
                }
            }
            if (r == 0.0 && i >= l) continue;
            d[l] -= p;
            e[l] = g;
            e[m] = 0.0;
        }
    }
    return true;
}

// ---- NME-SC pieces -----------------------------------------------------------------------------------------------------
// keep the p largest entries of every row (ties: lower index first), binarise, symmetrise: 0.5 (X + X^T)   (:76-103)
std::vector<float> neighbour_graph(const std::vector<float> &aff, int N, int p) {
    if (p <= 0) return aff;
    const int k = std::min(p, N);
    std::vector<unsigned char> keep((size_t)N * N, 0);
    std::vector<int> order(N);
    for (int i = 0; i < N; i++) {
        const float *row = &aff[(size_t)i * N];
        std::iota(order.begin(), order.end(), 0);
        std::partial_sort(order.begin(), order.begin() + k, order.end(),
                          [row](int x, int y) { return row[x] != row[y] ? row[x] > row[y] : x < y; });
        for (int j = 0; j < k; j++) keep[(size_t)i * N + order[j]] = 1;
    }
    std::vector<float> g((size_t)N * N);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) g[(size_t)i * N + j] = 0.5f * (float)(keep[(size_t)i * N + j] + keep[(size_t)j * N + i]);
    return g;
}

// L = D - A with the diagonal of A ignored (:105-114)
std::vector<double> laplacian(const std::vector<float> &g, int N) {
    std::vector<double> L((size_t)N * N);
    for (int i = 0; i < N; i++) {
        double deg = 0.0;
        for (int j = 0; j < N; j++) {
            const double w = i == j ? 0.0 : (double)g[(size_t)i * N + j];
            deg += std::fabs(w);
            L[(size_t)i * N + j] = -w;
        }
        L[(size_t)i * N + i] = deg;
    }
    return L;
}

bool connected(const std::vector<float> &g, int N) {
    if (N == 0) return true;
    std::vector<char> seen(N, 0);
    std::vector<int> stack(1, 0);
    seen[0] = 1;
    int reached = 1;
    while (!stack.empty()) {
        const int v = stack.back();
        stack.pop_back();
        for (int j = 0; j < N; j++)
            if (!seen[j] && g[(size_t)v * N + j] > 0.0f) { seen[j] = 1; reached++; stack.push_back(j); }
    }
    return reached == N;
}

// candidate p values: `steps` points from 1 to max(2, floor(N * max_rp_threshold)), truncated to int, duplicates removed (:168-190)
std::vector<int> candidate_p(int N, float max_rp_threshold, int sparse_search_volume) {
    const int max_p = std::max(MIN_P, (int)std::floor((double)N * (double)max_rp_threshold));
    const int steps = std::min(max_p, std::max(2, sparse_search_volume));
    std::vector<int> out;
    for (int i = 0; i < steps; i++) {
        const int p = (int)(1.0 + (double)i * ((double)max_p - 1.0) / (double)std::max(1, steps - 1));
        if (std::find(out.begin(), out.end(), p) == out.end()) out.push_back(p);
    }
    return out;
}

struct GapRatio { double g_p; int n_spk; };
GapRatio gap_ratio(const std::vector<float> &aff, int N, int p, int max_num_speakers) {      // :199-214
    std::vector<double> L = laplacian(neighbour_graph(aff, N, p), N), lam;
    nmesc_sym_eigen(L, N, lam, nullptr);
    const int K = std::min(N - 1, max_num_speakers);
    int best = 0;
    for (int i = 1; i < K; i++)
        if (lam[i + 1] - lam[i] > lam[best + 1] - lam[best]) best = i;
    const double gap = K > 0 ? (lam[best + 1] - lam[best]) / (lam[N - 1] + EIG_EPS) : 0.0;
    return {((double)p / (double)N) / (gap + EIG_EPS), best + 1};
}

struct Rng {                              // splitmix64
    uint64_t s;
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

// k-means++ seeding, Lloyd iterations until the inertia moves by less than 1e-4 (:221-307)
double kmeans(const std::vector<float> &X, int N, int D, int K, Rng &rng, std::vector<int> &labels) {
    labels.assign(N, 0);
    if (K <= 1) return 0.0;
    auto dist2 = [&](const float *a, const float *b) { double s = 0.0; for (int d = 0; d < D; d++) { const double t = (double)a[d] - b[d]; s += t * t; } return s; };
    std::vector<float> centre((size_t)K * D);
    std::vector<double> near(N, std::numeric_limits<double>::infinity());
    int pick = (int)(rng.next() % (uint64_t)N);
    for (int k = 0; k < K; k++) {
        std::copy_n(&X[(size_t)pick * D], D, &centre[(size_t)k * D]);
        if (k + 1 == K) break;
        double total = 0.0;
        for (int i = 0; i < N; i++) { near[i] = std::min(near[i], dist2(&X[(size_t)i * D], &centre[(size_t)k * D])); total += near[i]; }
        if (total <= 0.0) { pick = (int)(rng.next() % (uint64_t)N); continue; }
        double target = rng.uniform() * total, cum = 0.0;
        pick = N - 1;
        for (int i = 0; i < N; i++) { cum += near[i]; if (cum >= target) { pick = i; break; } }
    }
    std::vector<double> sum((size_t)K * D);
    std::vector<int> count(K);
    double prev = std::numeric_limits<double>::infinity();
    for (int it = 0; it < 300; it++) {
        std::fill(sum.begin(), sum.end(), 0.0);
        std::fill(count.begin(), count.end(), 0);
        double inertia = 0.0;
        for (int i = 0; i < N; i++) {
            int best = 0;
            double bd = std::numeric_limits<double>::infinity();
            for (int k = 0; k < K; k++) { const double dd = dist2(&X[(size_t)i * D], &centre[(size_t)k * D]); if (dd < bd) { bd = dd; best = k; } }
            labels[i] = best;
            inertia += bd;
            count[best]++;
            for (int d = 0; d < D; d++) sum[(size_t)best * D + d] += X[(size_t)i * D + d];
        }
        for (int k = 0; k < K; k++)
            if (count[k] > 0) for (int d = 0; d < D; d++) centre[(size_t)k * D + d] = (float)(sum[(size_t)k * D + d] / count[k]);
        if (std::fabs(prev - inertia) < 1e-4) break;
        prev = inertia;
    }
    return prev;
}

}  // namespace

void nmesc_sym_eigen(std::vector<double> &a, int n, std::vector<double> &values, std::vector<double> *vectors) {
    std::vector<double> e, qt;
    tridiagonalise(a, n, values, e, vectors ? &qt : nullptr);
    ql_implicit(values, e, n, vectors ? &qt : nullptr);
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int x, int y) { return values[x] < values[y]; });
    std::vector<double> sorted(n);
    for (int k = 0; k < n; k++) sorted[k] = values[order[k]];
    values.swap(sorted);
    if (vectors) {
        vectors->assign((size_t)n * n, 0.0);
        for (int k = 0; k < n; k++)
            for (int i = 0; i < n; i++) (*vectors)[(size_t)i * n + k] = qt[(size_t)order[k] * n + i];
    }
}

std::vector<float> nmesc_cosine_affinity(const float *emb, size_t N, size_t D) {
    if (N <= 1) return std::vector<float>(1, 1.0f);
    std::vector<float> unit(N * D);
    for (size_t i = 0; i < N; i++) {
        double s = 0.0;
        for (size_t d = 0; d < D; d++) s += (double)emb[i * D + d] * emb[i * D + d];
        const float inv = 1.0f / (std::sqrt((float)s) + AFF_EPS);
        for (size_t d = 0; d < D; d++) unit[i * D + d] = emb[i * D + d] * inv;
    }
    std::vector<float> aff(N * N);
    float lo = 1.0f, hi = 1.0f;
    for (size_t i = 0; i < N; i++) {
        aff[i * N + i] = 1.0f;                                 // NeMo fills the diagonal
        for (size_t j = i + 1; j < N; j++) {
            double s = 0.0;
            for (size_t d = 0; d < D; d++) s += (double)unit[i * D + d] * unit[j * D + d];
            aff[i * N + j] = aff[j * N + i] = (float)s;
            lo = std::min(lo, (float)s);
            hi = std::max(hi, (float)s);
        }
    }
    if (hi > lo) {
        const float range = hi - lo;
        for (float &v : aff) v = (v - lo) / range;             // a division: the largest entry scales to exactly 1
    }
    return aff;
}

nmesc_result nmesc_cluster(const float *emb, size_t N_, size_t D, const nmesc_cfg &cfg) {
    nmesc_result res;
    const int N = (int)N_;
    if (N <= 0) return res;
    if (N <= cfg.min_samples_for_nmesc) {                      // :325-331
        res.est_num_speakers = cfg.oracle_num_speakers > 0 ? cfg.oracle_num_speakers : 1;
        res.labels.assign(N, 0);
        res.p_hat = N - 1;
        return res;
    }
    const std::vector<float> aff = nmesc_cosine_affinity(emb, N, D);
    // the NME analysis runs on every ratio-th row / column when N exceeds nme_mat_size (:150-166)
    const int ratio = std::max(1, (int)std::ceil((double)N / (double)cfg.nme_mat_size));
    std::vector<float> sub;
    int Ns = N;
    if (ratio > 1) {
        std::vector<int> rows;
        for (int i = 0; i < N; i += ratio) rows.push_back(i);
        Ns = (int)rows.size();
        sub.resize((size_t)Ns * Ns);
        for (int i = 0; i < Ns; i++)
            for (int j = 0; j < Ns; j++) sub[(size_t)i * Ns + j] = aff[(size_t)rows[i] * N + rows[j]];
    }
    const std::vector<float> &ana = ratio > 1 ? sub : aff;
    const std::vector<int> ps = cfg.fixed_thres > 0.0f
        ? std::vector<int>(1, std::max(MIN_P, (int)std::floor((double)Ns * (double)cfg.fixed_thres)))
        : candidate_p(Ns, cfg.max_rp_threshold, cfg.sparse_search_volume);
    double best_g = std::numeric_limits<double>::infinity();
    int best_p = ps[0], est = 1;
    for (int p : ps) {
        const GapRatio gr = gap_ratio(ana, Ns, p, cfg.max_num_speakers);
        if (gr.g_p < best_g) { best_g = gr.g_p; best_p = p; est = gr.n_spk; }
    }
    int p_hat = ratio * best_p;
    std::vector<float> graph = neighbour_graph(aff, N, p_hat);
    if (!connected(graph, N)) {                                // grow p until the graph is connected (:367-381)
        for (int p : ps) {
            p_hat = ratio * p;
            graph = neighbour_graph(aff, N, p_hat);
            if (connected(graph, N)) break;
        }
    }
    int n_clusters = cfg.oracle_num_speakers > 0 ? cfg.oracle_num_speakers : est;
    n_clusters = std::max(1, std::min(n_clusters, cfg.max_num_speakers));
    res.est_num_speakers = n_clusters;
    res.p_hat = p_hat;
    if (n_clusters == 1) { res.labels.assign(N, 0); return res; }
    // spectral embedding = the eigenvectors of the n_clusters smallest Laplacian eigenvalues (:309-319), then k-means++
    std::vector<double> L = laplacian(graph, N), lam, vec;
    nmesc_sym_eigen(L, N, lam, &vec);
    std::vector<float> se((size_t)N * n_clusters);
    for (int i = 0; i < N; i++)
        for (int k = 0; k < n_clusters; k++) se[(size_t)i * n_clusters + k] = (float)vec[(size_t)i * N + (n_clusters - 1 - k)];
    Rng rng{cfg.kmeans_seed};
    double best_inertia = std::numeric_limits<double>::infinity();
    std::vector<int> labels;
    for (int t = 0; t < std::max(1, cfg.kmeans_random_trials); t++) {
        const double inertia = kmeans(se, N, n_clusters, n_clusters, rng, labels);
        if (inertia < best_inertia) { best_inertia = inertia; res.labels = labels; }
    }
    return res;
}

extern "C" int nasr_nmesc_affinity(const float *embeddings, int N, int D, float *out) {
    if (!embeddings || !out || N < 1 || D < 1) return -1;
    const std::vector<float> a = nmesc_cosine_affinity(embeddings, (size_t)N, (size_t)D);
    std::copy(a.begin(), a.end(), out);
    return 0;
}

extern "C" int nasr_nmesc_cluster(const float *embeddings, int N, int D, int max_num_speakers, float max_rp_threshold,
                                  int sparse_search_volume, int nme_mat_size, int oracle_num_speakers, uint64_t kmeans_seed,
                                  int32_t *labels_out, int32_t *est_num_speakers, int32_t *p_hat) {
    if (!embeddings || !labels_out || N < 1 || D < 1 || max_num_speakers < 1 || nme_mat_size < 1) return -1;
    nmesc_cfg cfg;
    cfg.max_num_speakers = max_num_speakers;
    cfg.max_rp_threshold = max_rp_threshold;
    cfg.sparse_search_volume = sparse_search_volume;
    cfg.nme_mat_size = nme_mat_size;
    cfg.oracle_num_speakers = oracle_num_speakers;
    cfg.kmeans_seed = kmeans_seed;
    const nmesc_result r = nmesc_cluster(embeddings, (size_t)N, (size_t)D, cfg);
    for (int i = 0; i < N; i++) labels_out[i] = r.labels[i];
    if (est_num_speakers) *est_num_speakers = r.est_num_speakers;
    if (p_hat) *p_hat = r.p_hat;
    return 0;
}

extern "C" int nasr_sym_eigen(const double *a, int n, double *values, double *vectors) {
    if (!a || !values || n < 1) return -1;
    std::vector<double> m(a, a + (size_t)n * n), val, vec;
    nmesc_sym_eigen(m, n, val, vectors ? &vec : nullptr);
    std::copy(val.begin(), val.end(), values);
    if (vectors) std::copy(vec.begin(), vec.end(), vectors);
    return 0;
}

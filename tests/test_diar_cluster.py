"""SURVEY.md section 8 f-4, the CPU stage: NME-SC clustering (host/diarize_cluster_amd.cpp) against the NeMo fixture the
reference's tests hold (tests/diarize/cluster_ref, checked by its tests/test_diarize_cluster.cpp: affinity diff printed,
permutation-invariant label accuracy >= 0.95), plus the eigen-solver it carries in place of Eigen."""
from itertools import permutations
from pathlib import Path

import numpy as np
import pytest

from nemotron_asr_amd import cluster

NEMO = Path(__file__).parent / "golden" / "nemo_diar_v1.npz"


def _accuracy(got, ref):
    K = int(max(got.max(), ref.max())) + 1
    return max(np.mean(np.asarray(perm)[got] == ref) for perm in permutations(range(K)))


def test_affinity_matches_nemo():
    g = np.load(NEMO)
    aff = cluster.cosine_affinity(g["cluster_embeddings"])
    assert np.abs(aff - g["cluster_affinity"]).max() < 2e-6
    assert aff.min() == 0.0 and aff.max() == 1.0 and (np.diag(aff) == 1.0).all() and (aff == aff.T).all()


def test_nmesc_matches_nemo_labels_speaker_count_and_p():
    g = np.load(NEMO)
    est_ref, p_ref, max_spk, volume, mat = (int(v) for v in g["cluster_meta"])
    labels, est, p_hat = cluster.nmesc_cluster(g["cluster_embeddings"], max_spk, float(g["cluster_max_rp_threshold"]), volume, mat)
    assert (est, p_hat) == (est_ref, p_ref) == (2, 15)
    assert _accuracy(labels, g["cluster_labels"]) == 1.0          # the reference's own bar is 0.95


@pytest.mark.parametrize("n", [1, 2, 3, 40, 200])
def test_eigen_solver_against_numpy(n):
    rng = np.random.default_rng(n)
    m = rng.standard_normal((n, n))
    m = (m + m.T) / 2
    val, vec = cluster.sym_eigen(m)
    assert np.abs(val - np.linalg.eigvalsh(m)).max() < 1e-11 * max(1.0, np.abs(m).max() * n)
    assert np.abs(m @ vec - vec * val).max() < 1e-11 * n and np.abs(vec.T @ vec - np.eye(n)).max() < 1e-11 * n
    assert np.abs(cluster.sym_eigen(m, vectors=False)[0] - val).max() < 1e-11 * n


def test_eigen_solver_degenerate_laplacian():
    """k disconnected cliques: eigenvalue 0 with multiplicity k (the case the speaker count is read from)."""
    k, size = 3, 7
    n = k * size
    a = np.zeros((n, n))
    for c in range(k):
        a[c * size:(c + 1) * size, c * size:(c + 1) * size] = 1.0
    np.fill_diagonal(a, 0.0)
    lap = np.diag(a.sum(1)) - a
    val, vec = cluster.sym_eigen(lap)
    assert np.abs(val[:k]).max() < 1e-12 and abs(val[k] - size) < 1e-10
    assert np.abs(lap @ vec[:, :k]).max() < 1e-10 and np.abs(vec.T @ vec - np.eye(n)).max() < 1e-10


def test_clusters_of_synthetic_speakers_and_edge_cases():
    rng = np.random.default_rng(5)
    centres = rng.standard_normal((3, 192))
    truth = np.repeat(np.arange(3), [25, 40, 18])
    emb = centres[truth] + 0.35 * rng.standard_normal((truth.size, 192))
    labels, est, _ = cluster.nmesc_cluster(emb.astype(np.float32))
    assert est == 3 and _accuracy(labels, truth) == 1.0
    # subsampled NME analysis (N > nme_mat_size): same answer
    labels2, est2, p2 = cluster.nmesc_cluster(emb.astype(np.float32), nme_mat_size=32)
    assert est2 == 3 and _accuracy(labels2, truth) == 1.0 and p2 % 3 == 0          # p_hat = ratio (3) x p
    # forced speaker count; tiny inputs fall back to one cluster (src/diarize_cluster.cpp:325-331)
    assert cluster.nmesc_cluster(emb.astype(np.float32), oracle_num_speakers=2)[1] == 2
    few, est_few, p_few = cluster.nmesc_cluster(emb[:5].astype(np.float32))
    assert (few == 0).all() and est_few == 1 and p_few == 4
    with pytest.raises(ValueError):
        cluster.nmesc_cluster(np.zeros((0, 192), np.float32))

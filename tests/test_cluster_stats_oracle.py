"""Pin oracle/cluster_stats_oracle.py to the scores the reference's own classes produced (oracle/make_golden_stats.py:
internal_eval.DunnIndex / Sihouette / CHIndex / DBIndex and p2's two gap-statistic inertia definitions).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import cluster_stats_oracle as CO

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, 'cluster_stats_*.npz')))


def test_fixture_inventory():
    assert len(CASES) == 3


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_scores(name):
    g = dict(np.load(os.path.join(GOLDEN, name)))
    x, lab = g['x'], g['labels']
    # the reference evaluates pairwise distances of float32 inputs in float32 for the inertia / Dunn terms: 1e-5 covers it
    np.testing.assert_allclose(CO.inertia_v1(x, lab), g['inertia_v1'], rtol=1e-5)
    np.testing.assert_allclose(CO.inertia_v2(x, lab), g['inertia_v2'], rtol=1e-5)
    np.testing.assert_allclose(CO.dunn(x, lab), g['dunn'], rtol=1e-5)
    np.testing.assert_allclose(CO.silhouette(x, lab), g['silhouette'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(CO.calinski_harabasz(x, lab), g['calinski_harabasz'], rtol=1e-5)
    np.testing.assert_allclose(CO.davies_bouldin(x, lab), g['davies_bouldin'], rtol=1e-5)


def test_pair_stats_consistency():
    rng = np.random.default_rng(0)
    x, lab = rng.normal(size=(50, 4)), rng.integers(0, 3, 50)
    _, K, S, Dmin, own_max = CO.pair_stats(x, lab)
    d = CO.distance_matrix(x)
    np.testing.assert_allclose(S.sum(1), d.sum(1), rtol=1e-12)
    assert (Dmin[np.arange(50), lab] == 0).all() and (own_max <= d.max(1) + 1e-12).all()


def test_reference_draws_into_a_buffer_replay_numpys_global_stream():
    """p2's gap statistic draws its uniform reference sets from NumPy's GLOBAL legacy stream (p2_clustering_optK.py:372-375 upstream:
    np.random.random_sample(shape)).  global_uniform_into fills a reused buffer instead: same doubles, and the stream continues
    exactly where the original call would have left it (the k-means++ seeds that follow depend on it)."""
    import numpy as np
    from deep_interpolation_clustering_amd.p2_clustering_optK import global_uniform_into
    np.random.seed(123)
    want, after = np.random.random_sample((1000, 37)), np.random.random_sample(4)
    np.random.seed(123)
    np.random.standard_normal(3)                     # (a cached Gaussian in the legacy state must survive the detour)
    np.random.seed(123)
    buf = np.full((1000, 37), -1.0)
    got = global_uniform_into(buf)
    assert got is buf and np.array_equal(buf, want)
    assert np.array_equal(np.random.random_sample(4), after)
    np.random.seed(7)
    g1 = np.random.standard_normal(1)                # leaves has_gauss = 1
    global_uniform_into(buf)
    tail = np.random.standard_normal(2)
    np.random.seed(7)
    assert np.array_equal(np.random.standard_normal(1), g1)
    np.random.random_sample((1000, 37))
    assert np.array_equal(np.random.standard_normal(2), tail)

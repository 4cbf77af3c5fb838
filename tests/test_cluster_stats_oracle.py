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

"""GPU parity of the K-sweep statistics (csrc/dic_pairdist.hip through cluster_stats.py / internal_eval.py) against the
reference's own scores (tests/golden/cluster_stats_*.npz), the CPU oracle, and scikit-learn.  ``-m gpu``."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import cluster_stats_oracle as CO

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, 'cluster_stats_*.npz')))


@pytest.fixture(scope='module')
def cs():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from deep_interpolation_clustering_amd import cluster_stats
    return cluster_stats


@pytest.mark.parametrize('name', CASES)
def test_scores_match_reference(cs, name):
    g = dict(np.load(os.path.join(GOLDEN, name)))
    x, lab = g['x'], g['labels']
    st = cs.pair_stats(x, lab)
    np.testing.assert_allclose(cs.inertia_v1(x, lab, st), g['inertia_v1'], rtol=1e-5)
    np.testing.assert_allclose(cs.inertia_v2(x, lab, st), g['inertia_v2'], rtol=1e-5)
    # the intra-cluster-only pass (dic_cluster_intra_sums: what the gap statistic's reference sets run) gives the reference's inertias too
    np.testing.assert_allclose(cs.inertia_v1(x, lab), g['inertia_v1'], rtol=1e-5)
    np.testing.assert_allclose(cs.inertia_v2(x, lab), g['inertia_v2'], rtol=1e-5)
    np.testing.assert_allclose(cs.dunn_index(x, lab, st), g['dunn'], rtol=1e-5)
    np.testing.assert_allclose(cs.silhouette_score(x, lab, st), g['silhouette'], rtol=1e-5, atol=1e-6)
    # ... and from the sums-only pass on the matrix cores (dic_cluster_pair_rowsums: what p2's sweep runs when no Dunn index is asked for)
    with pytest.MonkeyPatch.context() as mp:
        mp.setattr(cs, 'ROWSUM_MIN_POINTS', 0)
        np.testing.assert_allclose(cs.silhouette_score(x, lab, cs.pair_stats(x, lab, need_min=False, need_max=False)), g['silhouette'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cs.calinski_harabasz_score(x, lab), g['calinski_harabasz'], rtol=1e-5)
    np.testing.assert_allclose(cs.davies_bouldin_score(x, lab), g['davies_bouldin'], rtol=1e-5)


@pytest.mark.parametrize('n,d,k', [(1, 4, 1), (63, 4, 2), (64, 8, 1), (65, 36, 3), (1000, 256, 7), (1537, 20, 64), (300, 6, 5)])
def test_pair_stats_match_oracle(cs, n, d, k):
    """Every S / Dmin / own_max entry against the float64 oracle: ragged tile edges, D not a multiple of the 32-wide
    staging chunk (and not of 4: zero padded), the K limit, empty label values skipped by the encoder."""
    rng = np.random.default_rng(n * 1000 + d)
    x = rng.normal(0, 1, (n, d)).astype(np.float32)
    lab = rng.integers(0, k, n) * 3 - 4              # arbitrary label values; some may be unused
    st = cs.pair_stats(x, lab)
    olab, K, S, Dmin, own_max = CO.pair_stats(x, lab)
    assert st.K == K and np.array_equal(st.labels.cpu().numpy(), olab)
    np.testing.assert_allclose(st.S.cpu().numpy(), S, rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(st.Dmin.cpu().numpy(), Dmin, rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(st.own_max.cpu().numpy(), own_max, rtol=2e-6, atol=1e-6)


def test_validity_indices_match_sklearn(cs):
    """The drop-in callables of internal_eval.py against scikit-learn on k-means-like data (2 000 x 256)."""
    from sklearn import metrics
    from deep_interpolation_clustering_amd.internal_eval import CHIndex, DBIndex, DunnIndex, Sihouette
    rng = np.random.default_rng(5)
    cen = rng.normal(0, 1.0, (4, 256))
    lab = rng.integers(0, 4, 2000)
    x = (cen[lab] + rng.normal(0, 1.0, (2000, 256))).astype(np.float32)
    lab[rng.integers(0, 2000, 100)] = rng.integers(0, 4, 100)        # some misassigned points
    np.testing.assert_allclose(Sihouette()(x, lab), metrics.silhouette_score(x, lab), rtol=1e-5)
    np.testing.assert_allclose(CHIndex()(x, lab), metrics.calinski_harabasz_score(x, lab), rtol=1e-5)
    np.testing.assert_allclose(DBIndex()(x, lab), metrics.davies_bouldin_score(x, lab), rtol=1e-5)
    np.testing.assert_allclose(DunnIndex()(x, lab), CO.dunn(x, lab), rtol=1e-5)
    with pytest.raises(ValueError):
        Sihouette()(x, np.zeros(2000, dtype=np.int64))


def test_full_size_properties(cs):
    """75 000 x 256 latents (BASELINE cfg2), K = 4: properties that need no O(N^2) oracle -- row sums of S against
    torch.cdist for sampled rows, symmetry of the cluster-to-cluster sums, the Dmin / own_max invariants."""
    from oracle.synth import latent_blobs
    x, lab = latent_blobs(11, 75000, 256, 4)
    xd = torch.tensor(x, device='cuda')
    st = cs.pair_stats(xd, lab)
    pick = torch.arange(0, 75000, 293, device='cuda')
    ref = torch.cdist(xd[pick].double(), xd.double())
    np.testing.assert_allclose(st.S[pick].double().sum(1).cpu().numpy(), ref.sum(1).cpu().numpy(), rtol=2e-6)
    onehot = torch.nn.functional.one_hot(st.labels, st.K).double()
    between = onehot.t() @ st.S.double()                       # [a][b] = sum_{i in a, j in b} d_ij
    np.testing.assert_allclose(between.cpu().numpy(), between.t().cpu().numpy(), rtol=1e-6)
    assert float(st.Dmin.gather(1, st.labels[:, None]).abs().max()) == 0.0          # every point is its own nearest
    lab_t = st.labels[pick]
    own = torch.where(st.labels[None, :] == lab_t[:, None], ref, torch.zeros_like(ref)).max(1).values
    np.testing.assert_allclose(st.own_max[pick].cpu().numpy(), own.cpu().numpy(), rtol=2e-6)
    s = cs.silhouette_score(xd, lab, st)
    assert -1.0 <= s <= 1.0


@pytest.mark.parametrize('n,d,k', [(63, 4, 2), (65, 36, 3), (1000, 256, 7), (1537, 20, 64), (300, 6, 5), (5000, 256, 1)])
def test_intra_only_pass_equals_own_column_of_the_full_pass(cs, n, d, k):
    """dic_cluster_intra_sums visits only the column tiles of the clusters a row tile's own rows belong to (sum_c n_c^2 of the N^2 pairs):
    every point's own-cluster sum must be BIT-identical to the own column of the full pass (same tiles, same summation order), for
    row tiles that straddle cluster boundaries, singleton clusters and K = 1."""
    rng = np.random.default_rng(n + d + k)
    x = rng.normal(0, 1, (n, d)).astype(np.float32)
    lab = rng.integers(0, k, n)
    lab[:k] = np.arange(k)                       # every label occurs
    if k > 3:
        lab[lab == k - 1] = 0
        lab[k - 1] = k - 1                       # a singleton cluster
    full = cs.pair_stats(x, lab, need_min=True, need_max=False)          # (with Dmin: the difference-form kernel; sums only would take the matrix-core pass)
    intra = cs.pair_stats(x, lab, intra_only=True)
    own = full.S.gather(1, full.labels[:, None])[:, 0]
    assert torch.equal(intra.S_own, own)
    assert intra.S is None and torch.equal(intra.intra_sums(), full.intra_sums())


@pytest.mark.parametrize('n,d,k,spread', [(63, 4, 2, 1.0), (700, 36, 3, 1.0), (3000, 256, 7, 1.0), (1537, 20, 64, 1.0), (5000, 256, 1, 1.0), (2600, 256, 2, 30.0),
                                          (20000, 256, 4, 1.0), (513, 255, 2, 1.0)])
def test_intra_totals_on_the_matrix_cores_match_f64(cs, n, d, k, spread):
    """dic_cluster_intra_totals (centred hi / lo bf16 planes, norms folded into the MFMA inner product, block pairs I <= J only): every cluster's sum of pairwise
    distances against the f64 sum of the same pairs, for clusters smaller than a block, cluster ends inside a block, a singleton cluster, a width that needs
    zero-padding, and clusters far from the origin (``spread``: the norm form's cancellation is what the centring removes).  Measured <= 2e-7; the reference sums
    f32 distances (tests of the golden tables: rtol 1e-5).  Same result on a second call (no atomics)."""
    rng = np.random.default_rng(n + d + k)
    lab = rng.integers(0, k, n)
    lab[:k] = np.arange(k)
    if k > 3:
        lab[lab == k - 1] = 0
        lab[k - 1] = k - 1                       # a singleton cluster
    x = (rng.normal(0, 1, (n, d)) + spread * rng.normal(0, 1, (k, d))[lab]).astype(np.float32)
    st = cs.intra_totals(x, lab)
    assert st.S is None and st.S_own is None
    got = st.intra_sums().cpu().numpy()
    xd = torch.as_tensor(x, device='cuda', dtype=torch.float64)
    want = np.zeros(k)
    for c in range(k):
        xc = xd[torch.as_tensor(lab == c, device='cuda')]
        for lo in range(0, xc.shape[0], 128):          # (torch.cdist returned a third of this sum for 4096 x 5000 f64 rows on this image)
            want[c] += float((xc[lo:lo + 128, None, :] - xc[None, :, :]).square_().sum(-1).sqrt_().sum())
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)
    assert np.array_equal(cs.intra_totals(x, lab).intra_sums().cpu().numpy(), got)
    # and the two inertia definitions built on it agree with the per-point pass (dic_cluster_intra_sums)
    per_point = cs.pair_stats(x, lab, intra_only=True)
    np.testing.assert_allclose(cs.inertia_v1(x, lab), cs.inertia_v1(x, lab, per_point), rtol=2e-6)
    np.testing.assert_allclose(cs.inertia_v2(x, lab), cs.inertia_v2(x, lab, per_point), rtol=2e-6)


@pytest.mark.parametrize('n,d,k,spread', [(1, 4, 1, 1.0), (63, 4, 2, 1.0), (700, 36, 3, 1.0), (3000, 256, 7, 1.0), (1537, 20, 64, 1.0), (2600, 256, 2, 10.0),
                                          (70001, 256, 3, 1.0), (513, 255, 2, 1.0)])
def test_row_sums_on_the_matrix_cores_match_the_difference_form(cs, n, d, k, spread, monkeypatch):
    """dic_cluster_pair_rowsums (need_min = need_max = False: split bf16 planes relative to the mean, norms folded into the inner product, contiguous tile ranges per
    workgroup, slots added in order) against the direct-difference f32 pass (dic_cluster_pairdist): every S[i][c], with empty label values, a singleton cluster,
    clusters smaller than a block, more tiles than workgroups (70 001 points: slots that split a (row block, cluster) run between two workgroups).  Measured:
    1.1e-6 of the row's largest sum at 70 001 x 256 (the size the pass is used from: cluster_stats.ROWSUM_MIN_POINTS), up to 4.4e-6 on the tiny shapes forced
    through it here (63 points in 4 dimensions: few pairs, nothing averages).  The same result on a second call."""
    monkeypatch.setattr(cs, 'ROWSUM_MIN_POINTS', 0)
    rng = np.random.default_rng(n + d + k)
    lab = rng.integers(0, k, n)
    lab[:min(k, n)] = np.arange(min(k, n))
    if k > 3:
        lab[lab == k - 1] = 0
        lab[k - 1] = k - 1                       # a singleton cluster
    x = (rng.normal(0, 1, (n, d)) + spread * rng.normal(0, 1, (k, d))[lab]).astype(np.float32)
    want = cs.pair_stats(x, lab, need_min=True, need_max=False)
    got = cs.pair_stats(x, lab, need_min=False, need_max=False)
    assert got.Dmin is None and got.own_max is None and torch.equal(got.labels, want.labels)
    w, g_ = want.S.double(), got.S.double()
    scale = w.max(1, keepdim=True).values.clamp(min=1e-3)
    assert float(((g_ - w).abs() / scale).max()) <= (3e-6 if n >= 8192 else 1.5e-5)
    assert torch.equal(cs.pair_stats(x, lab, need_min=False, need_max=False).S, got.S)
    if n > k:
        np.testing.assert_allclose(cs.silhouette_score(x, lab, got), cs.silhouette_score(x, lab, want), rtol=1e-5, atol=2e-6 if n >= 8192 else 2e-5)


def test_gap_table_equals_reference(tmp_path):
    """KM.compute_gap_internal_metric against the table the REFERENCE's own method produced (oracle/make_golden_gap.py; p2:353-410):
    every column at rtol 1e-5 AND the position of NumPy's global stream afterwards -- i.e. the draw order reference set -> that fit's
    k-means++ seeds -> next reference set (which this package overlaps with the GPU work on a worker thread) is upstream's."""
    from deep_interpolation_clustering_amd import p2_clustering_optK as p2
    from oracle.synth import latent_blobs
    g = np.load(os.path.join(GOLDEN, 'gap_table_blobs.npz'))
    X, _ = latent_blobs(int(g['seed']), int(g['N']), int(g['D']), int(g['G']))
    assert str(X.dtype) == str(g['x_dtype'])
    cols = [str(c) for c in g['columns']]
    km = p2.KM(int(g['k_max']), str(tmp_path), cols[5:], int(g['n_init']), int(g['gap_b']))
    np.random.seed(int(g['np_seed']))
    df = km.compute_gap_internal_metric(X, int(g['k_max']), n_references=int(g['gap_b']), version=1).astype(float)
    pos = np.random.random()
    assert list(df.columns) == cols
    ref = g['table']
    assert pos == float(g['stream_pos']), 'the global stream was consumed differently'
    for j, c in enumerate(cols):
        np.testing.assert_allclose(df[c].to_numpy(), ref[:, j], rtol=1e-5, atol=1e-5 if c == 'gap' else 0, err_msg=c)   # gap = ref - act


def test_segment_sum_matches_numpy():
    """dic_segment_sum_f64 (the per-cluster f64 sums behind the centroid scores and the gap statistic's distance sums; round 6: no one-hot dgemm) against
    NumPy: matrices and vectors, K up to 20, labels in file order, an empty cluster; deterministic run to run."""
    from deep_interpolation_clustering_amd import cluster_stats
    rng = np.random.default_rng(3)
    for n, d, K in ((75000, 256, 20), (1001, 7, 3), (64, 1, 2), (5000, 256, 1)):
        lab = rng.integers(0, K, n)
        if K > 2:
            lab[lab == 1] = 0                                   # cluster 1 stays empty
        v = rng.normal(size=(n, d))
        ref = np.zeros((K, d))
        np.add.at(ref, lab, v)
        vt, lt = torch.tensor(v, device='cuda'), torch.tensor(lab, device='cuda')
        got = cluster_stats._segment_sum(vt, lt, K)
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-12, atol=1e-10)
        assert torch.equal(got, cluster_stats._segment_sum(vt, lt, K))
        g1 = cluster_stats._segment_sum(vt[:, 0], lt, K)
        np.testing.assert_allclose(g1.cpu().numpy(), ref[:, 0], rtol=1e-12, atol=1e-10)

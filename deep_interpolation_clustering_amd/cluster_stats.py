"""Cluster statistics of the K sweep on the device.

Upstream p2 evaluates, for every K and every gap-statistic reference set, ``pairwise_distances`` of each cluster
(p2_clustering_optK.py:334-351: an n_c x n_c float64 matrix per cluster) and scikit-learn's silhouette /
Calinski-Harabasz / Davies-Bouldin scores plus an O(N^2) Python-loop Dunn index (internal_eval.py:37-147).  Here
all distance work is ONE pass of ``dic_cluster_pairdist`` (csrc/dic_pairdist.hip) over the N^2 point pairs:

    S[i][k]    = sum_{j in cluster k} ||x_i - x_j||        Dmin[i][k] = min_{j in cluster k} ||x_i - x_j||
    own_max[i] = max_{j in cluster(i)} ||x_i - x_j||

from which every index above follows by O(N K) reductions (done here in f64 with torch on the device).  The centroid
based indices (CH, DB) need no pair pass at all.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native as N


def _device_points(x):
    x = torch.as_tensor(x)
    if not x.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError('deep_interpolation_clustering_amd.cluster_stats runs only on an MI355X; there is no CPU path by design')
        x = x.to('cuda')
    return x.to(torch.float32).contiguous()


def _encode(labels, device):
    """sklearn LabelEncoder semantics: sorted unique values -> 0..K-1."""
    lab = torch.as_tensor(np.asarray(labels) if not torch.is_tensor(labels) else labels).to(device).reshape(-1)
    uniq, inv = torch.unique(lab, sorted=True, return_inverse=True)
    return inv.to(torch.int64), int(uniq.numel())


def _segment_sum(values, lab, K):
    """(K, ...) f64 sums of ``values`` rows per label (``dic_segment_sum_f64``: private f64 sums per thread and cluster in LDS, fixed order -- until round 6 a
    one-hot f64 GEMM through rocBLAS; index_add_ on f64 is an atomic scatter that took 65 ms per call on 75 000 x 256)."""
    v = values.double()
    one_d = v.dim() == 1
    v2 = (v[:, None] if one_d else v).contiguous()
    n, d = v2.shape
    out = torch.empty((K, d), dtype=torch.float64, device=v2.device)
    L = N.lib()
    ws = torch.empty(max(16, L.dic_segment_sum_workspace(d, K)), dtype=torch.uint8, device=v2.device)
    labc = lab.to(torch.int64).contiguous()
    N.check(L.dic_segment_sum_f64(N.ptr(v2), v2.stride(0), N.ptr(labc), n, d, K, N.ptr(out), N.ptr(ws), ws.numel(), N.stream_of(v2)), 'dic_segment_sum_f64')
    return out[:, 0] if one_d else out


class PairStats:
    """Result of one pair pass: ``labels`` (N) encoded 0..K-1, ``counts`` (K), ``S`` / ``Dmin`` (N,K) f32 and ``own_max`` (N),
    all in the caller's row order."""

    def __init__(self, labels, counts, S, Dmin, own_max, S_own=None, totals=None):
        self.labels, self.counts, self.S, self.Dmin, self.own_max, self.S_own, self.totals = labels, counts, S, Dmin, own_max, S_own, totals
        self.K = int(counts.numel())

    def intra_sums(self):
        """(K,) f64: sum over ordered pairs (i, j) of one cluster of ||x_i - x_j||  (= np.sum(pairwise_distances(X_c)))."""
        if self.totals is not None:
            return self.totals
        own = (self.S_own if self.S is None else self.S.gather(1, self.labels[:, None])[:, 0]).double()
        return _segment_sum(own, self.labels, self.K)


TOTALS_MAX_D = 256          # dic_cluster_intra_totals: augmented rows of 256 coordinates


def _tile_list(counts_host):
    """(ntiles, 4) int32 for ``dic_cluster_intra_totals``: every pair of 256-row blocks I <= J of one cluster, sorted by cluster."""
    out, start = [], 0
    for c, n_c in enumerate(int(v) for v in counts_host):
        nb = -(-n_c // 256)
        if nb:
            bi, bj = np.triu_indices(nb)
            t = np.empty((bi.size, 4), np.int32)
            t[:, 0], t[:, 1], t[:, 2], t[:, 3] = start + 256 * bi, start + 256 * bj, start + n_c, c
            out.append(t)
        start += n_c
    return np.concatenate(out) if out else np.zeros((0, 4), np.int32)


# The sums-only all-pairs pass goes to the matrix cores from this many points on.  Its distances come from the norm form on split bf16 planes: a pair's d^2 is off
# by ~2^-17 |x - mean| |y - mean| (random sign), which the sums over thousands of pairs average away (tests: <= 2e-6 of a row's largest sum at 70 001 x 256) but
# which shows on a handful of close points in few dimensions (4e-6 at 63 x 4) -- and below this size the difference-form kernel takes well under a millisecond.
ROWSUM_MIN_POINTS = 8192
ROWSUM_WORKGROUPS = 256     # dic_cluster_pair_rowsums: one persistent workgroup per CU, each a contiguous range of the tile list


def _row_tile_list(counts_host, n):
    """Tile list of ``dic_cluster_pair_rowsums``: every 256-row block I of the (cluster-sorted) points against every 256-row block J of every cluster, sorted by
    (I, cluster, J); (tiles (ntiles, 4) int32 = (I0, J0, end of J's cluster, slot), group_start (blocks x K + 1) int32, n_slots).  A slot = a maximal run of one
    (I, cluster) inside one workgroup's contiguous range of the list."""
    K = len(counts_host)
    j0, jend, jc, start = [], [], [], 0
    for c, n_c in enumerate(int(v) for v in counts_host):
        nb = -(-n_c // 256)
        j0.append(start + 256 * np.arange(nb))
        jend.append(np.full(nb, start + n_c))
        jc.append(np.full(nb, c))
        start += n_c
    j0, jend, jc = (np.concatenate(a).astype(np.int64) for a in (j0, jend, jc))
    n_i, n_j = -(-n // 256), j0.size
    ntiles = n_i * n_j
    tiles = np.empty((ntiles, 4), np.int32)
    tiles[:, 0] = np.repeat(256 * np.arange(n_i), n_j)
    tiles[:, 1], tiles[:, 2] = np.tile(j0, n_i), np.tile(jend, n_i)
    group = np.repeat(np.arange(n_i) * K, n_j) + np.tile(jc, n_i)              # (I, cluster), ascending along the list
    nwg = min(ntiles, ROWSUM_WORKGROUPS)
    per = -(-ntiles // nwg)
    wg = np.arange(ntiles) // per
    new_slot = np.ones(ntiles, bool)
    new_slot[1:] = (group[1:] != group[:-1]) | (wg[1:] != wg[:-1])
    slot = np.cumsum(new_slot) - 1
    tiles[:, 3] = slot
    group_start = np.searchsorted(group[new_slot], np.arange(n_i * K + 1)).astype(np.int32)
    return tiles, group_start, int(slot[-1]) + 1


def intra_totals(x, labels):
    """PairStats with only ``totals`` (K,) f64 = np.sum(pairwise_distances(X[a == c])) per cluster (``dic_cluster_intra_totals``: the distances of a cluster's block
    pairs I <= J on the matrix cores) -- all the two inertia definitions of the gap statistic need (p2_clustering_optK.py:334-351)."""
    x = _device_points(x)
    n, d = x.shape
    lab, K = _encode(labels, x.device)
    if lab.numel() != n:
        raise ValueError('labels: %d entries for %d points' % (lab.numel(), n))
    if d > TOTALS_MAX_D:
        return pair_stats(x, lab, intra_only=True)
    if d % 4:
        x = torch.nn.functional.pad(x, (0, 4 - d % 4))
        d = x.shape[1]
    order = torch.argsort(lab, stable=True)
    xs = x[order].contiguous()
    counts = torch.bincount(lab, minlength=K)
    seg = torch.zeros(K + 1, dtype=torch.int32, device=x.device)
    seg[1:] = torch.cumsum(counts, 0).to(torch.int32)
    centres = (_segment_sum(xs, lab[order], K) / counts.double().clamp(min=1)[:, None]).float().contiguous()
    tiles = torch.from_numpy(_tile_list(counts.cpu().numpy())).to(x.device)
    L = N.lib()
    ws = torch.empty(L.dic_cluster_intra_totals_workspace(n, K), dtype=torch.uint8, device=x.device)
    totals = torch.empty(K, dtype=torch.float64, device=x.device)
    N.check(L.dic_cluster_intra_totals(N.ptr(xs), xs.stride(0), N.ptr(seg), N.ptr(centres), n, d, K, N.ptr(tiles) if tiles.numel() else None, tiles.shape[0],
                                       N.ptr(totals), N.ptr(ws), ws.numel(), N.stream_of(xs)), 'dic_cluster_intra_totals')
    return PairStats(lab, counts, None, None, None, None, totals)


def pair_stats(x, labels, need_min=True, need_max=True, intra_only=False):
    """``intra_only``: only the pairs inside a cluster (``dic_cluster_intra_sums``: sum_c n_c^2 of the N^2 pairs); ``S`` / ``Dmin`` / ``own_max`` are then None and
    ``S_own`` (N) holds each point's own-cluster sum.  (The gap statistic's inertias need only the per-cluster totals: ``intra_totals``.)"""
    x = _device_points(x)
    n, d = x.shape
    lab, K = _encode(labels, x.device)
    if lab.numel() != n:
        raise ValueError('labels: %d entries for %d points' % (lab.numel(), n))
    if d % 4:
        x = torch.nn.functional.pad(x, (0, 4 - d % 4))          # zero columns do not change distances
        d = x.shape[1]
    order = torch.argsort(lab, stable=True)
    xs = x[order].contiguous()
    counts = torch.bincount(lab, minlength=K)
    seg = torch.zeros(K + 1, dtype=torch.int32, device=x.device)
    seg[1:] = torch.cumsum(counts, 0).to(torch.int32)
    if intra_only:
        S_own = torch.empty(n, device=x.device, dtype=torch.float32)
        N.check(N.lib().dic_cluster_intra_sums(N.ptr(xs), N.ptr(seg), n, d, K, N.ptr(S_own), N.stream_of(xs)), 'dic_cluster_intra_sums')
        out = torch.empty_like(S_own)
        out[order] = S_own
        return PairStats(lab, counts, None, None, None, out)
    S = torch.empty((n, K), device=x.device, dtype=torch.float32)
    if not need_min and not need_max and d <= TOTALS_MAX_D and n >= ROWSUM_MIN_POINTS:
        # only the sums (the silhouette): the all-pairs pass on the matrix cores (dic_cluster_pair_rowsums), points relative to their mean
        tiles, group_start, n_slots = _row_tile_list(counts.cpu().numpy(), n)
        tiles, group_start = torch.from_numpy(tiles).to(x.device), torch.from_numpy(group_start).to(x.device)
        centre = xs.mean(0, keepdim=True, dtype=torch.float64).float().contiguous()
        L = N.lib()
        ws = torch.empty(L.dic_cluster_pair_rowsums_workspace(n, n_slots), dtype=torch.uint8, device=x.device)
        N.check(L.dic_cluster_pair_rowsums(N.ptr(xs), xs.stride(0), N.ptr(centre), n, d, K, N.ptr(tiles), tiles.shape[0], N.ptr(group_start), n_slots, N.ptr(S),
                                           N.ptr(ws), ws.numel(), N.stream_of(xs)), 'dic_cluster_pair_rowsums')
        out = torch.empty_like(S)
        out[order] = S
        return PairStats(lab, counts, out, None, None)
    Dmin = torch.empty_like(S) if need_min else None
    omax = torch.empty(n, device=x.device, dtype=torch.float32) if need_max else None
    N.check(N.lib().dic_cluster_pairdist(N.ptr(xs), N.ptr(seg), n, d, K, N.ptr(S), N.ptr(Dmin), N.ptr(omax), N.stream_of(xs)),
            'dic_cluster_pairdist')

    def unsort(t):
        if t is None:
            return None
        out = torch.empty_like(t)
        out[order] = t
        return out
    return PairStats(lab, counts, unsort(S), unsort(Dmin), unsort(omax))


# ---------------------------------------------------------------------------- gap-statistic inertia (p2:334-351)
def inertia_v1(x, labels, stats=None):
    """np.mean([np.mean(pairwise_distances(X[a == c])) for c in unique(a)])  (p2:334-342)."""
    st = stats or intra_totals(x, labels)
    n = st.counts.double()
    return float((st.intra_sums() / (n * n)).mean())


def inertia_v2(x, labels, stats=None):
    """sum_c sum(pairwise_distances(X[a == c])) / (2 n_c)  (p2:344-351)."""
    st = stats or intra_totals(x, labels)
    return float((st.intra_sums() / (2.0 * st.counts.double())).sum())


# ---------------------------------------------------------------------------- validity indices (internal_eval.py)
def silhouette_samples(x, labels, stats=None):
    st = stats or pair_stats(x, labels, need_min=False, need_max=False)
    n_pts = st.labels.numel()
    if not 1 < st.K < n_pts:
        raise ValueError('Number of labels is %d. Valid values are 2 to n_samples - 1 (inclusive)' % st.K)
    S = st.S.double()
    cnt = st.counts.double()
    own_mask = torch.nn.functional.one_hot(st.labels, st.K).bool()
    intra = S.gather(1, st.labels[:, None])[:, 0] / (cnt[st.labels] - 1)             # 0/0 = nan for singletons, as sklearn
    inter = (S / cnt).masked_fill(own_mask, float('inf')).min(1).values
    sil = (inter - intra) / torch.maximum(intra, inter)
    return torch.nan_to_num(sil, nan=0.0)


def silhouette_score(x, labels, stats=None):
    """sklearn.metrics.silhouette_score(x, labels, metric='euclidean') (internal_eval.py:112-123)."""
    return float(silhouette_samples(x, labels, stats).mean())


def dunn_index(x, labels, stats=None):
    """min non-zero nearest-point distance between two clusters / largest farthest-point diameter
    (internal_eval.py:84-110 with the 'nearest' / 'farthest' definitions of :21-82)."""
    st = stats or pair_stats(x, labels)
    K = st.K
    idx = st.labels[:, None].expand(-1, K)
    between = torch.full((K, K), float('inf'), device=st.Dmin.device, dtype=torch.float32)
    between = between.scatter_reduce(0, idx, st.Dmin, reduce='amin', include_self=True)   # [a][b] = min_{i in a} Dmin[i][b]
    off = between[~torch.eye(K, dtype=torch.bool, device=between.device)]
    off = off[off != 0]
    return float(off.min() / st.own_max.max())


def _centroids(x, lab, K):
    x64 = x.double()
    cnt = torch.bincount(lab, minlength=K).double()
    cen = _segment_sum(x64, lab, K) / cnt[:, None]
    return x64, cnt, cen


def calinski_harabasz_score(x, labels):
    """sklearn.metrics.calinski_harabasz_score (internal_eval.py:126-136)."""
    x = _device_points(x)
    lab, K = _encode(labels, x.device)
    n = x.shape[0]
    if not 1 < K < n:
        raise ValueError('Number of labels is %d. Valid values are 2 to n_samples - 1 (inclusive)' % K)
    x64, cnt, cen = _centroids(x, lab, K)
    extra = float((cnt * ((cen - x64.mean(0)) ** 2).sum(1)).sum())
    intra = float(((x64 - cen[lab]) ** 2).sum())
    return 1.0 if intra == 0.0 else extra * (n - K) / (intra * (K - 1.0))


def davies_bouldin_score(x, labels):
    """sklearn.metrics.davies_bouldin_score (internal_eval.py:139-147)."""
    x = _device_points(x)
    lab, K = _encode(labels, x.device)
    n = x.shape[0]
    if not 1 < K < n:
        raise ValueError('Number of labels is %d. Valid values are 2 to n_samples - 1 (inclusive)' % K)
    x64, cnt, cen = _centroids(x, lab, K)
    dist = ((x64 - cen[lab]) ** 2).sum(1).sqrt()
    intra = _segment_sum(dist, lab, K) / cnt
    cd = torch.cdist(cen, cen)
    if bool(torch.allclose(intra, torch.zeros_like(intra))) or bool(torch.allclose(cd, torch.zeros_like(cd))):
        return 0.0
    cd = cd.masked_fill(cd == 0, float('inf'))
    return float(((intra[:, None] + intra[None, :]) / cd).max(1).values.mean())

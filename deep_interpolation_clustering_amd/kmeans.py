"""``KMeans`` with scikit-learn's estimator surface, running on the HIP k-means kernels.

Drop-in for the ``sklearn.cluster.KMeans`` calls of the reference (clustering_trainer.py:75-82,
p2_clustering_optK.py:260-389, p4_clustering_final.py:159-174).  Control flow follows scikit-learn
1.7.2 (``_kmeans.py`` fit :1453-1551, ``_kmeans_single_lloyd`` :624-753, ``_kmeans_plusplus``
:176-262): centre the data, k-means++ seeding driven by NumPy's global ``RandomState`` in the same
draw order, Lloyd iterations with strict-convergence / ``tol`` stops, final E-step, lowest inertia
of ``n_init`` restarts wins.  The MI355X-specific part: all ``n_init`` restarts advance TOGETHER --
one launch per Lloyd iteration covers every restart (grid.y = restart), converged restarts drop out
on a device-side flag, and the host only polls that flag every few iterations.

Outputs (``cluster_centers_``, ``labels_``, ``inertia_``, ``n_iter_``) are NumPy, like sklearn's.
There is no CPU fallback: without a GPU ``fit``/``predict`` raise.
"""
from __future__ import annotations

import numbers

import numpy as np
import torch

from . import _native as N
from . import dist

_POLL_EVERY = 4      # Lloyd iterations enqueued between two host reads of the done flags


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('deep_interpolation_clustering_amd.KMeans needs an MI355X (no CPU fallback by design)')
    return torch.device('cuda', torch.cuda.current_device())


def _as_device_matrix(X, device):
    if isinstance(X, torch.Tensor):
        t = X.detach().to(device=device, dtype=torch.float32)
    else:
        t = torch.as_tensor(np.ascontiguousarray(X, dtype=np.float32), device=device)
    if t.dim() != 2:
        raise ValueError(f'expected a 2-D array, got shape {tuple(t.shape)}')
    return t.contiguous()


def _pad_features(t):
    """Kernels take D % 4 == 0 (one float4 per lane); zero columns leave every distance unchanged."""
    D = t.shape[1]
    if D > N.LATENT_MAX_DIM:
        raise ValueError(f'n_features={D} > {N.LATENT_MAX_DIM}: outside the compiled latent width')
    pad = (-D) % 4
    return torch.nn.functional.pad(t, (0, pad)) if pad else t


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def lloyd(X, xnorm, centers, tol, max_iter):
    """Batched Lloyd on centred device data.  centers (n_runs,K,D) is updated in place.
    Returns (labels (n_runs,N) int32, inertia (n_runs,) f32, n_iter (n_runs,) int list)."""
    L = N.lib()
    n_runs, K, D = centers.shape
    Nn = X.shape[0]
    dev = X.device
    labels = torch.full((n_runs, Nn), -1, dtype=torch.int32, device=dev)
    status = torch.zeros((n_runs, N.KM_STATUS_WORDS), dtype=torch.float32, device=dev)
    status[:, 6] = float(tol)
    status[:, 7] = float(max_iter)
    ws = _ws(L.dic_kmeans_workspace(Nn, D, K, n_runs), dev)
    st = N.stream_of(X)
    it = 0
    while it < max_iter:
        for _ in range(min(_POLL_EVERY, max_iter - it)):
            N.check(L.dic_kmeans_lloyd_iter(N.ptr(X), N.ptr(xnorm), Nn, D, K, n_runs, N.ptr(centers), N.ptr(labels),
                                            N.ptr(status), N.ptr(ws), ws.numel(), st), 'dic_kmeans_lloyd_iter')
            it += 1
        if bool((status[:, 0] != 0).all()):       # the only host sync of the loop
            break
    # final E-step: labels consistent with the final centres (a no-op change for strictly converged runs),
    # exact inertia (_kmeans.py:736-750)
    inertia = torch.empty(n_runs, dtype=torch.float32, device=dev)
    N.check(L.dic_kmeans_predict(N.ptr(X), Nn, D, K, n_runs, N.ptr(centers), N.ptr(labels), None, N.ptr(inertia),
                                 N.ptr(ws), ws.numel(), st), 'dic_kmeans_predict')
    return labels, inertia, status


def lloyd_sharded(X, xnorm, centers, tol, max_iter):
    """``lloyd`` with the points sharded over ranks ("centroid partial sums over RCCL").

    Every rank holds the same X (the trainers compute the latents unsharded) and the same initial centres; rank r iterates
    over rows [r N/P, (r+1) N/P) only.  Per iteration: dic_kmeans_lloyd_partial on the shard, ONE all-reduce of the f64
    statistics (K*D sums + K counts + #changed per restart), dic_kmeans_lloyd_finish on every rank -- so centres, the
    convergence decisions and the iteration counts are identical everywhere.  A restart that meets an empty cluster (its
    relocation needs the globally farthest point) is re-run on the full X by the unsharded kernels, on every rank alike.
    Returns what ``lloyd`` returns, with full-length labels on every rank."""
    L = N.lib()
    n_runs, K, D = centers.shape
    Nn, dev, st = X.shape[0], X.device, N.stream_of(X)
    lo, hi = dist.shard_bounds(Nn)
    if hi <= lo:
        raise ValueError(f'k-means shard of rank {dist.rank()} is empty ({Nn} points over {dist.world_size()} ranks)')
    Xl, xnl = X[lo:hi], xnorm[lo:hi]                       # contiguous row ranges: views
    n_loc = hi - lo
    init = centers.clone()
    labels_l = torch.full((n_runs, n_loc), -1, dtype=torch.int32, device=dev)
    status = torch.zeros((n_runs, N.KM_STATUS_WORDS), dtype=torch.float32, device=dev)
    status[:, 6] = float(tol)
    status[:, 7] = float(max_iter)
    ws = _ws(L.dic_kmeans_workspace(n_loc, D, K, n_runs), dev)
    stats = torch.zeros((n_runs, L.dic_kmeans_stats_words(D, K)), dtype=torch.float64, device=dev)
    it = 0
    while it < max_iter:
        for _ in range(min(_POLL_EVERY, max_iter - it)):
            N.check(L.dic_kmeans_lloyd_partial(N.ptr(Xl), N.ptr(xnl), n_loc, D, K, n_runs, N.ptr(centers), N.ptr(labels_l), N.ptr(status),
                                               N.ptr(stats), N.ptr(ws), ws.numel(), st), 'dic_kmeans_lloyd_partial')
            dist.all_reduce_sum_(stats)                  # converged restarts contribute their (stale, ignored) previous values
            N.check(L.dic_kmeans_lloyd_finish(D, K, n_runs, N.ptr(stats), N.ptr(centers), N.ptr(status), st), 'dic_kmeans_lloyd_finish')
            it += 1
        if bool((status[:, 0] != 0).all()):
            break
    # final E-step on the shard; inertia summed and labels assembled over ranks
    inertia = torch.empty(n_runs, dtype=torch.float32, device=dev)
    N.check(L.dic_kmeans_predict(N.ptr(Xl), n_loc, D, K, n_runs, N.ptr(centers), N.ptr(labels_l), None, N.ptr(inertia), N.ptr(ws),
                                 ws.numel(), st), 'dic_kmeans_predict')
    inertia64 = inertia.double()
    dist.all_reduce_sum_(inertia64)
    inertia = inertia64.float()
    labels = torch.zeros((n_runs, Nn), dtype=torch.int32, device=dev)
    labels[:, lo:hi] = labels_l
    dist.all_reduce_sum_(labels)                         # disjoint row ranges: the sum is the concatenation
    halted = (status[:, 0] == 2).nonzero().flatten().tolist()     # same on every rank (status is a function of reduced data)
    if halted:
        idx = torch.as_tensor(halted, device=dev)
        c_h = init[idx].contiguous()
        lab_h, in_h, st_h = lloyd(X, xnorm, c_h, tol, max_iter)
        centers[idx], labels[idx], inertia[idx], status[idx] = c_h, lab_h, in_h, st_h
    return labels, inertia, status


def _cumsum_f64(x):
    """(n_rows, n) f64 inclusive running sums along the rows of an f32 matrix (dic_cumsum_f64: one workgroup per row, fixed summation order --
    torch.cumsum's innermost-dimension scan took 175 us per call on the 10 x 75 000 potentials of p2's sweep, 2 090 calls)."""
    out = torch.empty(x.shape, dtype=torch.float64, device=x.device)
    xc = x.contiguous()
    N.check(N.lib().dic_cumsum_f64(N.ptr(xc), xc.shape[0], xc.shape[1], N.ptr(out), N.stream_of(xc)), 'dic_cumsum_f64')
    return out


def seed_draw_count(K, n_init):
    """Doubles a k-means++ fit takes from its random stream (``_pp_init``: per restart one ``choice`` = one double, then K-1 times
    ``uniform(size=trials)``) -- independent of the data, so a caller can advance a stream past a fit it does not run (p2's sharded sweep)."""
    return int(n_init) * (1 + (int(K) - 1) * (2 + int(np.log(K))))


def _pp_init(X, K, n_runs, rs, shard=False):
    """k-means++ (greedy, 2+log K local trials) for n_runs restarts at once.  Every random number is
    drawn from ``rs`` up front in scikit-learn's order (per restart: one ``choice`` then K-1
    ``uniform(size=trials)``), which is possible because the Lloyd runs consume no randomness."""
    L = N.lib()
    Nn, D = X.shape
    dev = X.device
    trials = 2 + int(np.log(K))
    p_uniform = np.full(Nn, 1.0 / Nn)
    first = np.empty(n_runs, dtype=np.int64)
    u = np.empty((n_runs, max(K - 1, 0), trials), dtype=np.float64)
    for r in range(n_runs):
        first[r] = rs.choice(Nn, p=p_uniform)
        for c in range(K - 1):
            u[r, c] = rs.uniform(size=trials)
    u = torch.as_tensor(u, device=dev)
    st = N.stream_of(X)
    centers_idx = torch.empty((n_runs, K), dtype=torch.int64, device=dev)
    centers_idx[:, 0] = torch.as_tensor(first, device=dev)
    inf = torch.full((n_runs, Nn), float('inf'), dtype=torch.float32, device=dev)
    closest = torch.empty((n_runs, Nn), dtype=torch.float32, device=dev)
    pot = torch.empty(n_runs, dtype=torch.float64, device=dev)
    ws = _ws(L.dic_kmeans_pp_workspace(Nn, n_runs * trials), dev)
    lo, hi = dist.shard_bounds(Nn) if shard else (0, Nn)

    def candidates(cand, n_cand, group, closest_in, dist_out, pot_out):
        """Squared distances to the candidate rows, min with the running closest distance, potentials.  With the points sharded
        over ranks each rank evaluates its row range and the (L,N) distances / L potentials are summed over ranks (disjoint rows:
        the sum of the zero-initialised buffers is the concatenation) -- every rank then draws the same next centre."""
        if shard:
            dist_out.zero_()
            N.check(L.dic_kmeans_pp_candidates_rows(N.ptr(X), Nn, D, lo, hi, N.ptr(cand), n_cand, group, N.ptr(closest_in), N.ptr(dist_out),
                                                    N.ptr(pot_out), N.ptr(ws), ws.numel(), st), 'dic_kmeans_pp_candidates_rows')
            dist.all_reduce_sum_(dist_out)
            dist.all_reduce_sum_(pot_out)
        else:
            N.check(L.dic_kmeans_pp_candidates(N.ptr(X), Nn, D, N.ptr(cand), n_cand, group, N.ptr(closest_in), N.ptr(dist_out), N.ptr(pot_out),
                                               N.ptr(ws), ws.numel(), st), 'dic_kmeans_pp_candidates')
    candidates(centers_idx[:, 0].contiguous(), n_runs, 1, inf, closest, pot)
    dist_c = torch.empty((n_runs * trials, Nn), dtype=torch.float32, device=dev)
    pot_c = torch.empty(n_runs * trials, dtype=torch.float64, device=dev)
    ar = torch.arange(n_runs, device=dev)
    for c in range(1, K):
        cur = pot.float().double()                                   # current_pot is an f32 scalar upstream
        cum = _cumsum_f64(closest)                                   # stable_cumsum(sample_weight * closest): f64 running sums of the f32 distances
        cand = torch.searchsorted(cum, u[:, c - 1] * cur[:, None]).clamp_(max=Nn - 1)      # (n_runs, trials)
        candidates(cand.contiguous(), n_runs * trials, trials, closest, dist_c, pot_c)
        best = torch.argmin(pot_c.view(n_runs, trials), dim=1)
        centers_idx[:, c] = cand[ar, best]
        closest = dist_c.view(n_runs, trials, Nn)[ar, best].contiguous()
        pot = pot_c.view(n_runs, trials)[ar, best]
    return X[centers_idx]                                            # (n_runs, K, D)


def _same_clustering(a, b, K):
    """sklearn _is_same_clustering: equal up to a permutation of the label ids."""
    pair = torch.unique(a.long() * K + b.long()).numel()
    return pair == torch.unique(a).numel() == torch.unique(b).numel()


class KMeans:
    def __init__(self, n_clusters=8, *, init='k-means++', n_init='auto', max_iter=300, tol=1e-4, verbose=0,
                 random_state=None, copy_x=True, algorithm='lloyd', shard_points=False):
        self.n_clusters = n_clusters
        self.init = init
        self.n_init = n_init
        self.max_iter = max_iter
        self.tol = tol
        self.verbose = verbose
        self.random_state = random_state
        self.copy_x = copy_x
        self.algorithm = algorithm
        # (extra) one process per GPU: every rank passes the SAME X to fit(); the Lloyd iterations then run on this rank's row
        # shard with one all-reduce of centroid partial sums per iteration (lloyd_sharded), and the k-means++ candidate distances
        # are evaluated per row shard too (the random draws stay replicated: same stream on every rank).
        self.shard_points = shard_points

    # -- helpers ------------------------------------------------------------------------------
    def _random_state(self):
        rs = self.random_state
        if rs is None or rs is np.random:
            return np.random.mtrand._rand            # sklearn check_random_state(None): NumPy's global state
        if isinstance(rs, numbers.Integral):
            return np.random.RandomState(rs)
        return rs

    def _resolve_n_init(self, init_is_array):
        n_init = self.n_init
        if n_init == 'auto':                          # sklearn >= 1.4 (_kmeans.py:_check_params_vs_input)
            n_init = 1 if (init_is_array or self.init == 'k-means++') else 10
        if init_is_array and n_init != 1:
            n_init = 1
        return int(n_init)

    # -- estimator API ------------------------------------------------------------------------
    def fit(self, X, y=None, sample_weight=None):
        if sample_weight is not None:
            raise NotImplementedError('sample_weight is not used by the reference call sites')
        if self.algorithm not in ('lloyd', 'auto', 'full'):
            raise NotImplementedError("only algorithm='lloyd' is implemented")
        dev = _device()
        K = int(self.n_clusters)
        if K > N.MAX_CLUSTERS:
            raise ValueError(f'n_clusters={K} > {N.MAX_CLUSTERS}: outside the compiled limit')
        Xd = _as_device_matrix(X, dev)
        Nn, D0 = Xd.shape
        if Nn < K:
            raise ValueError(f'n_samples={Nn} should be >= n_clusters={K}.')
        rs = self._random_state()
        init_is_array = not isinstance(self.init, str)
        n_init = self._resolve_n_init(init_is_array)

        tol_abs = float(Xd.var(dim=0, unbiased=False).mean()) * self.tol      # _tolerance (:279-288)
        mean = Xd.mean(dim=0)                                                 # :1479-1484
        Xc = _pad_features(Xd - mean)
        D = Xc.shape[1]
        xnorm = (Xc * Xc).sum(dim=1)

        if init_is_array:
            c0 = _as_device_matrix(self.init, dev)
            if tuple(c0.shape) != (K, D0):
                raise ValueError(f'The shape of the initial centers {tuple(c0.shape)} does not match (n_clusters, n_features)')
            centers = _pad_features(c0 - mean)[None].contiguous()
        elif self.init == 'k-means++':
            centers = _pp_init(Xc, K, n_init, rs, shard=self.shard_points and dist.is_sharded() and Nn >= dist.world_size()).contiguous()
        elif self.init == 'random':
            p = np.full(Nn, 1.0 / Nn)
            seeds = np.stack([rs.choice(Nn, size=K, replace=False, p=p) for _ in range(n_init)])
            centers = Xc[torch.as_tensor(seeds, device=dev)].contiguous()
        else:
            raise ValueError(f"init must be 'k-means++', 'random' or an array, got {self.init!r}")

        run = lloyd_sharded if (self.shard_points and dist.is_sharded()) else lloyd
        labels, inertia, status = run(Xc, xnorm, centers, tol_abs, int(self.max_iter))
        inertia_h = inertia.cpu().numpy()
        best = 0
        for i in range(1, centers.shape[0]):                                  # :1517-1532
            if inertia_h[i] < inertia_h[best] and not _same_clustering(labels[i], labels[best], K):
                best = i
        self.cluster_centers_ = (centers[best, :, :D0] + mean).cpu().numpy()
        self.labels_ = labels[best].cpu().numpy()
        self.inertia_ = float(inertia_h[best])
        self.n_iter_ = int(status[best, 1].item())
        self.n_features_in_ = D0
        self._status = status.cpu().numpy()
        return self

    def fit_predict(self, X, y=None, sample_weight=None):
        return self.fit(X, sample_weight=sample_weight).labels_

    def predict(self, X):
        """One E-step against the stored centres, no centring (_kmeans.py:1066-1090)."""
        dev = _device()
        Xd = _pad_features(_as_device_matrix(X, dev))
        c = _pad_features(_as_device_matrix(self.cluster_centers_, dev))[None].contiguous()
        Nn, D = Xd.shape
        labels = torch.empty((1, Nn), dtype=torch.int32, device=dev)
        N.check(N.lib().dic_kmeans_predict(N.ptr(Xd), Nn, D, c.shape[1], 1, N.ptr(c), N.ptr(labels), None, None, None, 0,
                                           N.stream_of(Xd)), 'dic_kmeans_predict')
        return labels[0].cpu().numpy()

    def nearest_distance(self, X):
        """(N,) f32 on the device: Euclidean distance of every row of X to its nearest centre -- what p2's elbow curve averages
        (``cdist(X, centers).min(1)``, p2_clustering_optK.py:253-270) -- from the E-step kernel's exact ||x - c_label||^2 instead of an N x K
        distance matrix out of a library GEMM."""
        dev = _device()
        Xd = _pad_features(_as_device_matrix(X, dev))
        c = _pad_features(_as_device_matrix(self.cluster_centers_, dev))[None].contiguous()
        Nn, D = Xd.shape
        labels = torch.empty((1, Nn), dtype=torch.int32, device=dev)
        mind = torch.empty((1, Nn), dtype=torch.float32, device=dev)
        L = N.lib()
        ws = _ws(L.dic_kmeans_workspace(Nn, D, c.shape[1], 1), dev)
        N.check(L.dic_kmeans_predict(N.ptr(Xd), Nn, D, c.shape[1], 1, N.ptr(c), N.ptr(labels), N.ptr(mind), None, N.ptr(ws), ws.numel(),
                                     N.stream_of(Xd)), 'dic_kmeans_predict')
        return mind[0].clamp_min_(0).sqrt_()

    def transform(self, X):
        raise NotImplementedError('transform is not on the reference path')

    def get_params(self, deep=True):
        return dict(n_clusters=self.n_clusters, init=self.init, n_init=self.n_init, max_iter=self.max_iter, tol=self.tol,
                    random_state=self.random_state)

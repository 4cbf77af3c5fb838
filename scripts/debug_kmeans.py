import sys, warnings
import numpy as np
sys.path.insert(0, '.')
warnings.filterwarnings('ignore')
from sklearn.cluster import KMeans as SK
from deep_interpolation_clustering_amd.kmeans import KMeans
from deep_interpolation_clustering_amd.synthetic import latent_blobs
N, D, K = 20000, 256, 8
X, _ = latent_blobs(N + K, N, D, max(2, K // 2), spread=0.3, noise=0.3)
init = X[np.random.default_rng(K).choice(N, K, replace=False)].copy()
ref = SK(n_clusters=K, init=init, n_init=1).fit(X)
km = KMeans(n_clusters=K, init=init, n_init=1).fit(X)
print('sk n_iter', ref.n_iter_, 'inertia', ref.inertia_, ' | hip n_iter', km.n_iter_, 'inertia', km.inertia_, 'status', km._status)
for it in [1, 2, 3, 5, 8, 12, 16, 20, 30, 40, 60, 80, 120]:
    a = SK(n_clusters=K, init=init, n_init=1, max_iter=it, tol=0).fit(X)
    b = KMeans(n_clusters=K, init=init, n_init=1, max_iter=it, tol=0).fit(X)
    bad = np.nonzero(a.labels_ != b.labels_)[0]
    msg = ''
    if bad.size:
        d = ((X[bad, None, :].astype(np.float64) - a.cluster_centers_[None].astype(np.float64)) ** 2).sum(-1)
        d.sort(axis=1)
        msg = 'max margin %.3e' % ((d[:, 1] - d[:, 0]) / d[:, 1]).max()
    print(it, 'n_iter', a.n_iter_, b.n_iter_, 'mismatch', bad.size, msg, 'cdiff %.3e' % np.abs(a.cluster_centers_ - b.cluster_centers_).max())

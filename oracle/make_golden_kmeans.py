#!/usr/bin/env python3
"""Golden vectors for row a10 (k-means) that need no import of the reference: the reference's k-means IS scikit-learn
(clustering_trainer.py:75-82, p2_clustering_optK.py:260-389, p4_clustering_final.py:159-174).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Run in the build container (scikit-learn 1.7.2, one BLAS / OpenMP
thread so that scikit-learn's per-thread partial sums have one reduction order):

    python oracle/make_golden_kmeans.py

Writes
  tests/golden/kmeans_seeded_*.npz   the reference's actual call, ``np.random.seed(7529)`` (utils.py:37-42) followed by
                                     ``KMeans(n_clusters=K, n_init=20).fit(X)`` (clustering_trainer.py:75-76): labels,
                                     centres, inertia, n_iter_ of the winning restart.  X is regenerated from the seed.
  tests/golden/kmeans_step_K8.npz    tie-prone data (8 centres on 4 blobs): scikit-learn's centres after i Lloyd
                                     iterations for six i (the inputs of a single-step audit), with scikit-learn's own
                                     E-step labels from those centres, and its labels / centres one iteration later.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
OUT = os.path.join(HERE, '..', 'tests', 'golden')

SEEDED = {
    # name: (blob seed, N, D, blobs, K, spread, noise)
    'K4_30k': (99, 30000, 256, 4, 4, 0.5, 0.25),
    'K8_6k': (314, 6000, 256, 8, 8, 0.45, 0.25),
    'K3_d64': (2718, 4500, 64, 3, 3, 0.6, 0.3),
    # overlapping blobs in 8 dimensions: the first restart ends in a bad local optimum and a later one replaces it
    # (_kmeans.py:1517-1532); every good restart reaches the same partition, so the choice does not hang on inertia rounding
    'K7_later_restart_wins': (138, 2000, 8, 7, 7, 0.6, 0.3),
}
STEP = dict(seed=20008, N=20000, D=256, blobs=4, K=8, spread=0.3, noise=0.3, its=(1, 4, 9, 14, 25, 40))


def main():
    import sklearn
    from sklearn.cluster import KMeans
    from threadpoolctl import threadpool_limits
    from oracle.synth import latent_blobs
    os.makedirs(OUT, exist_ok=True)
    with threadpool_limits(limits=1):
        for name, (seed, N, D, G, K, spread, noise) in SEEDED.items():
            X, comp = latent_blobs(seed, N, D, G, spread=spread, noise=noise)
            np.random.seed(7529)                                  # utils.set_seed's NumPy part (utils.py:37-42), p3's default seed
            km = KMeans(n_clusters=K, n_init=20).fit(X)           # clustering_trainer.py:75-76
            after = np.random.random_sample(4)                    # where the global stream stands after the fit
            np.savez_compressed(os.path.join(OUT, f'kmeans_seeded_{name}.npz'), seed=seed, N=N, D=D, blobs=G, K=K, spread=spread,
                                noise=noise, labels=km.labels_.astype(np.int8), centers=km.cluster_centers_,
                                inertia=np.float64(km.inertia_), n_iter=km.n_iter_, stream_after=after,
                                sklearn_version=sklearn.__version__)
            print(name, 'inertia', km.inertia_, 'n_iter', km.n_iter_, 'sizes', np.bincount(km.labels_))
        s = STEP
        X, _ = latent_blobs(s['seed'], s['N'], s['D'], s['blobs'], spread=s['spread'], noise=s['noise'])
        init_idx = np.random.default_rng(s['K']).choice(s['N'], s['K'], replace=False)
        init = X[init_idx].copy()
        cs, e_labels, next_labels, next_centers = [], [], [], []
        for it in s['its']:
            c = KMeans(n_clusters=s['K'], init=init, n_init=1, max_iter=it, tol=0).fit(X).cluster_centers_
            a = KMeans(n_clusters=s['K'], init=c, n_init=1, max_iter=1, tol=0).fit(X)
            next_labels.append(a.labels_.astype(np.int8))
            next_centers.append(a.cluster_centers_.copy())
            a.cluster_centers_ = c.copy()                         # predict = one E-step against c (_kmeans.py:1066-1090)
            e_labels.append(a.predict(X).astype(np.int8))
            cs.append(c)
        np.savez_compressed(os.path.join(OUT, 'kmeans_step_K8.npz'), init_idx=init_idx, its=np.array(s['its']),
                            centers=np.stack(cs), estep_labels=np.stack(e_labels), next_labels=np.stack(next_labels),
                            next_centers=np.stack(next_centers), sklearn_version=sklearn.__version__,
                            **{k: v for k, v in s.items() if k != 'its'})
        print('step fixture written')


if __name__ == '__main__':
    main()

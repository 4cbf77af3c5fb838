#!/usr/bin/env python3
"""Golden gap-statistic table from the reference's OWN ``p2_clustering_optK.KM.compute_gap_internal_metric`` (p2:353-410).

TEST INFRASTRUCTURE ONLY.  Run in the build container: ``PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_gap.py``.
Arrays only are written (tests/golden/gap_table_blobs.npz): the seeded inputs' generator arguments, the table (k, gap, ref, act, ref_s and the
internal indices) and the position of NumPy's GLOBAL random stream afterwards.  What it pins beyond the scores on fixed labellings
(oracle/make_golden_stats.py): the draw ORDER on the global stream -- reference set, that fit's k-means++ seeds (n_init restarts), next
reference set, ..., then the fit on the data -- which this package reproduces with the next draw running on a worker thread.
One BLAS / OpenMP thread: scikit-learn's Lloyd is then run-to-run reproducible.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..'))
from make_golden import OUT                              # noqa: E402
from make_golden_stats import import_reference           # noqa: E402

CASE = dict(seed=31, N=600, D=16, G=3, k_max=4, gap_b=3, n_init=2, np_seed=7529)
METRICS = ['Sihouette', 'Davies-Bouldin_Index', 'Calinski-Harabasz', 'Dunn_Index']


def main():
    from sklearn.cluster import KMeans
    from threadpoolctl import threadpool_limits
    from oracle.synth import latent_blobs
    _, p2 = import_reference()
    X, _ = latent_blobs(CASE['seed'], CASE['N'], CASE['D'], CASE['G'])
    with tempfile.TemporaryDirectory() as tmp, threadpool_limits(limits=1):
        km = p2.KM(CASE['k_max'], tmp, METRICS, CASE['n_init'], CASE['gap_b'])
        np.random.seed(CASE['np_seed'])
        df = km.compute_gap_internal_metric(KMeans(n_init=CASE['n_init']), X, CASE['k_max'], n_references=CASE['gap_b'], version=1)
        pos = np.random.random()
    df = df.astype(float)
    print(df)
    np.savez_compressed(os.path.join(OUT, 'gap_table_blobs.npz'), columns=np.array(list(df.columns)), table=df.to_numpy(),
                        stream_pos=np.float64(pos), x_dtype=str(X.dtype), **{k: np.int64(v) for k, v in CASE.items()})


if __name__ == '__main__':
    main()

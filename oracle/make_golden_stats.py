#!/usr/bin/env python3
"""Generate tests/golden/cluster_stats_*.npz by IMPORTING the reference's internal_eval / p2 modules (read-only) on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the build container: ``PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_stats.py``.
Only arrays are written (points, labels, the reference's scores).  p2 imports plotting / knee-finding packages that
are not installed (seaborn, kneed) and, through utils, tensorflow / warmup_scheduler: inert stand-ins, none of them
touches the arithmetic captured here.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import REF, OUT, _standin          # noqa: E402


def import_reference():
    from types import SimpleNamespace
    tf = _standin('tensorflow')
    tf.random = SimpleNamespace(set_seed=lambda s: None)
    _standin('warmup_scheduler', GradualWarmupScheduler=object)
    _standin('seaborn')
    _standin('kneed', KneeLocator=object)
    import matplotlib
    matplotlib.use('Agg')
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import internal_eval
    import p2_clustering_optK as p2
    return internal_eval, p2


def cases():
    rng = np.random.default_rng(20240517)
    # well separated blobs, labels = blob id
    cen = rng.normal(0, 4, (3, 8))
    lab = rng.integers(0, 3, 240)
    yield 'blobs_K3_D8', (cen[lab] + rng.normal(0, 1, (240, 8))).astype(np.float32), lab
    # overlapping clusters with arbitrary assignment, uneven sizes, latent width of the model
    lab = rng.choice(5, 203, p=[0.5, 0.2, 0.15, 0.1, 0.05])
    yield 'overlap_K5_D256', rng.normal(0, 1, (203, 256)).astype(np.float32) + 0.3 * lab[:, None].astype(np.float32), lab
    # a singleton cluster and duplicated points that sit in different clusters (zero inter-cluster distance)
    x = rng.normal(0, 1, (130, 6)).astype(np.float32)
    lab = rng.integers(0, 3, 130)
    lab[lab == 3] = 0
    lab[7] = 3                        # singleton
    x[11] = x[12]
    lab[11], lab[12] = 0, 1           # duplicate across clusters
    yield 'edge_K4_D6', x, lab


def main():
    ie, p2 = import_reference()
    km = p2.KM.__new__(p2.KM)         # only the two inertia methods are used; no constructor side effects (it makes directories)
    for name, x, lab in cases():
        lab = lab.astype(np.int64)
        out = dict(x=x, labels=lab,
                   inertia_v1=km.compute_inertia_v1(lab, x), inertia_v2=km.computer_intertia_v2(lab, x),
                   dunn=ie.DunnIndex()(x, lab), silhouette=ie.Sihouette()(x, lab),
                   calinski_harabasz=ie.CHIndex()(x, lab), davies_bouldin=ie.DBIndex()(x, lab))
        np.savez_compressed(os.path.join(OUT, 'cluster_stats_%s.npz' % name), **out)
        print(name, {k: float(v) for k, v in out.items() if k not in ('x', 'labels')})


if __name__ == '__main__':
    main()

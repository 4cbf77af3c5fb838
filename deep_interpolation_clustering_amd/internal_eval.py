"""Cluster validity indices used by p2 (internal_eval.py:112-147 upstream): thin callables over scikit-learn.
Post-hoc analysis, not on the accelerated path; ``DunnIndex`` (an O(N^2) Python loop
upstream, internal_eval.py:37-110) is evaluated from a chunked distance pass instead."""
import numpy as np
from sklearn import metrics


class Sihouette(object):
    def __call__(self, x, labels, *args, **kwargs):
        return metrics.silhouette_score(x, labels, metric=kwargs.get('metrics', 'euclidean'))


class CHIndex(object):
    def __call__(self, x, labels, *args, **kwargs):
        return metrics.calinski_harabasz_score(x, labels)


class DBIndex(object):
    def __call__(self, x, label, *args, **kwargs):
        return metrics.davies_bouldin_score(x, label)


class DunnIndex(object):
    """min nearest inter-cluster distance / max farthest intra-cluster diameter."""

    def __call__(self, x, labels, *args, **kwargs):
        x, labels = np.asarray(x), np.asarray(labels)
        ids = np.unique(labels)
        min_inter, max_diam = np.inf, 0.0
        for a, i in enumerate(ids):
            xi = x[labels == i]
            for d in metrics.pairwise_distances_chunked(xi):
                max_diam = max(max_diam, float(d.max()))
            for j in ids[a + 1:]:
                for d in metrics.pairwise_distances_chunked(xi, x[labels == j]):
                    min_inter = min(min_inter, float(d.min()))
        return min_inter / max_diam

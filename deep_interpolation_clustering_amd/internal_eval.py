"""Drop-in for the reference's ``internal_eval`` module (internal_eval.py:15-147): the four cluster validity
callables p2 looks up by name.  Same call convention (``metric(x, labels) -> float``) and error behaviour (scikit-learn's
ValueError for fewer than 2 / as many as N clusters); the arithmetic runs on the device (cluster_stats.py): all distance
work is one tiled pass of ``dic_cluster_pairdist`` instead of an (N,N) float64 matrix (silhouette) or an O(N^2) Python
loop (Dunn).  ``stats=`` lets a caller that evaluates several indices on one labelling share the pair pass."""
from . import cluster_stats as _cs


class Sihouette(object):
    def __call__(self, x, labels, *args, **kwargs):
        metric = kwargs.get('metrics', 'euclidean')
        if metric != 'euclidean':
            raise NotImplementedError("only the 'euclidean' silhouette of the reference path is provided")
        return _cs.silhouette_score(x, labels, kwargs.get('stats'))


class CHIndex(object):
    def __call__(self, x, labels, *args, **kwargs):
        return _cs.calinski_harabasz_score(x, labels)


class DBIndex(object):
    def __call__(self, x, label, *args, **kwargs):
        return _cs.davies_bouldin_score(x, label)


class DunnIndex(object):
    """min nearest inter-cluster distance / max farthest intra-cluster diameter (internal_eval.py:84-110)."""

    def __call__(self, x, labels, *args, **kwargs):
        return _cs.dunn_index(x, labels, kwargs.get('stats'))

"""p4: final cluster labels for every cohort (p4_clustering_final.py:30-313).

``kmeans``: KMeans(k, n_init=20) on the training latents (HIP kernels), clusters re-numbered by descending mean
systolic pressure so ids are comparable across runs (generate_align_map, p4:63-98), then ``predict`` on each cohort.
``dl``: argmax of the network's own soft assignment.  Reads Results/Clustering/out_feat/<metric>/<cohort>.npy,
writes .../<metric>_<method>_aligned/<cohort>_<k>.npy with an added ``cluster_id``.  The dbscan / consensus
branches of the upstream script are host-side alternatives and are not provided.
"""
import argparse
import copy
import os
import os.path as osp

import numpy as np

from .info import COHORTS
from .kmeans import KMeans
from .utils import logger, print_dict_byline

np.random.seed(123)        # p4_clustering_final.py:24


def get_arguments(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--cluster_method', default='kmeans', choices=['kmeans', 'dbscan', 'dl', 'optics', 'consensus'])
    p.add_argument('--num_clusters', type=int, default=4, help='The number of cluster centers')
    p.add_argument('--restore_metric', default=['ae_mse', 'loss', 'delta'])
    p.add_argument('--opt_eps', type=float, default=1.9)
    p.add_argument('--dl_cluster_label_type', default='pred', choices=['label', 'pred'])
    return p.parse_args(argv)


class Cluster(object):
    KEEP = ['encounter_id', 'hidden', 'ob', 'padding_mask']

    def __init__(self, args):
        self.args = args
        self.exp_path = os.path.join(os.getcwd(), 'Results', 'Clustering')

    def load_data(self):
        keep = self.KEEP + (['cluster_pred', 'cluster_label'] if self.args.cluster_method == 'dl' else [])
        cohorts = []
        for cohort in COHORTS:
            full = np.load(osp.join(self.feat_path, '{}.npy'.format(cohort)), allow_pickle=True).item()
            cohorts.append({k: full[k] for k in keep})
            logger.info('Cohort: {}, Sample: {}'.format(cohort, len(full['encounter_id'])))
        self.train_data, self.valid_data, self.test_data = cohorts
        self.feat_dim = self.train_data['hidden'].shape[-1]

    def generate_align_map(self, org_label, ob, padding, feat=None):
        """{old id -> new id} with new ids ordered by descending per-cluster mean of channel 0 (sbp)."""
        pad0 = padding[:, 0, :]
        per_enc = np.sum(ob[:, 0, :] * pad0, axis=1) / np.sum(pad0, axis=1)
        n_clusters = len(set(org_label)) - (1 if -1 in org_label else 0)
        members = [np.where(org_label == i) for i in range(n_clusters)]
        order = np.argsort([np.average(per_enc[m]) for m in members])[::-1]
        align_map = {int(prev): cur for cur, prev in enumerate(order)}
        align_map = {k: align_map[k] for k in sorted(align_map)}
        logger.info('Align_map: {}'.format(align_map))
        for old, new in align_map.items():
            org_label[members[old]] = new
        centers = [np.mean(feat[org_label == i], axis=0) for i in range(n_clusters)] if feat is not None else []
        return align_map, org_label, centers

    def pred(self, **kwargs):
        overwrite = kwargs.get('overwrite', False)
        for metric in self.args.restore_metric:
            self.feat_path = osp.join(self.exp_path, 'out_feat', metric)
            self.out_path = osp.join(self.exp_path, 'out_feat', '{}_{}'.format(metric, self.args.cluster_method)) + '_aligned'
            os.makedirs(self.out_path, exist_ok=True)
            self.load_data()
            cohorts = list(zip(COHORTS, [self.train_data, self.valid_data, self.test_data]))
            if self.args.cluster_method == 'kmeans':
                k = self.args.num_clusters
                logger.info('==> Generate the k-means results with opt-k: {}'.format(k))
                model = KMeans(n_clusters=k, init='k-means++', n_init=20).fit(self.train_data['hidden'])
                raw = model.predict(self.train_data['hidden'])
                align_map, _, _ = self.generate_align_map(raw, self.train_data['ob'], self.train_data['padding_mask'])
                independent = copy.deepcopy(model.cluster_centers_)
                for old, new in align_map.items():
                    model.cluster_centers_[new] = independent[old]
                for cohort, data in cohorts:
                    f = osp.join(self.out_path, '{}_{}.npy'.format(cohort, k))
                    if osp.exists(f) and not overwrite:
                        logger.info('Not Save for {}.'.format(f))
                        continue
                    data['cluster_id'] = model.predict(data['hidden'])
                    del data['ob'], data['padding_mask']
                    np.save(f, data)
                    logger.info('Cohort clustering: {} is done. Save to {}'.format(cohort, f))
            elif self.args.cluster_method == 'dl':
                for cohort, data in cohorts:
                    prob = data['cluster_label'] if self.args.dl_cluster_label_type == 'label' else data['cluster_pred']
                    data['cluster_id'] = np.argmax(prob, 1)
                    del data['cluster_pred'], data['cluster_label']
                    f = osp.join(self.out_path, '{}_{}.npy'.format(cohort, prob.shape[1]))
                    if osp.exists(f) and not overwrite:
                        logger.info('Not Save for {}.'.format(f))
                        continue
                    np.save(f, data)
                    logger.info('Cohort clustering: {} is done. Save to {}'.format(cohort, f))
            else:
                raise NotImplementedError("only 'kmeans' and 'dl' are on the accelerated path")


def main(args):
    Cluster(args).pred()


if __name__ == '__main__':
    _args = get_arguments()
    print_dict_byline(vars(_args))
    main(_args)

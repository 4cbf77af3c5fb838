"""p2: choose the number of clusters on the saved latents -- elbow and gap statistic over k = 2..k_max
(p2_clustering_optK.py:225-410) with every k-means fit on the HIP kernels.

    cd <run dir>;  python -m deep_interpolation_clustering_amd.p2_clustering_optK --k_max 10

Reads Results/Pretrain/out_feat/<metric>/<cohort>.npy (written by p1), writes
Results/Pretrain/out_feat/<metric>_kmeans_aligned/plot/{elbow.csv, gap_sts_v1.csv}.  The gap statistic's
"mean intra-cluster pairwise distance" and the validity indices come from one tiled all-pairs pass on the GPU
(cluster_stats.py / csrc/dic_pairdist.hip) instead of an n_c x n_c float64 matrix per cluster (multi-GB at 75 k points).  DBSCAN / OPTICS and the seaborn plots of the
upstream script are alternative algorithms / presentation and are not provided.
"""
import argparse
import concurrent.futures
import os
import os.path as osp

import numpy as np
import pandas as pd
import torch

from . import cluster_stats
from .info import COHORTS
from .internal_eval import CHIndex, DBIndex, DunnIndex, Sihouette
from .kmeans import KMeans
from .utils import logger, print_dict_byline

np.random.seed(123)        # p2_clustering_optK.py:23


def get_arguments(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--cluster_method', default='kmeans', choices=['kmeans', 'dbscan', 'dl', 'optics', 'consensus'])
    p.add_argument('--k_max', type=int, default=10, help='The max value of k, for k-means only.')
    p.add_argument('--select_opt_k', default=['gap_sts', 'elbow'])
    p.add_argument('--select_eps', type=str, default='k_distance_graph')
    p.add_argument('--n_init', type=int, default=10, help='The number of initialization for k-means.')
    p.add_argument('--gap_b', type=int, default=10, help='The number of randomly sampling for gap-sts.')
    p.add_argument('--restore_metric', default=['ae_mse', 'loss'])
    p.add_argument('--opt_eps', type=float, default=1.9)
    p.add_argument('--internal_metrics', default=['Sihouette', 'Davies-Bouldin_Index', 'Calinski-Harabasz'])
    p.add_argument('--metric_sample', type=int, default=0, help='(extra) subsample size for the O(N^2) validity indices; 0 = all')
    return p.parse_args(argv)


def global_uniform_into(out):
    """``out[...] = np.random.random_sample(out.shape)`` -- the same doubles from NumPy's GLOBAL legacy stream, which is left where that
    call would leave it -- written into a caller-owned buffer.  (A Generator over a copy of the global MT19937 state produces the
    identical sequence; filling a reused, pinned buffer instead of a fresh 154 MB array per reference set avoids its page faults:
    0.15 -> 0.03 s per 75 000 x 256 draw, which had become the bottleneck of the K sweep.)"""
    st = np.random.get_state()
    if st[0] != 'MT19937':
        out[...] = np.random.random_sample(out.shape)
        return out
    bg = np.random.MT19937()
    bg.state = {'bit_generator': 'MT19937', 'state': {'key': st[1], 'pos': st[2]}}
    np.random.Generator(bg).random(out=out)
    ns = bg.state['state']
    np.random.set_state(('MT19937', ns['key'], ns['pos'], st[3], st[4]))
    return out


class KM(object):
    def __init__(self, k_max, out_path, internal_metrics, n_init, gap_b, metric_sample=0):
        self.k_max, self.n_init, self.gap_b, self.metric_sample = k_max, n_init, gap_b, metric_sample
        self.out_path = osp.join(out_path, 'plot')
        os.makedirs(self.out_path, exist_ok=True)
        self.internal_metrics_names = internal_metrics
        table = {'Dunn_Index': DunnIndex, 'Sihouette': Sihouette, 'Davies-Bouldin_Index': DBIndex, 'Calinski-Harabasz': CHIndex}
        self.internal_metrics = [table[n]() for n in internal_metrics]

    # -- inertia definitions of the gap statistic (p2:334-351): one tiled pair pass on the device, nothing n x n
    def compute_inertia_v1(self, a, X, stats=None):
        return cluster_stats.inertia_v1(X, a, stats)

    def computer_intertia_v2(self, a, X, stats=None):
        return cluster_stats.inertia_v2(X, a, stats)

    def elbow(self, train_feat, valid_feat):
        rows = []
        dev = torch.device('cuda')
        tr, va = torch.as_tensor(train_feat, device=dev), torch.as_tensor(valid_feat, device=dev)
        for k in range(2, self.k_max + 1):
            logger.info('Running K: {}'.format(k))
            km = KMeans(n_clusters=k, init='k-means++').fit(tr)                  # n_init='auto' -> 1, as sklearn >= 1.4
            c = torch.as_tensor(km.cluster_centers_, device=dev)
            rows.append(dict(k=k, train=float(torch.cdist(tr, c).min(1).values.mean()),
                             valid=float(torch.cdist(va, c).min(1).values.mean())))
        df = pd.DataFrame(rows)
        df.to_csv(osp.join(self.out_path, 'elbow.csv'), index=False)
        return df

    def compute_gap_internal_metric(self, data, k_max=5, n_references=5, version=1):
        """Gap statistic (p2:353-410): uniform reference sets over the data's bounding range, NumPy global RNG."""
        data = np.asarray(data)
        lo, rng_ = float(data.min()), float(data.max() - data.min())
        logger.info('Data max: {}, min: {}, rng: {}'.format(data.max(), data.min(), rng_))
        dev = torch.device('cuda')
        Xd = torch.as_tensor(data, dtype=torch.float32, device=dev)
        inertia = self.compute_inertia_v1 if version == 1 else self.computer_intertia_v2
        rows = []
        bufs = [torch.empty(data.shape, dtype=torch.float64, pin_memory=True).numpy() for _ in range(2)]      # reference draws (double-buffered)
        for k in range(2, k_max + 1):
            local = []
            # Reference sets come from NumPy's global stream in upstream's order: draw(ref_1), seeds(fit_1), draw(ref_2), ...  A fit
            # draws all its k-means++ randomness before its GPU work starts, so draw(ref_{i+1}) runs on a worker thread during the
            # GPU work of fit_i / its pair pass (0.15 s of host time per 75 k x 256 set, about a third of the sweep otherwise)
            pending = None
            with concurrent.futures.ThreadPoolExecutor(max_workers=1) as pool:
                for i in range(n_references):
                    u = pending.result() if pending is not None else global_uniform_into(bufs[i & 1])
                    pending = None
                    refd = torch.as_tensor(u).to(dev).mul_(rng_).add_(lo).float()         # scale / shift / f32 cast on the device (f64 math as upstream)
                    torch.cuda.current_stream().synchronize()                              # the host buffer is refilled two draws later
                    km = KMeans(n_clusters=k, n_init=self.n_init)
                    if i + 1 < n_references and km.init == 'k-means++':
                        def start_next_draw(nxt=bufs[(i + 1) & 1]):
                            nonlocal pending
                            pending = pool.submit(global_uniform_into, nxt)
                        km._after_seeding = start_next_draw
                    local.append(inertia(km.fit_predict(refd), refd))
            ref_mean, ref_std = np.mean(np.log(local)), np.std(np.log(local))
            ref_s = np.sqrt(1 + 1 / n_references) * ref_std
            assignments = KMeans(n_clusters=k, n_init=self.n_init).fit_predict(Xd)
            need_minmax = any(isinstance(m, DunnIndex) for m in self.internal_metrics)
            stats = cluster_stats.pair_stats(Xd, assignments, need_min=need_minmax, need_max=need_minmax)
            act = np.log(inertia(assignments, Xd, stats))      # the same pair pass feeds the gap term and every index
            gap = ref_mean - act
            if self.metric_sample and self.metric_sample < len(data):
                pick = np.random.RandomState(0).choice(len(data), self.metric_sample, replace=False)
                vals = [m(Xd[torch.as_tensor(pick, device=dev)], assignments[pick]) for m in self.internal_metrics]
            else:
                vals = [m(Xd, assignments, stats=stats) for m in self.internal_metrics]
            logger.info('k: {}, gap: {:.4f}, ref: {:.4f}, act: {:.4f}, ref_s: {:.4f} '.format(k, gap, ref_mean, act, ref_s)
                        + ' '.join('{}: {:.4f}'.format(n, v) for n, v in zip(self.internal_metrics_names, vals)))
            rows.append([k, gap, ref_mean, act, ref_s] + vals)
        return pd.DataFrame(rows, columns=['k', 'gap', 'ref', 'act', 'ref_s'] + list(self.internal_metrics_names))

    def train(self, train_data, valid_data, select_opt_k, **kwargs):
        overwrite = kwargs.get('overwrite', False)
        out = {}
        for method in select_opt_k:
            if method == 'elbow':
                out['elbow'] = self.elbow(train_data['hidden'], valid_data['hidden'])
            elif method == 'gap_sts':
                csv = osp.join(self.out_path, 'gap_sts_v1.csv')
                if osp.exists(csv) and not overwrite:
                    logger.info('Load the previous gat_sts.csv')
                    out['gap_sts'] = pd.read_csv(csv)
                else:
                    df = self.compute_gap_internal_metric(train_data['hidden'], self.k_max, n_references=self.gap_b, version=1)
                    df = df.astype(float)
                    df.to_csv(csv, index=False)
                    out['gap_sts'] = df
        return out


class Cluster(object):
    def __init__(self, args):
        self.args = args
        self.exp_path = os.path.join(os.getcwd(), 'Results', 'Pretrain')

    def load_data(self):
        cohorts = []
        for cohort in COHORTS:
            full = np.load(osp.join(self.feat_path, '{}.npy'.format(cohort)), allow_pickle=True).item()
            cohorts.append({k: full[k] for k in ['encounter_id', 'hidden', 'ob', 'padding_mask']})
            logger.info('Cohort: {}, Sample: {}'.format(cohort, len(full['encounter_id'])))
        self.train_data, self.valid_data, self.test_data = cohorts
        self.feat_dim = self.train_data['hidden'].shape[-1]

    def select_opt_k(self):
        results = {}
        for metric in self.args.restore_metric:
            self.feat_path = osp.join(self.exp_path, 'out_feat', metric)
            self.out_path = osp.join(self.exp_path, 'out_feat', '{}_{}'.format(metric, self.args.cluster_method)) + '_aligned'
            os.makedirs(self.out_path, exist_ok=True)
            self.load_data()
            if self.args.cluster_method != 'kmeans':
                raise NotImplementedError("only --cluster_method kmeans is on the accelerated path")
            km = KM(self.args.k_max, self.out_path, self.args.internal_metrics, self.args.n_init, self.args.gap_b,
                    self.args.metric_sample)
            results[metric] = km.train(self.train_data, self.valid_data, self.args.select_opt_k)
        return results


def main(args):
    return Cluster(args).select_opt_k()


if __name__ == '__main__':
    _args = get_arguments()
    print_dict_byline(vars(_args))
    main(_args)

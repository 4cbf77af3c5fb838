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
import os
import queue
import threading
import os.path as osp

import numpy as np
import pandas as pd
import torch

from . import cluster_stats, dist
from .info import COHORTS
from .internal_eval import CHIndex, DBIndex, DunnIndex, Sihouette
from .kmeans import KMeans, seed_draw_count
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


def _random_state_at(state):
    rs = np.random.RandomState()
    rs.set_state(state)
    return rs


class _StreamWalker:
    """Walks NumPy's global stream through the gap statistic's draws in upstream's order on a worker thread (see
    ``KM.compute_gap_internal_metric``).  Iterating yields, for the problems of this rank only and in stream order,
    ``(k index, i, buffer, state)``: ``i < n_references`` is a reference fit (``buffer`` = its uniform draws, to be handed back with
    ``release``), ``i == n_references`` the fit on the data (``buffer`` None); ``state`` is the global stream's state where that fit's
    seeding starts.  Problems of other ranks only advance the stream (their uniform draws go to a scratch buffer)."""

    def __init__(self, shape, ks, n_references, n_init, rank=0, world=1, n_buffers=2):
        self.free = queue.Queue()
        for _ in range(n_buffers):
            self.free.put(torch.empty(shape, dtype=torch.float64, pin_memory=torch.cuda.is_available()).numpy())
        self.out = queue.Queue()
        self.shape, self.ks, self.n_ref, self.n_init, self.rank, self.world = tuple(shape), ks, n_references, n_init, rank, world
        self.stop = threading.Event()           # set by the consumer when it gives up: the walker must not keep moving the GLOBAL stream
        self.thread = threading.Thread(target=self._walk, daemon=True)
        self.thread.start()

    def _walk(self):
        try:
            scratch = None
            j = 0
            for ki, k in enumerate(self.ks):
                n_seed = seed_draw_count(k, self.n_init)
                for i in range(self.n_ref + 1):
                    if self.stop.is_set():
                        return
                    mine = j % self.world == self.rank
                    j += 1
                    buf = None
                    if i < self.n_ref:
                        if mine:
                            buf = None
                            while buf is None and not self.stop.is_set():        # (a consumer that raised never releases a buffer)
                                try:
                                    buf = self.free.get(timeout=0.2)
                                except queue.Empty:
                                    pass
                            if buf is None:
                                return
                        else:
                            buf = scratch = np.empty(self.shape, np.float64) if scratch is None else scratch
                        global_uniform_into(buf)
                    state = np.random.get_state()
                    np.random.random_sample(n_seed)               # what the fit's k-means++ seeding consumes
                    if mine:
                        self.out.put((ki, i, buf, state))
            self.out.put(None)
        except BaseException as e:                                # surfaced by the consumer
            self.out.put(e)

    def __iter__(self):
        while True:
            item = self.out.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def release(self, buf):
        self.free.put(buf)

    def close(self):
        """Stop walking (the consumer raised or is done) and wait for the thread: nothing touches NumPy's global stream afterwards."""
        self.stop.set()
        self.thread.join()

    def join(self):
        self.thread.join()


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
            # mean distance to the nearest centre (p2:253-270's cdist(...).min(1).mean()) from the E-step kernel: no N x K matrix, no library GEMM
            rows.append(dict(k=k, train=float(km.nearest_distance(tr).mean()), valid=float(km.nearest_distance(va).mean())))
        df = pd.DataFrame(rows)
        if dist.rank() == 0:
            df.to_csv(osp.join(self.out_path, 'elbow.csv'), index=False)
        return df

    @staticmethod
    def _staged(walker, dev, lo, rng_):
        """The walker's items with every reference set already on its way to the device: ``(k index, i, reference set (N, D) f32 on the device or None, state)``.
        The upload of the NEXT set (pinned buffer -> device, scale / shift in f64 as upstream, f32 cast) is issued on a side stream before the fits of the
        current one are queued, so the 154 MB transfer (3 ms of an otherwise idle GPU per set, 190 sets per sweep) runs under them; the pinned buffer goes
        back to the walker as soon as its upload has finished."""
        side = torch.cuda.Stream(device=dev)
        it = iter(walker)

        def stage(item):
            if item is None:
                return None
            ki, i, buf, state = item
            if buf is None:
                return ki, i, None, state, None
            with torch.cuda.stream(side):
                refd = torch.as_tensor(buf).to(dev, non_blocking=True).mul_(rng_).add_(lo).float()
                done = torch.cuda.Event()
                done.record(side)
            return ki, i, refd, state, (done, buf)

        nxt = stage(next(it, None))
        while nxt is not None:
            ki, i, refd, state, pending = nxt
            if pending is not None:
                done, buf = pending
                done.synchronize()                                   # (issued a whole fit ago)
                walker.release(buf)                                  # the walker refills it for a later set
                torch.cuda.current_stream().wait_event(done)
                refd.record_stream(torch.cuda.current_stream())      # allocated on the side stream, consumed on this one
            nxt = stage(next(it, None))
            yield ki, i, refd, state

    def _fit_problems(self, walker, table, ks, n_references, Xd, data, lo, rng_, dev, inertia, need_minmax):
        """The fits of this rank's (K, reference set) problems, in stream order (see compute_gap_internal_metric)."""
        for ki, i, refd, state in self._staged(walker, dev, lo, rng_):
            k = ks[ki]
            km = KMeans(n_clusters=k, n_init=self.n_init, random_state=_random_state_at(state))
            # the walker skipped seed_draw_count(k, n_init) doubles for this fit WITHOUT running it: only right for k-means++ seeding with
            # exactly that many restarts
            if km.init != 'k-means++' or km._resolve_n_init(False) != walker.n_init:
                raise RuntimeError(f'gap statistic: the stream walk assumes k-means++ seeding with n_init={walker.n_init}, got init={km.init!r}, '
                                   f'n_init={km._resolve_n_init(False)}')
            if i < n_references:
                table[ki, i] = np.log(inertia(km.fit_predict(refd), refd))
                continue
            assignments = km.fit_predict(Xd)
            stats = cluster_stats.pair_stats(Xd, assignments, need_min=need_minmax, need_max=need_minmax)
            table[ki, n_references] = np.log(inertia(assignments, Xd, stats))      # the same pair pass feeds the gap term and every index
            if self.metric_sample and self.metric_sample < len(data):
                pick = np.random.RandomState(0).choice(len(data), self.metric_sample, replace=False)
                vals = [m(Xd[torch.as_tensor(pick, device=dev)], assignments[pick]) for m in self.internal_metrics]
            else:
                vals = [m(Xd, assignments, stats=stats) for m in self.internal_metrics]
            table[ki, n_references + 1:] = vals
        return table

    def compute_gap_internal_metric(self, data, k_max=5, n_references=5, version=1):
        """Gap statistic (p2:353-410): uniform reference sets over the data's bounding range, NumPy global RNG.

        Upstream's draw order on the global stream is  draw(ref_1) seeds(fit_1) draw(ref_2) ... seeds(fit on the data)  for K = 2, 3, ...
        A fit draws all its k-means++ randomness before its first distance (``kmeans.seed_draw_count`` doubles: data-independent), so
        the stream can be walked without doing any fit: ``_StreamWalker`` does that on a worker thread, handing each fit the stream state
        its seeding starts from (the fit then draws from a private ``RandomState`` in that state) and each reference fit its uniform
        draws in a pinned buffer.  The walk therefore runs ahead of the GPU work (0.03 s of host time per 75 k x 256 set), and with
        one process per GPU every rank walks the SAME stream but fits only the (K, reference set) problems it owns
        (problem j -> rank j mod world; SURVEY.md 8e: no data-path collective, one all-reduce of the result table at the end)."""
        data = np.asarray(data)
        lo, rng_ = float(data.min()), float(data.max() - data.min())
        logger.info('Data max: {}, min: {}, rng: {}'.format(data.max(), data.min(), rng_))
        dev = torch.device('cuda', torch.cuda.current_device())
        Xd = torch.as_tensor(data, dtype=torch.float32, device=dev)
        inertia = self.compute_inertia_v1 if version == 1 else self.computer_intertia_v2
        world, rank = (dist.world_size(), dist.rank()) if dist.is_sharded() else (1, 0)
        ks = list(range(2, k_max + 1))
        n_met = len(self.internal_metrics)
        # per K: [log-inertia of every reference fit | act | validity indices]; a rank fills the entries of its own problems only
        table = np.zeros((len(ks), n_references + 1 + n_met), np.float64)
        need_minmax = any(isinstance(m, DunnIndex) for m in self.internal_metrics)
        walker = _StreamWalker(data.shape, ks, n_references, KMeans(n_init=self.n_init)._resolve_n_init(False), rank, world)
        try:
            table = self._fit_problems(walker, table, ks, n_references, Xd, data, lo, rng_, dev, inertia, need_minmax)
        finally:
            walker.close()
        if world > 1 or dist.is_sharded():

            t = torch.as_tensor(table, device=dev)
            dist.all_reduce_sum_(t)                 # disjoint entries, zeros elsewhere: the sum is the assembled table (x + 0.0 is exact)
            table = t.cpu().numpy()
        rows = []
        for ki, k in enumerate(ks):
            local = table[ki, :n_references]
            ref_mean, ref_std = np.mean(local), np.std(local)
            ref_s = np.sqrt(1 + 1 / n_references) * ref_std
            act = table[ki, n_references]
            gap = ref_mean - act
            vals = list(table[ki, n_references + 1:])
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
                    if dist.rank() == 0:
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
    dist.init_from_env()            # one process per GPU: the gap statistic's (K, reference set) problems are dealt over the ranks
    out = Cluster(args).select_opt_k()
    dist.barrier()
    return out


if __name__ == '__main__':
    _args = get_arguments()
    print_dict_byline(vars(_args))
    main(_args)

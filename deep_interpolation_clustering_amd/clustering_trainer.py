"""Drop-in for the reference's ``clustering_trainer`` module: ``TrainerCluster`` for the joint
interpolation + DEC stage (clustering_trainer.py:21-545).

The centroid initialisation runs on the GPU: latents never leave HBM between the feature pass and
k-means (upstream: per-batch D2H, NumPy concatenation, scikit-learn on the host), and
``KMeans(n_clusters=K, n_init=20)`` is this package's HIP implementation with scikit-learn's control
flow and NumPy's global random stream (so ``set_seed`` governs it exactly as upstream).
"""
import os

import numpy as np
import torch

from . import dist
from ._trainer_common import TrainerBase
from .info import COHORT2SCOPE
from .kmeans import KMeans
from .utils import format_metric_dict, logger, timer


class TrainerCluster(TrainerBase):
    restore_attr = 'dc_restore_metric'

    def __init__(self, args, model, dl_dict, exp_path, pretrain_exp_path, device, **kwags):
        super().__init__(args, model, dl_dict, exp_path, device, **kwags)
        self.pretrain_exp_path = pretrain_exp_path
        self.init_cluster_center = model.init_cluster_center
        self.get_cluster_center = model.get_cluster_center
        self.kl_loss_f = model.kl_loss
        torch.set_printoptions(precision=5)

    # ------------------------------------------------------------------------------ initialisation
    def load_pretrain_weight(self):
        """Warm start from the p1 checkpoint; only keys this model has are taken (clustering_trainer.py:431-447)."""
        logger.info('*******Restoring the pretrain model weight based on {}*******'.format(self.args.restore_metric))
        f = os.path.join(self.pretrain_exp_path, 'weight', '{}'.format(self.args.restore_metric), 'model.pth.tar')
        dist.barrier()
        pretrained = torch.load(f, map_location=self.device)['state_dict']
        own = self.model.state_dict()
        # a checkpoint written through nn.DataParallel upstream carries a 'module.' prefix: accept both
        pretrained = {(k[7:] if k.startswith('module.') and k[7:] in own else k): v for k, v in pretrained.items()}
        picked = {k: v for k, v in pretrained.items() if k in own}
        if not picked:
            raise RuntimeError('no parameter of {} matches this model (upstream would silently load nothing)'.format(f))
        self.model.load_state_dict(picked, strict=False)
        logger.info('=> loaded {} tensors from the pretrain checkpoint'.format(len(picked)))

    def _latents(self, cohort, denoise=False):
        """Feature pass that keeps the latents on the device: (hidden (N,256) cuda tensor, metrics)."""
        dl = self._eval_dl(cohort)                                    # sharded over ranks when there are several
        metrics, recs = self.eval_one_epoch(COHORT2SCOPE[cohort], dl, denoise)
        logger.info('{}, {}'.format(COHORT2SCOPE[cohort], format_metric_dict(metrics)))
        hidden = torch.cat([r['hidden'].float() for r in recs], dim=0)
        return self._all_rows(hidden, dl), recs                       # ONE collective: every rank gets all latents, in dataset order

    def generate_pretrain_feat(self, cohort, denoise=False):
        hidden, recs = self._latents(cohort, denoise)
        return self.merge_ob_pred(recs)

    def generate_pred_cluster(self, scope, dl, prev_pred, denoise=False):
        """argmax_j q_ij on the validation cohort and the fraction of labels that changed (clustering_trainer.py:473-484)."""
        metrics, recs = self.eval_one_epoch(scope, dl, denoise=denoise)
        logger.info('{}'.format(format_metric_dict(metrics)))
        cluster_pred = self._all_rows(torch.cat([r['cluster_pred'] for r in recs], dim=0).argmax(dim=1), dl)    # labels of ALL encounters
        if prev_pred is None:
            delta = 1.0
        else:
            prev = torch.as_tensor(np.asarray(prev_pred), device=cluster_pred.device)
            delta = float((cluster_pred != prev).sum()) / prev.numel()
        return delta, cluster_pred.cpu().numpy(), metrics

    # ------------------------------------------------------------------------------ training
    def train(self):
        logger.info('*******Building the model*******')
        args = self.args
        valid_prev = None
        if args.init_cluster_center == 'kmeans':
            self.load_pretrain_weight()
            train_hidden, _ = self._latents('training')
            kmeans = KMeans(n_clusters=args.cluster_number, n_init=20, shard_points=True)   # rows sharded over ranks when there are several
            kmeans.fit(train_hidden)                                            # clustering_trainer.py:75-76
            self.init_cluster_center(torch.tensor(kmeans.cluster_centers_, dtype=torch.float, device=self.device))
            valid_hidden, _ = self._latents('validation')
            valid_prev = kmeans.predict(valid_hidden)                           # :81-82
        elif args.init_cluster_center == 'random':
            self.load_pretrain_weight()
            train_hidden, _ = self._latents('training')
            lo, hi = train_hidden.min(0).values.cpu().numpy(), train_hidden.max(0).values.cpu().numpy()
            centers = np.random.uniform(low=lo, high=hi, size=(args.cluster_number, hi.shape[-1]))
            self.init_cluster_center(torch.tensor(centers, dtype=torch.float, device=self.device))
        elif args.init_cluster_center != 'none':
            raise ValueError(args.init_cluster_center)
        logger.info('*****Cluster initialize {} is done.*****'.format(args.init_cluster_center))

        with timer('Duration of training'):
            for epoch in range(1, args.max_epochs):
                train_metrics = self.train_one_epoch(self.train_dl, denoise=args.denoise)
                logger.info('==> Epoch: {}, Train, {}'.format(epoch, format_metric_dict(train_metrics)))
                delta, valid_pred, valid_metrics = self.generate_pred_cluster('valid', self._eval_dl('validation'), valid_prev)
                logger.info('Epoch: {}: valid delta of cluster label change: {}'.format(epoch, delta))
                valid_metrics['delta'] = delta
                self.aly_pred('valid', valid_metrics)
                if epoch % args.update_interval == 0:
                    if args.stopping_delta is not None and delta < args.stopping_delta:
                        logger.info('Early stopping as label delta "%1.5f" less than "%1.5f".' % (delta, args.stopping_delta))
                        break
                    valid_prev = valid_pred
                self.epoch += 1

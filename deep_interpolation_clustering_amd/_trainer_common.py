"""Epoch loop, evaluation, checkpointing and feature dump shared by ``Trainer`` and ``TrainerCluster``
(pretrain_trainer.py:17-438, clustering_trainer.py:21-545).

What is kept from upstream: constructor signatures, ``train/eval/train_one_epoch/eval_one_epoch/aly_pred/
load_weight/merge_ob_pred/re_norm_data`` names and semantics, checkpoint layout
(``<exp>/weight/<metric>/model.pth.tar`` = {'epoch','state_dict','optimizer'}), ``<exp>/out_feat/<metric>/
<cohort>.npy`` dictionaries, Adam(amsgrad)+clip 15, best-per-metric saving and early stopping.

What is different by design (MI355X): no ``nn.DataParallel`` -- with several GPUs every rank runs this same
code on its shard (``dist``), gradients go through one flat RCCL all-reduce inside ``Stepper``; loss terms
and evaluation outputs stay on the device and are read back once per epoch / per cohort, not once per batch.
"""
import os
from collections import defaultdict
from datetime import datetime

import numpy as np
import torch

from . import dist
from .info import COHORT2SCOPE, MAX_METRICS, METRICS, MIN_MAX_VALUES, MIN_METRICS, SUMMARY_ITEMS
from .step import Stepper, compute_losses
from .utils import (Summary, create_flag_dict, create_weight_dir, early_stop, format_metric_dict, logger,
                    make_summary_writer, pytorch_lr_scheduler, pytorch_optimizer, reduce_lr_on_plateau,
                    save_model_update_flag)

_TENSOR_KEYS = ('ob', 'padding_mask', 'timestamp', 'ae_mask')


class TrainerBase(object):
    restore_attr = 'restore_metric'          # which args field names the checkpoint/feature directory

    def __init__(self, args, model, dl_dict, exp_path, device, **kwargs):
        self.args, self.device = args, device
        self.model = model.to(device)
        autocast = torch.bfloat16 if (getattr(args, 'amp_bf16', False) and device.type == 'cuda') else None
        if autocast is not None and getattr(args, 'f32_products', None) is not None:
            raise ValueError('f32_products selects how the f32 step forms its dense products; it does not combine with amp_bf16')
        self.stepper = Stepper(self.model, lambda m: pytorch_optimizer(m, args.optimizer, args.init_lr, args.weight_decay_rate),
                               args, autocast_dtype=autocast, precision=(getattr(args, 'f32_products', None) if autocast is None else None),
                               use_graphs=False if getattr(args, 'no_hip_graph', False) else (True if getattr(args, 'hip_graph', None) else 'auto'))
        self.optimizer = self.stepper.optimizer
        self.lr_scheduler = pytorch_lr_scheduler(self.optimizer, args.lr_decay_mode, args.lr_decay_step_or_patience,
                                                 args.lr_decay_rate)
        # loss entry points, under upstream's attribute names
        self.rec_loss_f, self.sup_aux_loss_f = model.rec_loss, model.sup_aux_loss
        self.fake_det_loss_f, self.triplet_loss_f = model.fake_det_loss, model.triplet_loss
        self.multi_task_loss_f = model.multi_task_loss

        self.exp_path = exp_path
        self.weight_path = os.path.join(exp_path, 'weight')
        self.weight_path_dict = create_weight_dir(self.weight_path, METRICS)
        self.summary_path = os.path.join(exp_path, 'summary')
        self.out_feat_path = os.path.join(exp_path, 'out_feat', getattr(args, self.restore_attr))
        os.makedirs(self.out_feat_path, exist_ok=True)

        self.dl_dict = dl_dict
        self.train_dl, self.valid_dl, self.test_dl = dl_dict['training'], dl_dict['validation'], dl_dict['testing']
        self.epoch = 1
        self.flag_dict = create_flag_dict(METRICS, MIN_METRICS, MAX_METRICS)
        writer = make_summary_writer(self.summary_path, filename_suffix=datetime.now().strftime('_%m-%d-%y_%H-%M-%S'))
        self.summary = Summary(writer, METRICS, SUMMARY_ITEMS)

    # ------------------------------------------------------------------------------ batches
    def _get_dl(self, cohort):
        return {'training': self.train_dl, 'validation': self.valid_dl, 'testing': self.test_dl}.get(cohort)

    def _eval_dl(self, cohort):
        """Loader for evaluation / feature passes.  When the job is sharded over ranks (one process per GPU) the pass is sharded
        too: an unshuffled loader over the same device-resident cohort gives every rank its contiguous slice of each batch (no
        encounter dropped: a tail batch too small to shard joins the previous one).  The loss operators all-reduce their batch
        statistics, so every rank logs the global-batch metrics; what a caller needs in full (latents for k-means, predicted
        labels, the feature dumps) is assembled by ``_all_rows`` -- one collective per tensor and pass instead of P-fold
        redundant evaluation."""
        dl = self._get_dl(cohort)
        if dist.is_sharded() and hasattr(dl, 'ds'):
            cache = self.__dict__.setdefault('_eval_loaders', {})
            if cohort not in cache:
                from .dataloader import DeviceLoader
                cache[cohort] = DeviceLoader(dl.ds, dl.batch_size, False, dl.device, seed=0, shard=True, keep_every_row=True,
                                             store=getattr(dl, 'store', None), ragged='auto' if getattr(dl, 'store', None) is not None else False)
            return cache[cohort]
        return dl

    def _all_rows(self, local, dl):
        """Rows every rank produced over one pass of the sharded, unshuffled loader ``dl`` -> the full tensor, in dataset order, on
        every rank.  ONE all-gather (shards padded to the longest: every rank knows every rank's row list, the loader's sharding is
        a pure function of the rank) -- half the bytes of round 2's zero-padded sum all-reduce of the full tensor, and no adds."""
        world = getattr(dl, 'world', 1)
        if world <= 1:
            return local
        rows = dl.__dict__.get('_rank_rows')              # (a pure function of the loader: computed once, not per key of the record)
        if rows is None:
            rows = dl._rank_rows = [dl.shard_rows(r) for r in range(world)]
            if sum(r.numel() for r in rows) != len(dl.ds):
                raise RuntimeError(f'sharded pass: the loader covers {sum(r.numel() for r in rows)} of {len(dl.ds)} rows (it must keep every row: '
                                   'an unshuffled loader with keep_every_row=True)')
        if rows[dl.rank].numel() != local.shape[0]:
            raise RuntimeError(f'sharded pass produced {local.shape[0]} rows, the loader promises {rows[dl.rank].numel()}')
        kind = local.dtype
        work = local.to(torch.int32) if kind in (torch.bool, torch.uint8, torch.int8, torch.int16) else local.contiguous()
        longest = max(r.numel() for r in rows)
        mine = work
        if work.shape[0] < longest:
            mine = torch.zeros((longest,) + tuple(work.shape[1:]), dtype=work.dtype, device=work.device)
            mine[:work.shape[0]] = work
        gathered = dist.all_gather_rows(mine)                     # (world, longest, ...)
        full = torch.empty((len(dl.ds),) + tuple(work.shape[1:]), dtype=work.dtype, device=work.device)
        for r in range(world):
            full[rows[r].to(work.device)] = gathered[r, :rows[r].numel()]
        return full.to(kind)

    @staticmethod
    def _densify(s):
        """A ragged-only sample + the padded per-batch tensors upstream's loader yields, rebuilt from the store."""
        rb = s['ragged']
        rows, C = rb.store.dense_rows(rb.idx), rb.store.C
        s = dict(s)
        s.update(ob=rows[:, 0:C], padding_mask=rows[:, C:2 * C], timestamp=rows[:, 2 * C:3 * C], ae_mask=rows[:, 3 * C:4 * C])
        return s

    def _to_device(self, sample):
        return {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in sample.items()}

    def _stack(self, s, denoise):
        """[ob (x ae_mask when denoising) | padding_mask | timestamp | ae_mask] -> (B,4C,T), pretrain_trainer.py:132-143."""
        ob = s['ob'] * s['padding_mask']
        first = ob * s['ae_mask'] if denoise else ob
        return ob, torch.cat([first, s['padding_mask'], s['timestamp'], s['ae_mask']], dim=1)

    def add_gaussian_noise(self, tensor, padding_mask, gaussian_config_dict):
        mean, std = gaussian_config_dict.get('mean', 0.0), gaussian_config_dict.get('std', .1)
        return (tensor + torch.randn(tensor.size(), device=self.device) * std + mean) * padding_mask

    def _prepare(self, batch_sample, fake_batch_sample, denoise, train):
        """Everything the model and the loss switch need for one batch, on the device."""
        args = self.args
        s = self._to_device(batch_sample)
        rb = s.get('ragged')
        if rb is not None and 'ob' not in s and not train:
            s = self._densify(s)                       # evaluation records dump the padded tensors (clustering_trainer.py:409-416)
        if rb is not None and 'ob' not in s:
            # ragged-only sample (DeviceLoader on a ragged store, training pass of a plain objective): the kernels read the store in
            # place through the batch handle; no padded tensor is built.  `ob=None`: the observations are the batch itself
            out = dict(x=rb.with_denoise(denoise), ob=None, padding_mask=None, lengths=rb.lengths, sample=s,
                       fake_x=None, fake_perm_idx=None, positive_x=None, fake_det_label=None, aux_label_dict={}, future_vital_mask=None)
        else:
            ob, x = self._stack(s, denoise)
            if rb is not None:
                x = rb.with_denoise(denoise)           # same samples, read from the store instead of from the padded copy
                if train:
                    ob = None                          # (the training step takes its observations from the store too)
            out = dict(x=x, ob=ob, padding_mask=s['padding_mask'], lengths=s.get('lengths'), sample=s,
                       fake_x=None, fake_perm_idx=None, positive_x=None, fake_det_label=None,
                       aux_label_dict={}, future_vital_mask=None)
        if args.fake_detection:
            f = self._to_device(fake_batch_sample)
            assert np.array_equal(np.asarray(f['encounter_id']), np.asarray(s['encounter_id'])), 'Encounter_id dose not match.'
            mask = f['padding_mask'] if train else s['padding_mask']           # pretrain_trainer.py:152 vs :272
            fob = f['ob'] * mask
            first = fob * f['ae_mask'] if denoise else fob
            out['fake_x'] = torch.cat([first, f['padding_mask'], f['timestamp'], f['ae_mask']], dim=1)
            B = out['x'].size(0)
            label = torch.cat([torch.ones(B, device=self.device), torch.zeros(B, device=self.device)])
            perm = torch.randperm(2 * B, device=self.device)                   # shuffle the real / fake rows
            out['fake_perm_idx'] = perm
            out['fake_det_label'] = label[perm].to(torch.int64)
        if getattr(args, 'triple_margin', 0.) != 0. and args.fake_detection:
            assert args.scale == 20, 'The noise gaussian config should be adjusted.'
            ob = s['ob'] * s['padding_mask']
            nob = self.add_gaussian_noise(ob, s['padding_mask'], {'std': args.triple_pos_std})
            nts = self.add_gaussian_noise(s['timestamp'], s['padding_mask'], {'std': 0.01})
            out['positive_x'] = torch.cat([nob, s['padding_mask'], nts, s['ae_mask']], dim=1)
        if args.aux_tasks:
            if 'future_vital' in args.aux_tasks:
                out['future_vital_mask'] = s['future_vital_mask']
            out['aux_label_dict'] = {t: s[t] for t in args.aux_tasks.keys()}
        return out

    @staticmethod
    def _model_kwargs(b):
        return dict(fake_x=b['fake_x'], fake_perm_idx=b['fake_perm_idx'], positive_x=b['positive_x'],
                    aux_label_dict=b['aux_label_dict'], future_vital_mask=b['future_vital_mask'],
                    fake_det_label=b['fake_det_label'])

    # ------------------------------------------------------------------------------ epochs
    def train_one_epoch(self, dl, denoise=True):
        self.model.train()
        sums, n_batches = defaultdict(lambda: torch.zeros((), device=self.device)), 0
        total = len(dl)
        for i_batch, (batch_sample, fake_batch_sample) in enumerate(dl, start=1):
            b = self._prepare(batch_sample, fake_batch_sample, denoise, train=True)
            losses, _, _ = self.stepper.step(b['x'], b['ob'], b['padding_mask'], b['lengths'], global_rows=b['sample'].get('global_rows'),
                                              **self._model_kwargs(b))
            for k, v in losses.items():
                sums[k] = sums[k] + v.detach()
            n_batches += 1
            if i_batch % int(self.args.log_train_freq) == 1:
                now = {k: float(v.detach()) for k, v in losses.items()}  # the only host sync, at the logging cadence
                logger.info('{}-[{}/{} ({:.0f}%)]: train-{}'.format(self.epoch, i_batch, total, 100. * i_batch / total, now))
                self.summary.add_summary(self.epoch * total + i_batch, scope='train_batch', **now)
        out = {'scope': 'train'}
        out.update({k: float(v) / max(n_batches, 1) for k, v in sums.items()})   # epoch mean of the batch losses
        return out

    def eval_one_epoch(self, scope, dl, denoise=False):
        self.model.eval()
        if getattr(self.args, 'evaluate_interpolation', False):
            denoise = True
        sums, n_batches = defaultdict(lambda: torch.zeros((), device=self.device)), 0
        ob_pred_lst = []
        total = len(dl)
        with torch.no_grad():
            for i_batch, (batch_sample, fake_batch_sample) in enumerate(dl, start=1):
                b = self._prepare(batch_sample, fake_batch_sample, denoise, train=False)
                losses, hidden, rec_ob, aux_pred = self.stepper.forward_loss(
                    b['x'], b['ob'], b['padding_mask'], b['lengths'], **self._model_kwargs(b))
                for k, v in losses.items():
                    sums[k] = sums[k] + v
                n_batches += 1
                rec = {k: v for k, v in b['sample'].items() if k not in ('lengths', 'ragged', 'global_rows')}      # inputs + labels, as upstream dumps them
                if getattr(self.args, 'cpu_padded_ob', False) and torch.is_tensor(rec.get('ob')) and torch.is_tensor(rec.get('padding_mask')):
                    # upstream masks the observations IN PLACE before the forward (`ob *= padding_mask`, clustering_trainer.py:299-303): on a CPU run
                    # `.to(device)` is the identity, so the loader's tensor -- the one the dump keeps -- is the masked one (padded slots 0 ->
                    # mid-range after re_norm_data); on a GPU run the dump keeps the loader's -scale/2.  This switch writes the CPU run's bytes.
                    rec['ob'] = rec['ob'] * rec['padding_mask']
                rec.update(aux_pred)
                rec['hidden'], rec['rec_ob'] = hidden, rec_ob
                ob_pred_lst.append(rec)
                if i_batch % int(self.args.log_valid_freq) == 1:
                    now = {k: float(v) for k, v in losses.items()}
                    logger.info('{}-[{}/{} ({:.0f}%)]: {}-{}'.format(self.epoch, i_batch, total, 100. * i_batch / total, scope, now))
                    if self.args.mode == 'train':
                        self.summary.add_summary(self.epoch * total + i_batch, scope='{}_batch'.format(scope), **now)
        metrics = {'scope': scope}
        metrics.update({k: float(v) / max(n_batches, 1) for k, v in sums.items()})
        self._last_eval_dl = dl
        return metrics, ob_pred_lst

    def merge_ob_pred(self, ob_pred_lst, dl=None):
        """Concatenate the per-batch records; tensors leave the device here, once per key.  After a sharded evaluation pass
        (``dl``: its loader, default the last one used) every key is first assembled over ranks, in dataset order."""
        dl = dl if dl is not None else getattr(self, '_last_eval_dl', None)
        sharded = getattr(dl, 'world', 1) > 1
        merged = {}
        n_local = dl.shard_rows().numel() if sharded else None
        for k in (ob_pred_lst[0].keys() if ob_pred_lst else ()):
            vals = [d[k] for d in ob_pred_lst]
            if torch.is_tensor(vals[0]):
                t = torch.cat([v.float() if v.dtype.is_floating_point else v for v in vals], dim=0)     # (ids keep their integer type)
                if sharded and t.shape[0] != n_local:
                    # not one row per encounter: 'fake_det' holds 2 x batch rows in each batch's own random order (real and corrupted
                    # samples shuffled by fake_perm_idx, clustering_trainer.py:330-332) -- there is no dataset order to assemble it in
                    if k not in self.__dict__.setdefault('_skipped_dump_keys', set()):
                        self._skipped_dump_keys.add(k)
                        logger.warning('sharded feature pass: key {!r} ({} rows for {} encounters) is left out of the merged record'.format(k, t.shape[0], n_local))
                    continue
                merged[k] = (self._all_rows(t, dl) if sharded else t).cpu().numpy()
            elif sharded and k == 'encounter_id':
                merged[k] = np.asarray(dl.ids)                   # the unshuffled pass covers the cohort in dataset order
            elif sharded:
                local = torch.as_tensor(np.concatenate([np.asarray(v) for v in vals], axis=0), device=self.device)
                merged[k] = self._all_rows(local, dl).cpu().numpy()
            else:
                merged[k] = np.concatenate([np.asarray(v) for v in vals], axis=0)
        return merged

    def re_norm_data(self, ob_pred_dict):
        """Back from +-scale/2 to physiologic units (pretrain_trainer.py:416-429)."""
        if self.args.norm_method != 'minmax':
            raise NotImplementedError
        for k in ('ob', 'rec_ob'):
            data = ob_pred_dict[k]
            unit = (data + self.args.scale / 2) / self.args.scale
            for i, (lo, hi) in enumerate(list(MIN_MAX_VALUES.values())[:data.shape[1]]):
                data[:, i, :] = unit[:, i, :] * (hi - lo) + lo
            ob_pred_dict[k] = data
        return ob_pred_dict

    # ------------------------------------------------------------------------------ bookkeeping
    def aly_pred(self, scope, metric_dict):
        if scope == 'valid':
            if self.args.lr_decay_mode in ('step', 'warmup'):
                self.lr_scheduler.step()
            elif self.args.lr_decay_mode == 'plateau':
                reduce_lr_on_plateau(self.lr_scheduler, metric_dict, 'loss')
            for group in self.optimizer.param_groups:
                group['lr'] = max(group['lr'], self.args.min_lr)
                metric_dict.update({'lr': group['lr']})
            if dist.rank() == 0:
                save_model_update_flag(self.model, self.optimizer, self.weight_path_dict, self.flag_dict, metric_dict,
                                       MIN_METRICS, MAX_METRICS, self.epoch)
            else:                                  # keep every rank's early-stop bookkeeping identical
                for k, v in metric_dict.items():
                    if k in MIN_METRICS and v <= self.flag_dict[k]:
                        self.flag_dict[k], self.flag_dict[k + '_epoch'] = v, self.epoch
        self.summary.add_summary(self.epoch, **metric_dict)
        logger.info(metric_dict)
        return {'early_stop': early_stop(self.flag_dict, self.epoch, self.args.early_stopping, scope)}

    def _restore_file(self):
        return os.path.join(self.exp_path, 'weight', getattr(self.args, self.restore_attr), 'model.pth.tar')

    def load_weight(self):
        metric = getattr(self.args, self.restore_attr)
        dist.barrier()                              # rank 0 has finished writing the checkpoint every rank is about to read
        logger.info('*******Restoring the model weight based on {}*******'.format(metric))
        checkpoint = torch.load(self._restore_file(), map_location=self.device)     # upstream hard-codes cuda:0
        self.epoch = checkpoint['epoch']
        self.model.load_state_dict(checkpoint['state_dict'])
        self.optimizer.load_state_dict(checkpoint['optimizer'])
        logger.info('=> loaded checkpoint from model.path.tar')

    def eval(self, cohort, generate_feat=False, viz_feat=False, denoise=False):
        logger.info('*******Evaluating the model*******')
        self.load_weight()
        scope = COHORT2SCOPE[cohort]
        metrics, ob_pred_lst = self.eval_one_epoch(scope, self._eval_dl(cohort), denoise)
        logger.info('{}, {}'.format(scope, format_metric_dict(metrics)))
        ob_pred_dict = self.re_norm_data(self.merge_ob_pred(ob_pred_lst))
        if generate_feat and dist.rank() == 0:
            folder = os.path.join(self.exp_path, 'out_feat', getattr(self.args, self.restore_attr))
            os.makedirs(folder, exist_ok=True)
            suffix = '_interp_eval' if getattr(self.args, 'evaluate_interpolation', False) else ''
            np_f = os.path.join(folder, '{}{}.npy'.format(cohort, suffix))
            if os.path.exists(np_f):
                logger.info('No save, the npy file exists. {}'.format(np_f))
            else:
                np.save(np_f, ob_pred_dict)
                logger.info('The npy is saved to {}'.format(np_f))
        if viz_feat:
            self.summary.summary_writer.add_embedding(ob_pred_dict['hidden'], global_step=self.epoch, tag=cohort)
        return ob_pred_dict

"""Host glue with the behaviour of the reference's ``utils`` (utils.py:21-224): logger, seeding,
optimizer / scheduler factories, checkpoint helpers, early stopping, ``Summary`` and
``TimeDistributed``.  Unlike upstream, nothing here imports packages that are optional on the GPU
box (tensorflow, tensorboardX, warmup_scheduler): they are used when importable and skipped otherwise.
"""
import json
import logging
import os
import random
import time
from collections import OrderedDict
from contextlib import contextmanager

import numpy as np
import torch
import torch.optim as optim


def get_logger(log_level):
    log = logging.getLogger('deep_interpolation_clustering_amd')
    if not getattr(log, 'handler_set', None):
        handler = logging.StreamHandler()
        handler.setFormatter(logging.Formatter('%(asctime)s %(levelname)s - %(funcName)s(%(lineno)d): %(message)s',
                                               '%H:%M:%S'))
        handler.setLevel(log_level.upper())
        log.setLevel(log_level.upper())
        log.addHandler(handler)
        log.handler_set = True
    return log


logger = get_logger('INFO')


def set_seed(seed):
    """utils.py:37-42 seeds NumPy and ``random`` (and TensorFlow, absent here).  NumPy's global state
    is what drives k-means++ (sklearn random_state=None), so this is what makes p3's init reproducible.
    torch is left unseeded, as upstream."""
    logger.info('The global seed: {}'.format(seed))
    np.random.seed(int(seed))
    random.seed(seed)
    try:                                         # pragma: no cover - tensorflow is optional
        import tensorflow as tf
        tf.random.set_seed(seed)
    except Exception:
        pass


def load_config(dest_dir, f_name, cur_arg, *args):
    if not os.path.exists(dest_dir):
        raise Exception('The config folder does not exist. {}'.format(dest_dir))
    with open(os.path.join(dest_dir, '{}.json'.format(f_name))) as f:
        previous = json.load(f)[0]
    keep = args if args else ['mode', 'restore', 'restore_metric', 'log_level']
    current = vars(cur_arg)
    kept = {k: current[k] for k in keep}
    cur_arg.__dict__.update(previous)
    cur_arg.__dict__.update(kept)
    return cur_arg


def print_dict_byline(target_dict):
    for k, v in target_dict.items():
        logger.info('{}:'.format(k).ljust(15) + '{}'.format(v))
    logger.info('===============================')


def format_metric_dict(metric_dict, decimals=4):
    for k, v in metric_dict.items():
        if isinstance(v, float) and k != 'lr':
            metric_dict[k] = np.round(v, decimals=decimals)
    return metric_dict


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def pytorch_optimizer(model, optimizer, init_lr, weight_decay=0):
    """utils.py:76-83.  Adam is amsgrad with L2 weight decay; on a GPU it is flat_adam.FlatAdam (same update rule,
    same state_dict layout, one kernel)."""
    params = list(model.parameters())
    if optimizer == 'SGD':
        return optim.SGD(params, lr=init_lr, momentum=0.9, weight_decay=weight_decay, nesterov=True)
    if optimizer == 'RMSprop':
        return optim.RMSprop(params, lr=init_lr, momentum=0.9, weight_decay=weight_decay)
    if optimizer == 'Adam':
        if bool(params) and all(p.is_cuda for p in params):
            # one HIP kernel over the flat parameter bucket (clip scale + amsgrad update); state_dict-compatible with optim.Adam
            from .flat_adam import FlatAdam
            return FlatAdam(params, lr=init_lr, weight_decay=weight_decay)
        return optim.Adam(params, lr=init_lr, weight_decay=weight_decay, amsgrad=True)
    raise ValueError('unknown optimizer {}'.format(optimizer))


def pytorch_lr_scheduler(optimizer, lr_decay_mode, lr_decay_step_or_patience, lr_decay_rate):
    if lr_decay_mode == 'step':
        return optim.lr_scheduler.StepLR(optimizer, step_size=lr_decay_step_or_patience, gamma=lr_decay_rate)
    if lr_decay_mode == 'plateau':
        return optim.lr_scheduler.ReduceLROnPlateau(optimizer, 'min', factor=lr_decay_rate,
                                                    patience=lr_decay_step_or_patience)
    if lr_decay_mode == 'warmup':
        try:
            from warmup_scheduler import GradualWarmupScheduler
        except ImportError as e:                 # utils.py:18 imports it unconditionally upstream
            raise ImportError("lr_decay_mode='warmup' needs the warmup_scheduler package") from e
        after = optim.lr_scheduler.StepLR(optimizer, step_size=lr_decay_step_or_patience, gamma=lr_decay_rate)
        return GradualWarmupScheduler(optimizer, multiplier=8, total_epoch=10, after_scheduler=after)
    raise ValueError('No defined scheduler for {}.'.format(lr_decay_mode))


@contextmanager
def timer(message):
    tick = time.time()
    yield
    diff = time.time() - tick
    if diff >= 3600:
        duration = '{:.2f}h'.format(diff / 3600)
    elif diff >= 60:
        duration = '{:.2f}m'.format(round(diff / 60))
    else:
        duration = '{:.2f}s'.format(diff)
    logger.info('{}: {}'.format(message, duration))


def reduce_lr_on_plateau(lr_scheduler, metric_dict, monitor):
    lr_scheduler.step(metric_dict[monitor])      # call after validation


def save_checkpoint(epoch, model, optimizer, filename):
    """{'epoch','state_dict','optimizer'} -> model.pth.tar (utils.py:141-145)."""
    torch.save({'epoch': epoch, 'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict()}, filename)


def save_model_update_flag(model, optimizer, weight_dict, flag_dict, metric_dict, min_metrics, max_metrics, epoch):
    """Checkpoint once per metric that improved (ties count), utils.py:126-138."""
    for k, v in metric_dict.items():
        better = (k in min_metrics and v <= flag_dict[k]) or (k in max_metrics and v >= flag_dict[k])
        if better:
            flag_dict[k] = v
            flag_dict[k + '_epoch'] = epoch
            save_checkpoint(epoch, model, optimizer, os.path.join(weight_dict[k], 'model.pth.tar'))
            logger.info('Saving for {}'.format(k))


def early_stop(flag_dict, epoch, patience, scope):
    latest = max(v for k, v in flag_dict.items() if k.endswith('epoch'))
    if epoch - latest + 1 > patience:
        logger.info('==={} reaches early stop with best model==='.format(scope))
        logger.info('{}'.format(flag_dict))
        return True
    return False


def create_flag_dict(metrics, min_metrics, max_metrics):
    flags = OrderedDict()
    for m in metrics:
        if m in max_metrics:
            flags[m] = 0
        elif m in min_metrics:
            flags[m] = float('inf')
        flags[m + '_epoch'] = 0
    return flags


class NullSummaryWriter:
    """Stand-in when tensorboardX is not installed: same calls, no output."""

    def __init__(self, *args, **kwargs):
        pass

    def add_scalar(self, *args, **kwargs):
        pass

    def add_embedding(self, *args, **kwargs):
        pass


def make_summary_writer(path, **kwargs):
    try:
        from tensorboardX import SummaryWriter
        return SummaryWriter(path, **kwargs)
    except Exception:
        return NullSummaryWriter()


class Summary(object):
    def __init__(self, summary_writer, metric_items, summary_items):
        self.summary_writer = summary_writer
        self.metric_items = metric_items
        self.summary_items = summary_items

    def add_summary(self, step, **kwargs):
        scope = kwargs['scope']
        for k, v in kwargs.items():
            if k in self.metric_items or k in self.summary_items:
                self.summary_writer.add_scalar(tag=scope + '_' + k, scalar_value=float(v), global_step=step)


def create_dir(root_dir, x):
    target = os.path.join(root_dir, x)
    os.makedirs(target, exist_ok=True)
    return target


def create_weight_dir(root_weight_dir, metrics_lst):
    return {m: create_dir(root_weight_dir, m + '/') for m in metrics_lst}


class TimeDistributed(torch.nn.Module):
    """Fold (samples, steps, F) into (samples*steps, F) around ``module`` (utils.py:202-224)."""

    def __init__(self, module, batch_first=True):
        super().__init__()
        self.module = module
        self.batch_first = batch_first

    def forward(self, x):
        if x.dim() <= 2:
            return self.module(x)
        y = self.module(x.reshape(-1, x.size(-1)))
        if self.batch_first:
            return y.reshape(x.size(0), -1, y.size(-1))
        return y.reshape(-1, x.size(1), y.size(-1))

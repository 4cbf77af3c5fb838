"""Host-side support for the drivers and trainers: logging, seeding, optimiser / scheduler construction, best-checkpoint
bookkeeping with early stopping, scalar summaries, and the per-row module wrapper.

``utils.py`` re-exports these under the names the reference's ``utils`` module uses (utils.py:21-224), which is what the
trainers and drivers import; the implementations are organised around three small classes instead of free functions.
Optional packages of the reference environment (tensorflow, tensorboardX, warmup_scheduler) are used when importable and
skipped otherwise.
"""
from __future__ import annotations

import json
import logging
import os
import random
import time
from collections import OrderedDict
from contextlib import contextmanager

import numpy as np
import torch


# --------------------------------------------------------------------------------------------- logging / small helpers
_LOG_NAME = 'deep_interpolation_clustering_amd'
_LOG_FORMAT = '%(asctime)s %(levelname)s - %(funcName)s(%(lineno)d): %(message)s'


def package_logger(level='INFO'):
    """The package logger, configured once (a stream handler at ``level``)."""
    log = logging.getLogger(_LOG_NAME)
    if not log.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter(_LOG_FORMAT, '%H:%M:%S'))
        log.addHandler(h)
    lvl = str(level).upper()
    log.setLevel(lvl)
    for h in log.handlers:
        h.setLevel(lvl)
    return log


logger = package_logger()


def seed_host_rngs(seed):
    """Seeds ``random`` and NumPy's global state (upstream also seeds TensorFlow, which only it imports).  NumPy's global
    state is what k-means++ draws from (random_state=None), so this makes the p3 initialisation reproducible; torch is
    left unseeded, as upstream leaves it."""
    logger.info('The global seed: {}'.format(seed))
    random.seed(seed)
    np.random.seed(int(seed))
    try:                                             # pragma: no cover
        import tensorflow
        tensorflow.random.set_seed(seed)
    except Exception:
        pass


def merge_saved_config(folder, stem, args, *keep):
    """Overlay the argument namespace with ``<folder>/<stem>.json`` (a list holding one dict), except for the run-control
    fields in ``keep`` (default: mode, restore, restore_metric, log_level)."""
    if not os.path.isdir(folder):
        raise Exception('The config folder does not exist. {}'.format(folder))
    with open(os.path.join(folder, stem + '.json')) as fh:
        stored = json.load(fh)[0]
    protected = {k: getattr(args, k) for k in (keep or ('mode', 'restore', 'restore_metric', 'log_level'))}
    vars(args).update(stored)
    vars(args).update(protected)
    return args


def log_mapping(mapping):
    for key, value in mapping.items():
        logger.info((str(key) + ':').ljust(15) + str(value))
    logger.info('=' * 31)


def round_floats(metrics, decimals=4):
    """In-place rounding of the float entries (the learning rate keeps its digits)."""
    for key in metrics:
        if key != 'lr' and isinstance(metrics[key], float):
            metrics[key] = np.round(metrics[key], decimals=decimals)
    return metrics


def trainable_parameter_count(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


@contextmanager
def stopwatch(label):
    t0 = time.time()
    yield
    dt = time.time() - t0
    for unit, span in (('h', 3600.0), ('m', 60.0)):
        if dt >= span:
            logger.info('{}: {:.2f}{}'.format(label, dt / span, unit))
            return
    logger.info('{}: {:.2f}s'.format(label, dt))


def ensure_dir(root, name):
    path = os.path.join(root, name)
    os.makedirs(path, exist_ok=True)
    return path


# --------------------------------------------------------------------------------------------- optimiser / scheduler
def build_optimizer(model, kind, lr, weight_decay=0):
    """'Adam' is amsgrad with L2 weight decay (utils.py:76-83); on a GPU it is ``flat_adam.FlatAdam``: the same update rule and
    ``state_dict`` layout as ``torch.optim.Adam``, executed as one kernel over the flat parameter bucket."""
    params = list(model.parameters())
    if kind == 'Adam':
        if params and all(p.is_cuda for p in params):
            from .flat_adam import FlatAdam
            return FlatAdam(params, lr=lr, weight_decay=weight_decay)
        return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, amsgrad=True)
    momentum_family = {'SGD': dict(nesterov=True), 'RMSprop': {}}
    if kind in momentum_family:
        return getattr(torch.optim, kind)(params, lr=lr, momentum=0.9, weight_decay=weight_decay, **momentum_family[kind])
    raise ValueError('unknown optimizer {}'.format(kind))


def build_scheduler(optimizer, mode, step_or_patience, rate):
    sched = torch.optim.lr_scheduler
    if mode == 'plateau':
        return sched.ReduceLROnPlateau(optimizer, 'min', factor=rate, patience=step_or_patience)
    if mode in ('step', 'warmup'):
        stepped = sched.StepLR(optimizer, step_size=step_or_patience, gamma=rate)
        if mode == 'step':
            return stepped
        try:
            from warmup_scheduler import GradualWarmupScheduler
        except ImportError as err:
            raise ImportError("lr_decay_mode='warmup' needs the warmup_scheduler package") from err
        return GradualWarmupScheduler(optimizer, multiplier=8, total_epoch=10, after_scheduler=stepped)
    raise ValueError('No defined scheduler for {}.'.format(mode))


def plateau_step(scheduler, metrics, monitor):
    scheduler.step(metrics[monitor])


# --------------------------------------------------------------------------------------------- best-checkpoint bookkeeping
class BestBook:
    """Best value and epoch per metric, as one ordered mapping ``{metric: best, metric_epoch: epoch}`` -- the 'flag dict' the
    trainers carry around (kept as a plain mapping because they index it directly)."""

    @staticmethod
    def fresh(metrics, minimised, maximised):
        book = OrderedDict()
        for name in metrics:
            if name in maximised:
                book[name] = 0
            elif name in minimised:
                book[name] = float('inf')
            book[name + '_epoch'] = 0
        return book

    @staticmethod
    def improved(book, name, value, minimised, maximised):
        return (name in minimised and value <= book[name]) or (name in maximised and value >= book[name])      # ties count

    @staticmethod
    def stale(book, epoch, patience):
        newest = max(v for k, v in book.items() if k.endswith('epoch'))
        return epoch - newest + 1 > patience


def write_checkpoint(epoch, model, optimizer, path):
    """model.pth.tar = {'epoch', 'state_dict', 'optimizer'} (the reference's checkpoint layout, utils.py:141-145)."""
    torch.save(dict(epoch=epoch, state_dict=model.state_dict(), optimizer=optimizer.state_dict()), path)


def checkpoint_improved(model, optimizer, dirs, book, metrics, minimised, maximised, epoch):
    for name, value in metrics.items():
        if BestBook.improved(book, name, value, minimised, maximised):
            book[name], book[name + '_epoch'] = value, epoch
            write_checkpoint(epoch, model, optimizer, os.path.join(dirs[name], 'model.pth.tar'))
            logger.info('Saving for {}'.format(name))


def patience_exhausted(book, epoch, patience, scope):
    if not BestBook.stale(book, epoch, patience):
        return False
    logger.info('==={} reaches early stop with best model==='.format(scope))
    logger.info(str(book))
    return True


# --------------------------------------------------------------------------------------------- summaries
class NullSummaryWriter:
    """Used when tensorboardX is not installed: accepts the same calls, writes nothing."""

    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def add_embedding(self, *a, **k):
        pass


def open_summary_writer(path, **kwargs):
    try:
        from tensorboardX import SummaryWriter
    except Exception:
        return NullSummaryWriter()
    return SummaryWriter(path, **kwargs)


class ScalarSummary:
    """Forwards the metric / summary scalars of a step to the writer under '<scope>_<name>' tags."""

    def __init__(self, summary_writer, metric_items, summary_items):
        self.summary_writer = summary_writer
        self.metric_items, self.summary_items = metric_items, summary_items

    def add_summary(self, step, **scalars):
        prefix = scalars['scope'] + '_'
        wanted = set(self.metric_items) | set(self.summary_items)
        for name in scalars:
            if name in wanted:
                self.summary_writer.add_scalar(tag=prefix + name, scalar_value=float(scalars[name]), global_step=step)


# --------------------------------------------------------------------------------------------- per-row module wrapper
class RowsAsBatch(torch.nn.Module):
    """Applies ``module`` to the last dimension of an (n, steps, features) tensor by folding the first two dimensions into one
    batch of rows (the reference's TimeDistributed, utils.py:202-224; the attribute must stay ``module``: it is part of the
    ``state_dict`` keys ``rbf.compress_fc.module.model.*``)."""

    def __init__(self, module, batch_first=True):
        super().__init__()
        self.module, self.batch_first = module, batch_first

    def forward(self, x):
        if x.dim() <= 2:
            return self.module(x)
        rows = self.module(x.reshape(-1, x.size(-1)))
        lead = (x.size(0), -1) if self.batch_first else (-1, x.size(1))
        return rows.reshape(*lead, rows.size(-1))

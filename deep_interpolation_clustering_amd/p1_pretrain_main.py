"""p1: pretrain the interpolation auto-encoder, then dump latents per cohort (p1_pretrain_main.py:103-146).

    cd <run dir>;  python -m deep_interpolation_clustering_amd.p1_pretrain_main --mode train [...]

Reads ../Data/model_data/split_processed/<cohort>.pickle, writes Results/Pretrain/{weight,summary,out_feat}.
With several GPUs launch it under torchrun: one process per GPU, encounters sharded per batch.
"""
import os
import random

import torch

from . import _cli, dist
from .dataloader import DataSet, DeviceLoader
from .info import COHORTS, METRICS
from .pretrain_interp import Net
from .pretrain_trainer import Trainer
from .utils import count_parameters, logger, set_seed


def get_arguments(argv=None):
    parser = _cli.build_parser('Implementation of interpolation pretraining for irregular time series', _cli.P1_ONLY)
    return _cli.finalize(parser.parse_args(argv))


def build_loaders(args, device):
    loaders, n_train = {}, 1
    for cohort in COHORTS:
        ds = DataSet(args, cohort)
        shuffle = cohort == 'training'
        if shuffle:
            n_train = len(ds)
        if device.type == 'cuda' and not args.host_loader:
            loaders[cohort] = DeviceLoader(ds, args.batch_size, shuffle, device, seed=int(args.seed), shard=shuffle)
        else:
            from torch.utils.data import DataLoader
            loaders[cohort] = DataLoader(ds, batch_size=args.batch_size, num_workers=args.num_workers, shuffle=shuffle)
    return loaders, n_train


def main(args):
    if args.seed is None:
        args.seed = random.randint(1, 10000)
    set_seed(args.seed)
    rank, world, local = dist.init_from_env()
    exp_path = os.path.join(os.getcwd(), 'Results', 'Pretrain')
    os.makedirs(exp_path, exist_ok=True)
    logger.info('Root directory for saving and loading experiments: {}'.format(exp_path))
    device = torch.device('cuda', local) if args.num_gpus > 0 else torch.device('cpu')
    model = Net(args, device=device)
    dl_dict, n_train = build_loaders(args, device)
    n_param = count_parameters(model)
    logger.info('The ratio is {} ({} / {})'.format(n_param / n_train, n_param, n_train))
    trainer = Trainer(args, model, dl_dict, exp_path, device)
    if args.mode == 'train':
        trainer.train()
        trainer.args.mode = 'eval'
    for metric in METRICS[:2]:                      # 'loss', 'ae_mse'
        trainer.args.restore_metric = metric
        for cohort in COHORTS:
            trainer.eval(cohort, generate_feat=True, viz_feat=True, denoise=False)


if __name__ == '__main__':
    main(get_arguments())

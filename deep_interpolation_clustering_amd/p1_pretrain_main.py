"""p1: pretrain the interpolation auto-encoder, then dump latents per cohort (p1_pretrain_main.py:103-146).

    cd <run dir>;  python -m deep_interpolation_clustering_amd.p1_pretrain_main --mode train [...]

Reads ../Data/model_data/split_processed/<cohort>.pickle, writes Results/Pretrain/{weight,summary,out_feat}.
Several GPUs: ``--num_gpus N`` as upstream (p1_pretrain_main.py:27,118) -- the driver then starts its own N ranks, one process per GPU, encounters
sharded per batch -- or under ``python -m torch.distributed.run --nproc-per-node N -m ...``.
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
    # (LOCAL_RANK modulo the visible cards: under DIC_DIST_BACKEND=gloo several ranks may share one GPU -- the rehearsal of the N > 1 path on a 1-GPU box)
    device = torch.device('cuda', local % max(1, torch.cuda.device_count())) if args.num_gpus > 0 else torch.device('cpu')
    if world > 1:
        logger.info('rank {} of {} on {} ({}): encounters of every batch are sharded over the ranks'.format(rank, world, device, dist.td.get_backend()))
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


def cli(argv=None):
    """``python -m ...p1_pretrain_main --num_gpus N``: N > 1 without a launcher around it starts the N ranks itself (before anything here touches
    the GPU) and relays their exit code; a rank -- or a single-GPU run -- goes straight to ``main``."""
    args = get_arguments(argv)
    code = dist.ranks_for_num_gpus(args.num_gpus, __spec__.name if __spec__ else 'deep_interpolation_clustering_amd.p1_pretrain_main', argv)
    if code is not None:
        raise SystemExit(code)
    main(args)


if __name__ == '__main__':
    cli()

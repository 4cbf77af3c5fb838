"""p3: joint interpolation + DEC training from the p1 checkpoint (p3_clustering_main.py:107-143).

    cd <run dir>;  python -m deep_interpolation_clustering_amd.p3_clustering_main [...]

Centroids start from k-means (n_init=20) on the pretrained latents -- on the GPU, see ``clustering_trainer``.
"""
import os
import random

import torch

from . import _cli, dist
from .clustering_interp import Net
from .clustering_trainer import TrainerCluster
from .info import COHORTS, METRICS
from .p1_pretrain_main import build_loaders
from .utils import count_parameters, logger, set_seed


def get_arguments(argv=None):
    parser = _cli.build_parser('Implementation of deep clustering for time series data', _cli.P3_ONLY)
    return _cli.finalize(parser.parse_args(argv))


def main(args):
    if args.seed is None:
        args.seed = random.randint(1, 10000)
    set_seed(args.seed)
    rank, world, local = dist.init_from_env()
    pretrain_exp_path = os.path.join(os.getcwd(), 'Results', 'Pretrain')
    exp_path = os.path.join(os.getcwd(), 'Results', 'Clustering')
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
    trainer = TrainerCluster(args, model, dl_dict, exp_path, pretrain_exp_path, device)
    if args.mode == 'train':
        trainer.train()
        trainer.args.mode = 'eval'
    for metric in METRICS:                          # 'loss', 'ae_mse', 'delta'
        trainer.args.dc_restore_metric = metric
        for cohort in COHORTS:
            trainer.eval(cohort, generate_feat=True, viz_feat=True, denoise=False)


def cli(argv=None):
    """``python -m ...p3_clustering_main --num_gpus N``: N > 1 without a launcher around it starts the N ranks itself (before anything here touches
    the GPU) and relays their exit code; a rank -- or a single-GPU run -- goes straight to ``main``."""
    args = get_arguments(argv)
    code = dist.ranks_for_num_gpus(args.num_gpus, __spec__.name if __spec__ else 'deep_interpolation_clustering_amd.p3_clustering_main', argv)
    if code is not None:
        raise SystemExit(code)
    main(args)


if __name__ == '__main__':
    cli()

"""Command-line surface shared by the p1 / p3 drivers.

Option names, types and defaults are upstream's (p1_pretrain_main.py:18-101, p3_clustering_main.py:17-105),
declared as data.  Upstream quirks are kept where a caller could depend on them: ``--seed`` parses as float,
dict-valued defaults (``--aux_tasks`` ...) are not settable from the shell, ``--denoise`` /
``--fake_detection`` are un-typed (any string given is truthy).  Additions are listed in EXTRA.
"""
import argparse

LOSSES = ['ae_mse', 'ae_mse_sup', 'ae_mse_fake_detect', 'ae_mse_fake_detect_triplet', 'ae_mse_sup_fake_detect',
          'ae_mse_kl', 'ae_mse_fake_detect_kl', 'ae_mse_sup_kl', 'ae_mse_sup_fake_detect_kl']

# (group, flags, kwargs)
COMMON = [
    ('General options', ('-L', '--log-level'), dict(default='DEBUG', choices=['CRITICAL', 'ERROR', 'WARNING', 'INFO', 'DEBUG', 'NOTSET'], help='Logging levels.')),
    ('General options', ('-s', '--seed'), dict(type=float, default=7529)),
    ('General options', ('--num_gpus',), dict(type=int, default=1)),
    ('General options', ('--restore',), dict(action='store_true', help='Whether to restore or not.')),
    ('General options', ('--restore_metric',), dict(type=str, default='ae_mse', choices=['loss', 'ae_mse', 'ae_mse_sup', 'ae_mse_fake_detect', 'ae_mse_sup_fake_detect'], help='The metric used for restoring the weight')),
    ('General options', ('--log_train_freq',), dict(default=20, help='The log frequency for training.')),
    ('General options', ('--log_valid_freq',), dict(default=20, help='The log frequency for testing.')),
    ('Data specific options', ('--hours_from_admission',), dict(type=int, default=6, help='Hours of record to look at')),
    ('Data specific options', ('--num_workers',), dict(type=int, default=3, help='The number of workers used for loading data.')),
    ('Data specific options', ('--batch_size',), dict(type=int, default=256, help='batch size for the lstm training')),
    ('Data specific options', ('--norm_method',), dict(type=str, default='minmax', choices=['minmax'])),
    ('Data specific options', ('--aug_input',), dict(action='store_true', help='whether add gaussian noise to input ob and time point.')),
    ('Data specific options', ('--aug_std',), dict(type=float, default=0.1)),
    ('Data specific options', ('--scale',), dict(type=float, default=5, help='0: keep [0, 1]; otherwise scale the input to [-scale/2, +scale/2]')),
    ('Data specific options', ('--denoise',), dict(default=False, help='Whether to denoise the input.')),
    ('Data specific options', ('--num_variables',), dict(type=int, default=6)),
    ('Data specific options', ('--num_timestamps',), dict(type=int, default=354)),
    ('Data specific options', ('--data_filter',), dict(action='store_true')),
    ('Model specfic options', ('--ref_points',), dict(type=int, default=6, help='Number of reference points')),
    ('Model specfic options', ('--dropout',), dict(type=float, default=0.2)),
    ('Model specfic options', ('--fake_detection',), dict(default=True, help='Generate the fake samples and detect them')),
    ('Model specfic options', ('--triple_margin',), dict(type=float, default=0.)),
    ('Model specfic options', ('--triple_pos_std',), dict(type=float, default=0.1)),
    ('Training specific options', ('--unsup_aux_tasks',), dict(default={'fake_detection': 1., 'triplet': 1., 'kl': 10.})),
    ('Training specific options', ('--optimizer',), dict(default='Adam')),
    ('Training specific options', ('--init_lr', '-l'), dict(type=float, default=0.003)),
    ('Training specific options', ('--min_lr', '-mlr'), dict(type=float, default=1e-6)),
    ('Training specific options', ('--lr_decay_mode', '-lm'), dict(type=str, default='step', choices=['exp', 'anneal', 'plateau', 'step', 'warmup'])),
    ('Training specific options', ('--lr_decay_step_or_patience',), dict(type=int, default=20)),
    ('Training specific options', ('--lr_decay_rate', '-a'), dict(type=float, default=0.2)),
    ('Training specific options', ('--grad_clip',), dict(type=float, default=15)),
    ('Training specific options', ('--weight_decay_rate', '-wd'), dict(type=float, default=0.0004)),
    ('Training specific options', ('--early_stopping',), dict(default=50, help='The early stopping step.')),
]

P1_ONLY = [
    ('General options', ('--mode',), dict(type=str, default='eval', choices=['train', 'eval'])),
    ('Data specific options', ('--evaluate_interpolation',), dict(default=False)),
    ('Training specific options', ('--loss',), dict(default='ae_mse_sup_fake_detect', choices=LOSSES[:5])),
    ('Training specific options', ('--aux_tasks',), dict(default={'future_vital': .5})),
    ('Training specific options', ('--aux_pos_weights',), dict(default={'future_vital': 1, 'AKI_overall': 1, 'mort_status_30d': 1, 'ICU': 1})),
    ('Training specific options', ('--max_epochs',), dict(type=int, default=10000)),
]

P3_ONLY = [
    ('General options', ('--mode',), dict(type=str, default='train', choices=['train', 'eval'])),
    ('General options', ('--cluster_number',), dict(type=int, default=4, help='The number of cluster.')),
    ('General options', ('--dc_restore_metric',), dict(type=str, default='ae_mse')),
    ('General options', ('--init_cluster_center',), dict(type=str, default='kmeans', help='kmeans, random, none')),
    ('Model specfic options', ('--stopping_delta',), dict(type=float, default=0.0001)),
    ('Model specfic options', ('--update_interval',), dict(type=int, default=1)),
    ('Training specific options', ('--loss',), dict(default='ae_mse_sup_fake_detect_kl', choices=LOSSES)),
    ('Training specific options', ('--aux_tasks',), dict(default={'future_vital': .5})),
    ('Training specific options', ('--aux_pos_weights',), dict(default={'AKI_overall': 1, 'mort_status_30d': 1, 'ICU': 1, 'future_vital': 1})),
    ('Training specific options', ('--max_epochs',), dict(type=int, default=8000)),
]

EXTRA = [   # not upstream
    ('MI355X options', ('--host_loader',), dict(action='store_true', help='use torch DataLoader workers like upstream instead of the HBM-resident loader')),
    ('MI355X options', ('--amp_bf16',), dict(action='store_true', help='bf16 autocast for the bi-LSTMs / FC heads')),
    ('MI355X options', ('--f32_products',), dict(type=str, default=None, choices=['exact', 'x3'],
                                                  help="f32 step (no --amp_bf16): 'exact' = exact-f32 MFMA recurrence + f32 library GEMMs (the default); 'x3' = every "
                                                       "tensor f32, every dense product a three-term bf16 split on the matrix cores (losses within ~1e-6 of the reference at "
                                                       "1.6x the exact mode's throughput, no library GEMM)")),
    ('MI355X options', ('--hip_graph',), dict(action='store_true', default=None, help='always replay the training step from a captured hipGraph (single GPU); default: automatically for batches up to 8192 encounters, where the step is launch-bound')),
    ('MI355X options', ('--no_hip_graph',), dict(action='store_true', help='never capture the training step (eager launches)')),
    ('MI355X options', ('--cpu_padded_ob',), dict(action='store_true', help="feature dumps: write the padded slots of 'ob' as a CPU run of the reference does "
                                                                             "(masked in place before the dump) instead of as its GPU runs do")),
    ('MI355X options', ('--no_aux',), dict(action='store_true', help='shorthand: aux_tasks={} (the synthetic cohorts carry no outcome tables)')),
    ('MI355X options', ('--no_fake',), dict(action='store_true', help='shorthand: fake_detection=False')),
]


def build_parser(description, specific):
    parser = argparse.ArgumentParser(description=description)
    groups = {}
    for group, flags, kw in COMMON + specific + EXTRA:
        g = groups.setdefault(group, parser.add_argument_group(group))
        g.add_argument(*flags, **kw)
    return parser


def finalize(args):
    if args.no_aux:
        args.aux_tasks = {}
    if args.no_fake:
        args.fake_detection = False
    if not hasattr(args, 'cluster_number'):
        args.cluster_number = 0
    if getattr(args, 'amp_bf16', False) and getattr(args, 'f32_products', None) is not None:
        raise SystemExit('--f32_products selects how the f32 step forms its dense products; it does not combine with --amp_bf16')
    return args

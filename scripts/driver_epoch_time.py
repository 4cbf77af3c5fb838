#!/usr/bin/env python3
"""What a user of the drop-in drivers gets: seconds per training epoch of p1 (pretrain) and p3 (joint) on a 75 000-encounter synthetic
cohort (60 000 in the training split), through p1_pretrain_main / p3_clustering_main themselves, at upstream's default batch size 256
and at a large batch -- the host loop, the loader, logging and checkpointing included.
Usage: python3 scripts/driver_epoch_time.py [encounters] [batch ...]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

from deep_interpolation_clustering_amd import dataloader, synthetic  # noqa: E402
from deep_interpolation_clustering_amd import p1_pretrain_main as p1, p3_clustering_main as p3  # noqa: E402
from deep_interpolation_clustering_amd import _trainer_common as tc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 75000
batches = [int(b) for b in sys.argv[2:]] or [256, 8192]
base = tempfile.mkdtemp(prefix='dic_epoch_')
synthetic.write_split(base, n, C=6, T=96, H=24.0, lam=50.0, G=4)
dataloader.BASE_PATH = base
os.makedirs(os.path.join(base, 'run'))
os.chdir(os.path.join(base, 'run'))

spans = []
orig = tc.TrainerBase.train_one_epoch


def timed(self, dl, denoise=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = orig(self, dl, denoise)
    torch.cuda.synchronize()
    spans.append((len(dl), time.perf_counter() - t0))
    return out


tc.TrainerBase.train_one_epoch = timed
for B in batches:
    common = ['--hours_from_admission', '24', '--ref_points', '24', '--num_timestamps', '96', '--batch_size', str(B), '--dropout', '0', '--no_aux',
              '--no_fake', '--amp_bf16', '--log-level', 'WARNING']
    for name, mod, extra in (('p1', p1, ['--loss', 'ae_mse', '--max_epochs', '3']),
                             ('p3', p3, ['--loss', 'ae_mse_kl', '--cluster_number', '4', '--max_epochs', '5', '--stopping_delta', '-1'])):
        del spans[:]
        mod.main(mod.get_arguments(common + ['--mode', 'train'] + extra))
        steps, sec = spans[-1]                                # the last epoch: graphs captured, allocator warm
        print('%s  batch %5d  %4d steps/epoch  %.3f s/epoch  %.3f ms/step  %.0f encounters/s' % (name, B, steps, sec, sec / steps * 1e3, steps * B / sec), flush=True)

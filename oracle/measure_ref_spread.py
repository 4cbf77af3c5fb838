#!/usr/bin/env python3
"""Run-to-run spread of the REFERENCE's own trainers: oracle/make_golden_traj.py (which runs pretrain_trainer.Trainer.train() and
clustering_trainer.TrainerCluster.train() of /root/reference on the cfg1 cohort) executed twice in fresh processes with 8 and with 3 torch
threads; the maximum relative difference per quantity between the two runs and against the committed fixture goes to
tests/golden/ref_spread.json.  The GPU trajectory tests (tests/test_gpu_traj.py) take their tolerances from it: <= 3 x the larger of the
reference's own spread and the oracle's measured distance, stated next to each assert.

TEST INFRASTRUCTURE ONLY.  Run in the build container:  ``PYTHONDONTWRITEBYTECODE=1 python oracle/measure_ref_spread.py``.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, '..', 'tests', 'golden')

CHILD = r'''
import os, sys, torch
torch.set_num_threads(int(sys.argv[2]))
sys.path.insert(0, sys.argv[3])
import make_golden, make_golden_traj
make_golden.OUT = make_golden_traj.OUT = sys.argv[1]
import shutil
shutil.copy(os.path.join(sys.argv[4], 'netstep_plain.npz'), sys.argv[1])      # (the trajectory script checks its initial state against it)
make_golden_traj.main()
'''


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))) if a.size else 0.0


def compare(x, y):
    out = {}
    groups = {'p1_step_losses': ['p1/train_ae_mse'], 'p1_valid_losses': ['p1/valid_batch_ae_mse', 'p1/valid_ae_mse'],
              'p3_step_losses': ['p3/train_losses', 'p3k6/train_losses'], 'p3_valid_losses': ['p3/valid_batch_losses', 'p3k6/valid_batch_losses'],
              'kmeans_centers': ['p3/kmeans_centers', 'p3k6/kmeans_centers']}
    for g, keys in groups.items():
        out[g] = max(rel(x[k], y[k]) for k in keys)
    # per-step profile of the joint steps (loss column): how the two runs of one f32 Adam(amsgrad) loop separate
    out['p3_loss_by_step'] = [rel(x['p3/train_losses'][i, 0], y['p3/train_losses'][i, 0]) for i in range(len(x['p3/train_losses']))]
    out['p1_loss_by_step'] = [rel(x['p1/train_ae_mse'][i], y['p1/train_ae_mse'][i]) for i in range(len(x['p1/train_ae_mse']))]
    for pre in ('p1opt', 'p3opt', 'p3k6opt'):
        for st in ('exp_avg', 'exp_avg_sq', 'max_exp_avg_sq'):
            ks = [k for k in x if k.startswith(f'{pre}/{st}/') and 'compress_fc.module.model.0.bias' not in k]
            out[f'{pre}_{st}'] = max(rel(x[k], y[k]) for k in ks)
    ks = [k for k in x if (k.startswith('p3sdn/') or k.startswith('p3k6sdn/'))]
    out['p3_param_norms'] = max(rel(x[k], y[k]) for k in ks)
    out['labels_equal'] = bool(all(np.array_equal(x[k], y[k]) for k in ('p3/valid_labels', 'p3k6/valid_labels', 'p3/valid_prev_labels')))
    out['delta_equal'] = bool(np.array_equal(x['p3/delta'], y['p3/delta']) and np.array_equal(x['p3k6/delta'], y['p3k6/delta']))
    return out


def main():
    runs = []
    for threads in (8, 3):
        d = tempfile.mkdtemp(prefix='dic_spread_')
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
        subprocess.run([sys.executable, '-c', CHILD, d, str(threads), HERE, GOLD], check=True, env=env, stdout=subprocess.DEVNULL)
        runs.append(dict(np.load(os.path.join(d, 'traj_cfg1.npz'))))
    committed = dict(np.load(os.path.join(GOLD, 'traj_cfg1.npz')))
    res = {'_what': 'max relative difference between two runs of the reference trainers (oracle/make_golden_traj.py; 8 vs 3 torch threads) and of each '
                    'against the committed fixture tests/golden/traj_cfg1.npz', 'run8_vs_run3': compare(runs[0], runs[1]),
           'run8_vs_committed': compare(runs[0], committed), 'run3_vs_committed': compare(runs[1], committed)}
    with open(os.path.join(GOLD, 'ref_spread.json'), 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()

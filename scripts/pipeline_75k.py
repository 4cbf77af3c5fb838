#!/usr/bin/env python3
"""End-to-end p1 -> p3 on the BASELINE cohort size (75 k synthetic encounters) through the drop-in drivers.
Usage: python3 scripts/pipeline_75k.py [n_encounters] [batch] [bf16|f32]"""
import os, sys, time, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import numpy as np, torch
from deep_interpolation_clustering_amd import dataloader, synthetic
from deep_interpolation_clustering_amd import p1_pretrain_main as p1, p3_clustering_main as p3

n = int(sys.argv[1]) if len(sys.argv) > 1 else 75000
batch = sys.argv[2] if len(sys.argv) > 2 else '8192'
amp = ['--amp_bf16'] if (len(sys.argv) <= 3 or sys.argv[3] == 'bf16') else []
base = tempfile.mkdtemp(prefix='dic75k_')
t0 = time.time(); synthetic.write_split(base, n, C=6, T=96, H=24.0, lam=50.0, G=4); print('cohort written %.1fs' % (time.time() - t0), flush=True)
run = os.path.join(base, 'run'); os.makedirs(run); os.chdir(run); dataloader.BASE_PATH = base
common = ['--hours_from_admission', '24', '--ref_points', '24', '--num_timestamps', '96', '--batch_size', batch, '--dropout', '0',
          '--no_aux', '--log-level', 'INFO', '--log_train_freq', '1000000', '--log_valid_freq', '1000000'] + amp
t0 = time.time()
p1.main(p1.get_arguments(common + ['--mode', 'train', '--max_epochs', '4', '--loss', 'ae_mse_fake_detect']))
torch.cuda.synchronize(); print('p1 (3 epochs train+valid, feature dumps): %.1fs' % (time.time() - t0), flush=True)
t0 = time.time()
p3.main(p3.get_arguments(common + ['--mode', 'train', '--max_epochs', '4', '--loss', 'ae_mse_fake_detect_kl', '--cluster_number', '4']))
torch.cuda.synchronize(); print('p3 (k-means init n_init=20 + 3 joint epochs + feature dumps): %.1fs' % (time.time() - t0), flush=True)
f = np.load(os.path.join(run, 'Results/Clustering/out_feat/ae_mse/training.npy'), allow_pickle=True).item()
print('training latents', f['hidden'].shape, 'cluster sizes', np.bincount(f['cluster_pred'].argmax(1), minlength=4))

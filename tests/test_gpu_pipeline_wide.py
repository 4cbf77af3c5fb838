"""The drop-in drivers at BASELINE configs[3]'s shape -- 12 channels (encoder input 3C = 36: clustering_interp.py:102-111), T = 288 slots with ~200
observations per channel, K = 16 -- and in the three arithmetic modes of the step: p1 pretrain -> p3 joint clustering through the upstream file layout."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

COMMON = ['--hours_from_admission', '24', '--ref_points', '24', '--num_timestamps', '288', '--num_variables', '12', '--batch_size', '128',
          '--dropout', '0', '--no_aux', '--no_fake', '--log-level', 'WARNING']


@pytest.fixture(scope='module')
def wide_dir(tmp_path_factory):
    from deep_interpolation_clustering_amd import dataloader, synthetic
    base = tmp_path_factory.mktemp('dic_wide')
    synthetic.write_split(str(base), 640, C=12, T=288, H=24.0, lam=200.0, G=16)
    old_cwd, old_base = os.getcwd(), dataloader.BASE_PATH
    dataloader.BASE_PATH = str(base)
    yield base
    os.chdir(old_cwd)
    dataloader.BASE_PATH = old_base


@pytest.mark.parametrize('mode', ['bf16', 'f32x3', 'f32'])
def test_p1_to_p3_twelve_channels(wide_dir, mode):
    from deep_interpolation_clustering_amd import p1_pretrain_main as p1
    from deep_interpolation_clustering_amd import p3_clustering_main as p3
    run = wide_dir / f'run_{mode}'
    run.mkdir()
    os.chdir(run)
    flags = {'bf16': ['--amp_bf16'], 'f32x3': ['--f32_products', 'x3'], 'f32': []}[mode]
    a1 = p1.get_arguments(COMMON + flags + ['--mode', 'train', '--max_epochs', '4', '--loss', 'ae_mse'])
    p1.main(a1)
    feat = np.load(run / 'Results/Pretrain/out_feat/ae_mse/training.npy', allow_pickle=True).item()
    assert feat['hidden'].shape == (512, 256) and feat['rec_ob'].shape == (512, 12, 288)
    assert np.isfinite(feat['hidden']).all() and np.isfinite(feat['rec_ob']).all()
    ck = torch.load(run / 'Results/Pretrain/weight/ae_mse/model.pth.tar', map_location='cpu')
    assert tuple(ck['state_dict']['encoder.lstm.weight_ih_l0'].shape) == (512, 36)            # 3C = 36 input features
    a3 = p3.get_arguments(COMMON + flags + ['--mode', 'train', '--max_epochs', '3', '--loss', 'ae_mse_kl', '--cluster_number', '16'])
    p3.main(a3)
    cf = np.load(run / 'Results/Clustering/out_feat/ae_mse/validation.npy', allow_pickle=True).item()
    assert cf['cluster_pred'].shape == (64, 16)
    np.testing.assert_allclose(cf['cluster_pred'].sum(1), 1.0, rtol=1e-5)
    assert np.isfinite(cf['hidden']).all()

"""End-to-end drop-in surface on the GPU (BASELINE configs[0] shape: 1k synthetic encounters, 6 vitals, ~50 samples/24h,
K=4): p1 pretrain -> p3 joint clustering -> p2 K sweep -> p4 final labels, through the upstream file layout."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

COMMON = ['--hours_from_admission', '24', '--ref_points', '24', '--num_timestamps', '96', '--batch_size', '256',
          '--dropout', '0', '--no_aux', '--log-level', 'INFO']


@pytest.fixture(scope='module')
def run_dir(tmp_path_factory):
    from deep_interpolation_clustering_amd import dataloader, synthetic
    base = tmp_path_factory.mktemp('dic')
    synthetic.write_split(str(base), 1000, C=6, T=96, H=24.0, lam=50.0, G=4)
    run = base / 'run'
    run.mkdir()
    old_cwd, old_base = os.getcwd(), dataloader.BASE_PATH
    os.chdir(run)
    dataloader.BASE_PATH = str(base)
    yield run
    os.chdir(old_cwd)
    dataloader.BASE_PATH = old_base


def test_p1_to_p4(run_dir):
    from deep_interpolation_clustering_amd import p1_pretrain_main as p1, p2_clustering_optK as p2
    from deep_interpolation_clustering_amd import p3_clustering_main as p3, p4_clustering_final as p4
    # ---- p1: pretrain (with the fake-detection branch on, as upstream defaults) and dump latents
    a1 = p1.get_arguments(COMMON + ['--mode', 'train', '--max_epochs', '4', '--loss', 'ae_mse_fake_detect'])
    p1.main(a1)
    feat = np.load(run_dir / 'Results/Pretrain/out_feat/ae_mse/training.npy', allow_pickle=True).item()
    assert feat['hidden'].shape == (800, 256) and feat['rec_ob'].shape == (800, 6, 96)
    assert set(['encounter_id', 'ob', 'padding_mask', 'timestamp', 'ae_mask', 'hidden', 'rec_ob']) <= set(feat)
    ck = torch.load(run_dir / 'Results/Pretrain/weight/ae_mse/model.pth.tar', map_location='cpu')
    assert set(ck) == {'epoch', 'state_dict', 'optimizer'} and 'sci.kernel' in ck['state_dict']
    assert np.isfinite(feat['hidden']).all()
    # re_norm_data put ob back into physiologic units (sbp in [20, 300])
    m = feat['padding_mask'][:, 0] > 0
    assert 20 <= feat['ob'][:, 0][m].min() and feat['ob'][:, 0][m].max() <= 300

    # ---- p3: k-means initialised joint training
    a3 = p3.get_arguments(COMMON + ['--mode', 'train', '--max_epochs', '3', '--loss', 'ae_mse_fake_detect_kl', '--cluster_number', '4'])
    p3.main(a3)
    cf = np.load(run_dir / 'Results/Clustering/out_feat/ae_mse/validation.npy', allow_pickle=True).item()
    assert cf['cluster_pred'].shape == (100, 4)
    np.testing.assert_allclose(cf['cluster_pred'].sum(1), 1.0, rtol=1e-5)
    np.testing.assert_allclose(cf['cluster_label'].sum(1), 1.0, rtol=1e-5)

    # ---- p2: K sweep on the pretrain latents
    a2 = p2.get_arguments(['--k_max', '4', '--n_init', '2', '--gap_b', '2'])
    a2.restore_metric = ['ae_mse']
    a2.internal_metrics = ['Calinski-Harabasz', 'Davies-Bouldin_Index']
    res = p2.main(a2)['ae_mse']
    assert list(res['gap_sts']['k']) == [2.0, 3.0, 4.0] and np.isfinite(res['gap_sts'][['gap', 'ref', 'act']].to_numpy()).all()
    el = res['elbow']['train'].to_numpy()
    assert np.isfinite(el).all() and el[-1] < el[0]            # distortion falls from k = 2 to k_max (single k-means++ inits, as upstream's
                                                               # KMeans(k) with n_init='auto': a local optimum may make one step non-monotone)

    # ---- p4: final labels, both branches
    for method in ('kmeans', 'dl'):
        a4 = p4.get_arguments(['--cluster_method', method, '--num_clusters', '4'])
        a4.restore_metric = ['ae_mse']
        p4.main(a4)
        out = np.load(run_dir / f'Results/Clustering/out_feat/ae_mse_{method}_aligned/testing_4.npy', allow_pickle=True).item()
        assert out['cluster_id'].shape == (100,) and set(np.unique(out['cluster_id'])) <= {0, 1, 2, 3}


def test_device_loader_matches_dataset_semantics(run_dir):
    """HBM-resident loader vs the per-sample DataSet: same tensors for real samples; fake samples corrupt
    exactly max(1, n//2) valid slots per row with values in +-scale/2 and leave everything else untouched."""
    from deep_interpolation_clustering_amd import p1_pretrain_main as p1
    from deep_interpolation_clustering_amd.dataloader import DataSet, DeviceLoader
    args = p1.get_arguments(COMMON)
    ds = DataSet(args, 'validation')
    dl = DeviceLoader(ds, 64, False, torch.device('cuda'), seed=1)
    seen = 0
    for sample, fake in dl:
        B = sample['ob'].shape[0]
        for j in (0, B - 1):
            ref, _ = ds[seen + j]
            for k in ('ob', 'padding_mask', 'timestamp', 'ae_mask'):
                assert torch.equal(sample[k][j].cpu(), ref[k])
        changed = (fake['ob'] != sample['ob'])
        n = sample['lengths']
        assert torch.equal(changed.sum(-1), torch.clamp(n // 2, min=1))
        assert not changed[sample['padding_mask'] == 0].any()
        assert fake['ob'].abs().max() <= 2.5 + 1e-6
        assert torch.equal(fake['timestamp'], sample['timestamp'])
        seen += B
    assert seen == len(ds)

"""The package's DataSet (the host side of the input pipeline) against the REFERENCE's own dataloader.DataSet run on
the same cfg1 pickles (tests/golden/dataset_cfg1.npz, oracle/make_golden_traj.py): feed_data bit for bit, __getitem__ tensors,
the fake-sample rule on NumPy's global stream (same draws in the same order -> identical arrays and stream position), the
Gaussian augmentation on torch's stream."""
import hashlib
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _args(**over):
    a = dict(hours_from_admission=24, scale=5.0, aux_tasks={}, fake_detection=False, aug_input=False, aug_std=0.1, num_variables=6)
    a.update(over)
    return SimpleNamespace(**a)


@pytest.fixture(scope='module')
def cohort_dir(tmp_path_factory):
    from deep_interpolation_clustering_amd import dataloader, synthetic
    base = tmp_path_factory.mktemp('dic_ds')
    synthetic.write_split(str(base), 1000, C=6, T=96, H=24.0, lam=50.0, G=4)
    old = dataloader.BASE_PATH
    dataloader.BASE_PATH = str(base)
    yield base
    dataloader.BASE_PATH = old


def test_feed_data_equals_reference(cohort_dir):
    from deep_interpolation_clustering_amd.dataloader import DataSet
    g = np.load(os.path.join(GOLDEN, 'dataset_cfg1.npz'))
    for cohort in ('training', 'validation', 'testing'):
        ds = DataSet(_args(), cohort)
        fd = ds.feed_data
        assert tuple(g[f'{cohort}/shape']) == fd.shape and str(g[f'{cohort}/dtype']) == str(fd.dtype)
        assert hashlib.sha256(np.ascontiguousarray(fd.astype(np.float32)).tobytes()).hexdigest() == str(g[f'{cohort}/sha256_f32'])
        sums = [fd[:, 6 * i:6 * (i + 1)].sum(dtype=np.float64) for i in range(4)]
        np.testing.assert_allclose(sums, g[f'{cohort}/plane_sums'], rtol=1e-12)
        assert int(ds.encounter_ids[0]) == int(g[f'{cohort}/first_id']) and len(ds) == int(g[f'{cohort}/n'])


def test_getitem_fake_rule_and_augmentation_equal_reference(cohort_dir):
    from deep_interpolation_clustering_amd.dataloader import DataSet
    g = np.load(os.path.join(GOLDEN, 'dataset_cfg1.npz'))
    ds = DataSet(_args(), 'validation')
    for i in (0, 7, 99):
        s, f = ds[i]
        assert f is s and int(s['encounter_id']) == int(g[f'item{i}/encounter_id'])
        for k in ('ob', 'padding_mask', 'timestamp', 'ae_mask'):
            assert s[k].dtype == torch.float32
            np.testing.assert_array_equal(s[k].numpy(), g[f'item{i}/{k}'], err_msg=f'{i}:{k}')
    # dataloader.py:182-193: per channel np.random.choice(n, max(1, int(n*0.5)), replace=False) then np.random.rand(n_perm)
    ds_f = DataSet(_args(fake_detection=True), 'validation')
    np.random.seed(123)
    for i in (0, 7):
        s, f = ds_f[i]
        np.testing.assert_array_equal(f['ob'].numpy(), g[f'fake{i}/ob'])
        changed = (f['ob'] != s['ob']).sum(-1)
        n = s['padding_mask'].sum(-1)
        assert torch.equal(changed, torch.clamp((n * 0.5).floor(), min=1).to(changed.dtype))
    assert np.random.random() == float(g['fake/np_state_pos'])           # the same number of draws was taken from the global stream
    ds_0 = DataSet(_args(fake_detection=True, scale=0), 'validation')
    np.random.seed(124)
    np.testing.assert_array_equal(ds_0[3][1]['ob'].numpy(), g['fake_scale0/ob'])
    ds_a = DataSet(_args(aug_input=True, aug_std=0.25), 'training')
    torch.manual_seed(5)
    s, _ = ds_a[3]
    np.testing.assert_array_equal(s['ob'].numpy(), g['aug3/ob'])
    np.testing.assert_array_equal(s['timestamp'].numpy(), g['aug3/timestamp'])

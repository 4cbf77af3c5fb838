"""Sharded joint step == single-device joint step on the same global batch, on the GPU.
Two ranks share the one GPU of the test box and exchange through gloo (DIC_DIST_BACKEND=gloo); on an 8-GPU node the
same code runs one rank per GPU over RCCL."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args():
    return SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=False, triple_margin=0.0, cluster_number=4, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})


def _run(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      DIC_DIST_BACKEND='gloo')
    from deep_interpolation_clustering_amd import dist, synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    if world > 1:
        dist.init_from_env()
    dev = torch.device('cuda', 0)
    coh = synthetic.make_cohort(256, seed=21)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    lo, hi = dist.shard_bounds(256)
    x, ob, lens = (torch.tensor(a[lo:hi], device=dev) for a in (x_np, ob_np, n))
    torch.manual_seed(5)
    net = Net(_args(), dev).to(dev)
    net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), _args())
    res = []
    for _ in range(3):
        losses, gnorm, _ = st.step(x, ob, None, lens)
        res.append([float(losses['loss'].detach()), float(losses['ae_mse'].detach()), float(losses['kl'].detach()), float(gnorm)])
    torch.save({'traj': np.array(res), 'flat': st.flat.flat.detach().cpu(), 'bn_mean': net.rbf.compress_fc.module.model[1].running_mean.cpu()},
               os.path.join(out, f'w{world}_r{rank}.pt'))
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


def test_two_rank_step_equals_single_device(tmp_path):
    port = 29600 + (os.getpid() % 1000)
    mp.spawn(_run, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_run, args=(2, port + 1, str(tmp_path)), nprocs=2, join=True)
    one = torch.load(tmp_path / 'w1_r0.pt', weights_only=False)
    r0, r1 = torch.load(tmp_path / 'w2_r0.pt', weights_only=False), torch.load(tmp_path / 'w2_r1.pt', weights_only=False)
    assert torch.equal(r0['flat'], r1['flat'])                                     # replicas stay identical
    np.testing.assert_allclose(r0['traj'], r1['traj'], rtol=0, atol=0)
    np.testing.assert_allclose(r0['traj'][0, :2], one['traj'][0, :2], rtol=2e-6)   # first step: same loss to f32 rounding
    # later steps: Adam turns rounding noise on zero-gradient parameters (biases feeding BatchNorm) into O(lr) moves
    np.testing.assert_allclose(r0['traj'][:, :2], one['traj'][:, :2], rtol=2e-4)
    np.testing.assert_allclose(r0['traj'][:, 2], one['traj'][:, 2], rtol=2e-3, atol=1e-7)   # kl (tiny, ill-conditioned)
    np.testing.assert_allclose(r0['traj'][:, 3], one['traj'][:, 3], rtol=2e-3)     # gradient norm
    np.testing.assert_allclose(r0['bn_mean'].numpy(), one['bn_mean'].numpy(), rtol=1e-3, atol=1e-4)   # global-batch BatchNorm moments
    d = (r0['flat'] - one['flat']).abs()
    assert float(d.max()) < 2e-2 and float((d > 1e-4).float().mean()) < 0.01        # Adam amplifies noise only where grad ~ 0

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


def _pretrained(net, K=4):
    """The reference's own pretrained weights (p1 on the cfg1 cohort) with its scikit-learn centroids (tests/golden/traj_cfg1.npz,
    netstep_cfg_K*.npz): the p3 regime, where the KL term is O(0.1) and well conditioned -- not the 1e-4 of Xavier centroids."""
    g = os.path.join(ROOT, 'tests', 'golden')
    t = np.load(os.path.join(g, 'traj_cfg1.npz'))
    sd = {k[5:]: torch.tensor(t[k]) for k in t.files if k.startswith('p1sd/')}
    sd['cluster_assignment.cluster_centers'] = torch.tensor(np.load(os.path.join(g, f'netstep_cfg_K{K}.npz'))['centers'])
    missing = net.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.startswith('fake_det_head.') for k in missing.missing_keys)
    return net


def _env(rank, world, port, rccl):
    """gloo between ranks that share the test GPU -- or (rccl) ONE rank on RCCL with the sharded code paths switched on."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ['DIC_FC_BWD_MIN_ROWS'] = '1024'        # the shards of these small batches take the large-batch CompressFC path (one autograd node)
    if rccl:
        os.environ.pop('DIC_DIST_BACKEND', None)
        os.environ['DIC_DIST_SINGLE_RANK'] = '1'
    else:
        os.environ['DIC_DIST_BACKEND'] = 'gloo'


def _join(rccl):
    from deep_interpolation_clustering_amd import dist
    import torch.distributed as td
    if rccl:
        assert td.get_backend() == 'nccl' and dist.is_sharded() and dist.world_size() == 1
    return dist


def _leave():
    import torch.distributed as td
    if td.is_initialized():
        td.destroy_process_group()


def _run_fake(rank, world, port, out, rccl=False):
    """Upstream's default unsupervised objective (fake detection + KL) on an ODD global batch: shards of 127 and 128 rows."""
    sys.path.insert(0, ROOT)
    _env(rank, world, port, rccl)
    from deep_interpolation_clustering_amd import dist, synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    if world > 1 or rccl:
        dist.init_from_env()
        _join(rccl)
    dev = torch.device('cuda', 0)
    Bg = 255
    coh = synthetic.make_cohort(Bg, seed=22)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    rng = np.random.default_rng(1)
    fake_np = x_np.copy()
    hit = (x_np[:, 6:12] > 0) & (rng.random(x_np[:, :6].shape) < 0.5)
    fake_np[:, :6] = np.where(hit, rng.uniform(-2.5, 2.5, hit.shape).astype(np.float32), x_np[:, :6])
    lo, hi = dist.shard_bounds(Bg)
    x, fx, ob, lens = (torch.tensor(a[lo:hi], device=dev) for a in (x_np, fake_np, ob_np, n))
    args = _args()
    args.fake_detection, args.loss = True, 'ae_mse_fake_detect_kl'
    torch.manual_seed(5)
    net = _pretrained(Net(args, dev).to(dev))
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args)
    b = hi - lo
    perm = torch.arange(2 * b, device=dev)
    label = torch.cat([torch.ones(b, device=dev), torch.zeros(b, device=dev)]).to(torch.int64)
    res = {}
    for mode in ('train', 'eval'):
        net.train(mode == 'train')
        if mode == 'train':
            losses, gnorm, _ = st.step(x, ob, None, lens, fake_x=fx, fake_perm_idx=perm, fake_det_label=label)
            res['gnorm'] = float(gnorm)
        else:
            with torch.no_grad():
                losses, _, _, _ = st.forward_loss(x, ob, None, lens, fake_x=fx, fake_perm_idx=perm, fake_det_label=label)
        res[mode] = {k: float(v.detach()) for k, v in losses.items()}
    torch.save(res, os.path.join(out, f'f{world}{"x" if rccl else ""}_r{rank}.pt'))
    _leave()


def test_two_rank_fake_detection_odd_batch_losses_are_global(tmp_path):
    """every loss term a rank reports must be the GLOBAL-batch value (it feeds ReduceLROnPlateau, best-checkpoint
    selection and early stopping), and the sharded gradient must equal the single-device one, also when the global batch (255)
    does not divide by the world size."""
    port = 29900 + (os.getpid() % 1000)
    mp.spawn(_run_fake, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_run_fake, args=(2, port + 1, str(tmp_path)), nprocs=2, join=True)
    one = torch.load(tmp_path / 'f1_r0.pt', weights_only=False)
    r0, r1 = torch.load(tmp_path / 'f2_r0.pt', weights_only=False), torch.load(tmp_path / 'f2_r1.pt', weights_only=False)
    for mode in ('train', 'eval'):
        assert r0[mode] == r1[mode]
        for k, v in one[mode].items():
            np.testing.assert_allclose(r0[mode][k], v, rtol=1e-5, atol=0, err_msg=f'{mode}:{k}')          # kl included: O(0.1), no floor
        assert r0[mode]['kl'] > 0.05
    np.testing.assert_allclose(r0['gnorm'], one['gnorm'], rtol=1e-4)


def _run(rank, world, port, out, rccl=False):
    sys.path.insert(0, ROOT)
    _env(rank, world, port, rccl)
    from deep_interpolation_clustering_amd import dist, synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    if world > 1 or rccl:
        dist.init_from_env()
        _join(rccl)
    dev = torch.device('cuda', 0)
    coh = synthetic.make_cohort(256, seed=21)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    lo, hi = dist.shard_bounds(256)
    x, ob, lens = (torch.tensor(a[lo:hi], device=dev) for a in (x_np, ob_np, n))
    torch.manual_seed(5)
    net = _pretrained(Net(_args(), dev).to(dev))
    net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), _args())
    import torch.distributed as td
    calls, inner = [], td.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls.append(int(t.numel()))
        return inner(t, *a, **k)
    td.all_reduce = counting_all_reduce          # every collective of the step goes through it (dist.all_reduce_sum_, the gradient bucket's async pieces)
    res, per_step = [], []
    for _ in range(3):
        n0 = len(calls)
        losses, gnorm, _ = st.step(x, ob, None, lens)
        per_step.append(calls[n0:])
        res.append([float(losses['loss'].detach()), float(losses['ae_mse'].detach()), float(losses['kl'].detach()), float(gnorm)])
    td.all_reduce = inner
    torch.save({'traj': np.array(res), 'flat': st.flat.flat.detach().cpu(), 'bn_mean': net.rbf.compress_fc.module.model[1].running_mean.cpu(),
                'collectives': per_step, 'bucket': [int(st.flat._split or 0), int(st.flat.grad.numel())]},
               os.path.join(out, f'w{world}{"x" if rccl else ""}_r{rank}.pt'))
    _leave()


def test_two_rank_step_equals_single_device(tmp_path):
    port = 29600 + (os.getpid() % 1000)
    mp.spawn(_run, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_run, args=(2, port + 1, str(tmp_path)), nprocs=2, join=True)
    one = torch.load(tmp_path / 'w1_r0.pt', weights_only=False)
    r0, r1 = torch.load(tmp_path / 'w2_r0.pt', weights_only=False), torch.load(tmp_path / 'w2_r1.pt', weights_only=False)
    assert torch.equal(r0['flat'], r1['flat'])                                     # replicas stay identical
    np.testing.assert_allclose(r0['traj'], r1['traj'], rtol=0, atol=0)
    np.testing.assert_allclose(r0['traj'][0, :3], one['traj'][0, :3], rtol=1e-5, atol=0)   # first step: loss, ae_mse AND kl to f32 rounding
    assert one['traj'][0, 2] > 0.05
    # later steps: Adam turns rounding noise on zero-gradient parameters (biases feeding BatchNorm) into O(lr) moves
    np.testing.assert_allclose(r0['traj'][:, :3], one['traj'][:, :3], rtol=2e-4, atol=0)
    np.testing.assert_allclose(r0['traj'][:, 3], one['traj'][:, 3], rtol=2e-3)     # gradient norm
    np.testing.assert_allclose(r0['bn_mean'].numpy(), one['bn_mean'].numpy(), rtol=1e-3, atol=1e-4)   # global-batch BatchNorm moments
    d = (r0['flat'] - one['flat']).abs()
    assert float(d.max()) < 2e-2 and float((d > 1e-4).float().mean()) < 0.01        # Adam amplifies noise only where grad ~ 0
    # the exchanges of a sharded joint step, bounded by what the data dependencies force (at 4 096 rows per GPU every extra
    # latency-bound exchange is 1-2 % of the step): forward -- BatchNorm moments of CompressFC carrying the DEC column sums f_j (they are
    # known earlier and needed later: dist.deferred_sum_), ONE buffer with the reconstruction SSE + mask count (needs the BatchNorm output)
    # and the KL sum + batch rows (needs f_j) -- the former waits for the latter; backward -- BatchNorm's two column sums, the gradient
    # bucket in two pieces (the decoder-side piece overlaps the encoder backward): five
    assert one['collectives'] == [[], [], []]
    split, total = r0['bucket']
    assert 0 < split < total
    for step in r0['collectives']:
        assert len(step) <= 5, step
        assert sorted(step)[-2:] == sorted([split, total - split]), (step, split, total)                # the gradient bucket in its two pieces; the rest small statistics
        assert sum(1 for n in step if n > 10000) == 2
        assert any(n == 2 * 128 + 1 + 4 for n in step), step                                            # f_j (K = 4) rode with the 257 BatchNorm sums
        assert any(n == 2 + 2 for n in step), step                                                      # reconstruction SSE + count rode with the KL sum + row count


def _run_cfg3(rank, world, port, out, precision=None):
    """BASELINE configs[2]'s step shape: K = 8, encounters sharded over the ranks -- against the REFERENCE's own single-device step
    on the same 64 encounters (tests/golden/netstep_cfg_K8.npz)."""
    sys.path.insert(0, ROOT)
    _env(rank, world, port, False)
    from deep_interpolation_clustering_amd import dist
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    if world > 1:
        dist.init_from_env()
    dev = torch.device('cuda', 0)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'netstep_cfg_K8.npz'))
    args = _args()
    args.cluster_number = 8
    net = _pretrained(Net(args, dev).to(dev), K=8)
    net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, precision=precision)
    lo, hi = dist.shard_bounds(g['x'].shape[0])
    x, ob = torch.tensor(g['x'][lo:hi], device=dev), torch.tensor(g['ob'][lo:hi], device=dev)
    lens = x[:, 6:12].sum(-1).to(torch.int32)
    losses, gnorm, z = st.step(x, ob, None, lens)
    sd = net.state_dict()
    torch.save({'losses': {k: float(v.detach()) for k, v in losses.items()}, 'gnorm': float(gnorm), 'z': z.detach().cpu(), 'rows': (lo, hi),
                'centers': sd['cluster_assignment.cluster_centers'].cpu(), 'sci': sd['sci.kernel'].cpu()},
               os.path.join(out, f'c{world}_r{rank}.pt'))
    _leave()


@pytest.mark.parametrize('precision', [None, 'x3'])
def test_two_rank_K8_step_equals_reference_step(tmp_path, precision):
    """(precision 'x3': the same sharded step with every dense product a three-term bf16 split -- the parity-grade throughput mode -- held to the
    same 1e-5 on loss, ae_mse and kl against the reference's single-device step.)"""
    port = 29400 + (os.getpid() % 1000) + (7 if precision else 0)
    mp.spawn(_run_cfg3, args=(2, port, str(tmp_path), precision), nprocs=2, join=True)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'netstep_cfg_K8.npz'))
    for r in (0, 1):
        res = torch.load(tmp_path / f'c2_r{r}.pt', weights_only=False)
        for k in ('loss', 'ae_mse', 'kl'):                       # every rank reports the GLOBAL-batch terms: north_star's 1e-5, no floor
            np.testing.assert_allclose(res['losses'][k], float(g['loss_' + k]), rtol=1e-5, atol=0, err_msg=f'rank {r}: {k}')
        np.testing.assert_allclose(res['gnorm'], float(g['gnorm']), rtol=1e-4)
        lo, hi = res['rows']
        # (latents element by element: products good to 2^-24 in the exact mode, ~2^-17 per product in the x3 mode -- measured 4.9e-6 on values of O(0.3))
        np.testing.assert_allclose(res['z'].numpy(), g['z'][lo:hi], rtol=1e-4, atol=1.5e-5 if precision == 'x3' else 2e-6)
        np.testing.assert_allclose(res['centers'].numpy(), g['sd1/cluster_assignment.cluster_centers'], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(res['sci'].numpy(), g['sd1/sci.kernel'], rtol=1e-4, atol=2e-5)


COMMON = ['--hours_from_admission', '24', '--ref_points', '24', '--num_timestamps', '96', '--batch_size', '128',
          '--dropout', '0', '--no_aux', '--no_fake', '--amp_bf16', '--log-level', 'WARNING', '--seed', '11']


def _run_drivers(rank, world, port, base):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      DIC_DIST_BACKEND='gloo')
    from deep_interpolation_clustering_amd import dataloader
    from deep_interpolation_clustering_amd import p1_pretrain_main as p1, p3_clustering_main as p3
    run = os.path.join(base, 'run')
    os.makedirs(run, exist_ok=True)
    os.chdir(run)
    dataloader.BASE_PATH = base
    p1.main(p1.get_arguments(COMMON + ['--mode', 'train', '--max_epochs', '3', '--loss', 'ae_mse']))
    a3 = p3.get_arguments(COMMON + ['--mode', 'train', '--max_epochs', '3', '--loss', 'ae_mse_kl', '--cluster_number', '4'])
    # capture this rank's final replica before the evaluation passes reload checkpoints
    from deep_interpolation_clustering_amd import clustering_trainer as ct
    orig = ct.TrainerCluster.train

    def train_and_dump(self):
        orig(self)
        torch.save({'flat': self.stepper.flat.flat.detach().cpu(), 'centers': self.model.get_cluster_center().detach().cpu()},
                   os.path.join(base, f'replica_r{rank}.pt'))
    ct.TrainerCluster.train = train_and_dump
    p3.main(a3)
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


def test_two_rank_drivers_keep_replicas_identical_and_dump_every_encounter(tmp_path):
    """p1 -> p3 under two ranks (sharded training batches): k-means initialisation, label-change early stopping and the
    feature dumps must see ALL encounters on every rank, or the replicas diverge / the dumps lose rows."""
    sys.path.insert(0, ROOT)
    from deep_interpolation_clustering_amd import synthetic
    base = str(tmp_path)
    synthetic.write_split(base, 500, C=6, T=96, H=24.0, lam=50.0, G=4)
    port = 29700 + (os.getpid() % 1000)
    mp.spawn(_run_drivers, args=(2, port, base), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(base, f'replica_r{r}.pt'), weights_only=False) for r in (0, 1))
    assert torch.equal(r0['flat'], r1['flat']) and torch.equal(r0['centers'], r1['centers'])
    assert float(r0['centers'].abs().sum()) > 0
    feat = np.load(os.path.join(base, 'run/Results/Clustering/out_feat/ae_mse/training.npy'), allow_pickle=True).item()
    assert feat['hidden'].shape == (400, 256) and len(set(feat['encounter_id'].tolist())) == 400
    pre = np.load(os.path.join(base, 'run/Results/Pretrain/out_feat/ae_mse/training.npy'), allow_pickle=True).item()
    assert pre['hidden'].shape == (400, 256)
    # the evaluation / feature passes are SHARDED over the ranks and assembled by one collective per tensor: from the SAME checkpoints
    # a single process must write the same dumps (the two-rank dumps are moved aside, p1 --mode eval regenerates them)
    feat2 = os.path.join(base, 'run/Results/Pretrain/out_feat_2rank')
    os.rename(os.path.join(base, 'run/Results/Pretrain/out_feat'), feat2)
    mp.spawn(_run_p1_eval, args=(base,), nprocs=1, join=True)
    for cohort in ('training', 'validation', 'testing'):
        two = np.load(os.path.join(feat2, f'ae_mse/{cohort}.npy'), allow_pickle=True).item()
        one = np.load(os.path.join(base, f'run/Results/Pretrain/out_feat/ae_mse/{cohort}.npy'), allow_pickle=True).item()
        assert set(two.keys()) == set(one.keys())
        # (a single process dumps the training cohort in its shuffled loader's order, as upstream does: align by encounter)
        o2, o1 = np.argsort(two['encounter_id']), np.argsort(one['encounter_id'])
        assert np.array_equal(two['encounter_id'][o2], one['encounter_id'][o1])
        for k in one:
            a, b_ = two[k][o2], one[k][o1]
            if k in ('ob', 'padding_mask', 'timestamp', 'ae_mask', 'encounter_id'):
                np.testing.assert_array_equal(a, b_, err_msg=f'{cohort}/{k}')                        # inputs: the same rows
            else:
                np.testing.assert_allclose(a, b_, rtol=1e-3, atol=1e-3 * max(1.0, float(np.abs(b_).max())), err_msg=f'{cohort}/{k}')


def _run_p1_fake(rank, world, port, base):
    """p1 with upstream's DEFAULT fake-detection objective under two ranks, then the per-cohort feature dumps (p1:143-146)."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      DIC_DIST_BACKEND='gloo')
    from deep_interpolation_clustering_amd import dataloader
    from deep_interpolation_clustering_amd import p1_pretrain_main as p1
    run = os.path.join(base, 'run')
    os.makedirs(run, exist_ok=True)
    os.chdir(run)
    dataloader.BASE_PATH = base
    common = [a for a in COMMON if a != '--no_fake']
    p1.main(p1.get_arguments(common + ['--mode', 'train', '--max_epochs', '2', '--loss', 'ae_mse_fake_detect']))
    import torch.distributed as td
    td.destroy_process_group()


def test_two_rank_feature_dump_with_fake_detection(tmp_path):
    """with fake detection on (the default) the per-batch records carry 'fake_det' -- 2 x batch rows in a per-rank random
    order -- which the sharded row assembly cannot place; the sharded evaluation / feature passes must still run and dump every
    per-encounter tensor for ALL encounters."""
    sys.path.insert(0, ROOT)
    from deep_interpolation_clustering_amd import synthetic
    base = str(tmp_path)
    synthetic.write_split(base, 300, C=6, T=96, H=24.0, lam=50.0, G=4)
    mp.spawn(_run_p1_fake, args=(2, 29300 + (os.getpid() % 1000), base), nprocs=2, join=True)
    for cohort, n in (('training', 240), ('validation', 30), ('testing', 30)):
        d = np.load(os.path.join(base, f'run/Results/Pretrain/out_feat/ae_mse/{cohort}.npy'), allow_pickle=True).item()
        assert d['hidden'].shape == (n, 256) and d['rec_ob'].shape == (n, 6, 96) and np.isfinite(d['hidden']).all()
        assert len(set(d['encounter_id'].tolist())) == n and 'fake_det' not in d


def _run_p1_eval(rank, base):
    sys.path.insert(0, ROOT)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        os.environ.pop(k, None)
    from deep_interpolation_clustering_amd import dataloader
    from deep_interpolation_clustering_amd import p1_pretrain_main as p1
    os.chdir(os.path.join(base, 'run'))
    dataloader.BASE_PATH = base
    p1.main(p1.get_arguments(COMMON + ['--mode', 'eval', '--loss', 'ae_mse']))


def _run_kmeans(rank, world, port, out, rccl=False):
    sys.path.insert(0, ROOT)
    _env(rank, world, port, rccl)
    from deep_interpolation_clustering_amd import dist
    from deep_interpolation_clustering_amd.kmeans import KMeans
    from oracle.synth import latent_blobs
    if world > 1 or rccl:
        dist.init_from_env()
        _join(rccl)
    X, _ = latent_blobs(3, 20011, 256, 5)                       # odd size: uneven shards
    res = {}
    km = KMeans(n_clusters=5, n_init=4, random_state=7, shard_points=True).fit(X)
    res['pp'] = (km.labels_, km.cluster_centers_, km.inertia_, km.n_iter_)
    # an init that leaves a cluster empty (one centre far from all points): sharded runs fall back to the unsharded kernels
    init = np.concatenate([X[:3], X[:1] + 100.0], axis=0).astype(np.float32)
    km = KMeans(n_clusters=4, init=init, n_init=1, shard_points=True).fit(X)
    res['empty'] = (km.labels_, km.cluster_centers_, km.inertia_, km.n_iter_)
    torch.save(res, os.path.join(out, f'km_w{world}{"x" if rccl else ""}_r{rank}.pt'))
    _leave()


def test_sharded_kmeans_equals_single_device(tmp_path):
    """Points sharded over two ranks, centroid partial sums all-reduced every Lloyd iteration (dic_kmeans_lloyd_partial /
    _finish) == the single-device fit: same labels, same iteration count, centres to f32 rounding -- on every rank."""
    port = 29800 + (os.getpid() % 1000)
    mp.spawn(_run_kmeans, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_run_kmeans, args=(2, port + 1, str(tmp_path)), nprocs=2, join=True)
    one = torch.load(tmp_path / 'km_w1_r0.pt', weights_only=False)
    r0, r1 = torch.load(tmp_path / 'km_w2_r0.pt', weights_only=False), torch.load(tmp_path / 'km_w2_r1.pt', weights_only=False)
    for case in ('pp', 'empty'):
        for a, b in ((r0, r1), (r0, one)):
            la, ca, ia, na = a[case]
            lb, cb, ib, nb = b[case]
            assert np.array_equal(la, lb), case
            assert na == nb, case
            np.testing.assert_allclose(ca, cb, rtol=1e-5, atol=1e-6, err_msg=case)
            np.testing.assert_allclose(ia, ib, rtol=1e-5, err_msg=case)
    assert len(np.unique(one['empty'][0])) == 4               # the empty cluster was relocated, as scikit-learn does


def _run_gap(rank, world, port, out, rccl=False):
    sys.path.insert(0, ROOT)
    _env(rank, world, port, rccl)
    from deep_interpolation_clustering_amd import dist
    from deep_interpolation_clustering_amd import p2_clustering_optK as p2
    from oracle.synth import latent_blobs
    dist.init_from_env()
    _join(rccl)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'gap_table_blobs.npz'))
    X, _ = latent_blobs(int(g['seed']), int(g['N']), int(g['D']), int(g['G']))
    cols = [str(c) for c in g['columns']]
    km = p2.KM(int(g['k_max']), os.path.join(out, f'r{rank}'), cols[5:], int(g['n_init']), int(g['gap_b']))
    np.random.seed(int(g['np_seed']))
    fits = []
    fit0 = p2.KMeans.fit
    p2.KMeans.fit = lambda self, X, *a, **k: (fits.append(self.n_clusters), fit0(self, X, *a, **k))[1]
    df = km.compute_gap_internal_metric(X, int(g['k_max']), n_references=int(g['gap_b']), version=1).astype(float)
    torch.save(dict(table=df.to_numpy(), cols=list(df.columns), pos=np.random.random(), fits=fits),
               os.path.join(out, f'gap_w{world}{"x" if rccl else ""}_r{rank}.pt'))
    _leave()


def test_two_rank_gap_table_equals_reference(tmp_path):
    """The gap statistic's (K, reference set) problems dealt over two ranks (they shard with no data-path collective):
    every rank walks upstream's global NumPy stream in full but fits only its own problems; the assembled table on EVERY rank is the
    one the reference's own KM.compute_gap_internal_metric produced (oracle/make_golden_gap.py) and the stream ends where upstream's does."""
    port = 29400 + (os.getpid() % 1000)
    mp.spawn(_run_gap, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_run_gap, args=(1, port + 1, str(tmp_path), True), nprocs=1, join=True)       # and one rank with the table's all-reduce on RCCL
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'gap_table_blobs.npz'))
    cols = [str(c) for c in g['columns']]
    n_fits = (int(g['k_max']) - 1) * (int(g['gap_b']) + 1)
    res = [torch.load(tmp_path / f'gap_w2_r{r}.pt', weights_only=False) for r in range(2)]
    assert len(res[0]['fits']) + len(res[1]['fits']) == n_fits and abs(len(res[0]['fits']) - len(res[1]['fits'])) <= 1     # the work IS split
    for r in res:
        assert r['cols'] == cols
        assert r['pos'] == float(g['stream_pos']), 'the global stream was consumed differently'
        for j, c in enumerate(cols):
            np.testing.assert_allclose(r['table'][:, j], g['table'][:, j], rtol=1e-5, atol=1e-5 if c == 'gap' else 0, err_msg=c)
    assert np.array_equal(res[0]['table'], res[1]['table'])
    one = torch.load(tmp_path / 'gap_w1x_r0.pt', weights_only=False)
    assert len(one['fits']) == n_fits and one['pos'] == float(g['stream_pos'])
    np.testing.assert_allclose(one['table'], g['table'], rtol=1e-5, atol=1e-5)


def test_bench_runs_sharded_on_two_ranks(tmp_path):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one process per rank), two ranks sharing the
    test GPU over gloo: it must finish (every rank runs the traced steps rank 0 profiles -- a sharded step is full of collectives)
    and print ONE JSON line with the whole-job rate."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DIC_DIST_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '512',
           '--encounters', '4096', '--no-secondary', '--no-cpu-baseline', '--kernel-iters', '1']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['scaling'] == 'weak' and rec['config']['global_batch'] == 1024
    assert abs(rec['value'] - 2 * 512 * 3 / (rec['ms_per_step'] * 3e-3)) <= 0.01 * rec['value']      # whole-job rate over all ranks
    # the bench walks its cohort: 1 200 encounters per rank in batches of 512 = 512 + 512 + 176 per epoch, no drop_last; three
    # warm-up + three timed steps = exactly one timed epoch on each rank; `value` counts the encounters stepped, not steps x batch
    cmd2 = [c for c in cmd]
    cmd2[cmd2.index('--encounters') + 1], cmd2[cmd2.index('--warmup') + 1] = '1200', '3'
    res = subprocess.run(cmd2, env=env, capture_output=True, text=True, timeout=280, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    rec = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][0])
    assert rec['config']['batches_per_epoch'] == 3 and rec['config']['encounters_stepped'] == 2 * 1200
    assert 'incl. the last of 176' in rec['config']['workload']
    assert abs(rec['value'] - 2 * 1200 / (rec['ms_per_step'] * 3e-3)) <= 0.01 * rec['value']
    # strong scaling (BASELINE configs[2]'s shape of run): ONE cohort sharded over the ranks, the GLOBAL batch fixed
    res = subprocess.run(cmd + ['--scaling', 'strong'], env=env, capture_output=True, text=True, timeout=280, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    rec = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][0])
    assert rec['scaling'] == 'strong' and rec['config']['global_batch'] == 512 and rec['config']['per_gpu_batch'] == 256
    assert 'of ONE 4096-encounter' in rec['config']['workload']
    assert abs(rec['value'] - 512 * 3 / (rec['ms_per_step'] * 3e-3)) <= 0.01 * rec['value']


def test_bench_launches_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` with NO launcher around it (the shape of the driver's single-GPU command with another N; the reference's
    multi-GPU entry is a plain ``python p1...py --num_gpus N``, pretrain_trainer.py:21): the parent starts the ranks through
    torch.distributed.run without touching the GPU itself and relays rank 0's line; a WORLD_SIZE that contradicts --gpus still fails."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DIC_DIST_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '512',
           '--encounters', '4096', '--no-secondary', '--no-cpu-baseline', '--kernel-iters', '1']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['global_batch'] == 1024
    bad = subprocess.run(cmd, env=dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'), capture_output=True, text=True, timeout=280, cwd=root)
    assert bad.returncode != 0 and '--gpus 2 but WORLD_SIZE=1' in bad.stderr


def test_driver_with_num_gpus_starts_its_own_ranks(tmp_path):
    """``python -m ...p1_pretrain_main --num_gpus 2`` with NO launcher around it -- upstream's multi-GPU entry (p1_pretrain_main.py:27,118 ->
    DataParallel over N devices, pretrain_trainer.py:21).  Here N GPUs are N processes: the driver starts its own two ranks (over gloo they
    share the test GPU), both log their shard, rank 0 writes ONE complete feature dump; p3 the same way from that checkpoint; a launcher
    environment that contradicts --num_gpus is an error."""
    import subprocess
    sys.path.insert(0, ROOT)
    from deep_interpolation_clustering_amd import synthetic
    base = str(tmp_path)
    synthetic.write_split(base, 300, C=6, T=96, H=24.0, lam=50.0, G=4)
    run = os.path.join(base, 'run')
    os.makedirs(run)
    env = dict(os.environ, DIC_DIST_BACKEND='gloo', PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    common = [a for a in COMMON if a not in ('WARNING',)]
    common[common.index('--log-level') + 1:common.index('--log-level') + 1] = ['INFO']
    cmd = [sys.executable, '-m', 'deep_interpolation_clustering_amd.p1_pretrain_main', '--num_gpus', '2'] + common + \
          ['--mode', 'train', '--max_epochs', '2', '--loss', 'ae_mse']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=run)
    assert res.returncode == 0, res.stderr[-3000:]
    log = res.stdout + res.stderr
    assert 'starting 2 ranks' in log and 'rank 0 of 2' in log and 'rank 1 of 2' in log, log[-3000:]
    for cohort, n in (('training', 240), ('validation', 30), ('testing', 30)):
        d = np.load(os.path.join(run, f'Results/Pretrain/out_feat/ae_mse/{cohort}.npy'), allow_pickle=True).item()
        assert d['hidden'].shape == (n, 256) and len(set(d['encounter_id'].tolist())) == n and np.isfinite(d['hidden']).all()
    cmd3 = [sys.executable, '-m', 'deep_interpolation_clustering_amd.p3_clustering_main', '--num_gpus', '2'] + common + \
           ['--mode', 'train', '--max_epochs', '2', '--loss', 'ae_mse_kl', '--cluster_number', '4']
    res = subprocess.run(cmd3, env=env, capture_output=True, text=True, timeout=280, cwd=run)
    assert res.returncode == 0, res.stderr[-3000:]
    assert 'rank 1 of 2' in res.stdout + res.stderr
    d = np.load(os.path.join(run, 'Results/Clustering/out_feat/ae_mse/training.npy'), allow_pickle=True).item()
    assert d['hidden'].shape == (240, 256) and d['cluster_pred'].shape == (240, 4)
    bad = subprocess.run(cmd, env=dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'), capture_output=True, text=True, timeout=120, cwd=run)
    assert bad.returncode != 0 and '--num_gpus 2 but WORLD_SIZE=1' in bad.stderr


def test_sharded_paths_on_rccl_with_one_rank(tmp_path):
    """The box has one GPU, and RCCL wants one GPU per rank: so ONE rank joins a `nccl` process group (dist.init_from_env as on a real
    node: device_id, current device) with DIC_DIST_SINGLE_RANK=1, which makes that world count as sharded.  Every collective of the
    sharded joint step (loss sums, global BatchNorm moments and their backward, DEC column sums, the split gradient bucket), of the
    default fake-detection objective and of the sharded k-means then runs on RCCL, on device tensors, on the HIP streams the
    kernels use -- and has to reproduce the unsharded run (a one-rank sum is the identity)."""
    port = 29900 + (os.getpid() % 1000)
    for worker in (_run, _run_fake, _run_kmeans):
        mp.spawn(worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
        mp.spawn(worker, args=(1, port + 1, str(tmp_path), True), nprocs=1, join=True)
        port += 2
    a, b = torch.load(tmp_path / 'w1_r0.pt', weights_only=False), torch.load(tmp_path / 'w1x_r0.pt', weights_only=False)
    np.testing.assert_allclose(b['traj'][0, :2], a['traj'][0, :2], rtol=2e-6)
    np.testing.assert_allclose(b['traj'][:, :2], a['traj'][:, :2], rtol=2e-4)
    np.testing.assert_allclose(b['traj'][:, 3], a['traj'][:, 3], rtol=2e-3)
    np.testing.assert_allclose(b['bn_mean'].numpy(), a['bn_mean'].numpy(), rtol=1e-3, atol=1e-4)
    a, b = torch.load(tmp_path / 'f1_r0.pt', weights_only=False), torch.load(tmp_path / 'f1x_r0.pt', weights_only=False)
    for mode in ('train', 'eval'):
        for k, v in a[mode].items():
            np.testing.assert_allclose(b[mode][k], v, rtol=1e-5, atol=0, err_msg=f'{mode} {k}')
    np.testing.assert_allclose(b['gnorm'], a['gnorm'], rtol=1e-3)
    a, b = torch.load(tmp_path / 'km_w1_r0.pt', weights_only=False), torch.load(tmp_path / 'km_w1x_r0.pt', weights_only=False)
    for case in ('pp', 'empty'):
        assert np.array_equal(a[case][0], b[case][0]) and a[case][3] == b[case][3], case
        np.testing.assert_allclose(a[case][1], b[case][1], rtol=1e-5, atol=1e-6, err_msg=case)


def _run_graphed(rank, world, port, out):
    """The SHARDED step (every collective on RCCL: one rank, DIC_DIST_SINGLE_RANK=1) captured in a hipGraph against the same step
    launched eagerly."""
    sys.path.insert(0, ROOT)
    _env(rank, world, port, True)
    from deep_interpolation_clustering_amd import dist, synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    dist.init_from_env()
    _join(True)
    assert dist.graph_capturable()
    dev = torch.device('cuda', 0)
    coh = synthetic.make_cohort(512, seed=31)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    X, OB, LEN = (torch.tensor(a, device=dev) for a in (x_np, ob_np, n))
    res = {}
    for graphs in (False, True, 'capture_fails', 'no_global_rows'):
        torch.manual_seed(5)
        net = _pretrained(Net(_args(), dev).to(dev))
        net.train()
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), _args(), autocast_dtype=torch.bfloat16, use_graphs=bool(graphs))
        assert st.use_graphs == bool(graphs)
        if graphs == 'capture_fails':           # (round 6: a capture that fails is agreed on -- dist.all_agree on RCCL -- and every rank steps eagerly from there)
            def failing(run):
                raise RuntimeError('injected: capture failed')
            st._capture = failing
        traj = []
        for i in range(6):
            lo = (i % 2) * 256
            # the replay-or-capture decision of a SHARDED step is keyed on the global batch's row count (the loaders' samples carry it)
            losses, gnorm, _ = st.step(X[lo:lo + 256], OB[lo:lo + 256], None, LEN[lo:lo + 256], global_rows=None if graphs == 'no_global_rows' else 256 * world)
            traj.append([float(losses['loss'].detach()), float(losses['ae_mse'].detach()), float(losses['kl'].detach()), float(gnorm)])
        if graphs is True:
            assert len(st._graphs) == 1 and not st._sharded_capture_off
        elif graphs == 'capture_fails':
            assert len(st._graphs) == 0 and st._sharded_capture_off
        elif graphs == 'no_global_rows':
            assert len(st._graphs) == 0 and not st._sharded_capture_off       # nothing rank-invariant to key on: eager
        res[graphs] = np.array(traj)
    torch.save(res, os.path.join(out, 'graphed.pt'))
    _leave()


def test_sharded_step_is_hip_graph_capturable_on_rccl(tmp_path):
    """a strong-scaled batch of a few thousand encounters per rank is launch-bound, and Stepper refused to capture a
    sharded step.  With the collectives on RCCL (stream operations) an explicit use_graphs=True records them with the kernels: the
    replayed trajectory must follow the eager one (bf16 mode; same bar as the single-GPU graph test)."""
    mp.spawn(_run_graphed, args=(1, 29200 + (os.getpid() % 1000), str(tmp_path)), nprocs=1, join=True)
    res = torch.load(tmp_path / 'graphed.pt', weights_only=False)
    np.testing.assert_allclose(res[True], res[False], rtol=2e-3)
    assert np.isfinite(res[True]).all() and not np.allclose(res[True][0], res[True][-1])        # the replays do advance the parameters
    # a capture that failed (agreed through dist.all_agree on RCCL) and a step without a rank-invariant key both ARE the eager trajectory: the two
    # warm-up steps before the failed capture left no trace
    np.testing.assert_array_equal(res['capture_fails'], res[False])
    np.testing.assert_array_equal(res['no_global_rows'], res[False])


def test_bench_runs_on_rccl_with_one_rank():
    """bench.py under torch.distributed.run as the driver launches it, one rank, the process group on RCCL and the sharded code paths
    on (DIC_DIST_SINGLE_RANK=1): barriers, the max-over-ranks timing all-reduce and the lockstep trace steps all execute on `nccl`."""
    import json
    import subprocess
    env = dict(os.environ, DIC_DIST_SINGLE_RANK='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'DIC_DIST_BACKEND'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29577', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--batch', '512',
           '--encounters', '4096', '--no-secondary', '--no-cpu-baseline', '--kernel-iters', '1']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    assert 'process group: nccl' in res.stderr, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 1 and rec['value'] > 0 and set(rec) == {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                                                     'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'}
    side = json.load(open(os.path.join(ROOT, 'bench_secondary.json')))          # everything else goes to the side file
    assert side['headline']['value'] == rec['value'] and np.isfinite(side['whole_step']['final_loss']) and 'kernels' in side

"""The ragged, device-resident encounter store (ragged.RaggedStore, dic_*_store entry points) on the GPU.

The store path runs the SAME kernels on the SAME samples as the padded (B,4C,T) path -- only the addresses differ -- so every result
must be bit-identical to the dense entry points (which are pinned against the reference, tests/test_gpu_ops.py), for a SHUFFLED batch
read in place through the encounter index; the reference-generated fixtures are also run through the store entry points directly.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def cohort(n, C, T, lam, seed):
    from deep_interpolation_clustering_amd import synthetic
    coh = synthetic.make_cohort(n, C=C, T=T, H=24.0, lam=lam, G=4, seed=seed)
    if seed % 2:                                   # a few rows with a single observation, one channel at full length
        coh['padding_mask'][1, 0, 1:] = 0
        coh['feat'][1, 0, 1:] = 0
        coh['time_step'][1, 0, 1:] = 0
        coh['drop_mask'][1, 0, 1:] = 0
        coh['lengths'][1, 0] = 1
    x, ob, n_ = synthetic.stacked_batch(coh)
    return x, ob, n_


def make_store(x, C, dev):
    from deep_interpolation_clustering_amd.ragged import RaggedStore
    assert RaggedStore.fits(x, C)
    return RaggedStore(x, C, dev)


@pytest.mark.parametrize('C,T,R,lam', [(6, 96, 24, 50.0), (12, 288, 24, 200.0), (5, 40, 7, 12.0), (6, 354, 6, 60.0)])
@pytest.mark.parametrize('denoise', [False, True])
def test_store_path_equals_dense_path_bit_for_bit(C, T, R, lam, denoise):
    from deep_interpolation_clustering_amd import ops
    from deep_interpolation_clustering_amd.ragged import RaggedBatch
    dev = torch.device('cuda')
    x_np, ob_np, n_np = cohort(97, C, T, lam, seed=11)
    store = make_store(x_np, C, dev)
    g = torch.Generator().manual_seed(3)
    idx = torch.randperm(97, generator=g)[:61].to(dev)              # a shuffled batch, read in place
    rb = RaggedBatch(store, idx, denoise=denoise)
    X = torch.tensor(x_np, device=dev).index_select(0, idx)
    if denoise:
        X[:, :C] *= X[:, 3 * C:]
    OB = torch.tensor(ob_np, device=dev).index_select(0, idx)
    LEN = torch.tensor(n_np, device=dev).index_select(0, idx)
    assert torch.equal(rb.lengths, LEN) and torch.equal(rb.dense(), X) and torch.equal(rb.ob_dense(), OB)
    torch.manual_seed(0)
    grid = ops.ref_grid(24.0, R, dev)

    def params():
        sk = torch.rand(C, device=dev, requires_grad=True)
        ck = (torch.eye(C, device=dev) + 0.1 * torch.randn(C, C, device=dev)).requires_grad_(True)
        rk = torch.rand(C, device=dev, requires_grad=True)
        v = torch.randn(61, C, R, device=dev, requires_grad=True)
        return sk, ck, rk, v
    torch.manual_seed(1)
    pd = params()
    torch.manual_seed(1)
    ps = params()
    cot = torch.randn(61, R, 3 * C, device=dev)
    # k1, f32 output + parameter gradients (through the saved moments)
    od = ops.sci_cci(X, pd[0], pd[1], grid, LEN)
    os_ = ops.sci_cci(rb, ps[0], ps[1], grid)
    assert torch.equal(od, os_)
    (od * cot).sum().backward()
    (os_ * cot).sum().backward()
    assert torch.equal(pd[0].grad, ps[0].grad) and torch.equal(pd[1].grad, ps[1].grad)
    # k1, packed bf16 rows for the encoder LSTM
    if ops.packed_width(3 * C):
        assert torch.equal(ops.sci_cci_packed(X, pd[0], pd[1], grid, LEN), ops.sci_cci_packed(rb, ps[0], ps[1], grid))
    # k2 with the reconstruction loss riding along (training step) -- the store supplies time stamps AND observations
    plain = RaggedBatch(store, idx)                                  # (the target is never the denoised input)
    yd, md = ops.rbf_rec_loss(pd[3], X, pd[2], grid, LEN, OB)
    ys, ms = ops.rbf_rec_loss(ps[3], plain, ps[2], grid, plain.lengths, plain)
    assert torch.equal(md, ms)
    m = torch.arange(T, device=dev) < LEN[..., None]
    assert torch.equal(yd[m], ys[m])
    md.backward()
    ms.backward()
    assert torch.equal(pd[3].grad, ps[3].grad) and torch.equal(pd[2].grad, ps[2].grad)
    # k2 alone (evaluation / module API): zero-padded reconstruction, incoming gradient given
    for p in (pd, ps):
        p[2].grad = p[3].grad = None
    gy = torch.randn(61, C, T, device=dev)
    yd = ops.rbf_deinterp(pd[3], X, pd[2], grid, LEN)
    ys = ops.rbf_deinterp(ps[3], plain, ps[2], grid)
    assert torch.equal(yd, ys)
    (yd * gy).sum().backward()
    (ys * gy).sum().backward()
    assert torch.equal(pd[3].grad, ps[3].grad) and torch.equal(pd[2].grad, ps[2].grad)


@pytest.mark.parametrize('name', ['cfg_6_96_24_24', 'default_6_354_6_6', 'wide_12_288_24_24'])
def test_reference_fixtures_through_the_store_entry_points(name):
    """interp_* / rbf_* fixtures (outputs of the reference's own modules, oracle/make_golden.py) with the inputs packed into a store."""
    from deep_interpolation_clustering_amd import ops
    from deep_interpolation_clustering_amd.ragged import RaggedBatch
    dev = torch.device('cuda')
    g = dict(np.load(os.path.join(GOLDEN, f'interp_{name}.npz')))
    x = g['x'].copy()
    C = g['sci_kernel'].shape[0]
    x[:, :C] *= x[:, C:2 * C]                                        # (what SingleChannelInterp sees: value x mask)
    x[:, 3 * C:] *= x[:, C:2 * C]                                    # (the fixture's hold-out plane is random everywhere; it is never read)
    store = make_store(x, C, dev)
    rb = RaggedBatch(store, torch.arange(x.shape[0], device=dev))
    sk = torch.tensor(g['sci_kernel'], device=dev, requires_grad=True)
    ck = torch.tensor(g['cci_kernel'], device=dev, requires_grad=True)
    grid = ops.ref_grid(float(g['H']), int(g['R']), dev)
    out = ops.sci_cci(rb, sk, ck, grid)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g['cci_out'], rtol=3e-5, atol=3e-6)
    (out * torch.tensor(g['cot'], device=dev)).sum().backward()
    np.testing.assert_allclose(sk.grad.cpu().numpy(), g['g_sci'], rtol=2e-4, atol=2e-4 * np.abs(g['g_sci']).max())
    np.testing.assert_allclose(ck.grad.cpu().numpy(), g['g_cci'], rtol=2e-4, atol=2e-4 * np.abs(g['g_cci']).max())
    # ---- de-interpolation + reconstruction loss
    g = dict(np.load(os.path.join(GOLDEN, f'rbf_{name}.npz')))
    x = g['x'].copy()
    x[:, :C] = g['ob']                                               # the store's value plane doubles as the observations of rec_loss
    x[:, 3 * C:] *= x[:, C:2 * C]
    store = make_store(x, C, dev)
    rb = RaggedBatch(store, torch.arange(x.shape[0], device=dev))
    rk = torch.tensor(g['kernel'], device=dev, requires_grad=True)
    v = torch.tensor(g['v'], device=dev, requires_grad=True)
    y, mse = ops.rbf_rec_loss(v, rb, rk, grid, rb.lengths, rb)
    np.testing.assert_allclose(float(mse), float(g['loss']), rtol=1e-5)
    m = x[:, C:2 * C] > 0
    np.testing.assert_allclose(y.detach().cpu().numpy()[m], g['y'][m], rtol=3e-5, atol=3e-6)
    mse.backward()
    np.testing.assert_allclose(v.grad.cpu().numpy(), g['g_v'], rtol=2e-4, atol=2e-4 * np.abs(g['g_v']).max())
    np.testing.assert_allclose(rk.grad.cpu().numpy(), g['g_kernel'], rtol=2e-4, atol=2e-4 * np.abs(g['g_kernel']).max())


@pytest.mark.parametrize('mode', ['f32', 'bf16', 'bf16_graph'])
def test_joint_step_on_the_store_equals_the_padded_step(mode):
    """Three optimisation steps of the joint objective with the batch as a RaggedBatch against the same steps on padded tensors:
    identical losses, gradient norms and parameters (f32 parity mode, the bf16 fast mode, and replayed from a hipGraph)."""
    from types import SimpleNamespace
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.ragged import RaggedBatch
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=False, triple_margin=0.0, cluster_number=4, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
    dev = torch.device('cuda')
    x_np, ob_np, n_np = cohort(600, 6, 96, 50.0, seed=4)
    store = make_store(x_np, 6, dev)
    X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n_np, device=dev)
    order = torch.randperm(600, generator=torch.Generator().manual_seed(9)).to(dev)
    res = {}
    for ragged in (False, True):
        torch.manual_seed(4)
        net = Net(args, dev).to(dev)
        net.train()
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=None if mode == 'f32' else torch.bfloat16,
                     use_graphs=mode == 'bf16_graph')
        out = []
        for i in range(3):
            idx = order[i * 200:(i + 1) * 200]
            if ragged:
                losses, gnorm, _ = st.step(RaggedBatch(store, idx), None, None)
            else:
                losses, gnorm, _ = st.step(X.index_select(0, idx), OB.index_select(0, idx), None, LEN.index_select(0, idx))
            out.append([float(losses['loss'].detach()), float(losses['ae_mse'].detach()), float(losses['kl'].detach()), float(gnorm)])
        torch.cuda.synchronize()
        res[ragged] = (np.array(out), st.flat.flat.detach().clone())
    np.testing.assert_array_equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])


def test_device_loader_on_the_store_yields_what_the_padded_loader_yields(tmp_path):
    """DeviceLoader(ragged) against DeviceLoader(padded array) over one cohort: the same padded tensors where it rebuilds them (evaluation /
    dump passes), ragged-only samples for the plain training pass, and a fraction of the resident bytes."""
    from types import SimpleNamespace
    from deep_interpolation_clustering_amd import dataloader, synthetic
    from deep_interpolation_clustering_amd.dataloader import DataSet, DeviceLoader
    synthetic.write_split(str(tmp_path), 400, C=6, T=96, H=24.0, lam=50.0, G=4)
    old = dataloader.BASE_PATH
    dataloader.BASE_PATH = str(tmp_path)
    try:
        a = SimpleNamespace(hours_from_admission=24, scale=5.0, aux_tasks={}, fake_detection=False, aug_input=False, aug_std=0.1, num_variables=6)
        ds = DataSet(a, 'training')
        dev = torch.device('cuda')
        dl_r = DeviceLoader(ds, 64, False, dev, seed=1, shard=False)                 # 'auto' -> ragged store
        dl_d = DeviceLoader(ds, 64, False, dev, seed=1, shard=False, ragged=False)
        assert dl_r.store is not None and dl_r.data is None and dl_d.store is None
        assert dl_r.store.nbytes() < 0.35 * dl_d.data.numel() * 4
        for (sr, _), (sd, _) in zip(dl_r, dl_d):
            for k in ('ob', 'padding_mask', 'timestamp', 'ae_mask', 'lengths'):
                assert torch.equal(sr[k], sd[k]), k
            assert (sr['encounter_id'] == sd['encounter_id']).all()
            assert torch.equal(sr['ragged'].dense()[:, :6], sd['ob'] * sd['padding_mask'])
        full = torch.as_tensor(ds.feed_data, dtype=torch.float32, device=dev)
        assert torch.equal(dl_r.store.dense_rows(torch.arange(len(ds), device=dev)), full)     # the padded array, bit for bit
        train = DeviceLoader(ds, 64, True, dev, seed=1, shard=False)
        s, f = next(iter(train))
        assert f is s and set(s) == {'encounter_id', 'lengths', 'ragged', 'global_rows'} and s['ragged'].shape == (64, 24, 96)
        assert s['global_rows'] == 64          # rows of the GLOBAL batch (round 6: what a sharded Stepper keys its captured steps on)
    finally:
        dataloader.BASE_PATH = old


@pytest.mark.parametrize('C,T,R,lam', [(6, 96, 24, 50.0), (12, 288, 24, 200.0)])
def test_unsorted_and_duplicate_time_stamps_through_the_store(C, T, R, lam):
    """k1 takes each grid point's nearest sample by BISECTION when the store certifies sorted time stamps (RaggedStore.times_sorted) and by a
    pass over the row otherwise.  (a) a cohort whose rows are shuffled in time: the store must say so and the store path must
    equal the dense path bit for bit; (b) a sorted cohort with ties -- duplicated stamps, some of them exactly on a grid point: the
    bisection and the full pass (flag forced to 0) must agree bit for bit, both with the dense path."""
    from deep_interpolation_clustering_amd import ops
    from deep_interpolation_clustering_amd.ragged import RaggedBatch
    dev = torch.device('cuda')
    x_np, _, n_np = cohort(53, C, T, lam, seed=8)
    rng = np.random.default_rng(5)
    grid = ops.ref_grid(24.0, R, dev)
    sk = torch.rand(C, device=dev)
    ck = torch.eye(C, device=dev) + 0.1 * torch.randn(C, C, device=dev)
    idx = torch.randperm(53, generator=torch.Generator().manual_seed(1)).to(dev)

    def both_paths(x, expect_sorted):
        store = make_store(x, C, dev)
        assert store.times_sorted is expect_sorted
        rb = RaggedBatch(store, idx)
        X = torch.tensor(x, device=dev).index_select(0, idx)
        LEN = torch.tensor(n_np, device=dev).index_select(0, idx)
        dense = ops.sci_cci(X, sk, ck, grid, LEN)
        got = ops.sci_cci(rb, sk, ck, grid)
        assert torch.equal(dense, got)
        if ops.packed_width(3 * C):
            assert torch.equal(ops.sci_cci_packed(X, sk, ck, grid, LEN), ops.sci_cci_packed(rb, sk, ck, grid))
        return store, rb, got
    # (a) every row's observed prefix permuted in time (values and hold-out flags travel with their stamps)
    xs = x_np.copy()
    for b in range(xs.shape[0]):
        for c in range(C):
            n = int(n_np[b, c])
            p = rng.permutation(n)
            for plane in (0, 2, 3):
                xs[b, plane * C + c, :n] = xs[b, plane * C + c, :n][p]
    both_paths(xs, False)
    # (b) sorted with ties: every third stamp repeats its predecessor, a few sit exactly on grid points
    xt = x_np.copy()
    gridv = np.linspace(0, 24.0, R).astype(np.float32)
    for b in range(xt.shape[0]):
        for c in range(C):
            n = int(n_np[b, c])
            t = xt[b, 2 * C + c, :n].copy()
            t[2::3] = t[1:-1:3][:len(t[2::3])]
            if n > 4:
                t[n // 2] = gridv[np.abs(gridv - t[n // 2]).argmin()]
            xt[b, 2 * C + c, :n] = np.sort(t)
    store, rb, with_bisection = both_paths(xt, True)
    store.times_sorted = False                                        # the same rows through the full pass
    assert torch.equal(ops.sci_cci(rb, sk, ck, grid), with_bisection)


def test_device_built_cohort_store_equals_the_host_built_one_and_steps():
    """bench.py times BASELINE configs[3] on a cohort built ON the device (synthetic.device_cohort_store -> RaggedStore.from_device / concat: the padded planes of
    300 000 x 12 x 288 would be 16.6 GB).  The device-built store must equal the host constructor on the same planes field for field, and a joint step on a
    shuffled batch of it must equal the step on the padded batch bit for bit (same kernels, same samples)."""
    from types import SimpleNamespace

    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.ragged import RaggedBatch, RaggedStore
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    dev = torch.device('cuda')
    C, T, K = 12, 288, 16
    store, pheno = synthetic.device_cohort_store(300, C, T, 24.0, 200.0, K, 4, dev, chunk=128)      # three chunks, joined
    assert store.N == 300 and store.times_sorted and pheno.shape == (300,) and int(pheno.max()) < K
    x = store.dense_rows(torch.arange(300, device=dev), masked_values=True)                          # (300, 4C, T) planes rebuilt from the packed rows
    host = RaggedStore(x.cpu().numpy(), C, dev)
    for name in ('t_pk', 'v_pk', 'hold_pk', 'row_off', 'lengths', 'pad_value'):
        assert torch.equal(getattr(store, name), getattr(host, name)), name
    whole = RaggedStore.from_device(x, C)
    assert torch.equal(whole.t_pk, store.t_pk) and torch.equal(whole.row_off, store.row_off)
    args = SimpleNamespace(num_variables=C, num_timestamps=T, ref_points=24, hours_from_admission=24.0, dropout=0.0, aux_tasks={}, fake_detection=False,
                           triple_margin=0.0, cluster_number=K, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.}, aux_pos_weights={})
    idx = torch.randperm(300, device=dev, generator=torch.Generator(device=dev).manual_seed(2))[:192].to(torch.int32)
    res = []
    for ragged in (True, False):
        torch.manual_seed(3)
        net = Net(args, dev).to(dev)
        net.train()
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16)
        if ragged:
            losses, gnorm, _ = st.step(RaggedBatch(store, idx), None, None)
        else:
            xb = x.index_select(0, idx.to(torch.int64))
            losses, gnorm, _ = st.step(xb, xb[:, :C].contiguous(), None, store.lengths.index_select(0, idx.to(torch.int64)))
        res.append((float(losses['loss'].detach()), float(losses['ae_mse'].detach()), float(losses['kl'].detach()), float(gnorm)))
    assert res[0] == res[1] and all(np.isfinite(res[0])), res

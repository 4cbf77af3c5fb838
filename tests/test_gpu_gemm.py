"""Row-streaming MFMA products (csrc/dic_gemm.hip: dic_gemm_nt / dic_gemm_tn) against f64 products of the same operands: bf16 inputs (one
MFMA per product) and f32 inputs (the three-term bf16 split, 'x3'), tile tails in every dimension, strided row views, bias, the ReLU on load,
written and accumulated weight gradients -- and the f32 step in the 'x3' mode against the reference's own joint step."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _tol(dtype):
    # bf16 operands are exact inputs (the products are exact, f32 accumulation): only the accumulation order differs from f64.
    # f32 operands: the split drops lo.lo and rounds lo: ~2^-17 per product, a random walk over K
    return 2e-6 if dtype == torch.bfloat16 else 2e-5


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('M,N,K,bias,relu', [(24 * 200, 1024, 256, True, False), (33, 128, 256, True, True), (70001, 64, 1024, False, False),
                                               (1, 1024, 40, True, False), (5000, 36, 1024, False, False), (129, 200, 72, True, True),
                                               (4096, 1024, 64, False, False)])
def test_gemm_nt_matches_f64(dtype, M, N, K, bias, relu):
    from deep_interpolation_clustering_amd import ops
    torch.manual_seed(M + N + K)
    dev = torch.device('cuda')
    a = (torch.randn(M, K + 8, device=dev) * 0.7).to(dtype)[:, :K]                 # a strided row view
    w = (torch.randn(N, K, device=dev) * 0.1).to(dtype)
    b = torch.randn(N, device=dev) if bias else None
    for out_dtype in (torch.float32, torch.bfloat16):
        y = ops.gemm_nt(a, w, b, out_dtype=out_dtype, relu_a=relu)
        assert y.dtype == out_dtype and tuple(y.shape) == (M, N)
        ad = a.double().clamp_min(0) if relu else a.double()
        want = ad @ w.double().t() + (b.double() if bias else 0.0)
        err = float((y.double() - want).abs().max()) / float(want.abs().max())
        assert err <= (_tol(dtype) if out_dtype == torch.float32 else 5e-3), (dtype, out_dtype, err)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('M,N,K,kcols,accumulate', [(24 * 200, 512, 256, 256, False), (70001, 512, 128, 128, True), (31, 128, 256, 256, False),
                                                      (5000, 512, 40, 36, True), (4099, 1024, 64, 37, False), (24 * 4096, 128, 256, 256, False),
                                                      (1, 512, 24, 18, False)])
def test_gemm_tn_matches_f64(dtype, M, N, K, kcols, accumulate):
    from deep_interpolation_clustering_amd import ops
    torch.manual_seed(M + N + K)
    dev = torch.device('cuda')
    a = (torch.randn(M, 2 * N, device=dev) * 0.3).to(dtype)[:, N:]                 # the second direction's gate columns: a strided view
    x = (torch.randn(M + 7, K, device=dev) * 0.5).to(dtype)[7:]                     # row-shifted view (h_prev of the reverse direction)
    dst = torch.randn(N, kcols, device=dev)
    before = dst.clone()
    ops.gemm_tn_into(a, x, dst, kcols=kcols, accumulate=accumulate)
    want = a.double().t() @ x.double()[:, :kcols] + (before.double() if accumulate else 0.0)
    err = float((dst.double() - want).abs().max()) / float(want.abs().max())
    assert err <= 10 * _tol(dtype), (dtype, err)                                    # (f32 partial sums over up to 100 k rows before the f64 stage)
    again = torch.randn(N, kcols, device=dev) if not accumulate else before.clone()
    ops.gemm_tn_into(a, x, again, kcols=kcols, accumulate=accumulate)
    assert torch.equal(again, dst)                                                  # fixed-order reductions: run-to-run identical


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('M,K,kcols,K2,accumulate', [(24 * 200, 40, 36, 128, False), (70001, 256, 256, 128, True), (33, 24, 18, 128, False)])
def test_gemm_tn_two_products_from_one_pass(dtype, M, K, kcols, K2, accumulate):
    """dW_ih = dG^T.x and dW_hh = dG^T.h_prev of one LSTM direction from ONE pass over the gate gradients (dic_gemm_tn's second operand)."""
    from deep_interpolation_clustering_amd import ops
    torch.manual_seed(M + K)
    dev = torch.device('cuda')
    N = 512
    a = (torch.randn(M, 2 * N, device=dev) * 0.3).to(dtype)[:, :N]
    x = (torch.randn(M, K, device=dev) * 0.5).to(dtype)
    h = (torch.randn(M + 3, 2 * K2, device=dev) * 0.5).to(dtype)[3:, K2:]          # a row-shifted half-row view, as h_prev of the reverse direction
    d1, d2 = torch.randn(N, kcols, device=dev), torch.randn(N, K2, device=dev)
    b1, b2 = d1.clone(), d2.clone()
    ops.gemm_tn_into(a, x, d1, kcols=kcols, accumulate=accumulate, x2=h, dst2=d2)
    w1 = a.double().t() @ x.double()[:, :kcols] + (b1.double() if accumulate else 0.0)
    w2 = a.double().t() @ h.double() + (b2.double() if accumulate else 0.0)
    for got, want in ((d1, w1), (d2, w2)):
        assert float((got.double() - want).abs().max()) / float(want.abs().max()) <= 10 * _tol(dtype)


@pytest.mark.parametrize('n,nout,bias,relu', [(24 * 200, 1024, True, False), (33, 128, True, True), (70001, 1024, False, True), (1, 256, True, False),
                                             (4097, 128, False, False)])
def test_x3_row_proj_matches_f64(n, nout, bias, relu):
    """dic_x3_row_proj (resident split weights, 256-input projections of the x3 step) against the f64 product, and against dic_gemm_nt."""
    from deep_interpolation_clustering_amd import ops
    torch.manual_seed(n + nout)
    dev = torch.device('cuda')
    x = torch.randn(n, 256, device=dev) * 0.7
    w = torch.randn(nout, 256, device=dev) * 0.1
    b = torch.randn(nout, device=dev) if bias else None
    assert ops.x3_row_proj_ok(x, w)
    y = ops.x3_row_proj(x, w, b, relu_a=relu)
    xd = x.double().clamp_min(0) if relu else x.double()
    want = xd @ w.double().t() + (b.double() if bias else 0.0)
    scale = float(want.abs().max())
    assert float((y.double() - want).abs().max()) / scale <= 2e-5
    y2 = ops.gemm_nt(x, w, b, relu_a=relu)
    assert float((y - y2).abs().max()) / scale <= 4e-6              # the same three products per term, another accumulation order


def test_split_products_are_f32_grade_where_bf16_is_not():
    """The point of the three-term split: f32 operands with a full mantissa -- bf16 rounding of the same operands is off by ~1e-3."""
    from deep_interpolation_clustering_amd import ops
    torch.manual_seed(0)
    dev = torch.device('cuda')
    a, w = torch.randn(4096, 256, device=dev), torch.randn(512, 256, device=dev) * 0.1
    want = a.double() @ w.double().t()
    scale = float(want.abs().max())
    e3 = float((ops.gemm_nt(a, w).double() - want).abs().max()) / scale
    e1 = float((ops.gemm_nt(a.bfloat16(), w.bfloat16(), out_dtype=torch.float32).double() - want).abs().max()) / scale
    ef = float(((a @ w.t()).double() - want).abs().max()) / scale
    assert e3 < 1e-5 and e1 > 20 * e3, (e3, e1, ef)


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def _args(C, T, K):
    return SimpleNamespace(num_variables=C, num_timestamps=T, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=False, triple_margin=0.0, cluster_number=K, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.}, aux_pos_weights={})


@pytest.mark.parametrize('case', ['cfg_K4', 'cfg_K8', 'wide_K16'])
def test_f32_step_on_split_products_matches_reference_step(case):
    """The f32 step with every dense product on the bf16 matrix cores as a three-term split (Stepper(precision='x3'): dic_gemm_nt / dic_gemm_tn,
    the split recurrence): loss / ae_mse / kl of the reference's own joint step at rtol 1e-5 with no floor at the configured shape (K = 4, 8;
    measured ~1e-6), 2e-5 at BASELINE configs[3]'s (the oracle's emulation of the split: kl 7e-6 there); argmax of q exact."""
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    dev = torch.device('cuda')
    if case == 'wide_K16':
        g = _load('netstep_wide_K16.npz')
        sd = {k[4:]: torch.tensor(v) for k, v in g.items() if k.startswith('sd0/')}
        C, T, K, rtol = 12, 288, 16, 2e-5
    else:
        g = _load(f'netstep_{case}.npz')
        t = _load('traj_cfg1.npz')
        sd = {k[5:]: torch.tensor(v) for k, v in t.items() if k.startswith('p1sd/')}
        sd['cluster_assignment.cluster_centers'] = torch.tensor(g['centers'])
        C, T, K, rtol = 6, 96, int(g['K']), 1e-5
    args = _args(C, T, K)
    net = Net(args, dev).to(dev)
    net.load_state_dict(sd, strict=True)
    net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, precision='x3')
    x, ob = torch.tensor(g['x'], device=dev), torch.tensor(g['ob'], device=dev)
    lens = x[:, C:2 * C].sum(-1).to(torch.int32)
    losses, gnorm, z = st.step(x, ob, None, lens)
    dev_rel = {k: abs(float(losses[k]) - float(g['loss_' + k])) / abs(float(g['loss_' + k])) for k in ('loss', 'ae_mse', 'kl')}
    print('x3 deviation', case, dev_rel)
    for k in ('loss', 'ae_mse', 'kl'):
        np.testing.assert_allclose(float(losses[k]), float(g['loss_' + k]), rtol=rtol, atol=0, err_msg=k)
    np.testing.assert_allclose(float(gnorm), float(g['gnorm']), rtol=1e-4)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g['z'], rtol=1e-3, atol=2e-4 if case == 'wide_K16' else 3e-5)
    q = net.cluster_assignment(z.detach())
    assert (q.argmax(1).cpu().numpy() == g['q'].argmax(1)).all()
    for k, v in net.state_dict().items():
        if 'sd1n/' + k in g:
            np.testing.assert_allclose(np.linalg.norm(v.detach().cpu().numpy().astype(np.float64)), float(g['sd1n/' + k]), rtol=1e-4, err_msg=k)


def test_x3_step_tracks_the_exact_f32_step_at_a_large_batch():
    """The x3 mode where its kernels run many tiles (B = 8192: 196 608 rows through dic_x3_row_proj / dic_gemm_nt / dic_gemm_tn's row chunks, 256 tiles of
    the x3 recurrence, the ReLU between the LSTMs applied on load, ragged store input): three optimisation steps against the exact-f32 mode from the
    same state -- first-step losses to 2e-6, the trajectories to 1e-4 (two f32-grade implementations of one Adam loop)."""
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.ragged import RaggedBatch, RaggedStore
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    dev = torch.device('cuda')
    B = 8192
    coh = synthetic.make_cohort(B, G=4, seed=35)
    x_np, _, _ = synthetic.stacked_batch(coh)
    store = RaggedStore(x_np, 6, dev)
    rb = RaggedBatch(store, torch.randperm(B, device=dev, generator=torch.Generator(device=dev).manual_seed(3)))
    g = torch.tensor(coh['phenotype'].astype(np.int64), device=dev)[rb.idx.long()]
    args = _args(6, 96, 4)
    traj = {}
    for mode in ('exact', 'x3'):
        torch.manual_seed(4)
        net = Net(args, dev).to(dev)
        net.eval()
        with torch.no_grad():            # centroids on the latents' clusters: the p3 regime
            z = net(rb)[0]
            net.init_cluster_center(torch.stack([z[g == j].mean(0) for j in range(4)]))
        net.train()
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, precision=mode)
        out = []
        for _ in range(3):
            losses, gnorm, _ = st.step(rb, None, None)
            out.append([float(losses[k].detach()) for k in ('loss', 'ae_mse', 'kl')] + [float(gnorm)])
        traj[mode] = np.array(out)
    assert traj['exact'][0, 2] > 0.01
    np.testing.assert_allclose(traj['x3'][0, :3], traj['exact'][0, :3], rtol=2e-6)
    np.testing.assert_allclose(traj['x3'][0, 3], traj['exact'][0, 3], rtol=2e-5)
    np.testing.assert_allclose(traj['x3'], traj['exact'], rtol=1e-4)

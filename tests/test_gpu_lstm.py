"""Fused bi-LSTM recurrence (csrc/dic_lstm.hip + lstm.py, the bf16 fast path) against
(1) an f32 emulation that rounds to bf16 at exactly the kernel's rounding points, and
(2) torch.nn.LSTM in f32 (loose: bf16 operands)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H = 128


def rb(x):
    return x.to(torch.bfloat16).float()


def emulate(x, lstm, h0=None, c0=None, gx_bf16=False):
    """Differentiable f32 torch loop with the kernel's bf16 rounding points: X, W_ih, W_hh, gx, the h fed back."""
    R, B, I = x.shape
    outs, hn, cn = [None, None], [], []
    for d, sfx in enumerate(('', '_reverse')):
        w_ih, w_hh = getattr(lstm, 'weight_ih_l0' + sfx), getattr(lstm, 'weight_hh_l0' + sfx)
        bias = getattr(lstm, 'bias_ih_l0' + sfx) + getattr(lstm, 'bias_hh_l0' + sfx)
        gx = (rb(x).reshape(R * B, I) @ rb(w_ih).t() + rb(bias)).reshape(R, B, 4 * H)
        if I >= 64 or gx_bf16:      # separate projection kernel: gx itself is stored in bf16; I < 64 is projected in-kernel (f32) by the 64-row kernels
            gx = rb(gx)
        h = torch.zeros(B, H, device=x.device) if h0 is None else h0[d]
        c = torch.zeros(B, H, device=x.device) if c0 is None else c0[d]
        seq = [None] * R
        for t in (range(R) if d == 0 else range(R - 1, -1, -1)):
            g = gx[t] + rb(h) @ rb(w_hh).t()
            i, f, gg, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H]), torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:])
            c = f * c + i * gg
            h = o * torch.tanh(c)
            seq[t] = h
        outs[d] = torch.stack(seq)
        hn.append(h)
        cn.append(c)
    return torch.cat(outs, dim=-1), torch.stack(hn), torch.stack(cn)


@pytest.fixture(params=['pipelined64', 'tile32'])
def kernel_family(request, monkeypatch):
    """Both bf16 recurrence kernel families on every case: the 64-row software-pipelined kernels (csrc/dic_lstm.hip, large batches)
    and the one-tile-per-workgroup kernels (csrc/dic_lstm32.hip, batches up to lstm.SMALL_BATCH)."""
    from deep_interpolation_clustering_amd import lstm as L
    monkeypatch.setattr(L, 'SMALL_BATCH', 0 if request.param == 'pipelined64' else 1 << 30)
    return request.param


@pytest.mark.parametrize('R,B,I,init', [(24, 200, 18, False), (24, 96, 256, True), (6, 64, 18, False), (5, 1, 256, True), (3, 130, 40, True),
                                          (7, 70, 31, True), (4, 33, 32, False), (2, 5, 1, False), (24, 200, 36, True), (4, 130, 63, False),
                                          (3, 70, 80, True)])
def test_fused_bilstm_matches_emulation(R, B, I, init, kernel_family):
    from deep_interpolation_clustering_amd import lstm as L
    torch.manual_seed(R * 1000 + B)
    dev = torch.device('cuda')
    net = torch.nn.LSTM(I, H, num_layers=1, bidirectional=True).to(dev)
    x = torch.randn(R, B, I, device=dev) * (1.0 if I < 100 else 0.5)
    h0 = (torch.randn(2, B, H, device=dev) * 0.5).requires_grad_() if init else None
    c0 = (torch.randn(2, B, H, device=dev) * 0.5).requires_grad_() if init else None
    x1 = x.clone().requires_grad_()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        assert L.fused_available(x1, net)
        out, (hn, cn) = L.bilstm(x1, net, h0, c0)
    assert out.dtype == torch.bfloat16 and hn.dtype == torch.float32
    go = torch.randn(R, B, 2 * H, device=dev)
    ghn, gcn = torch.randn(2, B, H, device=dev), torch.randn(2, B, H, device=dev)
    ((out.float() * rb(go)).sum() + (hn * ghn).sum() + (cn * gcn).sum()).backward()
    got = {k: p.grad.clone() for k, p in net.named_parameters()}
    gx1 = x1.grad.clone()
    gh0 = None if not init else (h0.grad.clone(), c0.grad.clone())
    net.zero_grad()
    if init:
        h0.grad = c0.grad = None

    x2 = x.clone().requires_grad_()
    eo, ehn, ecn = emulate(x2, net, h0, c0, gx_bf16=kernel_family == 'tile32' and L.packed_width(I) == 0)      # (packed rows: projected in-kernel by both families)
    ((eo * rb(go)).sum() + (ehn * ghn).sum() + (ecn * gcn).sum()).backward()
    # forward: same rounding points -> differences only from accumulation order / fast sigmoid
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), eo.detach().cpu().numpy(), rtol=2e-2, atol=6e-3)
    np.testing.assert_allclose(hn.detach().cpu().numpy(), ehn.detach().cpu().numpy(), rtol=1e-2, atol=4e-3)
    np.testing.assert_allclose(cn.detach().cpu().numpy(), ecn.detach().cpu().numpy(), rtol=1e-2, atol=6e-3)

    def close(a, b, name, tol=3e-2):
        a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
        err = np.abs(a - b).max() / (np.abs(b).max() + 1e-12)
        assert err < tol, f'{name}: max err {err:.3e} of the max magnitude'
    close(gx1, x2.grad, 'dx')
    for k, p in net.named_parameters():
        close(got[k], p.grad, k)
    if init:
        close(gh0[0], h0.grad, 'dh0')
        close(gh0[1], c0.grad, 'dc0')


@pytest.mark.parametrize('sixteen', ['0', '1'])
@pytest.mark.parametrize('R,B,I,init', [(24, 256, 18, False), (5, 70, 36, True), (3, 33, 63, True)])
def test_small_batch_in_kernel_projection_equals_the_64_row_kernel(R, B, I, init, sixteen, monkeypatch):
    """dic_lstm_rec_fwd_proj (32-row tiles, the encoder's input projection inside the recurrence kernel: round 4) against dic_lstm_fwd_proj (the 64-row
    pipelined kernel): the same products in the same order, the same gate math -- outputs, final states and saved-state-driven gradients bit for bit.
    Both small-batch tilings: 32 rows per workgroup (DIC_REC_SIXTEEN=0) and the 16-row kernels that serve batches up to 2048."""
    from deep_interpolation_clustering_amd import lstm as L
    monkeypatch.setenv('DIC_REC_SIXTEEN', sixteen)
    torch.manual_seed(R * 1000 + B + I)
    dev = torch.device('cuda')
    net = torch.nn.LSTM(I, H, num_layers=1, bidirectional=True).to(dev)
    x = torch.randn(R, B, I, device=dev)
    h0 = (torch.randn(2, B, H, device=dev) * 0.5) if init else None
    c0 = (torch.randn(2, B, H, device=dev) * 0.5) if init else None
    go = rb(torch.randn(R, B, 2 * H, device=dev))
    res = {}
    for fam, small in (('tile32', 1 << 30), ('pipelined64', 0)):
        monkeypatch.setattr(L, 'SMALL_BATCH', small)
        net.zero_grad()
        xi = x.clone().requires_grad_()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out, (hn, cn) = L.bilstm(xi, net, h0, c0)
        ((out.float() * go).sum() + hn.sum() + (cn * 0.5).sum()).backward()
        res[fam] = (out.detach().clone(), hn.detach().clone(), cn.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in net.parameters()])
    a_, b_ = res['tile32'], res['pipelined64']
    assert torch.equal(a_[0], b_[0]) and torch.equal(a_[1], b_[1]) and torch.equal(a_[2], b_[2])
    torch.testing.assert_close(a_[3], b_[3], rtol=2e-2, atol=2e-3)          # (the two families' backward kernels and dX paths differ in summation order)
    for ga, gb in zip(a_[4], b_[4]):
        torch.testing.assert_close(ga, gb, rtol=2e-2, atol=2e-2 * float(gb.abs().max()))


def test_fused_bilstm_close_to_f32_lstm(kernel_family):
    from deep_interpolation_clustering_amd import lstm as L
    torch.manual_seed(7)
    dev = torch.device('cuda')
    R, B, I = 24, 300, 18
    net = torch.nn.LSTM(I, H, num_layers=1, bidirectional=True).to(dev)
    x = torch.randn(R, B, I, device=dev)
    ref, (rh, rc) = net(x)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out, (hn, cn) = L.bilstm(x, net)
    assert float((out.float() - ref).abs().max()) < 3e-2
    assert float((hn - rh).abs().max()) < 3e-2 and float((cn - rc).abs().max()) < 5e-2


def test_joint_step_bf16_fused_tracks_f32():
    """Whole joint step under bf16 autocast with the fused LSTM vs the f32 step: losses agree to bf16 accuracy."""
    from types import SimpleNamespace
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=False, triple_margin=0.0, cluster_number=4, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
    dev = torch.device('cuda')
    coh = synthetic.make_cohort(512, seed=3)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    x, ob, lens = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
    res = {}
    for mode in ('f32', 'bf16'):
        torch.manual_seed(11)
        net = Net(args, dev).to(dev)
        net.train()
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args,
                     autocast_dtype=torch.bfloat16 if mode == 'bf16' else None)
        traj = []
        for _ in range(5):
            losses, gnorm, _ = st.step(x, ob, None, lens)
            traj.append([float(losses['loss'].detach()), float(losses['ae_mse'].detach()), float(gnorm)])
        res[mode] = np.array(traj)
    assert np.isfinite(res['bf16']).all()
    np.testing.assert_allclose(res['bf16'][:, 0], res['f32'][:, 0], rtol=3e-2)
    assert res['bf16'][-1, 0] < res['bf16'][0, 0]           # it trains


@pytest.mark.parametrize('N,C', [(5000, 6), (257, 12), (3, 1), (70000, 16)])
def test_head_linear_matches_torch(N, C):
    """CompressFC's small-output Linear as a streaming kernel (csrc/dic_head.hip) vs torch on the same bf16 input."""
    from deep_interpolation_clustering_amd import ops
    torch.manual_seed(N)
    dev = torch.device('cuda')
    lin = torch.nn.Linear(128, C).to(dev)
    h = (torch.randn(N, 128, device=dev)).to(torch.bfloat16)
    h1 = h.clone().requires_grad_()
    v = ops.head_linear(h1, lin.weight, lin.bias)
    cot = torch.randn(N, C, device=dev)
    (v * cot).sum().backward()
    got = (h1.grad.float().clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    lin.zero_grad()
    h2 = h.float().clone().requires_grad_()
    ref = torch.nn.functional.linear(h2, lin.weight, lin.bias)
    (ref * cot).sum().backward()
    np.testing.assert_allclose(v.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[0].cpu().numpy(), h2.grad.cpu().numpy(), rtol=1e-2, atol=1e-2 * float(h2.grad.abs().max()))   # dh is bf16
    np.testing.assert_allclose(got[1].cpu().numpy(), lin.weight.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(lin.weight.grad.abs().max()))
    np.testing.assert_allclose(got[2].cpu().numpy(), lin.bias.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(lin.bias.grad.abs().max()))


@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('N,C,training', [(1, 6, False), (77, 6, True), (4099, 1, True), (20000, 8, True), (20000, 6, False)])
def test_bn_relu_head_matches_torch(N, C, training, relu):
    """Fused BatchNorm1d -> ReLU -> Linear tail of CompressFC (csrc/dic_bnhead.hip, rbf.py:116-123) vs the torch modules in
    f32 on the same bf16 input: output, running statistics, and all five gradients."""
    import copy
    from deep_interpolation_clustering_amd import ops
    torch.manual_seed(N + C)
    dev = torch.device('cuda')
    bn = torch.nn.BatchNorm1d(128).to(dev)
    lin = torch.nn.Linear(128, C).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.2)
        bn.running_var.uniform_(0.5, 2.0)
    bn.train(training)
    bn_ref, lin_ref = copy.deepcopy(bn), copy.deepcopy(lin)
    z = (torch.randn(N, 128, device=dev) * 1.5 + 0.3).to(torch.bfloat16)
    cot = torch.randn(N, C, device=dev)
    z1 = z.clone().requires_grad_()
    v = ops.bn_relu_head(z1, bn, lin, relu=relu)
    (v * cot).sum().backward()
    z2 = z.float().clone().requires_grad_()
    ref = lin_ref(torch.relu(bn_ref(z2)) if relu else bn_ref(z2))
    (ref * cot).sum().backward()

    def close(a, b, tol):
        a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=tol, atol=tol * max(float(np.abs(b).max()), 1e-6))
    close(v, ref, 2e-5)
    close(bn.running_mean, bn_ref.running_mean, 1e-5)
    close(bn.running_var, bn_ref.running_var, 1e-5)
    assert int(bn.num_batches_tracked) == int(bn_ref.num_batches_tracked)
    close(z1.grad, z2.grad, 1e-2)                       # dz is rounded to bf16
    close(lin.weight.grad, lin_ref.weight.grad, 2e-4)
    close(lin.bias.grad, lin_ref.bias.grad, 2e-4)
    close(bn.weight.grad, bn_ref.weight.grad, 2e-4)
    close(bn.bias.grad, bn_ref.bias.grad, 2e-4)


def test_compress_fc_fused_tail_matches_module_path():
    """CompressFC under bf16 autocast takes the fused tail; with dropout active it must not (rbf.py:111-125)."""
    from deep_interpolation_clustering_amd.rbf import CompressFC
    torch.manual_seed(5)
    dev = torch.device('cuda')
    fc = CompressFC(256, 6, 0.0).to(dev).train()
    x = torch.randn(3000, 256, device=dev)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        got = fc(x)
    ref = fc.model(x)                                    # f32 module path (updates running stats a second time: irrelevant here)
    np.testing.assert_allclose(got.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=0, atol=3e-2)
    assert got.dtype == torch.float32


def test_aux_head_fast_path_matches_module_path():
    """The auxiliary / fake-detection heads (Linear -> BatchNorm -> Dropout -> Linear [-> LogSoftmax], clustering_interp.py:43-87)
    take the streaming BatchNorm+Linear kernels under bf16 autocast: same outputs and gradients as the module path."""
    import copy
    from deep_interpolation_clustering_amd._net_common import FakeDetFc
    torch.manual_seed(9)
    dev = torch.device('cuda')
    head = FakeDetFc(256, 2, 0.0).to(dev).train()
    ref_head = copy.deepcopy(head)
    x = torch.randn(5000, 256, device=dev)
    cot = torch.randn(5000, 2, device=dev)
    x1 = x.clone().requires_grad_()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        got = head(x1)
    (got * cot).sum().backward()
    x2 = x.clone().requires_grad_()
    ref = ref_head.model(x2)                              # f32 module path
    (ref * cot).sum().backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=0, atol=3e-2)
    np.testing.assert_allclose(np.exp(got.detach().cpu().numpy()).sum(1), 1.0, rtol=1e-5)          # LogSoftmax tail applied
    g1, g2 = x1.grad.cpu().numpy(), x2.grad.cpu().numpy()
    assert np.abs(g1 - g2).max() < 3e-2 * np.abs(g2).max()
    for (k, p), (_, q) in zip(head.named_parameters(), ref_head.named_parameters()):
        if k == 'model.0.bias':
            continue                                      # identically zero in front of a training-mode BatchNorm (noise upstream)
        a, b = p.grad.cpu().numpy(), q.grad.cpu().numpy()
        assert np.abs(a - b).max() < 3e-2 * max(np.abs(b).max(), 1e-6), k


def _host_keep_mask(seed, counter, n_rows, p):
    """NumPy restatement of the in-kernel dropout decision (csrc/dic_bnhead.hip drop_factors): murmur3's 64-bit finaliser of
    key + (row * 64 + column pair); low word -> even column, high word -> odd column; keep iff word >= p * 2^32."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over='ignore'):
        key = np.uint64(seed) ^ (np.uint64(counter) * np.uint64(0x9E3779B97F4A7C15))
        idx = (np.arange(n_rows, dtype=np.uint64)[:, None] * np.uint64(64) + np.arange(64, dtype=np.uint64)[None, :])
        x = (key + idx) & M
        x ^= x >> np.uint64(33); x = (x * np.uint64(0xff51afd7ed558ccd)) & M
        x ^= x >> np.uint64(33); x = (x * np.uint64(0xc4ceb9fe1a85ec53)) & M
        x ^= x >> np.uint64(33)
    thresh = np.uint64(min(np.float32(p) * np.float32(4294967296.0), np.float32(4294967040.0)))
    lo, hi = x & np.uint64(0xFFFFFFFF), x >> np.uint64(32)
    keep = np.empty((n_rows, 128), dtype=bool)
    keep[:, 0::2], keep[:, 1::2] = lo >= thresh, hi >= thresh
    return keep


@pytest.mark.parametrize('relu', [True, False])
def test_bn_head_dropout_matches_torch_with_the_same_mask(relu):
    """nn.Dropout between the activation and the Linear, drawn inside the kernels: the mask is re-derived on the host from the
    (seed, counter) the call used, and the fused forward / backward must equal torch with exactly that mask."""
    import copy
    from deep_interpolation_clustering_amd import ops
    torch.manual_seed(4)
    dev = torch.device('cuda')
    N_, C, p = 6000, 6, 0.3
    bn = torch.nn.BatchNorm1d(128).to(dev).train()
    lin = torch.nn.Linear(128, C).to(dev)
    drop = torch.nn.Dropout(p).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
    bn_ref, lin_ref = copy.deepcopy(bn), copy.deepcopy(lin)
    z = (torch.randn(N_, 128, device=dev) * 1.5 + 0.3).to(torch.bfloat16)
    cot = torch.randn(N_, C, device=dev)
    z1 = z.clone().requires_grad_()
    v = ops.bn_relu_head(z1, bn, lin, relu=relu, dropout=drop)
    seed, counter = (int(t) for t in ops._DROP_STATE[(z.device.type, z.device.index)].tolist())      # the state this call used
    (v * cot).sum().backward()
    keep = torch.tensor(_host_keep_mask(seed, counter, N_, p), device=dev)
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.01
    z2 = z.float().clone().requires_grad_()
    a = bn_ref(z2)
    a = torch.relu(a) if relu else a
    ref = lin_ref(a * keep / (1 - p))
    (ref * cot).sum().backward()

    def close(x, y, tol):
        x, y = x.detach().float().cpu().numpy(), y.detach().float().cpu().numpy()
        np.testing.assert_allclose(x, y, rtol=tol, atol=tol * max(float(np.abs(y).max()), 1e-6))
    close(v, ref, 2e-5)
    close(z1.grad, z2.grad, 1e-2)
    close(lin.weight.grad, lin_ref.weight.grad, 2e-4)
    close(lin.bias.grad, lin_ref.bias.grad, 2e-4)
    close(bn.weight.grad, bn_ref.weight.grad, 2e-4)
    close(bn.bias.grad, bn_ref.bias.grad, 2e-4)
    # a second call draws a different mask; eval mode draws none
    v2 = ops.bn_relu_head(z.clone(), bn, lin, relu=relu, dropout=drop)
    assert not torch.equal(v2, v.detach())
    drop.eval(); bn.eval()
    e1, e2 = ops.bn_relu_head(z.clone(), bn, lin, relu=relu, dropout=drop), ops.bn_relu_head(z.clone(), bn, lin, relu=relu, dropout=drop)
    assert torch.equal(e1, e2)


@pytest.mark.parametrize('n', [8192 * 3, 8192 * 5 + 1000, 70000, 1000])
def test_splitk_tn_matches_plain_product(n):
    """a^T b over row chunks (+ tail) == the plain product, for row counts that are and are not multiples of the chunk."""
    from deep_interpolation_clustering_amd import ops
    torch.manual_seed(n)
    a = (torch.randn(n, 96, device='cuda') * 0.1).to(torch.bfloat16)
    b = torch.randn(n, 40, device='cuda').to(torch.bfloat16)
    ref = a.double().t() @ b.double()
    got = ops.splitk_tn(a, b)
    assert got.dtype == torch.float32
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=6e-3 * float(ref.abs().max()))
    s = ops.splitk_tn(a[100:, :64], b[:-100, 8:])              # strided views, as the per-direction dW_hh uses them
    np.testing.assert_allclose(s.cpu().numpy(), (a[100:, :64].double().t() @ b[:-100, 8:].double()).cpu().numpy(), rtol=0,
                               atol=6e-3 * float(ref.abs().max()))


# ------------------------------------------------------------------ parameter-side kernels (csrc/dic_lstmgrad.hip)
def _lstm_params(net):
    from deep_interpolation_clustering_amd import lstm as L
    return [getattr(net, n) for n in L.PARAM_NAMES]


@pytest.mark.parametrize('I', [18, 256, 40, 1, 36, 80])
def test_lstm_pack_matches_torch(I):
    """dic_lstm_pack: the eight nn.LSTM parameters -> the bf16 operands, against the torch stack / add / cast / pad sequence."""
    from deep_interpolation_clustering_amd import _native as N
    from deep_interpolation_clustering_amd import lstm as L
    torch.manual_seed(I)
    dev = torch.device('cuda')
    net = torch.nn.LSTM(I, H, num_layers=1, bidirectional=True).to(dev)
    proj = L.packed_width(I) > 0
    Ip = L.packed_width(I) if proj else (I + 15) // 16 * 16
    bf = torch.bfloat16
    wih = torch.full((8 * H, Ip), 7.0, device=dev, dtype=bf)
    whh, whh_t = torch.empty((2, 4 * H, H), device=dev, dtype=bf), torch.empty((2, H, 4 * H), device=dev, dtype=bf)
    bias = torch.empty(8 * H, device=dev, dtype=bf)
    ps = [p.detach() for p in _lstm_params(net)]
    wih_t = torch.full((Ip, 8 * H), 7.0, device=dev, dtype=bf)
    N.check(N.lib().dic_lstm_pack(N.DTYPE_BF16, N.ptr_array(ps), H, I, Ip, int(proj), N.ptr(wih), N.ptr(whh), N.ptr(whh_t), N.ptr(bias), N.ptr(wih_t),
                                  N.stream_of(wih)), 'dic_lstm_pack')
    w_ih = torch.stack([net.weight_ih_l0, net.weight_ih_l0_reverse]).detach()
    w_hh = torch.stack([net.weight_hh_l0, net.weight_hh_l0_reverse]).detach()
    b = torch.stack([net.bias_ih_l0 + net.bias_hh_l0, net.bias_ih_l0_reverse + net.bias_hh_l0_reverse]).detach()
    want = torch.zeros((8 * H, Ip), device=dev)
    want[:, :I] = w_ih.reshape(8 * H, I)
    if proj:
        want[:, I] = b.reshape(8 * H)
    assert torch.equal(wih, want.to(bf)) and torch.equal(wih_t, want.to(bf).t().contiguous())
    assert torch.equal(whh, w_hh.to(bf))
    assert torch.equal(whh_t, w_hh.transpose(1, 2).contiguous().to(bf))
    assert torch.equal(bias, b.reshape(8 * H).to(bf))
    # f32 outputs: the parameters themselves, re-arranged
    wih32, whh32, bias32 = torch.empty((8 * H, I), device=dev), torch.empty((2, 4 * H, H), device=dev), torch.empty(8 * H, device=dev)
    N.check(N.lib().dic_lstm_pack(N.DTYPE_F32, N.ptr_array(ps), H, I, I, 0, N.ptr(wih32), N.ptr(whh32), None, N.ptr(bias32), None, N.stream_of(wih)),
            'dic_lstm_pack')
    assert torch.equal(wih32, w_ih.reshape(8 * H, I)) and torch.equal(whh32, w_hh) and torch.equal(bias32, b.reshape(8 * H))


@pytest.mark.parametrize('R,B,I,init,accumulate', [(24, 200, 18, False, False), (3, 64, 18, True, True), (5, 130, 31, True, False),
                                                     (1, 37, 6, False, True), (2, 4099, 18, True, False), (3, 11, 18, True, True),
                                                     (24, 200, 36, True, False), (2, 4099, 36, False, True), (3, 11, 63, True, False),
                                                     (5, 130, 32, False, False)])
def test_lstm_dw_matches_matmul(R, B, I, init, accumulate):
    """dic_lstm_dw (one pass over dG, MFMA with transposed LDS reads) against f64 products of the same bf16 operands:
    dW_hh[d] = sum_t dG_t[d]^T h_prev_t[d], dW_ih[d] = sum_t dG_t[d]^T x_t; written or accumulated into the eight gradients."""
    from deep_interpolation_clustering_amd import _native as N
    torch.manual_seed(R * 100 + B)
    dev = torch.device('cuda')
    bf = torch.bfloat16
    dg = (torch.randn(R, B, 2, 4 * H, device=dev) * 0.3).to(bf)
    out = (torch.randn(R, B, 2 * H, device=dev) * 0.5).to(bf)
    XW = 32 if I < 32 else 64                                        # packed row width: 3C = 18 -> 32, 3C = 36 (BASELINE configs[3]) -> 64
    x = torch.zeros(R, B, XW, device=dev)
    x[..., :I] = torch.randn(R, B, I, device=dev)
    x[..., I] = 1.0
    x = x.to(bf)
    h0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    grads = [torch.randn(4 * H, I, device=dev), torch.randn(4 * H, H, device=dev), torch.randn(4 * H, device=dev), torch.randn(4 * H, device=dev)] * 2
    grads = [g.clone() for g in grads]
    before = [g.clone() for g in grads]
    L = N.lib()
    ws = torch.empty(L.dic_lstm_dw_workspace(R, B), dtype=torch.uint8, device=dev)
    out_ext = torch.full((R + 2, B, 2 * H), float('nan'), device=dev, dtype=bf)      # the halves the kernel must not read stay NaN
    out_ext[1:R + 1] = out
    out_ext[0, :, :H] = h0[0] if init else 0.0
    out_ext[R + 1, :, H:] = h0[1] if init else 0.0
    fuse_dx = I <= 19
    wih = (torch.randn(2 * 4 * H, XW, device=dev) * 0.2).to(bf)
    dxp = torch.full((2, R * B, XW), float('nan'), device=dev, dtype=bf) if fuse_dx else None
    N.check(L.dic_lstm_dw(N.ptr(dg), N.ptr(out_ext), N.ptr(x), N.ptr(wih) if fuse_dx else None, N.ptr(dxp), R, B, H, I, XW, N.ptr_array(grads),
                          int(accumulate), N.ptr(ws), ws.numel(), N.stream_of(dg)), 'dic_lstm_dw')
    if fuse_dx:      # per-direction input gradients dG[d] . W_ih[d], columns [0, 19)
        for d in range(2):
            want = dg.double()[:, :, d].reshape(R * B, 4 * H) @ wih.double()[d * 4 * H:(d + 1) * 4 * H, :19]
            got = dxp[d, :, :19].double()
            assert float((got - want).abs().max()) <= 1e-2 * float(want.abs().max()) + 1e-3        # bf16 output
    torch.cuda.synchronize()
    d64, o64, x64 = dg.double(), out.double(), x.double()
    for d in range(2):
        hp = torch.zeros(R, B, H, device=dev, dtype=torch.float64)
        h0d = h0[d].to(bf).double() if init else torch.zeros(B, H, device=dev, dtype=torch.float64)
        if d == 0:
            hp[0], hp[1:] = h0d, o64[:-1, :, :H]
        else:
            hp[-1], hp[:-1] = h0d, o64[1:, :, H:]
        w_hh = torch.einsum('tbg,tbh->gh', d64[:, :, d], hp)
        w_ih = torch.einsum('tbg,tbi->gi', d64[:, :, d], x64[..., :I])
        base_ih = before[4 * d].double() if accumulate else 0.0
        base_hh = before[4 * d + 1].double() if accumulate else 0.0
        scale = float(w_hh.abs().max())
        assert float((grads[4 * d + 1].double() - (w_hh + base_hh)).abs().max()) <= 2e-5 * scale + 1e-5
        assert float((grads[4 * d].double() - (w_ih + base_ih)).abs().max()) <= 2e-5 * float(w_ih.abs().max()) + 1e-5
        assert torch.equal(grads[4 * d + 2], before[4 * d + 2]) and torch.equal(grads[4 * d + 3], before[4 * d + 3])     # biases untouched


@pytest.mark.parametrize('R,B,init,accumulate', [(24, 200, False, False), (3, 64, True, True), (5, 130, True, False), (1, 37, False, True),
                                                   (2, 4099, True, False), (24, 1024, True, True)])
def test_lstm_dw_wide_matches_matmul(R, B, init, accumulate):
    """dic_lstm_dw_wide (decoder: 256-wide input, one pass over dG by four workgroups per row chunk) against f64 products of the
    same bf16 operands, incl. the shifted last tile of a row count that is not a multiple of 32."""
    from deep_interpolation_clustering_amd import _native as N
    torch.manual_seed(R * 100 + B)
    dev = torch.device('cuda')
    bf, I = torch.bfloat16, 256
    dg = (torch.randn(R, B, 2, 4 * H, device=dev) * 0.3).to(bf)
    out = (torch.randn(R, B, 2 * H, device=dev) * 0.5).to(bf)
    x = torch.randn(R, B, I, device=dev).clamp_min(0).to(bf)
    h0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    grads = [torch.randn(4 * H, I, device=dev), torch.randn(4 * H, H, device=dev), torch.randn(4 * H, device=dev), torch.randn(4 * H, device=dev)] * 2
    grads = [g.clone() for g in grads]
    before = [g.clone() for g in grads]
    L = N.lib()
    ws = torch.empty(L.dic_lstm_dw_wide_workspace(R, B), dtype=torch.uint8, device=dev)
    out_ext = torch.full((R + 2, B, 2 * H), float('nan'), device=dev, dtype=bf)      # the halves the kernel must not read stay NaN
    out_ext[1:R + 1] = out
    out_ext[0, :, :H] = h0[0] if init else 0.0
    out_ext[R + 1, :, H:] = h0[1] if init else 0.0
    N.check(L.dic_lstm_dw_wide(N.ptr(dg), N.ptr(out_ext), N.ptr(x), 0, R, B, H, I, N.ptr_array(grads), int(accumulate), N.ptr(ws), ws.numel(),
                               N.stream_of(dg)), 'dic_lstm_dw_wide')
    torch.cuda.synchronize()
    d64, o64, x64 = dg.double(), out.double(), x.double()
    for d in range(2):
        hp = torch.zeros(R, B, H, device=dev, dtype=torch.float64)
        h0d = h0[d].to(bf).double() if init else torch.zeros(B, H, device=dev, dtype=torch.float64)
        if d == 0:
            hp[0], hp[1:] = h0d, o64[:-1, :, :H]
        else:
            hp[-1], hp[:-1] = h0d, o64[1:, :, H:]
        w_hh = torch.einsum('tbg,tbh->gh', d64[:, :, d], hp)
        w_ih = torch.einsum('tbg,tbi->gi', d64[:, :, d], x64)
        base_ih = before[4 * d].double() if accumulate else 0.0
        base_hh = before[4 * d + 1].double() if accumulate else 0.0
        assert float((grads[4 * d + 1].double() - (w_hh + base_hh)).abs().max()) <= 2e-5 * float(w_hh.abs().max()) + 1e-5
        assert float((grads[4 * d].double() - (w_ih + base_ih)).abs().max()) <= 2e-5 * float(w_ih.abs().max()) + 1e-5
        assert torch.equal(grads[4 * d + 2], before[4 * d + 2]) and torch.equal(grads[4 * d + 3], before[4 * d + 3])     # biases untouched


@pytest.mark.parametrize('I,init', [(18, False), (256, True)])
def test_lstm_param_grads_written_in_place(I, init):
    """With ``.grad`` tensors present (the flat bucket's views) the kernels ADD the parameter gradients into them and hand autograd
    None: same values as the autograd-returned path, on top of what the buffers held."""
    from deep_interpolation_clustering_amd import lstm as L
    torch.manual_seed(5)
    dev = torch.device('cuda')
    R, B = 6, 96
    net = torch.nn.LSTM(I, H, num_layers=1, bidirectional=True).to(dev)
    x = torch.randn(R, B, I, device=dev)
    h0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    go = torch.randn(R, B, 2 * H, device=dev)

    def run():
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out, (hn, cn) = L.bilstm(x, net, h0, h0)
        ((out.float() * go).sum() + hn.sum()).backward()
    run()                                                   # .grad is None: gradients come back through autograd
    ref = {k: p.grad.clone() for k, p in net.named_parameters()}
    for p in net.parameters():
        p.grad = torch.full_like(p, 0.25)                   # now the sinks exist: the kernels accumulate
    ptrs = {k: p.grad.data_ptr() for k, p in net.named_parameters()}
    run()
    for k, p in net.named_parameters():
        assert p.grad.data_ptr() == ptrs[k]
        np.testing.assert_allclose((p.grad - 0.25).cpu().numpy(), ref[k].cpu().numpy(), rtol=1e-4, atol=1e-4 * float(ref[k].abs().max()))


def test_packed_interp_encoder_path_matches_unpacked():
    """ops.sci_cci_packed -> lstm.bilstm_packed (the interpolation kernel writes the recurrence kernel's input rows, the input
    gradient returns in that layout) against the module path sci_cci -> permute -> bilstm on the same parameters."""
    from deep_interpolation_clustering_amd import lstm as L
    from deep_interpolation_clustering_amd import ops
    from oracle.synth import vitals_stack
    torch.manual_seed(3)
    dev = torch.device('cuda')
    B, C, T, R, Hh = 150, 6, 96, 24, 24.0
    x_np, n = vitals_stack(5, B, C, T, Hh, 50)
    x, lens = torch.tensor(x_np, device=dev), torch.tensor(n, device=dev)
    net = torch.nn.LSTM(3 * C, H, num_layers=1, bidirectional=True).to(dev)
    grid = ops.ref_grid(Hh, R, dev)
    go = torch.randn(R, B, 2 * H, device=dev)
    res = {}
    for mode in ('module', 'packed'):
        ks = torch.rand(C, device=dev).requires_grad_()
        kc = (torch.eye(C, device=dev) + 0.1 * torch.randn(C, C, device=dev)).requires_grad_()
        torch.manual_seed(9)
        ks.data.copy_(torch.rand(C))
        kc.data.copy_(torch.eye(C) + 0.1 * torch.randn(C, C))
        net.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            if mode == 'packed':
                out, (hn, cn) = L.bilstm_packed(ops.sci_cci_packed(x, ks, kc, grid, lens), net)
            else:
                out, (hn, cn) = L.bilstm(ops.sci_cci(x, ks, kc, grid, lens).permute(1, 0, 2), net)
        ((out.float() * go).sum() + (hn * 0.5).sum() + cn.sum()).backward()
        res[mode] = dict(out=out.detach().float(), hn=hn.detach(), ks=ks.grad.clone(), kc=kc.grad.clone(),
                         **{k: p.grad.clone() for k, p in net.named_parameters()})
    assert torch.equal(res['module']['out'], res['packed']['out'])          # same bf16 input rows -> the same recurrence, bit for bit
    assert torch.equal(res['module']['hn'], res['packed']['hn'])
    for k in res['module']:
        if k in ('out', 'hn'):
            continue
        a, b = res['packed'][k].cpu().numpy(), res['module'][k].cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=2e-2, atol=4e-3 * np.abs(b).max(), err_msg=k)     # dX stays bf16 on the packed path


@pytest.mark.parametrize('I', [18, 256])
def test_batch_major_state_layout_is_a_relabelling(I):
    """batch_major_state=True moves h0/c0/h_n/c_n and their gradients to (B,2,H): same numbers as the (2,B,H) layout, bit for bit."""
    from deep_interpolation_clustering_amd import lstm as L
    torch.manual_seed(12)
    dev = torch.device('cuda')
    R, B = 5, 70
    net = torch.nn.LSTM(I, H, num_layers=1, bidirectional=True).to(dev)
    x = torch.randn(R, B, I, device=dev)
    h0, c0 = torch.randn(2, B, H, device=dev) * 0.5, torch.randn(2, B, H, device=dev) * 0.5
    go, gh, gc = torch.randn(R, B, 2 * H, device=dev), torch.randn(2, B, H, device=dev), torch.randn(2, B, H, device=dev)
    res = {}
    for bm in (False, True):
        net.zero_grad()
        xi = x.clone().requires_grad_()
        hi = (h0.transpose(0, 1).contiguous() if bm else h0.clone()).requires_grad_()
        ci = (c0.transpose(0, 1).contiguous() if bm else c0.clone()).requires_grad_()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out, (hn, cn) = L.bilstm(xi, net, hi, ci, batch_major_state=bm)
        assert tuple(hn.shape) == ((B, 2, H) if bm else (2, B, H))
        hn_, cn_ = (hn.transpose(0, 1), cn.transpose(0, 1)) if bm else (hn, cn)
        ((out.float() * go).sum() + (hn_ * gh).sum() + (cn_ * gc).sum()).backward()
        res[bm] = dict(out=out.detach().clone(), hn=hn_.detach().clone(), cn=cn_.detach().clone(), dx=xi.grad.clone(),
                       dh0=(hi.grad.transpose(0, 1) if bm else hi.grad).clone(), dc0=(ci.grad.transpose(0, 1) if bm else ci.grad).clone(),
                       **{k: p.grad.clone() for k, p in net.named_parameters()})
    for k in res[False]:
        assert torch.equal(res[False][k], res[True][k]), k


@pytest.mark.parametrize('R,B,I,init,bm', [(24, 200, 18, False, False), (24, 96, 256, True, True), (5, 1, 256, True, False), (3, 130, 40, True, True),
                                             (2, 33, 1, False, False)])
def test_f32_recurrence_matches_nn_lstm(R, B, I, init, bm):
    """The f32 step's recurrence (v_mfma_f32_32x32x2_f32: exact f32 products and sums) against torch.nn.LSTM in f32 on the same
    device: outputs, final states and every gradient to f32 rounding -- the configuration of the 1e-5 loss parity."""
    from deep_interpolation_clustering_amd import lstm as L
    torch.manual_seed(R * 100 + B)
    dev = torch.device('cuda')
    net = torch.nn.LSTM(I, H, num_layers=1, bidirectional=True).to(dev)
    x = torch.randn(R, B, I, device=dev) * (1.0 if I < 100 else 0.5)
    h0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    c0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    go, gh, gc = torch.randn(R, B, 2 * H, device=dev), torch.randn(2, B, H, device=dev), torch.randn(2, B, H, device=dev)
    res = {}
    for mode in ('hip', 'torch'):
        net.zero_grad()
        xi = x.clone().requires_grad_()
        hi = None if not init else h0.clone().requires_grad_()
        ci = None if not init else c0.clone().requires_grad_()
        if mode == 'hip':
            assert L.f32_available(xi, net)
            hin = hi.transpose(0, 1).contiguous() if (init and bm) else hi
            cin = ci.transpose(0, 1).contiguous() if (init and bm) else ci
            out, (hn, cn) = L.bilstm(xi, net, hin, cin, batch_major_state=bm)
            assert out.dtype == torch.float32
            if bm:
                hn, cn = hn.transpose(0, 1), cn.transpose(0, 1)
        else:
            out, (hn, cn) = net(xi) if not init else net(xi, (hi, ci))
        ((out * go).sum() + (hn * gh).sum() + (cn * gc).sum()).backward()
        res[mode] = dict(out=out.detach(), hn=hn.detach(), cn=cn.detach(), dx=xi.grad, **{k: p.grad.clone() for k, p in net.named_parameters()})
        if init:
            res[mode].update(dh0=hi.grad, dc0=ci.grad)
    for k, ref in res['torch'].items():
        a, b = res['hip'][k].cpu().numpy(), ref.cpu().numpy()
        tol = 2e-6 if k in ('out', 'hn', 'cn') else 1e-5
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=tol * max(1.0, float(np.abs(b).max())), err_msg=k)


@pytest.mark.parametrize('R,B,I,init,bm,proj', [(24, 200, 18, False, False, True), (24, 200, 18, True, True, False), (24, 96, 256, True, True, True),
                                                  (5, 1, 256, True, False, True), (3, 130, 28, True, True, True), (2, 33, 1, False, False, True),
                                                  (1, 4099, 18, True, True, True), (6, 65, 36, False, True, True),
                                                  (24, 200, 18, True, True, 'generic'), (24, 96, 256, True, False, 'generic')])
def test_x3_recurrence_matches_nn_lstm(R, B, I, init, bm, proj, monkeypatch):
    """The x3 recurrence kernels of round 6 (csrc/dic_lstm32.hip: lstm_rec_fwd8x3 / lstm_rec_bwd8x3 -- eight waves per 32-row tile, every product hi.hi + lo.hi
    + hi.lo on the bf16 matrix cores, gate non-linearities on the transcendental unit, the narrow encoder input projected INSIDE the kernel, gate gradients
    leaving as split planes for dic_gemm_tn_planes / dic_gemm_nt_planes) against torch.nn.LSTM in f32: outputs, final states and every gradient to the
    2^-17-per-product accuracy of the split -- ragged last tiles, R = 1, one row, with / without initial states, the in-kernel projection on and off (I = 36
    does not fit its 32 columns: the gx path), the wide decoder input."""
    from deep_interpolation_clustering_amd import lstm as L, ops
    monkeypatch.setattr(L, 'X3_REC_PROJ', bool(proj))
    if proj == 'generic':          # the fallbacks behind DIC_X3_DW=0 / DIC_X3_DX_TILE=0: dic_gemm_tn_planes / dic_gemm_nt_planes on the split planes of dG
        monkeypatch.setattr(L, 'X3_DW', False)
        monkeypatch.setattr(L, 'X3_DX_TILE', False)
    torch.manual_seed(R * 100 + B)
    dev = torch.device('cuda')
    net = torch.nn.LSTM(I, H, num_layers=1, bidirectional=True).to(dev)
    x = torch.randn(R, B, I, device=dev) * (1.0 if I < 100 else 0.5)
    h0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    c0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    go, gh, gc = torch.randn(R, B, 2 * H, device=dev), torch.randn(2, B, H, device=dev), torch.randn(2, B, H, device=dev)
    res = {}
    for mode in ('x3', 'torch'):
        net.zero_grad()
        xi = x.clone().requires_grad_()
        hi = None if not init else h0.clone().requires_grad_()
        ci = None if not init else c0.clone().requires_grad_()
        if mode == 'x3':
            hin = hi.transpose(0, 1).contiguous() if (init and bm) else hi
            cin = ci.transpose(0, 1).contiguous() if (init and bm) else ci
            with ops.f32_products_mode('x3'):
                out, (hn, cn) = L.bilstm(xi, net, hin, cin, batch_major_state=bm)
                assert out.dtype == torch.float32
                if bm:
                    hn, cn = hn.transpose(0, 1), cn.transpose(0, 1)
                ((out * go).sum() + (hn * gh).sum() + (cn * gc).sum()).backward()
        else:
            out, (hn, cn) = net(xi) if not init else net(xi, (hi, ci))
            ((out * go).sum() + (hn * gh).sum() + (cn * gc).sum()).backward()
        res[mode] = dict(out=out.detach(), hn=hn.detach(), cn=cn.detach(), dx=xi.grad, **{k: p.grad.clone() for k, p in net.named_parameters()})
        if init:
            res[mode].update(dh0=hi.grad, dc0=ci.grad)
    for k, ref in res['torch'].items():
        a, b = res['x3'][k].double().cpu().numpy(), ref.double().cpu().numpy()
        assert np.isfinite(a).all(), k
        # 2^-17 ~ 8e-6 per product term; sums over up to R * B rows average it down, recurrences compound it
        tol = 2e-5 if k in ('out', 'hn', 'cn') else 5e-5
        assert float(np.abs(a - b).max()) <= tol * max(1.0, float(np.abs(b).max())), (k, float(np.abs(a - b).max()), float(np.abs(b).max()))


def test_split_plane_products_equal_the_f32_operand_products():
    """dic_gemm_nt_planes / dic_gemm_tn_planes (A = two bf16 planes hi + lo, as the x3 backward writes the gate gradients) against dic_gemm_nt / dic_gemm_tn
    on the f32 tensor the planes stand for: the same three-term products, so equal to f32 summation-order noise -- and against the f64 product."""
    from deep_interpolation_clustering_amd import ops
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(3)
    M, Ncol, K = 4099, 1024, 20
    a = torch.randn(M, Ncol, device=dev, generator=g)
    hi = a.to(torch.bfloat16)
    lo = (a - hi.float()).to(torch.bfloat16)
    planes = torch.stack([hi, lo])                          # (2, M, 1024)
    a2 = ops.planes_to_f32(planes)                          # what the planes stand for (a to 2^-17)
    assert float((a2 - a).abs().max()) <= 2.0 ** -16 * float(a.abs().max())
    w = torch.randn(K, Ncol, device=dev, generator=g) * 0.1  # dX = dG . W_ih: (M, 1024) x (20, 1024)^T
    y = ops.gemm_nt_planes(planes, w)
    ref = (a2.double() @ w.double().t())
    assert float((y.double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
    assert float((y - ops.gemm_nt(a2, w)).abs().max()) <= 2e-5 * float(ref.abs().max())
    w2 = torch.randn(256, Ncol, device=dev, generator=g) * 0.1      # the 128-column-tile kernel
    y2, ref2 = ops.gemm_nt_planes(planes, w2), a2.double() @ w2.double().t()
    assert float((y2.double() - ref2).abs().max()) <= 3e-5 * float(ref2.abs().max())
    x = torch.randn(M, 20, device=dev, generator=g)
    h = torch.randn(M, 128, device=dev, generator=g)
    for d in range(2):                                      # dW_ih / dW_hh of one direction from its half of the planes (strided column views)
        ad = planes[..., 512 * d:512 * (d + 1)]
        dst, dst2 = torch.zeros(512, 18, device=dev), torch.full((512, 128), 0.5, device=dev)
        ops.gemm_tn_into(ad, x, dst, kcols=18, x2=h, dst2=dst2, accumulate=True)
        r1 = a2[:, 512 * d:512 * (d + 1)].double().t() @ x[:, :18].double()
        r2 = a2[:, 512 * d:512 * (d + 1)].double().t() @ h.double() + 0.5
        assert float((dst.double() - r1).abs().max()) <= 3e-5 * float(r1.abs().max())
        assert float((dst2.double() - r2).abs().max()) <= 3e-5 * float(r2.abs().max())


@pytest.mark.parametrize('R,B,init,relu_in', [(24, 200, True, True), (7, 70, False, False), (1, 64, True, True), (5, 333, True, False), (3, 1, False, True)])
def test_decoder_forward_with_the_projection_inside_tracks_the_gx_path(R, B, init, relu_in, monkeypatch):
    """dic_lstm_fwd_xproj (the decoder's input projection inside the recurrence kernel: no gx tensor; round 4) against dic_row_proj + dic_lstm_fwd on the same
    LSTM (I = 256, large-batch path forced): the new kernel keeps x W_ih^T + b in f32 where the gx path rounds it to bf16 on its way through HBM, so outputs
    and states agree to a bf16 rounding or two, and -- through the saved state, which dic_lstm_bwd reads in the same lane-native order -- every gradient to
    bf16 noise.  Ragged last tiles (B = 70, 333, 1), R = 1, with / without initial states and the rectified input."""
    from deep_interpolation_clustering_amd import lstm as L
    monkeypatch.setattr(L, 'SMALL_BATCH', 0)
    torch.manual_seed(R * 100 + B)
    dev = torch.device('cuda')
    net = torch.nn.LSTM(2 * H, H, num_layers=1, bidirectional=True).to(dev)
    x = torch.randn(R, B, 2 * H, device=dev) * 0.5
    h0 = (torch.randn(2, B, H, device=dev) * 0.5) if init else None
    c0 = (torch.randn(2, B, H, device=dev) * 0.5) if init else None
    go = rb(torch.randn(R, B, 2 * H, device=dev))
    res = {}
    for inside in (False, True):
        monkeypatch.setattr(L, 'FWD_XPROJ', inside)
        net.zero_grad()
        xi = x.clone().requires_grad_()
        hi = None if h0 is None else h0.clone().requires_grad_()
        ci = None if c0 is None else c0.clone().requires_grad_()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out, (hn, cn) = L.bilstm(xi, net, hi, ci, input_rectify=relu_in)
        ((out.float() * go).sum() + hn.sum() + (cn * 0.5).sum()).backward()
        res[inside] = dict(out=out.detach().float(), hn=hn.detach(), cn=cn.detach(), dx=xi.grad.clone(),
                           dh0=None if hi is None else hi.grad.clone(), **{k: p.grad.clone() for k, p in net.named_parameters()})
    a_, b_ = res[True], res[False]
    for k in ('out', 'hn', 'cn'):
        torch.testing.assert_close(a_[k], b_[k], rtol=0, atol=2.0 ** -6)
        assert float((a_[k] - b_[k]).abs().mean()) < 5e-4, k
    for k in a_:
        if k in ('out', 'hn', 'cn') or a_[k] is None:
            continue
        err = float((a_[k].float() - b_[k].float()).abs().max()) / (float(b_[k].float().abs().max()) + 1e-12)
        assert err < 3e-2, (k, err)


def test_decoder_forward_with_the_projection_inside_without_saved_state(monkeypatch):
    """The same kernel under no_grad (evaluation passes: no gates / cell states are written, no boundary rows) and with the rectified-output variant
    (out_r): against the gx path, and relu(out) == out_r."""
    from deep_interpolation_clustering_amd import lstm as L
    monkeypatch.setattr(L, 'SMALL_BATCH', 0)
    torch.manual_seed(11)
    dev = torch.device('cuda')
    R, B = 9, 157
    net = torch.nn.LSTM(2 * H, H, num_layers=1, bidirectional=True).to(dev)
    x = torch.randn(R, B, 2 * H, device=dev) * 0.5
    h0, c0 = torch.randn(B, 2, H, device=dev) * 0.3, torch.randn(B, 2, H, device=dev) * 0.3
    res = {}
    for inside in (False, True):
        monkeypatch.setattr(L, 'FWD_XPROJ', inside)
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            out, (hn, cn) = L.bilstm(x, net, h0, c0, batch_major_state=True, input_rectify=True)
            outr, _ = L.bilstm(x, net, h0, c0, batch_major_state=True, input_rectify=True, rectified_out=True)
        assert torch.equal(outr, torch.relu(out))
        res[inside] = (out.float(), hn, cn)
    for a_, b_ in zip(res[True], res[False]):
        torch.testing.assert_close(a_, b_, rtol=0, atol=2.0 ** -6)


@pytest.mark.parametrize('mode', ['pipelined64', 'tile32', 'f32'])
@pytest.mark.parametrize('I,B', [(18, 200), (256, 70)])
def test_rectified_output_equals_relu_applied_outside(mode, I, B, monkeypatch):
    """rectified_out=True hands out relu(out) and applies the ReLU's backward inside the recurrence kernel (mask = sign of tanh(c_t),
    which is the sign of h_t since o_t is a sigmoid): same numbers as relu(bilstm(...)) with torch's threshold backward."""
    from deep_interpolation_clustering_amd import lstm as L
    monkeypatch.setattr(L, 'SMALL_BATCH', 0 if mode == 'pipelined64' else 1 << 30)
    torch.manual_seed(5)
    dev = torch.device('cuda')
    R = 7
    net = torch.nn.LSTM(I, H, num_layers=1, bidirectional=True).to(dev)
    x = torch.randn(R, B, I, device=dev) * (1.0 if I < 100 else 0.5)
    go, gh = torch.randn(R, B, 2 * H, device=dev), torch.randn(2, B, H, device=dev)
    res = {}
    for inside in (False, True):
        net.zero_grad()
        xi = x.clone().requires_grad_()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=mode != 'f32'):
            out, (hn, cn) = L.bilstm(xi, net, rectified_out=inside)
            if not inside:
                out = torch.relu(out)
        ((out.float() * go).sum() + (hn * gh).sum()).backward()
        res[inside] = dict(out=out.detach().float(), hn=hn.detach(), dx=xi.grad.clone(), **{k: p.grad.clone() for k, p in net.named_parameters()})
    assert float((res[True]['out'] == 0).float().mean()) > 0.3            # the mask is doing something
    for k in res[False]:
        if k in ('out', 'hn'):
            assert torch.equal(res[False][k], res[True][k]), k
        else:       # (an element whose h rounds to exactly +-0 may take the other branch: none in practice, but not a bit-level contract)
            a, b = res[True][k].cpu().numpy(), res[False][k].cpu().numpy()
            np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6 * float(np.abs(b).max()), err_msg=k)


@pytest.mark.parametrize('n,want_dx', [(24 * 200, True), (32, True), (70000, True), (8192 * 3 + 5, False), (1, True)])
def test_fc_bwd_matches_matmul(n, want_dx):
    """dic_fc_bwd (first CompressFC layer's backward: dx = dz . W and dW = dz^T . x from one pass over the rows) against f64
    products of the same bf16 operands, for row counts that are not a multiple of the 32-row tile."""
    from deep_interpolation_clustering_amd import _native as N
    torch.manual_seed(n)
    dev, bf = torch.device('cuda'), torch.bfloat16
    dz = (torch.randn(n, 128, device=dev) * 0.3).to(bf)
    x = (torch.randn(n, 256, device=dev) * 0.5).to(bf)
    w = (torch.randn(128, 256, device=dev) * 0.1).to(bf)
    L = N.lib()
    dx = torch.full((n, 256), float('nan'), device=dev, dtype=bf) if want_dx else None
    dw = torch.full((128, 256), float('nan'), device=dev)
    ws = torch.empty(L.dic_fc_bwd_workspace(n, 256, 128), dtype=torch.uint8, device=dev)
    N.check(L.dic_fc_bwd(N.ptr(dz), N.ptr(x), N.ptr(w), n, 256, 128, N.ptr(dx), N.ptr(dw), N.ptr(ws), ws.numel(), N.stream_of(dz)), 'dic_fc_bwd')
    torch.cuda.synchronize()
    want_w = dz.double().t() @ x.double()
    assert float((dw.double() - want_w).abs().max()) <= 2e-5 * float(want_w.abs().max()) + 1e-5
    if want_dx:
        want_x = dz.double() @ w.double()
        assert float((dx.double() - want_x).abs().max()) <= 8e-3 * float(want_x.abs().max()) + 1e-3        # bf16 output


def test_rows_linear_backward_uses_the_one_pass_kernel_and_matches_the_gemm_path(monkeypatch):
    from deep_interpolation_clustering_amd import ops
    torch.manual_seed(3)
    dev = torch.device('cuda')
    n = 24 * 512
    x0 = torch.randn(n, 256, device=dev).to(torch.bfloat16)
    lin = torch.nn.Linear(256, 128).to(dev)
    g = torch.randn(n, 128, device=dev).to(torch.bfloat16)
    res = {}
    for rows in (1 << 40, 1):          # library GEMMs, then the fused kernel
        monkeypatch.setattr(ops, 'FC_BWD_MIN_ROWS', rows)
        lin.zero_grad()
        x = x0.clone().requires_grad_()
        ops.rows_linear(x, lin.weight, lin.bias).backward(g)
        res[rows] = (x.grad.float().clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    for a, b in zip(res[1], res[1 << 40]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-2, atol=2e-3 * float(b.abs().max()))


@pytest.mark.parametrize('n,nout,bias', [(24 * 200, 1024, True), (33, 256, False), (70001, 1024, True), (1, 512, True)])
def test_row_proj_matches_addmm(n, nout, bias):
    """dic_row_proj (weights resident in registers, x tiles through LDS) against f64 products of the same bf16 operands."""
    from deep_interpolation_clustering_amd import _native as N
    torch.manual_seed(n)
    dev, bf = torch.device('cuda'), torch.bfloat16
    x = (torch.randn(n, 256, device=dev) * 0.5).clamp_min(0).to(bf)
    w = (torch.randn(nout, 256, device=dev) * 0.1).to(bf)
    b = (torch.randn(nout, device=dev) * 0.3).to(bf) if bias else None
    out = torch.full((n, nout), float('nan'), device=dev, dtype=bf)
    N.check(N.lib().dic_row_proj(N.ptr(x), N.ptr(w), N.ptr(b), n, 256, nout, N.ptr(out), 0, 0, N.stream_of(x)), 'dic_row_proj')
    want = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    assert float((out.double() - want).abs().max()) <= 8e-3 * float(want.abs().max()) + 1e-3        # bf16 output
    ref = torch.addmm(b, x, w.t()) if bias else x @ w.t()
    assert float((out.float() - ref.float()).abs().max()) <= 8e-3 * float(want.abs().max()) + 1e-3


@pytest.mark.parametrize('n', [24 * 512, 70001, 8192])
def test_row_proj_stats_matches_addmm_and_colstats(n):
    """dic_row_proj_stats: z = x W^T + b for Linear(256, 128) and the column sums of z, z^2 (of the stored bf16 values) + row count."""
    from deep_interpolation_clustering_amd import _native as N
    torch.manual_seed(n)
    dev, bf = torch.device('cuda'), torch.bfloat16
    x = (torch.randn(n, 256, device=dev) * 0.5).to(bf)
    w = (torch.randn(128, 256, device=dev) * 0.1).to(bf)
    b = (torch.randn(128, device=dev) * 0.3).to(bf)
    L = N.lib()
    z = torch.full((n, 128), float('nan'), device=dev, dtype=bf)
    sums = torch.full((257,), float('nan'), device=dev, dtype=torch.float64)
    ws = torch.empty(L.dic_row_proj_stats_workspace(n, 128), dtype=torch.uint8, device=dev)
    N.check(L.dic_row_proj_stats(N.ptr(x), N.ptr(w), N.ptr(b), n, 256, 128, N.ptr(z), N.ptr(sums), N.ptr(ws), ws.numel(), N.stream_of(x)), 'dic_row_proj_stats')
    want = x.double() @ w.double().t() + b.double()
    assert float((z.double() - want).abs().max()) <= 8e-3 * float(want.abs().max()) + 1e-3
    zd = z.double()
    assert float(sums[256]) == n
    np.testing.assert_allclose(sums[:128].cpu().numpy(), zd.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-5 * n)
    np.testing.assert_allclose(sums[128:256].cpu().numpy(), (zd * zd).sum(0).cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize('R,B', [(5, 128), (24, 64), (3, 192)])
def test_row_proj_lane_native_output_is_a_reordering_of_the_row_major_one(R, B):
    """dic_row_proj(lane_native_batch = B): the decoder's gx in the order of the recurrence kernel's MFMA accumulators -- element
    (t, 32-row tile, direction, wave, gate, unit half, hh, row, e) -- holds exactly the values of the row-major (R*B, 2*4*128) product."""
    from deep_interpolation_clustering_amd import _native as N
    torch.manual_seed(R * B)
    dev, bf = torch.device('cuda'), torch.bfloat16
    n = R * B
    x = (torch.randn(n, 256, device=dev) * 0.5).clamp_min(0).to(bf)
    w = (torch.randn(1024, 256, device=dev) * 0.1).to(bf)
    b = (torch.randn(1024, device=dev) * 0.3).to(bf)
    rows = torch.full((n, 1024), float('nan'), device=dev, dtype=bf)
    nat = torch.full((n, 1024), float('nan'), device=dev, dtype=bf)
    N.check(N.lib().dic_row_proj(N.ptr(x), N.ptr(w), N.ptr(b), n, 256, 1024, N.ptr(rows), 0, 0, N.stream_of(x)), 'dic_row_proj')
    N.check(N.lib().dic_row_proj(N.ptr(x), N.ptr(w), N.ptr(b), n, 256, 1024, N.ptr(nat), B, 0, N.stream_of(x)), 'dic_row_proj')
    # (t, tile, dir, wave, gate, qp, hh, row, e / 4, e % 4) -> (t, tile, row, dir, gate, wave, qp, e / 4, hh, e % 4): unit = 32 wave + 16 qp + 8 (e / 4) + 4 hh + e % 4
    back = nat.view(R, B // 32, 2, 4, 4, 2, 2, 32, 2, 4).permute(0, 1, 7, 2, 4, 3, 5, 8, 6, 9).reshape(n, 1024)
    assert not torch.isnan(back.float()).any()
    assert torch.equal(back, rows)
    with pytest.raises(RuntimeError):          # batches that do not tile by 64 rows are refused, not mangled
        N.check(N.lib().dic_row_proj(N.ptr(x), N.ptr(w), N.ptr(b), n, 256, 1024, N.ptr(nat), 96, 0, N.stream_of(x)), 'dic_row_proj')


@pytest.mark.parametrize('R,B,init', [(24, 64, True), (7, 192, False)])
def test_lane_native_gx_path_equals_row_major_path(R, B, init, monkeypatch):
    """Decoder-shaped LSTM (input width 256) on the 64-row kernels: gx handed from dic_row_proj to dic_lstm_fwd in the lane-native form
    (registers, a step ahead) or row-major (LDS-staged tile) -- the same numbers go into the same accumulators: bit-equal results."""
    from deep_interpolation_clustering_amd import lstm as L
    monkeypatch.setattr(L, 'SMALL_BATCH', 0)
    torch.manual_seed(R + B)
    dev = torch.device('cuda')
    net = torch.nn.LSTM(256, H, num_layers=1, bidirectional=True).to(dev)
    x = torch.randn(R, B, 256, device=dev) * 0.5
    h0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    c0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    go, gh = torch.randn(R, B, 2 * H, device=dev), torch.randn(2, B, H, device=dev)
    res = {}
    for native in (False, True):
        monkeypatch.setattr(L, 'GX_LANE_NATIVE', native)
        net.zero_grad()
        xi = x.clone().requires_grad_()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out, (hn, cn) = L.bilstm(xi, net, h0, c0)
        ((out.float() * go).sum() + (hn * gh).sum() + cn.sum()).backward()
        res[native] = dict(out=out.detach().float(), hn=hn.detach(), cn=cn.detach(), dx=xi.grad.clone(), **{k: p.grad.clone() for k, p in net.named_parameters()})
    for k in res[False]:
        assert torch.equal(res[False][k], res[True][k]), k


@pytest.mark.parametrize('n,p', [(24 * 512, 0.0), (8192 + 37, 0.0), (24 * 512, 0.25)])
def test_compress_fc_as_one_node_equals_the_two_node_path(n, p, monkeypatch):
    """CompressFC's training step as one autograd node (ops._CompressFC: dic_fc_bwd_bnhead forms the gradient of the 128-wide
    pre-activation inside the kernel) against rows_linear + bn_relu_head (dic_bnhead_bwd_input writes it, dic_fc_bwd reads it): the
    same arithmetic in the same order -- outputs, every gradient and the BatchNorm running statistics are bit-equal; also with the
    in-kernel dropout mask (same seed and call counter: same mask)."""
    from deep_interpolation_clustering_amd import ops
    from deep_interpolation_clustering_amd.rbf import CompressFC
    dev = torch.device('cuda')
    torch.manual_seed(n)
    base = CompressFC(256, 6, p).to(dev).train()
    x = torch.randn(n, 256, device=dev) * 0.7
    gv = torch.randn(n, 6, device=dev)
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(ops, 'COMPRESS_FUSED', fused)
        ops._DROP_STATE.clear()                             # both runs draw the first mask of the same seed
        torch.manual_seed(99)
        fc = CompressFC(256, 6, p).to(dev).train()
        fc.load_state_dict(base.state_dict())
        xi = x.clone().requires_grad_()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            assert ops.compress_fc_fused_ok(xi.to(torch.bfloat16), fc.model[0], fc.model[1], fc.model[4]) == fused
            v = fc(xi.to(torch.bfloat16))
        (v * gv).sum().backward()
        res[fused] = dict(v=v.detach(), dx=xi.grad.clone(), rm=fc.model[1].running_mean.clone(), rv=fc.model[1].running_var.clone(),
                          **{k: q.grad.clone() for k, q in fc.named_parameters()})
    assert float(res[True]['dx'].abs().max()) > 0
    for k in res[False]:
        assert torch.equal(res[False][k], res[True][k]), k


def test_decoder_projection_and_recurrence_at_bench_size():
    """BASELINE configs[1] size (B = 32768 rows per step, R = 24): the lane-native hand-over between dic_row_proj and dic_lstm_fwd
    against the row-major one -- a pure reordering of gx, so out / h_n / c_n and the saved gates are bit-equal; plus linearity of the
    projection in its bias (gx(b) - gx(0) == b to bf16 rounding) as a size-independent check of the 786 432 x 1024 product."""
    from deep_interpolation_clustering_amd import _native as N
    L = N.lib()
    R, B = 24, 32768
    dev, bf = torch.device('cuda'), torch.bfloat16
    torch.manual_seed(3)
    x = (torch.randn(R * B, 256, device=dev) * 0.5).clamp_min(0).to(bf)
    wih = (torch.randn(8 * H, 256, device=dev) * 0.06).to(bf)
    bias = (torch.randn(8 * H, device=dev) * 0.1).to(bf)
    whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf)
    st = N.stream_of(x)
    res = {}
    for native in (0, B):
        gx = torch.empty(R * B, 8 * H, device=dev, dtype=bf)
        out = torch.empty(R, B, 2 * H, device=dev, dtype=bf)
        gates, cs = torch.empty(R, B, 2, 4, H, device=dev, dtype=bf), torch.empty(R, B, 2, H, device=dev, dtype=bf)
        hn, cn = torch.empty(2, B, H, device=dev), torch.empty(2, B, H, device=dev)
        N.check(L.dic_row_proj(N.ptr(x), N.ptr(wih), N.ptr(bias), R * B, 256, 8 * H, N.ptr(gx), native, 0, st), 'dic_row_proj')
        N.check(L.dic_lstm_fwd(N.ptr(gx), int(native > 0), N.ptr(whh), None, None, R, B, H, N.ptr(out), None, N.ptr(hn), N.ptr(cn), N.ptr(gates),
                               N.ptr(cs), 0, 0, st), 'dic_lstm_fwd')
        res[native] = (out, hn, cn, gates, cs)
        if not native:
            g0 = torch.empty_like(gx)
            N.check(L.dic_row_proj(N.ptr(x), N.ptr(wih), None, R * B, 256, 8 * H, N.ptr(g0), 0, 0, st), 'dic_row_proj')
            d = (gx[::97].float() - g0[::97].float()) - bias.float()
            assert float(d.abs().max()) <= 2.0 ** -7 * float(gx[::97].float().abs().max())      # two bf16 roundings
            del g0
        del gx
    for a, b in zip(res[0], res[B]):
        assert torch.equal(a, b)
    assert torch.isfinite(res[0][0].float()).all() and float(res[0][0].float().abs().mean()) > 0.01


@pytest.mark.parametrize('native', [False, True])
def test_row_proj_rectifies_its_input_on_load(native):
    """relu_input: the product of relu(x), x raw (negative halves zeroed by an int16 max on the packed bf16 pairs) == the product of a
    rectified copy of x, bit for bit."""
    from deep_interpolation_clustering_amd import _native as N
    torch.manual_seed(11)
    dev, bf = torch.device('cuda'), torch.bfloat16
    R, B = 6, 128
    n = R * B
    x = (torch.randn(n, 256, device=dev) * 0.5).to(bf)
    x[5, :8] = torch.tensor([-0.0, 0.0, -3e38, 3e38, -1e-38, 1e-38, -3.0, 3.0], device=dev).to(bf)       # signed zeros, extremes, subnormals
    w = (torch.randn(1024, 256, device=dev) * 0.1).to(bf)
    w[:, 2:4] = 0                                   # (keep the two huge values out of the sums)
    b = (torch.randn(1024, device=dev) * 0.3).to(bf)
    a, c = torch.empty(n, 1024, device=dev, dtype=bf), torch.empty(n, 1024, device=dev, dtype=bf)
    lane = B if native else 0
    N.check(N.lib().dic_row_proj(N.ptr(x), N.ptr(w), N.ptr(b), n, 256, 1024, N.ptr(a), lane, 1, N.stream_of(x)), 'dic_row_proj')
    xr = torch.relu(x)
    N.check(N.lib().dic_row_proj(N.ptr(xr), N.ptr(w), N.ptr(b), n, 256, 1024, N.ptr(c), lane, 0, N.stream_of(x)), 'dic_row_proj')
    assert torch.equal(a, c)
    assert float((x < 0).float().mean()) > 0.4


def test_lstm_dw_wide_rectifies_x_on_load():
    """x_relu: dW_ih = dG^T relu(x) from the raw x == from a rectified copy (the h columns of the same operand stay untouched)."""
    from deep_interpolation_clustering_amd import _native as N
    torch.manual_seed(12)
    dev, bf = torch.device('cuda'), torch.bfloat16
    R, B, I = 5, 96, 256
    L = N.lib()
    dg = (torch.randn(R, B, 2, 4, H, device=dev) * 0.1).to(bf)
    out_ext = (torch.randn(R + 2, B, 2 * H, device=dev) * 0.5).to(bf)          # (h has negative entries: they must survive)
    x = (torch.randn(R, B, I, device=dev) * 0.5).to(bf)
    ws = torch.empty(max(16, L.dic_lstm_dw_wide_workspace(R, B)), dtype=torch.uint8, device=dev)
    res = []
    for xin, flag in ((x, 1), (torch.relu(x), 0)):
        grads = [torch.zeros(4 * H, I, device=dev), torch.zeros(4 * H, H, device=dev), torch.zeros(4 * H, device=dev), torch.zeros(4 * H, device=dev)]
        grads = grads + [g.clone() for g in grads]
        N.check(L.dic_lstm_dw_wide(N.ptr(dg), N.ptr(out_ext), N.ptr(xin), flag, R, B, H, I, N.ptr_array(grads), 0, N.ptr(ws), ws.numel(),
                                   N.stream_of(dg)), 'dic_lstm_dw_wide')
        res.append([g.clone() for g in grads])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert float(res[0][0].abs().max()) > 0 and float(res[0][1].abs().max()) > 0


@pytest.mark.parametrize('B', [128, 200])
def test_relu_between_the_lstms_left_to_the_decoder_kernels(B, monkeypatch):
    """Encoder -> F.relu -> decoder (clustering_interp.py:38-41) on the 64-row kernels, two ways: the encoder writes a rectified copy of
    its output (rectified_out=True) / the encoder hands out the raw rows and the decoder's projection and weight-gradient kernels rectify
    them on load (rectified_out='deferred' + input_rectify=True).  Same numbers everywhere: outputs and all gradients bit-equal."""
    from deep_interpolation_clustering_amd import lstm as L
    monkeypatch.setattr(L, 'SMALL_BATCH', 0)
    torch.manual_seed(B)
    dev = torch.device('cuda')
    R = 6
    enc = torch.nn.LSTM(18, H, num_layers=1, bidirectional=True).to(dev)
    dec = torch.nn.LSTM(2 * H, H, num_layers=1, bidirectional=True).to(dev)
    x = torch.randn(R, B, 18, device=dev)
    go = torch.randn(R, B, 2 * H, device=dev)
    res = {}
    for defer in (False, True):
        enc.zero_grad(); dec.zero_grad()
        xi = x.clone().requires_grad_()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            assert L.deferred_relu_ok(B)
            ctxt, (hn, cn) = L.bilstm(xi, enc, batch_major_state=True, rectified_out='deferred' if defer else True)
            assert bool((ctxt < 0).any()) == defer              # raw rows when deferred
            y, (h2, c2) = L.bilstm(ctxt, dec, hn, cn, batch_major_state=True, input_rectify=defer)
        ((y.float() * go).sum() + h2.sum()).backward()
        res[defer] = dict(y=y.detach().float(), h2=h2.detach(), dx=xi.grad.clone(), **{'e.' + k: p.grad.clone() for k, p in enc.named_parameters()},
                          **{'d.' + k: p.grad.clone() for k, p in dec.named_parameters()})
    for k in res[False]:
        assert torch.equal(res[False][k], res[True][k]), k


@pytest.mark.parametrize('R,B,init,relu', [(24, 128, True, True), (5, 70, False, False), (3, 1, True, True), (7, 333, False, True)])
def test_eight_wave_encoder_recurrence_equals_the_four_wave_kernel(R, B, init, relu):
    """dic_lstm_fwd_proj(eight_waves=1): 512-thread workgroups, 16 hidden units per wave with two gates stacked in one MFMA block --
    the same k order in every accumulator, so out, the rectified copy, h_n, c_n and the saved gates / cell states (in the layout
    lstm_bwd reads) are bit-equal to the 4-wave kernel's, for ragged batches and boundary rows too."""
    from deep_interpolation_clustering_amd import _native as N
    L = N.lib()
    dev, bf = torch.device('cuda'), torch.bfloat16
    torch.manual_seed(R * 100 + B)
    x = torch.randn(R, B, 32, device=dev).to(bf)
    wih = (torch.randn(2, 4 * H, 32, device=dev) * 0.1).to(bf)
    whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf)
    h0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    c0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    Bp = (B + 63) // 64 * 64
    res = []
    for eight in (0, 1):
        ext = torch.full((R + 2, B, 2 * H), float('nan'), device=dev, dtype=bf)
        out = ext[1:R + 1]
        outr = torch.full((R, B, 2 * H), float('nan'), device=dev, dtype=bf) if relu else None
        gates = torch.zeros(R, Bp, 2, 4, H, device=dev, dtype=bf)
        cs = torch.zeros(R, Bp, 2, H, device=dev, dtype=bf)
        hn, cn = torch.empty(2, B, H, device=dev), torch.empty(2, B, H, device=dev)
        N.check(L.dic_lstm_fwd_proj(N.ptr(x), N.ptr(wih), N.ptr(whh), N.ptr(h0), N.ptr(c0), R, B, H, 32, N.ptr(out), N.ptr(outr), N.ptr(hn), N.ptr(cn),
                                    N.ptr(gates), N.ptr(cs), 0, 1, eight, N.stream_of(x)), 'dic_lstm_fwd_proj')
        # (padded rows of the saved state hold whatever the kernel computed for the clamped input row: compare the live rows only)
        res.append([out.clone(), ext[0, :, :H].clone(), ext[R + 1, :, H:].clone(), hn, cn] + ([outr] if relu else []))
        res[-1] += [gates.view(R, Bp // 32, 2, 4, 4, 4, 2, 32, 4), cs.view(R, Bp // 32, 2, 4, 1, 4, 2, 32, 4)]
    live = torch.arange(Bp, device=dev).view(Bp // 32, 32) < B                     # [tile][row]
    for a, b in zip(res[0][:-2], res[1][:-2]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][-2:], res[1][-2:]):
        m = live.view(1, Bp // 32, 1, 1, 1, 1, 1, 32, 1).expand_as(a)
        assert torch.equal(a[m], b[m])
    assert torch.isfinite(res[0][0].float()).all()


@pytest.mark.parametrize('R,B,init', [(24, 128, True), (5, 64, False), (3, 320, True)])
def test_eight_wave_decoder_recurrence_equals_the_four_wave_kernel(R, B, init):
    """dic_lstm_fwd(gx_lane_native=2): eight waves per workgroup on the lane-native gx (every wave stages the 4 pieces of its 16 units per
    32-row half; two permutation MFMAs per accumulator block) == gx_lane_native=1: out, h_n, c_n, saved gates and cell states bit-equal."""
    from deep_interpolation_clustering_amd import _native as N
    L = N.lib()
    dev, bf = torch.device('cuda'), torch.bfloat16
    torch.manual_seed(R * 100 + B)
    gx = (torch.randn(R * B, 8 * H, device=dev) * 0.5).to(bf)
    whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf)
    h0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    c0 = torch.randn(2, B, H, device=dev) * 0.5 if init else None
    res = []
    for mode in (1, 2):
        ext = torch.full((R + 2, B, 2 * H), float('nan'), device=dev, dtype=bf)
        out = ext[1:R + 1]
        gates = torch.zeros(R, B, 2, 4, H, device=dev, dtype=bf)
        cs = torch.zeros(R, B, 2, H, device=dev, dtype=bf)
        hn, cn = torch.empty(2, B, H, device=dev), torch.empty(2, B, H, device=dev)
        N.check(L.dic_lstm_fwd(N.ptr(gx), mode, N.ptr(whh), N.ptr(h0), N.ptr(c0), R, B, H, N.ptr(out), None, N.ptr(hn), N.ptr(cn), N.ptr(gates),
                               N.ptr(cs), 0, 1, N.stream_of(gx)), 'dic_lstm_fwd')
        res.append([out.clone(), ext[0, :, :H].clone(), ext[R + 1, :, H:].clone(), hn, cn, gates, cs])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert torch.isfinite(res[0][0].float()).all() and float(res[0][0].float().abs().mean()) > 0.01


@pytest.mark.parametrize('rows', [256, 257, 1000, 24 * 4160, 70000])
def test_decoder_input_gradient_tile_kernel_equals_matmul(rows):
    """dic_lstm_dx_tile (csrc/dic_dxproj.hip, round 5: dX = dG . W_ih on 256 x 256 macro-tiles, dG and W_ih^T through LDS-DMA rings)
    against the f64 product and torch's bf16 matmul: one tile, ragged row counts (the shifted last tile), more tiles than workgroups (the ring runs on
    across tile boundaries), every output element checked."""
    from deep_interpolation_clustering_amd import _native as N
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(rows)
    dg = (torch.randn(rows, 1024, device=dev, generator=g) * 0.3).to(torch.bfloat16)
    w = (torch.randn(1024, 256, device=dev, generator=g) * 0.06).to(torch.bfloat16)
    wt = w.t().contiguous()
    dx = torch.full((rows, 256), float('nan'), device=dev, dtype=torch.bfloat16)
    N.check(N.lib().dic_lstm_dx_tile(N.ptr(dg), N.ptr(wt), rows, 1024, 256, N.ptr(dx), N.stream_of(dg)), 'dic_lstm_dx_tile')
    ref64 = dg.double() @ w.double()
    assert torch.isfinite(dx.float()).all()
    err = (dx.double() - ref64).abs()
    assert float(err.max()) <= 2.0 ** -8 * float(ref64.abs().max()) + 1e-6              # one bf16 rounding of an f32-accumulated sum
    lib = (dg @ w).double()
    assert float((dx.double() - lib).abs().max()) <= 2.0 ** -7 * float(ref64.abs().max())
    # a structured operand catches a permuted fragment or a misplaced output column that random data could hide behind the tolerance
    dg2 = torch.zeros(rows, 1024, device=dev, dtype=torch.bfloat16)
    dg2[torch.arange(rows, device=dev), torch.arange(rows, device=dev) % 1024] = 1.0      # dX row i = row (i mod 1024) of W
    dx2 = torch.empty((rows, 256), device=dev, dtype=torch.bfloat16)
    N.check(N.lib().dic_lstm_dx_tile(N.ptr(dg2), N.ptr(wt), rows, 1024, 256, N.ptr(dx2), N.stream_of(dg)), 'dic_lstm_dx_tile')
    assert torch.equal(dx2, w[torch.arange(rows, device=dev) % 1024])
    assert N.lib().dic_lstm_dx_tile(N.ptr(dg), N.ptr(wt), 255, 1024, 256, N.ptr(dx), N.stream_of(dg)) == -1      # fewer rows than a tile: rejected


@pytest.mark.parametrize('rows', [256, 257, 1000, 24 * 1030 + 7])
def test_decoder_input_gradient_tile_kernel_on_split_planes(rows):
    """dic_lstm_dx_tile_x3 (csrc/dic_dxproj.hip: dX = dG . W_ih on 256 x 256 macro-tiles, dG and W_ih^T as hi / lo bf16 planes, three MFMAs per product,
    f32 output) against the f64 product of the f32 tensors the planes stand for and against dic_gemm_nt_planes: one tile, ragged row counts (the shifted last
    tile), more tiles than workgroups; and a structured operand that would expose a permuted fragment, plane or column."""
    from deep_interpolation_clustering_amd import _native as N, ops
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(rows)
    dg = torch.randn(rows, 1024, device=dev, generator=g) * 0.3
    w = torch.randn(1024, 256, device=dev, generator=g) * 0.06
    dgp, wtp = ops.split_planes(dg), ops.split_planes(w.t())
    dx = torch.full((rows, 256), float('nan'), device=dev)
    N.check(N.lib().dic_lstm_dx_tile_x3(N.ptr(dgp), dgp.stride(0), N.ptr(wtp), wtp.stride(0), rows, 1024, 256, N.ptr(dx), N.stream_of(dg)), 'dic_lstm_dx_tile_x3')
    ref = ops.planes_to_f32(dgp).double() @ ops.planes_to_f32(wtp).double().t()
    assert torch.isfinite(dx).all()
    scale = float(ref.abs().max())
    assert float((dx.double() - ref).abs().max()) <= 3e-5 * scale                       # the dropped lo.lo term and f32 accumulation
    assert float((dx - ops.gemm_nt_planes(dgp, w.t().contiguous())).abs().max()) <= 2e-5 * scale
    # row i of dG = e_(i mod 1024) (exactly representable: lo plane zero): dX row i = row (i mod 1024) of W to the split's accuracy
    dg2 = torch.zeros(rows, 1024, device=dev)
    dg2[torch.arange(rows, device=dev), torch.arange(rows, device=dev) % 1024] = 1.0
    dgp2 = ops.split_planes(dg2)
    N.check(N.lib().dic_lstm_dx_tile_x3(N.ptr(dgp2), dgp2.stride(0), N.ptr(wtp), wtp.stride(0), rows, 1024, 256, N.ptr(dx), N.stream_of(dg)), 'dic_lstm_dx_tile_x3')
    assert torch.equal(dx, ops.planes_to_f32(wtp).t()[torch.arange(rows, device=dev) % 1024])
    assert N.lib().dic_lstm_dx_tile_x3(N.ptr(dgp), dgp.stride(0), N.ptr(wtp), wtp.stride(0), 255, 1024, 256, N.ptr(dx), N.stream_of(dg)) == -1


def test_eight_wave_small_batch_recurrence_equals_four_wave(tmp_path):
    """The eight-waves-per-tile bf16 recurrence kernels of csrc/dic_lstm32.hip (a wave owns 16 hidden units; the backward's dh on
    16x16x32 MFMAs with a permlane16 swap between the two batch blocks) against the four-wave kernels on the same inputs, batch sizes
    with and without a ragged last tile: outputs, saved state and every gradient bit for bit; the bias gradient (a sum over eight
    waves instead of four) to f32 rounding.  One process per variant: the switch DIC_REC_EIGHT_WAVES is read once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for mode in ('0', '1'):
        f = str(tmp_path / f'rec{mode}.pt')
        res = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'rec8_ab.py'), 'run', f], env=dict(os.environ, DIC_REC_EIGHT_WAVES=mode, DIC_REC_SIXTEEN='0'),
                             capture_output=True, text=True, timeout=280, cwd=root)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(torch.load(f))
    four, eight = outs
    for B in four:
        for k in ('out', 'hn', 'cn', 'gates', 'cs', 'dgx', 'dh0', 'dc0'):
            assert torch.equal(four[B][k], eight[B][k]), (B, k)
        torch.testing.assert_close(eight[B]['db'], four[B]['db'], rtol=1e-5, atol=1e-5)


def test_sixteen_row_small_batch_recurrence_equals_the_32_row_kernels(tmp_path):
    """The 16-row-tile kernels of csrc/dic_lstm32.hip (batches up to 2048: v_mfma_f32_16x16x32_bf16, twice the workgroups at half the work per step)
    against the 32-row eight-wave kernels (DIC_REC_SIXTEEN=0) on the same inputs, batches with and without a ragged last tile and R = 1, 2, 24: the two MFMA
    shapes add a gate's products in the same k order, so outputs, final states, dG, dh0 and dc0 agree bit for bit; the bias gradient (one partial per 16
    rows instead of 32) to f32 rounding.  The saved gates / cell states are layout-private to each pair of kernels and not compared."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for mode in ('0', '1'):
        f = str(tmp_path / f'rec16_{mode}.pt')
        res = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'rec8_ab.py'), 'run', f],
                             env=dict(os.environ, DIC_REC_SIXTEEN=mode, REC_AB_SHAPES='24x256,24x300,24x1024,24x2048,1x40,2x96,5x7'),
                             capture_output=True, text=True, timeout=280, cwd=root)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(torch.load(f))
    wide, narrow = outs
    assert len(wide) == 7
    for key in wide:
        for k in ('out', 'hn', 'cn', 'dgx', 'dh0', 'dc0'):
            assert torch.equal(wide[key][k], narrow[key][k]), (key, k)
        torch.testing.assert_close(narrow[key]['db'], wide[key]['db'], rtol=1e-5, atol=1e-5)
        assert not torch.equal(wide[key]['gates'], narrow[key]['gates'])         # (the two runs did take different kernels)


def test_eight_wave_backward_equals_four_wave_on_odd_shapes(tmp_path):
    """lstm_bwd8_kernel (eight waves per 64-row workgroup, dh on 16x16x32 MFMAs) against lstm_bwd_kernel on R = 1, 2, 3, 5, 24, batches with a
    ragged last tile, with / without c0 and the ReLU mask: dG, dh0, dc0 bit for bit; the bias gradient (summed over eight waves instead of
    four) to f32 rounding.  scripts/bwd8_ab.py, one process per variant (DIC_BWD_EIGHT_WAVES is read once)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = []
    for mode in ('0', '1'):
        f = str(tmp_path / f'bwd{mode}.pt')
        res = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'bwd8_ab.py'), 'run', f], env=dict(os.environ, DIC_BWD_EIGHT_WAVES=mode),
                             capture_output=True, text=True, timeout=280, cwd=root)
        assert res.returncode == 0, res.stderr[-2000:]
        files.append(f)
    four, eight = torch.load(files[0]), torch.load(files[1])
    assert set(four) == set(eight) and len(four) == 5
    for key in four:
        for k in ('dgx', 'dh0', 'dc0'):
            assert torch.equal(four[key][k], eight[key][k]), (key, k)
        torch.testing.assert_close(eight[key]['db'], four[key]['db'], rtol=1e-5, atol=1e-5)

"""GPU parity tests: HIP kernels (through the C ABI via ops.py) vs the golden vectors captured from
the reference and vs the CPU oracle on seeded inputs.  Run with ``-m gpu`` on an MI355X."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import dic_oracle as O
from oracle.synth import latent_blobs, vitals_stack

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
INTERP = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, 'interp_*.npz')))
RBF = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, 'rbf_*.npz')))
DEC = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, 'dec_K*.npz')))

# fp32 tolerances (north_star: losses within 1e-5 relative; element-wise tensors get a small absolute floor
# because the kernels use v_exp_f32 / a different summation order than ATen)
RT, AT = 3e-5, 3e-5


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from deep_interpolation_clustering_amd import ops as _ops
    return _ops


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def G(a, grad=False, dtype=torch.float32):
    return torch.tensor(a, dtype=dtype, device='cuda', requires_grad=grad)


def prefix_lengths(x, C):
    """(B,C) int32 lengths if every mask row is a prefix of ones, else None."""
    m = x[:, C:2 * C]
    n = m.sum(-1).astype(np.int32)
    ok = (m == (np.arange(m.shape[-1])[None, None] < n[..., None])).all()
    return n if ok else None


# ------------------------------------------------------------------------------------ k1 golden
@pytest.mark.parametrize('name', INTERP)
def test_sci_cci_golden(ops, name):
    g = load(name)
    R, H = int(g['R']), float(g['H'])
    C = g['sci_kernel'].shape[0]
    grid = ops.ref_grid(H, R, 'cuda')
    x = G(g['x'])
    ks, kc = G(g['sci_kernel'], True), G(g['cci_kernel'], True)
    s = ops.sci_only(x, ks, grid)
    o = ops.sci_cci(x, ks, kc, grid)
    o2 = ops.cci(s, kc)
    np.testing.assert_allclose(s.detach().cpu().numpy(), g['sci_out'], rtol=RT, atol=AT, equal_nan=True)
    np.testing.assert_allclose(o.detach().cpu().numpy(), g['cci_out'], rtol=RT, atol=AT, equal_nan=True)
    np.testing.assert_allclose(o2.detach().cpu().numpy(), g['cci_out'], rtol=RT, atol=AT, equal_nan=True)
    lens = prefix_lengths(g['x'], C)
    if lens is not None:       # prefix-length fast path == mask path, bit for bit
        o3 = ops.sci_cci(x, ks, kc, grid, lengths=G(lens, dtype=torch.int32))
        assert torch.equal(torch.nan_to_num(o3, nan=7.0), torch.nan_to_num(o, nan=7.0))
    if 'edge' in name:
        return
    (o * G(g['cot'])).sum().backward()
    sc = np.abs(g['g_sci']).max()
    np.testing.assert_allclose(ks.grad.cpu().numpy(), g['g_sci'], rtol=2e-4, atol=2e-4 * sc)
    np.testing.assert_allclose(kc.grad.cpu().numpy(), g['g_cci'], rtol=2e-4, atol=2e-4 * np.abs(g['g_cci']).max())
    # unfused composition cci(sci(x)) gives the same parameter gradients
    ks2, kc2 = G(g['sci_kernel'], True), G(g['cci_kernel'], True)
    (ops.cci(ops.sci_only(x, ks2, grid), kc2) * G(g['cot'])).sum().backward()
    np.testing.assert_allclose(ks2.grad.cpu().numpy(), ks.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * sc)
    np.testing.assert_allclose(kc2.grad.cpu().numpy(), kc.grad.cpu().numpy(), rtol=1e-4, atol=1e-5 * np.abs(g['g_cci']).max())


def test_sci_cci_ragged_packed_matches_dense(ops):
    from deep_interpolation_clustering_amd import _native as N
    B, C, T, R, H = 37, 6, 96, 24, 24
    x, n = vitals_stack(11, B, C, T, H, 50)
    grid = ops.ref_grid(H, R, 'cuda')
    ks = G(np.linspace(-0.3, 1.1, C).astype(np.float32))
    kc = G((np.eye(C) + 0.1 * np.random.default_rng(3).normal(size=(C, C))).astype(np.float32))
    dense = ops.sci_cci(G(x), ks, kc, grid, lengths=G(n, dtype=torch.int32))
    off = np.zeros(B * C + 1, np.int64)
    off[1:] = np.cumsum(n.reshape(-1))
    tp = np.concatenate([x[b, 2 * C + c, :n[b, c]] for b in range(B) for c in range(C)])
    vp = np.concatenate([x[b, c, :n[b, c]] for b in range(B) for c in range(C)])
    out = torch.empty_like(dense)
    tpk, vpk, offd = G(tp), G(vp), torch.tensor(off, device='cuda')
    N.check(N.lib().dic_sci_cci_fwd_ragged(N.ptr(tpk), N.ptr(vpk), N.ptr(offd), int(n.max()), B, C, R, N.ptr(grid),
                                           N.ptr(ks), N.ptr(kc), N.ptr(out), None, N.stream_of(out)), 'ragged')
    assert torch.equal(out, dense)


@pytest.mark.parametrize('shape', [(64, 6, 96, 24, 24, 50), (300, 6, 354, 6, 6, 60), (33, 12, 288, 24, 24, 200),
                                   (1, 6, 96, 24, 24, 50), (517, 3, 40, 9, 12, 20), (5, 16, 64, 64, 24, 30)])
def test_sci_cci_vs_oracle(ops, shape):
    B, C, T, R, H, lam = shape
    x, n = vitals_stack(5 + B, B, C, T, H, lam)
    rng = np.random.default_rng(B)
    ks_np = rng.uniform(-0.5, 1.5, C).astype(np.float32)
    kc_np = (np.eye(C) + rng.normal(0, 0.2, (C, C))).astype(np.float32)
    cot_np = rng.normal(0, 1, (B, R, 3 * C)).astype(np.float32)
    grid = ops.ref_grid(H, R, 'cuda')
    ks, kc = G(ks_np, True), G(kc_np, True)
    out = ops.sci_cci(G(x), ks, kc, grid, lengths=G(n, dtype=torch.int32))
    (out * G(cot_np)).sum().backward()
    # oracle in fp64: the kernel must be at least as close to it as fp32 rounding allows
    x64 = torch.tensor(x, dtype=torch.float64)
    k1, k2 = torch.tensor(ks_np, dtype=torch.float64, requires_grad=True), torch.tensor(kc_np, dtype=torch.float64, requires_grad=True)
    ref = O.sci_cci_forward(x64, k1, k2, R, H)
    (ref * torch.tensor(cot_np, dtype=torch.float64)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=RT, atol=AT)
    sc = float(k1.grad.abs().max())
    np.testing.assert_allclose(ks.grad.cpu().numpy(), k1.grad.numpy(), rtol=2e-4, atol=2e-4 * sc)
    np.testing.assert_allclose(kc.grad.cpu().numpy(), k2.grad.numpy(), rtol=2e-4, atol=2e-4 * float(k2.grad.abs().max()))


@pytest.mark.parametrize('B,T,lam,packed', [(1, 30, 9, False), (37, 96, 80, True), (1027, 96, 50, True), (130, 64, 40, False), (9, 100, 3, True)])
def test_sci_cci_forward_reference_shape_corner_cases(ops, B, T, lam, packed):
    """The reference's own shape (C = 6, R = 24, prefix masks): rows longer than one 64-slot chunk next to short ones, an empty channel
    (NaN / -inf exactly as upstream), odd batch sizes, both output layouts and the saved moments (through the parameter
    gradients) -- against the fp64 oracle."""
    C, R, H = 6, 24, 24.0
    x, n = vitals_stack(500 + B, B, C, T, H, lam)
    if B > 8:
        n[3, 2] = 0                                                  # an empty channel
        for pl in range(4):
            x[3, pl * C + 2, :] = 0.0
        n[5, 1] = min(T, int(n[5, 1]) + 30)                          # a long row next to short ones
        k = int(n[5, 1])
        rng5 = np.random.default_rng(1)
        x[5, C + 1, :k] = 1.0
        x[5, 2 * C + 1, :k] = np.sort(rng5.uniform(0, H, k)).astype(np.float32)
        x[5, 1, :k] = rng5.normal(0, 1.2, k).astype(np.float32)
    rng = np.random.default_rng(B)
    ks_np = rng.uniform(-0.5, 1.5, C).astype(np.float32)
    kc_np = (np.eye(C) + rng.normal(0, 0.2, (C, C))).astype(np.float32)
    cot_np = rng.normal(0, 1, (B, R, 3 * C)).astype(np.float32)
    grid = ops.ref_grid(H, R, 'cuda')
    ks, kc = G(ks_np, True), G(kc_np, True)
    lens = G(n, dtype=torch.int32)
    x64 = torch.tensor(x, dtype=torch.float64)
    k1, k2 = torch.tensor(ks_np, dtype=torch.float64, requires_grad=True), torch.tensor(kc_np, dtype=torch.float64, requires_grad=True)
    ref = O.sci_cci_forward(x64, k1, k2, R, H)
    refn = ref.detach().numpy()
    if packed:
        rows = ops.sci_cci_packed(G(x), ks, kc, grid, lens)                     # (R,B,32) bf16 [features | 1 | 0...]
        got = rows[:, :, :3 * C].float().permute(1, 0, 2).detach().cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(refn))
        np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(refn), rtol=1e-2, atol=1e-2)      # bf16 rows
        tail = rows[:, :, 3 * C:].float().detach().cpu().numpy()
        assert (tail[:, :, 0] == 1).all() and (tail[:, :, 1:] == 0).all()
        return
    out = ops.sci_cci(G(x), ks, kc, grid, lengths=lens)
    assert np.array_equal(np.isnan(out.detach().cpu().numpy()), np.isnan(refn))
    np.testing.assert_allclose(np.nan_to_num(out.detach().cpu().numpy()), np.nan_to_num(refn), rtol=RT, atol=AT)
    if B > 8:
        return                                                       # (an empty channel makes the parameter gradients NaN, upstream too)
    (out * G(cot_np)).sum().backward()
    (ref * torch.tensor(cot_np, dtype=torch.float64)).sum().backward()
    np.testing.assert_allclose(ks.grad.cpu().numpy(), k1.grad.numpy(), rtol=2e-4, atol=2e-4 * float(k1.grad.abs().max()))
    np.testing.assert_allclose(kc.grad.cpu().numpy(), k2.grad.numpy(), rtol=2e-4, atol=2e-4 * float(k2.grad.abs().max()))


@pytest.mark.parametrize('B,R,packed', [(1, 24, False), (2, 24, True), (37, 24, True), (1027, 24, False), (1027, 24, True), (130, 6, False), (9, 11, True),
                                        (64, 32, True)])
def test_sci_cci_backward_lane_kernel_equals_tile_kernel(B, R, packed, monkeypatch):
    """sci_cci_bwd_lane_kernel (C = 6: a grid point per lane, half a wave per encounter; round 4) against the tile kernel (DIC_K1_BWD_LANES=0) on the same
    saved planes and incoming gradient -- odd batch (a half-empty last pair), R below / at the 32 lanes of a half, the f32 (B,R,3C) and the packed bf16
    (R,B,32) gradient layouts: dL/dkernel of both layers to summation order."""
    from deep_interpolation_clustering_amd import _native as N
    L = N.lib()
    C = 6
    torch.manual_seed(B * 31 + R)
    dev = 'cuda'
    saved = torch.randn(B, 7, C, R, device=dev)
    saved[:, 3:] = saved[:, 3:].abs()
    sk, ck = torch.randn(C, device=dev) * 0.5, torch.randn(C, C, device=dev) * 0.4
    g32 = torch.randn(B, R, 3 * C, device=dev)
    gp = torch.zeros(R, B, 32, device=dev, dtype=torch.bfloat16)
    gp[:, :, :3 * C] = g32.permute(1, 0, 2).to(torch.bfloat16)
    if packed:
        g32 = gp[:, :, :3 * C].float().permute(1, 0, 2).contiguous()          # (the same bf16-rounded values through the f32 entry point as reference)
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('DIC_K1_BWD_LANES', mode)
        ws = torch.empty(max(16, L.dic_sci_cci_bwd_workspace(B, C, R)), dtype=torch.uint8, device=dev)
        gs, gc = torch.zeros(C, device=dev), torch.zeros(C, C, device=dev)
        if packed:
            N.check(L.dic_sci_cci_bwd_packed(N.ptr(gp), 32, N.ptr(saved), N.ptr(sk), N.ptr(ck), B, C, R, N.ptr(gs), N.ptr(gc), N.ptr(ws), ws.numel(),
                                             N.stream_of(gp)), 'bwd_packed')
        else:
            N.check(L.dic_sci_cci_bwd(N.ptr(g32), N.ptr(saved), N.ptr(sk), N.ptr(ck), B, C, R, N.ptr(gs), N.ptr(gc), N.ptr(ws), ws.numel(), N.stream_of(g32)), 'bwd')
        res[mode] = (gs.clone(), gc.clone())
    for a, b in zip(res['1'], res['0']):
        assert torch.isfinite(a).all()
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5 * float(b.abs().max()))


# ------------------------------------------------------------------------------------ k2 golden
@pytest.mark.parametrize('name', RBF)
def test_rbf_golden(ops, name):
    g = load(name)
    R, H = int(g['R']), float(g['H'])
    C = g['kernel'].shape[0]
    grid = ops.ref_grid(H, R, 'cuda')
    x = G(g['x'])
    v, k = G(g['v'], True), G(g['kernel'], True)
    y = ops.rbf_deinterp(v, x, k, grid)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g['y'], rtol=RT, atol=3e-6)
    mask = x[:, C:2 * C]
    loss = ops.masked_mse(G(g['ob']), y, mask)
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(v.grad.cpu().numpy(), g['g_v'], rtol=2e-4, atol=2e-6 * np.abs(g['g_v']).max() + 1e-9)
    np.testing.assert_allclose(k.grad.cpu().numpy(), g['g_kernel'], rtol=3e-4, atol=3e-5 * np.abs(g['g_kernel']).max())
    lens = prefix_lengths(g['x'], C)
    if lens is not None:
        ld = G(lens, dtype=torch.int32)
        v2, k2 = G(g['v'], True), G(g['kernel'], True)
        y2 = ops.rbf_deinterp(v2, x, k2, grid, lengths=ld)
        assert torch.equal(y2, y)
        l2 = ops.masked_mse(G(g['ob']), y2, None, lengths=ld)
        np.testing.assert_allclose(float(l2), float(loss), rtol=1e-6)
        l2.backward()
        np.testing.assert_allclose(v2.grad.cpu().numpy(), v.grad.cpu().numpy(), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(k2.grad.cpu().numpy(), k.grad.cpu().numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('slot_mode', ['0', '2'])
@pytest.mark.parametrize('shape', [(64, 6, 96, 24, 24, 50), (130, 6, 354, 6, 6, 60), (9, 12, 288, 24, 24, 200), (1, 6, 30, 11, 6, 9), (70, 3, 50, 20, 12, 30)])
def test_rbf_vs_oracle(ops, shape, slot_mode, monkeypatch):
    monkeypatch.setenv('DIC_RBF_BWD_SLOT', slot_mode)         # (0: tile / wave-per-encounter kernels; 2: slots-on-lanes)
    B, C, T, R, H, lam = shape
    x, n = vitals_stack(77 + B, B, C, T, H, lam)
    rng = np.random.default_rng(B)
    v_np = rng.normal(0, 1, (B, C, R)).astype(np.float32)
    k_np = rng.uniform(-0.5, 1.5, C).astype(np.float32)
    cot = rng.normal(0, 1, (B, C, T)).astype(np.float32)
    grid = ops.ref_grid(H, R, 'cuda')
    v, k = G(v_np, True), G(k_np, True)
    y = ops.rbf_deinterp(v, G(x), k, grid, lengths=G(n, dtype=torch.int32))
    (y * G(cot)).sum().backward()
    v64 = torch.tensor(v_np, dtype=torch.float64, requires_grad=True)
    k64 = torch.tensor(k_np, dtype=torch.float64, requires_grad=True)
    ref = O.rbf_deinterp(v64, torch.tensor(x, dtype=torch.float64), k64, R, H)
    (ref * torch.tensor(cot, dtype=torch.float64)).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().numpy(), rtol=RT, atol=3e-6)
    np.testing.assert_allclose(v.grad.cpu().numpy(), v64.grad.numpy(), rtol=2e-4, atol=2e-5 * float(v64.grad.abs().max()))
    np.testing.assert_allclose(k.grad.cpu().numpy(), k64.grad.numpy(), rtol=3e-4, atol=3e-5 * float(k64.grad.abs().max()))


@pytest.mark.parametrize('B,T,lam,time_major,prefix_only', [(1, 30, 9, False, False), (37, 96, 80, True, True), (1027, 96, 50, True, True),
                                                           (130, 64, 40, False, True), (9, 100, 3, True, False)])
@pytest.mark.parametrize('slot_mode', ['1', '2'])
def test_rbf_backward_reference_shape_kernel(ops, B, T, lam, time_major, prefix_only, slot_mode, monkeypatch):
    """The wave-per-encounter backward (C = 6, R = 24, prefix masks; slot_mode 2: the slots-on-lanes kernel that serves every other shape, forced
    onto this one): rows longer than one 64-slot chunk, empty rows, a batch that is not a multiple of the wave count, v / grad_v in CompressFC's
    (R,B,C) row order, and padding the forward never wrote (prefix_only: poisoned with NaN here) -- against the fp64 oracle."""
    monkeypatch.setenv('DIC_RBF_BWD_SLOT', slot_mode)
    C, R, H = 6, 24, 24.0
    x, n = vitals_stack(300 + B, B, C, T, H, lam)
    if B > 8:
        n[3, 2] = 0                                                  # an empty channel
        x[3, 2, :] = 0.0; x[3, C + 2, :] = 0.0; x[3, 2 * C + 2, :] = 0.0; x[3, 3 * C + 2, :] = 0.0
        n[5, 1] = min(T, int(n[5, 1]) + 30)                          # a long row next to short ones
        x[5, C + 1, :n[5, 1]] = 1.0
        x[5, 2 * C + 1, :n[5, 1]] = np.sort(np.random.default_rng(1).uniform(0, H, n[5, 1])).astype(np.float32)
    rng = np.random.default_rng(B)
    v_np = rng.normal(0, 1, (B, C, R)).astype(np.float32)
    k_np = rng.uniform(-0.5, 1.5, C).astype(np.float32)
    cot = rng.normal(0, 1, (B, C, T)).astype(np.float32)
    grid = ops.ref_grid(H, R, 'cuda')
    k = G(k_np, True)
    if time_major:
        v_rows = G(np.ascontiguousarray(v_np.transpose(2, 0, 1)), True)          # (R,B,C) leaf; the op sees the permuted view
        v = v_rows.permute(1, 2, 0)
    else:
        v_rows = v = G(v_np, True)
    lens = G(n, dtype=torch.int32)
    y = ops.rbf_deinterp(v, G(x), k, grid, lengths=lens, prefix_only=prefix_only)
    keep = torch.arange(T, device='cuda')[None, None, :] < lens[:, :, None]
    g = torch.where(keep, G(cot), torch.full_like(y, float('nan')) if prefix_only else torch.zeros_like(y))
    y.backward(g)
    v64 = torch.tensor(v_np, dtype=torch.float64, requires_grad=True)
    k64 = torch.tensor(k_np, dtype=torch.float64, requires_grad=True)
    ref = O.rbf_deinterp(v64, torch.tensor(x, dtype=torch.float64), k64, R, H)
    (ref * torch.tensor(cot, dtype=torch.float64)).sum().backward()
    gv = v_rows.grad.permute(1, 2, 0) if time_major else v_rows.grad
    np.testing.assert_allclose(gv.cpu().numpy(), v64.grad.numpy(), rtol=2e-4, atol=2e-5 * float(v64.grad.abs().max()))
    np.testing.assert_allclose(k.grad.cpu().numpy(), k64.grad.numpy(), rtol=3e-4, atol=3e-5 * float(k64.grad.abs().max()))


@pytest.mark.parametrize('B,C,T,R,H,lam,time_major', [(1027, 6, 96, 24, 24.0, 50, True), (37, 6, 96, 24, 24.0, 80, False), (33, 12, 288, 24, 24.0, 200, True),
                                                       (5, 3, 40, 9, 12.0, 20, False)])
@pytest.mark.parametrize('slot_mode', ['0', '1', '2'])
def test_rbf_with_fused_reconstruction_loss_matches_separate_kernels(ops, B, C, T, R, H, lam, time_major, slot_mode, monkeypatch):
    """ops.rbf_rec_loss (k2 forward emitting the masked SSE, k2 backward forming dL/dy from the observations) against
    rbf_deinterp + masked_mse: same reconstruction on the observed slots, same loss and same gradients (to summation order).  slot_mode: which
    backward kernels serve the shape (DIC_RBF_BWD_SLOT: 0 = tile / wave-per-encounter only, 1 = default, 2 = slots-on-lanes wherever lengths are given)."""
    monkeypatch.setenv('DIC_RBF_BWD_SLOT', slot_mode)
    x, n = vitals_stack(900 + B, B, C, T, H, lam)
    rng = np.random.default_rng(B)
    v_np = rng.normal(0, 1, (B, C, R)).astype(np.float32)
    k_np = rng.uniform(-0.5, 1.5, C).astype(np.float32)
    ob_np = (rng.normal(0, 1, (B, C, T)) * x[:, C:2 * C]).astype(np.float32)
    grid = ops.ref_grid(H, R, 'cuda')
    lens, xg, ob = G(n, dtype=torch.int32), G(x), G(ob_np)
    res = {}
    for fused in (False, True):
        k = G(k_np, True)
        if time_major:
            leaf = G(np.ascontiguousarray(v_np.transpose(2, 0, 1)), True)
            v = leaf.permute(1, 2, 0)
        else:
            leaf = v = G(v_np, True)
        if fused:
            y, mse = ops.rbf_rec_loss(v, xg, k, grid, lens, ob)
        else:
            y = ops.rbf_deinterp(v, xg, k, grid, lengths=lens, prefix_only=True)
            mse = ops.masked_mse(ob, y, None, lengths=lens, prefix_only=True)
        (3.0 * mse).backward()
        keep = torch.arange(T, device='cuda')[None, None, :] < lens[:, :, None]
        res[fused] = dict(y=torch.where(keep, y.detach(), torch.zeros_like(y)), mse=mse.detach(),
                          gv=(leaf.grad.permute(1, 2, 0) if time_major else leaf.grad).clone(), gk=k.grad.clone())
    assert torch.equal(res[True]['y'], res[False]['y'])
    np.testing.assert_allclose(float(res[True]['mse']), float(res[False]['mse']), rtol=2e-6)
    for key, tol in (('gv', 2e-5), ('gk', 1e-4)):
        a, b = res[True][key].cpu().numpy(), res[False][key].cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=tol, atol=tol * np.abs(b).max(), err_msg=key)


@pytest.mark.parametrize('B,C,T,R,H,lam,time_major', [(1027, 6, 96, 24, 24.0, 50, True), (37, 6, 354, 6, 6.0, 80, False), (33, 12, 288, 24, 24.0, 200, True),
                                                       (5, 3, 40, 9, 12.0, 20, False), (70, 6, 64, 24, 24.0, 70, False)])
def test_rbf_forward_row_kernel_equals_tile_kernel(ops, B, C, T, R, H, lam, time_major, monkeypatch):
    """rbf_fwd_row_kernel (prefix masks: a row per wave, round 4) against the tile kernel (DIC_RBF_FWD_ROW=0): the reconstruction bit for bit -- written
    slots, zeros (or, prefix_only, the caller's bytes) in the padding --, an empty row, rows of exactly T slots; the fused loss to summation order."""
    x, n = vitals_stack(500 + B, B, C, T, H, lam)
    if B > 8:
        n[3, 1] = 0
        x[3, 1, :] = 0.0; x[3, C + 1, :] = 0.0; x[3, 2 * C + 1, :] = 0.0
        n[4, 0] = T
        x[4, C, :] = 1.0
        x[4, 2 * C, :] = np.sort(np.random.default_rng(2).uniform(0, H, T)).astype(np.float32)
    rng = np.random.default_rng(B)
    v_np = rng.normal(0, 1, (B, C, R)).astype(np.float32)
    k = G(rng.uniform(-0.5, 1.5, C).astype(np.float32))
    ob = G((rng.normal(0, 1, (B, C, T)) * x[:, C:2 * C]).astype(np.float32))
    v = G(np.ascontiguousarray(v_np.transpose(2, 0, 1))).permute(1, 2, 0) if time_major else G(v_np)
    grid = ops.ref_grid(H, R, 'cuda')
    lens, xg = G(n, dtype=torch.int32), G(x)
    keep = torch.arange(T, device='cuda')[None, None, :] < lens[:, :, None]
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('DIC_RBF_FWD_ROW', mode)
        y_full = ops.rbf_deinterp(v, xg, k, grid, lengths=lens)
        y_pref = ops.rbf_deinterp(v, xg, k, grid, lengths=lens, prefix_only=True)
        y_loss, mse = ops.rbf_rec_loss(v, xg, k, grid, lens, ob)
        res[mode] = (y_full, torch.where(keep, y_pref, torch.zeros_like(y_pref)), torch.where(keep, y_loss, torch.zeros_like(y_loss)), float(mse))
    for j in range(3):
        assert torch.equal(res['0'][j], res['1'][j]), j
    assert float(res['1'][0][~keep].abs().max()) == 0.0
    np.testing.assert_allclose(res['1'][3], res['0'][3], rtol=2e-6)


# ------------------------------------------------------------------------------------ k3 golden
@pytest.mark.parametrize('name', DEC)
def test_dec_golden(ops, name):
    g = load(name)
    z, mu = G(g['z'], True), G(g['mu'], True)
    q, colsum = ops.dec_soft_assign(z, mu, 1.0, return_colsum=True)
    p = ops.dec_target(q, colsum)
    np.testing.assert_allclose(q.detach().cpu().numpy(), g['q'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(colsum.cpu().numpy(), g['q'].sum(0), rtol=1e-5)
    np.testing.assert_allclose(p.cpu().numpy(), g['p'], rtol=2e-5, atol=1e-7)
    kl = ops.kl_batchmean(p, q)
    np.testing.assert_allclose(float(kl), float(g['kl']), rtol=1e-5, atol=1e-8)
    kl.backward()
    np.testing.assert_allclose(z.grad.cpu().numpy(), g['g_z'], rtol=2e-4, atol=2e-6 * np.abs(g['g_z']).max())
    np.testing.assert_allclose(mu.grad.cpu().numpy(), g['g_mu'], rtol=2e-4, atol=2e-5 * np.abs(g['g_mu']).max())
    # predicted label = argmax q = argmin distance (clustering_trainer.py:477): exact
    assert (q.argmax(1).cpu().numpy() == g['q'].argmax(1)).all()


@pytest.mark.parametrize('B,D,K,alpha', [(1000, 256, 4, 1.0), (777, 256, 20, 1.0), (64, 128, 3, 2.5), (40, 256, 32, 1.0), (4099, 64, 7, 0.5)])
def test_dec_vs_oracle(ops, B, D, K, alpha):
    X, _ = latent_blobs(B + K, B, D, K, spread=0.3, noise=0.2)
    rng = np.random.default_rng(K)
    mu_np = (X[rng.choice(B, K, replace=False)] + rng.normal(0, 0.05, (K, D))).astype(np.float32)
    cot = rng.normal(0, 1, (B, K)).astype(np.float32)
    z, mu = G(X, True), G(mu_np, True)
    q = ops.dec_soft_assign(z, mu, alpha)
    (q * G(cot)).sum().backward()
    z64 = torch.tensor(X, dtype=torch.float64, requires_grad=True)
    m64 = torch.tensor(mu_np, dtype=torch.float64, requires_grad=True)
    ref = O.dec_soft_assign(z64, m64, alpha)
    (ref * torch.tensor(cot, dtype=torch.float64)).sum().backward()
    np.testing.assert_allclose(q.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(z.grad.cpu().numpy(), z64.grad.numpy(), rtol=2e-4, atol=2e-6 * float(z64.grad.abs().max()))
    np.testing.assert_allclose(mu.grad.cpu().numpy(), m64.grad.numpy(), rtol=2e-4, atol=2e-5 * float(m64.grad.abs().max()))
    p = ops.dec_target(q.detach())
    np.testing.assert_allclose(p.cpu().numpy(), O.dec_target(ref.detach()).numpy(), rtol=5e-5, atol=1e-7)


# ------------------------------------------------------------------------------------ k4
MARGIN = 1e-6      # a label may differ from the fp64 argmin only below this relative fp64 margin (fp32 resolution of ||c||^2 - 2 x.c)


def _margin_of(X, centers, rows, lab_a, lab_b):
    """fp64 relative gap between the squared distances of ``rows`` to the two centres ``lab_a`` / ``lab_b`` name."""
    x = X[rows].astype(np.float64)
    c = centers.astype(np.float64)
    da = ((x - c[lab_a]) ** 2).sum(-1)
    db = ((x - c[lab_b]) ** 2).sum(-1)
    return np.abs(da - db) / np.maximum(np.maximum(da, db), 1e-30)


def _fp64_argmin(X, centers, chunk=4096):
    c = centers.astype(np.float64)
    out = np.empty(X.shape[0], dtype=np.int64)
    for lo in range(0, X.shape[0], chunk):
        x = X[lo:lo + chunk].astype(np.float64)
        out[lo:lo + chunk] = ((x * x).sum(1)[:, None] - 2.0 * x @ c.T + (c * c).sum(1)[None]).argmin(1)
    return out


def _label_audit(X, centers, got, want, margin=MARGIN):
    """``got`` must equal ``want`` except where the fp64 margin between the two centres in question (measured against
    ``centers``, the centres the E-step under audit actually used) is below fp32 resolution.  Returns #sub-resolution flips."""
    bad = np.nonzero(np.asarray(got) != np.asarray(want))[0]
    if bad.size == 0:
        return 0
    m = _margin_of(X, centers, bad, np.asarray(got)[bad], np.asarray(want)[bad])
    assert (m < margin).all(), f'{bad.size} label mismatches, {int((m >= margin).sum())} with a decisive fp64 margin (max {m.max():.3e})'
    return bad.size


def _one_thread():
    from threadpoolctl import threadpool_limits
    return threadpool_limits(limits=1)     # scikit-learn's per-thread partial sums: one reduction order on every host


@pytest.mark.parametrize('K', [4, 16])
def test_kmeans_golden_fixed_init(K):
    from deep_interpolation_clustering_amd.kmeans import KMeans
    g = load(f'kmeans_K{K}.npz')
    X, _ = latent_blobs(int(g['seed']), int(g['N']), int(g['D']), K)
    km = KMeans(n_clusters=K, init=X[g['init_idx']].copy(), n_init=1).fit(X)
    assert (km.labels_ == g['labels']).all()          # bit-exact assignments for a fixed init
    np.testing.assert_allclose(km.cluster_centers_, g['centers'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(km.inertia_, float(g['inertia']), rtol=1e-5)
    assert km.n_iter_ == int(g['n_iter'])
    Xv, _ = latent_blobs(int(g['seed']) + 1, 500, int(g['D']), K, centers_seed=int(g['seed']))
    assert (km.predict(Xv) == g['pred']).all()


SEEDED = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, 'kmeans_seeded_*.npz')))


@pytest.mark.parametrize('name', SEEDED)
def test_kmeans_seeded_plusplus_golden(name):
    """The reference's actual call (clustering_trainer.py:75-76 after utils.set_seed, utils.py:37-42):
    ``np.random.seed(7529); KMeans(n_clusters=K, n_init=20).fit(X)``.  The k-means++ seeding replays NumPy's global
    stream in scikit-learn's draw order, so the SAME restarts are run and the same one wins: labels bit-exact
    (oracle/make_golden_kmeans.py wrote the fixture from scikit-learn 1.7.2)."""
    from deep_interpolation_clustering_amd.kmeans import KMeans
    g = load(name)
    K = int(g['K'])
    X, _ = latent_blobs(int(g['seed']), int(g['N']), int(g['D']), int(g['blobs']), spread=float(g['spread']), noise=float(g['noise']))
    np.random.seed(7529)
    km = KMeans(n_clusters=K, n_init=20).fit(X)
    after = np.random.random_sample(4)
    assert (after == g['stream_after']).all()          # the fit consumed exactly the random numbers scikit-learn consumes
    assert (km.labels_ == g['labels']).all()           # bit-exact cluster assignments for the fixed seed
    np.testing.assert_allclose(km.cluster_centers_, g['centers'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(km.inertia_, float(g['inertia']), rtol=1e-5)
    assert km.n_iter_ == int(g['n_iter'])


@pytest.mark.parametrize('N,D,K', [(20000, 256, 8), (5000, 256, 2), (3001, 64, 20), (1500, 6, 3), (900, 256, 32)])
def test_kmeans_vs_sklearn_fixed_init(N, D, K):
    """K separated blobs, fixed init, scikit-learn run live on one thread: the same optimum, bit-exact labels
    (margin-audited against the centres both final E-steps used), same inertia, same iteration count."""
    from sklearn.cluster import KMeans as SK
    from deep_interpolation_clustering_amd.kmeans import KMeans
    X, _ = latent_blobs(N + K, N, D, K, spread=0.3, noise=0.3)
    init = X[np.random.default_rng(K).choice(N, K, replace=False)].copy()
    with _one_thread():
        ref = SK(n_clusters=K, init=init, n_init=1).fit(X)
    km = KMeans(n_clusters=K, init=init, n_init=1).fit(X)
    flips = _label_audit(X, km.cluster_centers_, km.labels_, ref.labels_)
    np.testing.assert_allclose(km.inertia_, ref.inertia_, rtol=2e-5)
    np.testing.assert_allclose(km.cluster_centers_, ref.cluster_centers_, rtol=1e-4, atol=1e-5)
    assert km.n_iter_ == ref.n_iter_ or flips > 0
    Xv, _ = latent_blobs(N + K + 1, 777, D, K, centers_seed=N + K, spread=0.3, noise=0.3)
    with _one_thread():
        want = ref.predict(Xv)
    _label_audit(Xv, ref.cluster_centers_, km.predict(Xv), want)


def _lloyd_stages(X, c):
    """One Lloyd iteration of the HIP path from centres ``c``, stage by stage through the C ABI, exactly as ``KMeans.fit``
    drives it: centred data -> dic_kmeans_lloyd_iter (E-step labels + M-step centres) -> dic_kmeans_predict (final E-step).
    Returns the centred data / centres the kernels saw and what each stage produced (host arrays)."""
    from deep_interpolation_clustering_amd import _native as N
    from deep_interpolation_clustering_amd import kmeans as KM
    L = N.lib()
    Xd = torch.tensor(X, device='cuda')
    mean = Xd.mean(dim=0)
    Xc = (Xd - mean).contiguous()
    xnorm = (Xc * Xc).sum(dim=1)
    c0 = (torch.tensor(c, device='cuda') - mean)[None].contiguous()
    n, D = Xc.shape
    K = c0.shape[1]
    centers = c0.clone()
    labels = torch.full((1, n), -1, dtype=torch.int32, device='cuda')
    status = torch.zeros((1, N.KM_STATUS_WORDS), dtype=torch.float32, device='cuda')
    status[:, 7] = 300.0
    ws = KM._ws(L.dic_kmeans_workspace(n, D, K, 1), Xc.device)
    st = N.stream_of(Xc)
    N.check(L.dic_kmeans_lloyd_iter(N.ptr(Xc), N.ptr(xnorm), n, D, K, 1, N.ptr(centers), N.ptr(labels), N.ptr(status), N.ptr(ws),
                                    ws.numel(), st), 'dic_kmeans_lloyd_iter')
    e_labels = labels[0].cpu().numpy().copy()
    m_centers = centers[0].cpu().numpy().copy()
    inertia = torch.empty(1, dtype=torch.float32, device='cuda')
    N.check(L.dic_kmeans_predict(N.ptr(Xc), n, D, K, 1, N.ptr(centers), N.ptr(labels), None, N.ptr(inertia), N.ptr(ws), ws.numel(), st),
            'dic_kmeans_predict')
    return dict(Xc=Xc.cpu().numpy(), c0=c0[0].cpu().numpy(), e_labels=e_labels, m_centers=m_centers,
                f_labels=labels[0].cpu().numpy(), inertia=float(inertia), mean=mean.cpu().numpy())


def test_kmeans_single_step_parity_on_tie_prone_data():
    """More clusters than blobs: Lloyd wanders along flat directions and whole-fit equality is not a meaningful bar (one
    sub-resolution tie flip moves two centres by |x-c|/n and every later label near their bisector with them).  The STEP is
    what must be right, and each of its stages is audited against ITS OWN inputs, so nothing depends on the host:
      (i)   E-step: ``predict`` from scikit-learn's pinned centres after i iterations (tests/golden/kmeans_step_K8.npz) against
            the fp64 argmin and against scikit-learn's own E-step labels -- differences only below fp32 resolution;
      (ii)  M-step: the centres the HIP iteration produces = the fp64 mean of the points ITS E-step assigned to each cluster;
      (iii) final E-step: labels against the fp64 argmin over the centres the HIP path itself produced in (ii).
    ``KMeans(init=c, max_iter=1, tol=0).fit`` must return exactly the stage-(iii) result."""
    from deep_interpolation_clustering_amd.kmeans import KMeans
    g = load('kmeans_step_K8.npz')
    N_, D, K = int(g['N']), int(g['D']), int(g['K'])
    X, _ = latent_blobs(int(g['seed']), N_, D, int(g['blobs']), spread=float(g['spread']), noise=float(g['noise']))
    flips = 0
    for i in range(len(g['its'])):
        c = g['centers'][i]
        # (i) E-step on the raw data, as KMeans.predict runs it (_kmeans.py:1066-1090: no centring)
        km = KMeans(n_clusters=K)
        km.cluster_centers_ = c
        pred = km.predict(X)
        flips += _label_audit(X, c, pred, _fp64_argmin(X, c))
        flips += _label_audit(X, c, pred, g['estep_labels'][i])
        s = _lloyd_stages(X, c)
        # (i') the same E-step inside the iteration (centred data, as fit runs it)
        flips += _label_audit(s['Xc'], s['c0'], s['e_labels'], _fp64_argmin(s['Xc'], s['c0']))
        # (ii) M-step against the fp64 means of the HIP E-step's own clusters
        want = np.stack([s['Xc'][s['e_labels'] == k].astype(np.float64).mean(0) for k in range(K)])
        np.testing.assert_allclose(s['m_centers'], want, rtol=0, atol=1e-6)
        # (iii) final E-step against the centres it used
        flips += _label_audit(s['Xc'], s['m_centers'], s['f_labels'], _fp64_argmin(s['Xc'], s['m_centers']))
        exact = ((s['Xc'].astype(np.float64) - s['m_centers'][s['f_labels']].astype(np.float64)) ** 2).sum()
        np.testing.assert_allclose(s['inertia'], exact, rtol=2e-6)
        # the estimator is those three stages
        b = KMeans(n_clusters=K, init=c, n_init=1, max_iter=1, tol=0).fit(X)
        assert (b.labels_ == s['f_labels']).all()
        np.testing.assert_array_equal(b.cluster_centers_, s['m_centers'] + s['mean'])
        # scikit-learn's own next iterate (pinned) = the fp64 means under OUR E-step labels, up to re-labelling points whose
        # fp64 margin is below fp32 resolution (its in-fit E-step and its own predict disagree on such a point at i = 14:
        # point 6488, margin 1.3e-7); nothing else may differ
        sk_next = g['next_centers'][i] - s['mean']
        x64, c64 = s['Xc'].astype(np.float64), s['c0'].astype(np.float64)
        d = (x64 * x64).sum(1)[:, None] - 2.0 * x64 @ c64.T + (c64 * c64).sum(1)[None]
        order = np.argsort(d, axis=1)[:, :2]
        d0, d1 = d[np.arange(N_), order[:, 0]], d[np.arange(N_), order[:, 1]]
        ties = np.nonzero((d1 - d0) / np.maximum(d1, 1e-30) < MARGIN)[0]
        assert ties.size <= 6
        explained = False
        for subset in range(1 << ties.size):
            lab = s['e_labels'].copy()
            for j, t in enumerate(ties):
                if subset >> j & 1:
                    lab[t] = order[t, 1] if lab[t] == order[t, 0] else order[t, 0]
            means = np.stack([x64[lab == k].mean(0) for k in range(K)])
            if np.abs(means - sk_next).max() < 5e-6:
                explained = True
                break
        assert explained, f'i={i}: scikit-learn\'s next centres are not the means of our E-step clusters (ties: {ties.tolist()})'
    assert flips <= 24       # sub-resolution near-ties only (each audited above); 20 000 points x 6 centres sets x 4 audits


def test_kmeans_plusplus_restarts_quality():
    """n_init=20 k-means++ against scikit-learn run live (one thread) from the same NumPy global seed: the same labels
    (margin-audited), inertia and iteration count; the bit-exact pin of this path is test_kmeans_seeded_plusplus_golden."""
    from sklearn.cluster import KMeans as SK
    from sklearn.metrics import adjusted_rand_score
    from deep_interpolation_clustering_amd.kmeans import KMeans
    X, comp = latent_blobs(99, 30000, 256, 4, spread=0.5, noise=0.25)
    np.random.seed(7529)
    with _one_thread():
        ref = SK(n_clusters=4, n_init=20).fit(X)
    np.random.seed(7529)
    km = KMeans(n_clusters=4, n_init=20).fit(X)
    _label_audit(X, km.cluster_centers_, km.labels_, ref.labels_)
    np.testing.assert_allclose(km.inertia_, ref.inertia_, rtol=1e-5)
    assert adjusted_rand_score(comp, km.labels_) > 0.99


def test_kmeans_empty_cluster_relocation():
    """An initial centre far from all data gets no points: it must be relocated to the farthest point (sklearn
    _relocate_empty_clusters_dense), and the fit must still match sklearn."""
    from sklearn.cluster import KMeans as SK
    from deep_interpolation_clustering_amd.kmeans import KMeans
    X, _ = latent_blobs(5, 4000, 256, 3, spread=0.4, noise=0.2)
    init = X[[0, 1, 2, 3]].copy()
    init[3] += 50.0
    with _one_thread():
        ref = SK(n_clusters=4, init=init, n_init=1).fit(X)
    km = KMeans(n_clusters=4, init=init, n_init=1).fit(X)
    assert km._status[0, 5] >= 1                      # a relocation happened
    _label_audit(X, km.cluster_centers_, km.labels_, ref.labels_)
    np.testing.assert_allclose(km.inertia_, ref.inertia_, rtol=2e-5)


def test_kmeans_full_size_properties():
    """75k x 256 (BASELINE size): idempotence of the E-step and optimality conditions of the fixed point."""
    from deep_interpolation_clustering_amd.kmeans import KMeans
    X, _ = latent_blobs(2024, 75000, 256, 4, spread=0.35, noise=0.3)
    np.random.seed(1)
    km = KMeans(n_clusters=4, n_init=3).fit(X)
    lab = km.predict(X)
    assert (lab == km.labels_).all()                                   # predict(fit data) == labels_
    cent = np.stack([X[lab == k].mean(0) for k in range(4)])
    np.testing.assert_allclose(cent, km.cluster_centers_, rtol=0, atol=5e-5)   # centres are the cluster means
    d = ((X[:2000, None] - km.cluster_centers_[None]) ** 2).sum(-1)
    assert (d.argmin(1) == lab[:2000]).all()
    np.testing.assert_allclose(km.inertia_, ((X - km.cluster_centers_[lab]) ** 2).sum(dtype=np.float64), rtol=1e-5)


@pytest.mark.parametrize('K,n_init', [(16, 10), (20, 3)])
def test_kmeans_full_size_many_clusters(K, n_init):
    """BASELINE configs[4]'s shape on the K > 8 path (exact-f32 MFMA assign kernel; 75 000 rows leave a ragged last row block and, at 10
    restarts of K = 16, a ragged last group of restarts): fixed-point properties at full size, and the same optimum as scikit-learn from the
    same single init (the reference's estimator, run live on the box's host cores)."""
    from sklearn.cluster import KMeans as SK
    from deep_interpolation_clustering_amd.kmeans import KMeans
    X, truth = latent_blobs(77, 75000, 256, K, spread=0.35, noise=0.3)
    np.random.seed(3)
    km = KMeans(n_clusters=K, n_init=n_init).fit(X)
    lab = km.predict(X)
    assert (lab == km.labels_).all() and len(np.unique(lab)) == K
    cent = np.stack([X[lab == k].mean(0) for k in range(K)])
    np.testing.assert_allclose(cent, km.cluster_centers_, rtol=0, atol=5e-5)          # centres are the cluster means
    d = ((X[:1500, None].astype(np.float64) - km.cluster_centers_[None].astype(np.float64)) ** 2).sum(-1)
    assert (d.argmin(1) == lab[:1500]).all()                                            # labels are the nearest centres
    np.testing.assert_allclose(km.inertia_, ((X - km.cluster_centers_[lab]) ** 2).sum(dtype=np.float64), rtol=1e-5)
    # (one seed point per blob: from a random init with two seeds in one blob Lloyd's path is chaotic -- near-tied points decide which
    #  local optimum is reached, and scikit-learn itself is not run-to-run reproducible there)
    init = np.stack([X[np.flatnonzero(truth == k)[0]] for k in range(K)])
    ref = SK(n_clusters=K, init=init, n_init=1).fit(X)
    one = KMeans(n_clusters=K, init=init, n_init=1).fit(X)
    _label_audit(X, one.cluster_centers_, one.labels_, ref.labels_)
    np.testing.assert_allclose(one.inertia_, ref.inertia_, rtol=2e-5)
    assert one.n_iter_ == ref.n_iter_


# ------------------------------------------------------------------ full BASELINE sizes: size-independent properties
def _device_vitals(B, C, T, H, lam, seed):
    """Stacked (B,4C,T) ragged input generated on the device (prefix masks, sorted times), plus lengths."""
    g = torch.Generator(device='cuda').manual_seed(seed)
    n = torch.poisson(torch.full((B, C), float(lam), device='cuda'), generator=g).clamp_(1, T).to(torch.int32)
    mask = (torch.arange(T, device='cuda')[None, None, :] < n[..., None]).float()
    # sorted observation times in the first n slots: sort uniform draws with the padded slots pushed to the end
    t = torch.rand((B, C, T), device='cuda', generator=g) * H
    t = torch.where(mask > 0, t, torch.full_like(t, 2 * H)).sort(dim=-1).values * mask
    val = torch.randn((B, C, T), device='cuda', generator=g) * 1.2 * mask
    hold = (torch.rand((B, C, T), device='cuda', generator=g) < 0.2).float()
    x = torch.cat([val, mask, t, hold], dim=1)
    del val, mask, t, hold
    return x, n


@pytest.mark.parametrize('B,C,T,R,H,lam', [(75000, 6, 96, 24, 24, 50), (300000, 12, 288, 24, 24, 200)], ids=['cfg2_75k', 'cfg4_300k'])
def test_interp_full_size_properties(ops, B, C, T, R, H, lam):
    """BASELINE configs[1] and [3] at full size (cfg4: 16.6 GB of input), k1 and k2 forward:
    (a) encounters are independent -- any slice computed alone reproduces its rows of the full launch (to f32 rounding), and
        a small slice is checked against the fp64 oracle; (b) the smoothers are weighted means: a channel whose observed values
        all equal v interpolates to v at every grid point; (c) both operators are linear in the values they average."""
    x, n = _device_vitals(B, C, T, H, lam, seed=B)
    rng = np.random.default_rng(C)
    ks = G(rng.uniform(-0.5, 1.5, C).astype(np.float32))
    kc = G((np.eye(C) + rng.normal(0, 0.2, (C, C))).astype(np.float32))
    kr = G(rng.uniform(-0.5, 1.5, C).astype(np.float32))
    grid = ops.ref_grid(H, R, 'cuda')
    with torch.no_grad():
        full = ops.sci_cci(x, ks, kc, grid, lengths=n)
        assert bool(torch.isfinite(full).all())
        pick = torch.arange(17, B, max(1, B // 300), device='cuda')[:256]
        part = ops.sci_cci(x[pick].contiguous(), ks, kc, grid, lengths=n[pick].contiguous())
        # (a) (not bit-equal: the launch shape -- encounters per workgroup, time-axis split -- follows the batch size)
        np.testing.assert_allclose(part.cpu().numpy(), full[pick].cpu().numpy(), rtol=2e-5, atol=2e-6)
        ref = O.sci_cci_forward(x[pick[:16]].double().cpu(), ks.double().cpu(), kc.double().cpu(), R, H)
        np.testing.assert_allclose(part[:16].cpu().numpy(), ref.numpy(), rtol=RT, atol=AT)
        # (b) constant channels, SCI alone: y == y_trans == the constant
        const = torch.randn((B, C, 1), device='cuda')
        xc = x.clone()
        xc[:, :C] = const * x[:, C:2 * C]
        s = ops.sci_only(xc, ks, grid, lengths=n)                                                # (B,R,3C) = [y | w | y_trans]
        np.testing.assert_allclose(s[:, :, :C].amax(1).cpu().numpy(), const[:, :, 0].cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(s[:, :, 2 * C:].amin(1).cpu().numpy(), const[:, :, 0].cpu().numpy(), rtol=1e-5, atol=1e-5)
        del xc
        # (c) linearity of SCI in the values (weights depend on times only): sci(2.5 x) = 2.5 sci(x) for y and y_trans, same w
        s1 = ops.sci_only(x, ks, grid, lengths=n)
        x2 = x.clone()
        x2[:, :C] *= 2.5
        s2 = ops.sci_only(x2, ks, grid, lengths=n)
        del x2
        assert torch.equal(s1[:, :, C:2 * C], s2[:, :, C:2 * C])
        lin = torch.cat([s1[:, :, :C], s1[:, :, 2 * C:]], dim=-1) * 2.5
        got = torch.cat([s2[:, :, :C], s2[:, :, 2 * C:]], dim=-1)
        assert float((got - lin).abs().max()) <= 2e-6 * float(lin.abs().max())
        del s1, s2, lin, got, full
        # k2: slice consistency, oracle on a small slice, linearity in v
        v = torch.randn((B, C, R), device='cuda')
        y = ops.rbf_deinterp(v, x, kr, grid, lengths=n)
        assert bool(torch.isfinite(y).all())
        yp = ops.rbf_deinterp(v[pick].contiguous(), x[pick].contiguous(), kr, grid, lengths=n[pick].contiguous())
        np.testing.assert_allclose(yp.cpu().numpy(), y[pick].cpu().numpy(), rtol=2e-5, atol=2e-6)
        ref = O.rbf_deinterp(v[pick[:16]].double().cpu(), x[pick[:16]].double().cpu(), kr.double().cpu(), R, H)
        np.testing.assert_allclose(yp[:16].cpu().numpy(), ref.numpy(), rtol=RT, atol=3e-6)
        y2 = ops.rbf_deinterp(v * -3.0, x, kr, grid, lengths=n)
        assert float((y2 + 3.0 * y).abs().max()) <= 2e-6 * float(y.abs().max()) * 3.0


@pytest.mark.parametrize('shape', [(64, 6, 96, 24, 24, 50), (9, 12, 288, 24, 24, 200)])
def test_rbf_backward_with_an_underflowing_bandwidth(ops, shape):
    """a raw bandwidth parameter so negative that softplus underflows to 0 (beta = 0: every basis function is 1).  The slots-on-lanes
    backward folds sqrt(beta) into the time stamps and scales its u-weighted sums back by 1 / beta: that must not put 0 * inf = NaN into
    dL/dbeta (sigmoid(raw) = 0 there, so the parameter gradient is exactly 0), on any backward kernel; the other channels are unaffected."""
    B, C, T, R, H, lam = shape
    x, n = vitals_stack(5 + B, B, C, T, H, lam)
    rng = np.random.default_rng(B)
    v_np = rng.normal(0, 1, (B, C, R)).astype(np.float32)
    k_np = rng.uniform(-0.5, 1.5, C).astype(np.float32)
    k_np[1] = -200.0
    cot = rng.normal(0, 1, (B, C, T)).astype(np.float32)
    grid = ops.ref_grid(H, R, 'cuda')
    res = {}
    for mode in ('0', '2'):
        os.environ['DIC_RBF_BWD_SLOT'] = mode
        try:
            v, k = G(v_np, True), G(k_np, True)
            y = ops.rbf_deinterp(v, G(x), k, grid, lengths=G(n, dtype=torch.int32))
            (y * G(cot)).sum().backward()
        finally:
            os.environ.pop('DIC_RBF_BWD_SLOT', None)
        assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(v.grad).all()) and bool(torch.isfinite(k.grad).all()), mode
        assert float(k.grad[1]) == 0.0
        res[mode] = (v.grad.cpu().numpy(), k.grad.cpu().numpy())
    v64 = torch.tensor(v_np, dtype=torch.float64, requires_grad=True)
    k64 = torch.tensor(k_np, dtype=torch.float64, requires_grad=True)
    ref = O.rbf_deinterp(v64, torch.tensor(x, dtype=torch.float64), k64, R, H)
    (ref * torch.tensor(cot, dtype=torch.float64)).sum().backward()
    for mode in res:
        np.testing.assert_allclose(res[mode][0], v64.grad.numpy(), rtol=2e-4, atol=2e-5 * float(v64.grad.abs().max()))
        np.testing.assert_allclose(res[mode][1], k64.grad.numpy(), rtol=3e-4, atol=3e-5 * float(k64.grad.abs().max()))


@pytest.mark.parametrize('rows,n', [(1, 1), (3, 63), (10, 75000), (2, 4737), (20, 1000)])
def test_cumsum_f64_matches_numpy(rows, n):
    """dic_cumsum_f64 (the stable_cumsum of k-means++ sampling: sklearn/cluster/_kmeans.py:218-243) against NumPy's sequential f64 running sums: equal to
    f64 summation-order noise -- and the draw it feeds (searchsorted against thresholds away from that noise) identical."""
    from deep_interpolation_clustering_amd import kmeans
    rng = np.random.default_rng(n)
    x = (rng.random((rows, n)) ** 2).astype(np.float32)
    got = kmeans._cumsum_f64(torch.tensor(x, device='cuda')).cpu().numpy()
    ref = np.cumsum(x.astype(np.float64), axis=1)
    assert got.dtype == np.float64 and got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-13, atol=0)
    assert (np.diff(got, axis=1) >= 0).all()
    thr = rng.random((rows, 5)) * ref[:, -1:]
    for r in range(rows):
        assert np.array_equal(np.searchsorted(got[r], thr[r]), np.searchsorted(ref[r], thr[r]))


def test_nearest_distance_equals_cdist_min():
    """KMeans.nearest_distance (p2's elbow curve: cdist(X, centers).min(1), p2_clustering_optK.py:253-270) from the E-step kernel against the distance matrix."""
    from deep_interpolation_clustering_amd.kmeans import KMeans
    rng = np.random.default_rng(5)
    X = rng.normal(size=(3001, 256)).astype(np.float32)
    km = KMeans(n_clusters=7, n_init=1, random_state=0).fit(X)
    got = km.nearest_distance(X[:1999]).cpu().numpy()
    ref = np.sqrt(((X[:1999, None, :].astype(np.float64) - km.cluster_centers_[None].astype(np.float64)) ** 2).sum(-1)).min(1)
    np.testing.assert_allclose(got, ref, rtol=2e-5)

"""GPU parity of the assembled network and of one full optimisation step against the golden
step captured from the reference (tests/golden/netstep_*.npz) and against the CPU oracle."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import dic_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def make_args(g, fake, loss):
    return SimpleNamespace(num_variables=6, num_timestamps=g['x'].shape[-1], ref_points=int(g['R']),
                           hours_from_admission=float(g['H']), dropout=0.0, aux_tasks={}, fake_detection=fake,
                           triple_margin=0.0, cluster_number=int(g['K']), loss=loss, grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.}, aux_pos_weights={})


def initial_state(name):
    plain = load('netstep_plain.npz')
    sd = {k[4:]: v for k, v in plain.items() if k.startswith('sd0/')}
    g = plain
    if name == 'fake':
        g = load('netstep_fake.npz')
        sd.update({k[4:]: v for k, v in g.items() if k.startswith('sd0/')})
    return g, {k: torch.tensor(v) for k, v in sd.items()}


@pytest.mark.parametrize('name', ['plain', 'fake'])
@pytest.mark.parametrize('use_lengths', [False, True])
def test_joint_step_matches_reference(name, use_lengths):
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    g, sd0 = initial_state(name)
    fake = name == 'fake'
    args = make_args(g, fake, 'ae_mse_fake_detect_kl' if fake else 'ae_mse_kl')
    dev = torch.device('cuda')
    net = Net(args, dev).to(dev)
    net.load_state_dict(sd0, strict=True)            # reference checkpoint keys load as they are
    net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args)
    x = torch.tensor(g['x'], device=dev)
    ob = torch.tensor(g['ob'], device=dev)
    mask = x[:, 6:12].contiguous()
    lengths = mask.sum(-1).to(torch.int32) if use_lengths else None
    kw = {}
    if fake:
        kw = dict(fake_x=torch.tensor(g['fake_x'], device=dev), fake_perm_idx=torch.tensor(g['fake_perm_idx'], device=dev),
                  fake_det_label=torch.tensor(g['fake_label'], device=dev))
    losses, gnorm, z = st.step(x, ob, mask, lengths, **kw)
    # --- the north-star bar: losses within 1e-5 relative (kl is ~1e-4 here and ill-conditioned in fp32: abs floor)
    for k in ('loss', 'ae_mse'):
        np.testing.assert_allclose(float(losses[k]), float(g['loss_' + k]), rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(float(losses['kl']), float(g['loss_kl']), rtol=1e-5, atol=5e-8)
    if fake:
        np.testing.assert_allclose(float(losses['fake_detection']), float(g['loss_fake_detection']), rtol=1e-5)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g['z'], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(float(gnorm), float(g['gnorm']), rtol=1e-4)
    # parameters after clip + Adam(amsgrad, wd): compare where the gradient is decisively non-zero
    for k, v in net.state_dict().items():
        got = v.detach().cpu().numpy()
        if 'sd1/' + k in g:
            ref = g['sd1/' + k]
            if 'g/' + k in g:
                live = np.abs(g['g/' + k]) >= 1e-4 * float(g['gnorm'])
                got, ref = got[live], ref[live]
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5, err_msg=k)
        elif 'sd1n/' + k in g:
            np.testing.assert_allclose(np.linalg.norm(got.astype(np.float64)), float(g['sd1n/' + k]), rtol=2e-5, err_msg=k)


def test_forward_matches_oracle_bigger_batch():
    """B=300 ragged batch at the BASELINE shape (C=6,T=96,R=24,H=24): whole forward + losses vs the CPU oracle
    carrying the same weights."""
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    coh = synthetic.make_cohort(300, seed=5)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0,
                           aux_tasks={}, fake_detection=False, triple_margin=0.0, cluster_number=4)
    torch.manual_seed(3)
    ref = O.OracleNet(6, 24, 24, 4, 0.0)
    ref.train()
    dev = torch.device('cuda')
    net = Net(args, dev).to(dev)
    net.load_state_dict(ref.state_dict(), strict=True)
    net.train()
    x, ob = torch.tensor(x_np), torch.tensor(ob_np)
    terms, z_ref, y_ref, aux_ref = O.joint_loss(ref, x, ob, x[:, 6:12], 10.0)
    xd, obd = x.to(dev), ob.to(dev)
    lengths = torch.tensor(n, device=dev)
    z, y, aux = net(xd, lengths=lengths)
    rec = net.rec_loss(obd, y, None, lengths)['ae_mse']
    kl = net.kl_loss(aux['cluster_label'], aux['cluster_pred'])['kl']
    np.testing.assert_allclose(float(rec), float(terms['ae_mse']), rtol=1e-5)
    np.testing.assert_allclose(float(kl), float(terms['kl']), rtol=1e-5, atol=5e-8)
    np.testing.assert_allclose(z.detach().cpu().numpy(), z_ref.detach().numpy(), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(aux['cluster_pred'].detach().cpu().numpy(), aux_ref['cluster_pred'].detach().numpy(), rtol=1e-4)
    assert (aux['cluster_pred'].argmax(1).cpu() == aux_ref['cluster_pred'].argmax(1)).all()


@pytest.mark.parametrize('bs', [256, 512])
def test_hip_graph_step_matches_eager(bs):
    """Stepper(use_graphs=True) captures the whole step in a hipGraph; the trajectory must equal the eager one.  (512 rows x 24 steps
    reach the row count from which CompressFC runs as one autograd node on the resident-weight kernels.)"""
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=False, triple_margin=0.0, cluster_number=4, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
    dev = torch.device('cuda')
    coh = synthetic.make_cohort(2 * bs, seed=8)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
    traj = {}
    for mode in (False, True):
        torch.manual_seed(4)
        net = Net(args, dev).to(dev)
        net.train()
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16, use_graphs=mode)
        out = []
        for i in range(8):
            lo = (i % 2) * bs
            losses, gnorm, _ = st.step(X[lo:lo + bs], OB[lo:lo + bs], None, LEN[lo:lo + bs])
            out.append([float(losses['loss'].detach()), float(losses['ae_mse'].detach()), float(gnorm)])
        traj[mode] = np.array(out)
        if mode:
            assert len(st._graphs) == 1
    np.testing.assert_allclose(traj[True], traj[False], rtol=2e-3)


def _overlay_state(name):
    plain = load('netstep_plain.npz')
    g = load('netstep_%s.npz' % name)
    sd = {k[4:]: v for k, v in plain.items() if k.startswith('sd0/')}
    sd.update({k[4:]: v for k, v in g.items() if k.startswith('sd0/')})
    return g, sd


def _check_after_step(net, g):
    for k, v in net.state_dict().items():
        got = v.detach().cpu().numpy()
        if 'sd1/' + k in g:
            ref = g['sd1/' + k]
            if 'g/' + k in g:
                live = np.abs(g['g/' + k]) >= 1e-4 * float(g['gnorm'])
                got, ref = got[live], ref[live]
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5, err_msg=k)
        elif 'sd1n/' + k in g:
            np.testing.assert_allclose(np.linalg.norm(got.astype(np.float64)), float(g['sd1n/' + k]), rtol=2e-5, err_msg=k)


def test_supervised_heads_step_matches_reference():
    """clustering_interp.Net with the supervised auxiliary heads (masked future-vital MSE, two weighted-BCE tasks), fake detection
    and the KL term -- upstream's default p3 objective ae_mse_sup_fake_detect_kl (p3:78) -- against the reference's own step
    (oracle/make_golden_sup.py; synthetic labels stand in for the private cohort's)."""
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    g, sd0 = _overlay_state('sup')
    tasks = {k: float(g['w_' + k]) for k in ('future_vital', 'AKI_overall', 'ICU_24h')}
    args = SimpleNamespace(num_variables=6, num_timestamps=g['x'].shape[-1], ref_points=int(g['R']), hours_from_admission=float(g['H']),
                           dropout=0.0, aux_tasks=tasks, fake_detection=True, triple_margin=0.0, cluster_number=int(g['K']),
                           loss='ae_mse_sup_fake_detect_kl', grad_clip=15.0, unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.},
                           aux_pos_weights={'AKI_overall': float(g['pos_w_AKI_overall']), 'ICU_24h': float(g['pos_w_ICU_24h'])})
    dev = torch.device('cuda')
    net = Net(args, dev).to(dev)
    net.load_state_dict({k: torch.tensor(v) for k, v in sd0.items()}, strict=True)
    net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args)
    G = lambda a: torch.tensor(a, device=dev)                                   # noqa: E731
    x = G(g['x'])
    losses, gnorm, z = st.step(x, G(g['ob']), x[:, 6:12].contiguous(), None, fake_x=G(g['fake_x']), fake_perm_idx=G(g['fake_perm_idx']),
                               fake_det_label=G(g['fake_label']), future_vital_mask=G(g['fv_mask']),
                               aux_label_dict={k: G(g['label_' + k]) for k in tasks})
    for k in ('loss', 'ae_mse', 'future_vital', 'AKI_overall', 'ICU_24h', 'fake_detection'):
        np.testing.assert_allclose(float(losses[k].detach()), float(g['loss_' + k]), rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(float(losses['kl'].detach()), float(g['loss_kl']), rtol=1e-5, atol=5e-8)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g['z'], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(float(gnorm), float(g['gnorm']), rtol=1e-4)
    _check_after_step(net, g)


def test_triplet_step_matches_reference():
    """Fake detection + triplet margin on clustering_interp.Net (the only net with the triplet branch, :171-180, :234-236), loss
    ae_mse_fake_detect_triplet, hinge active on part of the batch."""
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    g, sd0 = _overlay_state('triplet')
    args = SimpleNamespace(num_variables=6, num_timestamps=g['x'].shape[-1], ref_points=int(g['R']), hours_from_admission=float(g['H']),
                           dropout=0.0, aux_tasks={}, fake_detection=True, triple_margin=float(g['margin']), cluster_number=int(g['K']),
                           loss='ae_mse_fake_detect_triplet', grad_clip=15.0, unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.},
                           aux_pos_weights={})
    dev = torch.device('cuda')
    net = Net(args, dev).to(dev)
    net.load_state_dict({k: torch.tensor(v) for k, v in sd0.items()}, strict=True)
    net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args)
    G = lambda a: torch.tensor(a, device=dev)                                   # noqa: E731
    x = G(g['x'])
    losses, gnorm, z = st.step(x, G(g['ob']), x[:, 6:12].contiguous(), None, fake_x=G(g['fake_x']), fake_perm_idx=G(g['fake_perm_idx']),
                               positive_x=G(g['positive_x']), fake_det_label=G(g['fake_label']))
    assert float(g['loss_triplet']) > 0.05
    for k in ('loss', 'ae_mse', 'fake_detection', 'triplet'):
        np.testing.assert_allclose(float(losses[k].detach()), float(g['loss_' + k]), rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(float(gnorm), float(g['gnorm']), rtol=1e-4)
    _check_after_step(net, g)


def test_hip_graph_replays_draw_fresh_dropout_masks():
    """With dropout active the captured step must not freeze its mask: the per-device call counter advances inside the graph, so
    two replays on identical inputs and (restored) identical parameters see different masks -> different losses."""
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.5, aux_tasks={},
                           fake_detection=False, triple_margin=0.0, cluster_number=4, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
    dev = torch.device('cuda')
    coh = synthetic.make_cohort(256, seed=8)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
    torch.manual_seed(4)
    net = Net(args, dev).to(dev)
    net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 0.0, 0.0), args, autocast_dtype=torch.bfloat16, use_graphs=True)   # lr 0: parameters stay put
    vals = []
    for _ in range(4):
        losses, _, _ = st.step(X, OB, None, LEN)
        vals.append(float(losses['ae_mse'].detach()))
    assert len(st._graphs) == 1 and len(set(vals)) == 4, vals
    net.eval()                                    # dropout off: the (re-captured, eval-mode) step is deterministic
    e = [float(st.forward_loss(X, OB, None, LEN)[0]['ae_mse'].detach()) for _ in range(2)]
    assert e[0] == e[1]


@pytest.mark.parametrize('graphs', [False, True])
def test_bf16_step_updates_every_parameter(graphs):
    """The fast mode writes some gradients (the LSTMs') straight into the flat bucket, past autograd's accumulate hooks that tell
    the optimiser which parameters a backward reached: after two steps EVERY parameter tensor must have moved, and by the same
    amount as when all gradients travel through autograd (the f32 mode), up to bf16 effects on their direction."""
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=False, triple_margin=0.0, cluster_number=4, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
    dev = torch.device('cuda')
    coh = synthetic.make_cohort(256, seed=21)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
    moved = {}
    for mode in ('bf16', 'f32'):
        torch.manual_seed(4)
        net = Net(args, dev).to(dev)
        net.train()
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16 if mode == 'bf16' else None,
                     use_graphs=graphs and mode == 'bf16')
        before = {k: v.detach().clone() for k, v in net.named_parameters()}
        for _ in range(2):
            st.step(X, OB, None, LEN)
        moved[mode] = {k: float((v.detach() - before[k]).abs().mean()) for k, v in net.named_parameters()}
    for k, d in moved['bf16'].items():
        assert d > 0, f'{k} did not move'
        assert 0.5 * moved['f32'][k] < d < 2.0 * moved['f32'][k], f"{k}: mean |update| {d:.3e} vs {moved['f32'][k]:.3e} through autograd"


@pytest.mark.parametrize('B', [1, 31, 257, 1000, 4097, 9000])
def test_bf16_step_tracks_f32_step_at_odd_batch_sizes(B):
    """The bf16 fast mode switches kernels with the batch size (32-row / 64-row recurrences, one-pass weight gradients from 32 rows up,
    resident-weight projections and the fused CompressFC layer from 8192 rows up, the wave-per-encounter k2 backward, tails of
    every tile size): its losses over three steps must follow the f32 mode's to bf16 accuracy at sizes that are multiples of nothing."""
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=False, triple_margin=0.0, cluster_number=4, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
    dev = torch.device('cuda')
    coh = synthetic.make_cohort(B, seed=100 + B)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
    traj = {}
    for mode in ('bf16', 'f32'):
        torch.manual_seed(4)
        net = Net(args, dev).to(dev)
        net.train()
        if B == 1:
            net.eval()                     # BatchNorm needs more than one row in training mode (upstream raises too)
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16 if mode == 'bf16' else None,
                     use_graphs=False)
        traj[mode] = [float(st.step(X, OB, None, LEN)[0]['ae_mse']) for _ in range(3)]
    for a, b in zip(traj['bf16'], traj['f32']):
        assert np.isfinite(a) and abs(a - b) <= 2e-2 * abs(b) + 1e-3, (traj['bf16'], traj['f32'])
    if B > 1:
        assert traj['f32'][2] < traj['f32'][0] and traj['bf16'][2] < traj['bf16'][0]


def test_side_stream_weight_gradients_leave_the_trajectory_unchanged(monkeypatch):
    """The decoder's weight-gradient kernel on a side stream (lstm.side_stream_session inside Stepper's backward) against the same step
    with everything on one stream: the same kernels on the same data -- identical losses, gradient norms and parameters."""
    from deep_interpolation_clustering_amd import lstm as L
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=False, triple_margin=0.0, cluster_number=4, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
    dev = torch.device('cuda')
    monkeypatch.setattr(L, 'SMALL_BATCH', 0)                 # the 64-row kernels and the one-pass weight-gradient kernels at a test-sized batch
    coh = synthetic.make_cohort(512, seed=9)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
    res = {}
    for side in (False, True):
        monkeypatch.setattr(L, 'DW_SIDE_STREAM', side)
        torch.manual_seed(4)
        net = Net(args, dev).to(dev)
        net.train()
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16)
        out = []
        for i in range(4):
            losses, gnorm, _ = st.step(X, OB, None, LEN)
            out.append([float(losses['loss'].detach()), float(gnorm)])
        torch.cuda.synchronize()
        res[side] = (np.array(out), st.flat.flat.detach().clone())
    np.testing.assert_array_equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])


@pytest.mark.parametrize('loss,fake', [('ae_mse_kl', False), ('ae_mse_fake_detect_kl', True)])
def test_queued_small_gradients_equal_autograd_accumulation(loss, fake, monkeypatch):
    """ops.grad_sink_session (one dic_accumulate_many launch for the small parameter gradients of the k1 / k2 / DEC / CompressFC nodes)
    against autograd's own AccumulateGrad path: adding into the zeroed bucket is exact, so losses, norms and parameters are identical --
    also with the fake-detection branch, whose second interpolation pass adds a second gradient to the same bandwidth parameters."""
    from deep_interpolation_clustering_amd import ops, synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=fake, triple_margin=0.0, cluster_number=4, loss=loss, grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
    dev = torch.device('cuda')
    B = 384
    coh = synthetic.make_cohort(B, seed=10)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
    FX = X.clone()
    FX[:, :6] = torch.where(X[:, 6:12] > 0, X[:, :6].flip(0), X[:, :6])
    kw = {}
    if fake:
        kw = dict(fake_x=FX, fake_perm_idx=torch.arange(2 * B, device=dev),
                  fake_det_label=torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)]).to(torch.int64))
    res = {}
    for sinks in (False, True):
        monkeypatch.setattr(ops, 'GRAD_SINKS', sinks)
        torch.manual_seed(4)
        net = Net(args, dev).to(dev)
        net.train()
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16)
        out = []
        for i in range(3):
            losses, gnorm, _ = st.step(X, OB, None, LEN, **kw)
            out.append([float(losses['loss'].detach()), float(gnorm)])
        res[sinks] = (np.array(out), st.flat.flat.detach().clone())
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=1e-6)
    d = (res[True][1] - res[False][1]).abs()
    assert float(d.max()) <= 1e-6 + 1e-5 * float(res[False][1].abs().max())


@pytest.mark.parametrize('module,switch', [('lstm', 'DEFER_RELU'), ('lstm', 'GX_LANE_NATIVE'), ('ops', 'COMPRESS_FUSED'), ('lstm', 'DW_SIDE_STREAM'),
                                           ('lstm', 'FWD_EIGHT_WAVES')])
@pytest.mark.parametrize('loss,fake', [('ae_mse_kl', False), ('ae_mse_fake_detect_kl', True)])
def test_fast_path_switches_leave_the_step_unchanged(module, switch, loss, fake, monkeypatch):
    """Each large-batch fast path of this round was built to be bit-identical to the path it replaces: the ReLU left to the decoder's
    kernels, gx in accumulator order, CompressFC as one node, the side-stream weight gradients.  Three optimisation steps on the 64-row
    kernels with the switch off and on: identical losses, gradient norms and parameters -- for the plain objective and for upstream's
    default one (second encoder pass on corrupted samples + detection head)."""
    import importlib
    from deep_interpolation_clustering_amd import lstm as L
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    mod = importlib.import_module('deep_interpolation_clustering_amd.' + module)
    args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=fake, triple_margin=0.0, cluster_number=4, loss=loss, grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
    dev = torch.device('cuda')
    monkeypatch.setattr(L, 'SMALL_BATCH', 0)                 # the large-batch kernels at a test-sized batch
    B = 384
    coh = synthetic.make_cohort(B, seed=12)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
    kw = {}
    if fake:
        FX = X.clone()
        FX[:, :6] = torch.where(X[:, 6:12] > 0, X[:, :6].flip(0), X[:, :6])
        kw = dict(fake_x=FX, fake_perm_idx=torch.arange(2 * B, device=dev),
                  fake_det_label=torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)]).to(torch.int64))
    res = {}
    for on in (False, True):
        monkeypatch.setattr(mod, switch, on)
        torch.manual_seed(4)
        net = Net(args, dev).to(dev)
        net.train()
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16)
        out = []
        for i in range(3):
            losses, gnorm, _ = st.step(X, OB, None, LEN, **kw)
            out.append([float(losses['loss'].detach()), float(gnorm)])
        torch.cuda.synchronize()
        res[on] = (np.array(out), st.flat.flat.detach().clone())
    np.testing.assert_array_equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])


def test_joint_step_K8_large_batch_bf16_tracks_f32():
    """BASELINE configs[2]'s model (K = 8) at a batch that takes every large-batch kernel (64-row recurrences, one-pass weight gradients,
    resident-weight projections, fused CompressFC, ragged store input): three steps of the bf16 fast mode follow the f32 parity mode, whose
    single step is pinned against the reference at this K by tests/test_gpu_traj.py::test_joint_step_cfg_shape_kmeans_centroids[8]."""
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.ragged import RaggedBatch, RaggedStore
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=False, triple_margin=0.0, cluster_number=8, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
    dev = torch.device('cuda')
    B = 8192
    coh = synthetic.make_cohort(B, G=8, seed=33)
    x_np, _, n = synthetic.stacked_batch(coh)
    store = RaggedStore(x_np, 6, dev)
    rb = RaggedBatch(store, torch.arange(B, device=dev))
    g = torch.tensor(coh['phenotype'].astype(np.int64), device=dev)
    traj = {}
    for mode in ('bf16', 'f32'):
        torch.manual_seed(4)
        net = Net(args, dev).to(dev)
        net.eval()
        with torch.no_grad():            # centroids on the latents' clusters: the p3 regime
            z = net(rb)[0]
            net.init_cluster_center(torch.stack([z[g == j].mean(0) for j in range(8)]))
        net.train()
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16 if mode == 'bf16' else None,
                     use_graphs=False)
        out = []
        for _ in range(3):
            losses, gnorm, _ = st.step(rb, None, None)
            out.append([float(losses['loss'].detach()), float(losses['ae_mse'].detach()), float(losses['kl'].detach())])
        traj[mode] = np.array(out)
    assert traj['f32'][0, 2] > 0.01
    np.testing.assert_allclose(traj['bf16'], traj['f32'], rtol=2e-2)
    assert traj['f32'][2, 1] < traj['f32'][0, 1]


@pytest.mark.parametrize('shape', ['cfg2-256', 'cfg2-4096', 'cfg2-32768', 'cfg4-8192', 'x3-4096'])
def test_bf16_step_launches_no_library_gemm(shape):
    """(one way to form each product.)  Every dense product of the bf16 step -- and of the f32 step on split products -- runs
    on a kernel of this library at every batch size: one traced step (the ROCm tracer behind torch.profiler) at the reference's batch of
    256, at 4 096 (the 32-row recurrence path), at the headline's 32 768 and at configs[3]'s shape must show no hipBLASLt / rocBLAS / Tensile
    kernel.  (The exact-f32 parity mode keeps its f32 library GEMMs: not traced here.)"""
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    cfg, B = shape.split('-')
    B = int(B)
    C, T, lam, K = (12, 288, 200.0, 16) if cfg == 'cfg4' else (6, 96, 50.0, 4)
    args = SimpleNamespace(num_variables=C, num_timestamps=T, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=False, triple_margin=0.0, cluster_number=K, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
    dev = torch.device('cuda')
    coh = synthetic.make_cohort(B, C=C, T=T, lam=lam, G=K, seed=77)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
    del x_np, ob_np
    torch.manual_seed(4)
    net = Net(args, dev).to(dev)
    net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=None if cfg == 'x3' else torch.bfloat16,
                 precision='x3' if cfg == 'x3' else None, use_graphs=False)
    st.step(X, OB, None, LEN)                                # (lazy initialisation outside the trace)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        losses, _, _ = st.step(X, OB, None, LEN)
        torch.cuda.synchronize()
    names = {ev.name for ev in prof.events() if ev.device_type == DeviceType.CUDA}
    assert any('dic' in n_ for n_ in names), sorted(names)[:5]                 # the trace saw this library's kernels
    library = sorted(n_ for n_ in names if 'Cijk' in n_ or 'rocblas' in n_.lower() or 'hipblas' in n_.lower() or 'tensile' in n_.lower())
    assert not library, library
    assert np.isfinite(float(losses['loss']))

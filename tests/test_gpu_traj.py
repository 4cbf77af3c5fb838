"""The trainers, the input pipeline and the feature dump on the GPU against what the REFERENCE's own trainers did on the cfg1 cohort
(tests/golden/{traj_cfg1,featdump_cfg1,netstep_cfg_K4,netstep_cfg_K8}.npz, written by oracle/make_golden_traj.py running
pretrain_trainer.Trainer.train(), clustering_trainer.TrainerCluster.train() / .eval() and one joint step of clustering_interp.Net).

  * joint step at the configured shape (C,T,R,H) = (6,96,24,24), pretrained weights, k-means centroids, K = 4 and K = 8:
    loss / ae_mse / kl at rtol 1e-5 with NO absolute floor (north_star's bar; KL is ~0.1 here);
  * p1: 16 optimiser steps over 2 epochs with the learning-rate schedule acting in between, both loaders (HBM-resident DeviceLoader
    and upstream-style torch DataLoader over the DataSet);
  * p3: k-means initialisation from the reference's checkpoint (centres, validation labels), 24 joint steps, per-epoch label delta,
    for K = 4 (cfg1) and K = 6 (labels keep moving);
  * feature dump: the .npy dictionary eval() writes from the reference's checkpoint -- keys, dtypes, shapes, values.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


DEV_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'traj_deviation.jsonl')


def log_deviation(test, **measured):
    """The measured distance of this run from the reference's trajectory, one JSON line per test (gpurun_out/traj_deviation.jsonl: the
    directory gpurun merges back; also on stdout with -s).  The tolerances below are <= 3 x the larger of these numbers (last measured
    values: profiles/r4_traj_deviation.json) and of the reference's own run-to-run spread (tests/golden/ref_spread.json: the same trainers
    on 8 and on 3 CPU threads)."""
    import json
    rec = {'test': test}
    rec.update({k: (v.tolist() if hasattr(v, 'tolist') else v) for k, v in measured.items()})
    line = json.dumps(rec)
    print('[traj-deviation]', line)
    try:
        os.makedirs(os.path.dirname(DEV_LOG), exist_ok=True)
        with open(DEV_LOG, 'a') as f:
            f.write(line + '\n')
    except OSError:
        pass


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.abs(b))) if a.size else 0.0


# Tolerances of the trajectory tests: <= 3 x max(measured distance of the GPU run from the reference's trajectory, the reference's own
# run-to-run spread between 8 and 3 CPU threads).  Measured values: profiles/r4_traj_deviation.json / tests/golden/ref_spread.json; both are
# restated in DESIGN.md section 2.
TOL = {       # tolerance (measured on the GPU, round 4 / the reference's own spread): profiles/r4_traj_deviation.json
    'p1_first2': 2.5e-6,        # 1.5e-7 / 8.2e-7
    'p1_first8': 1.1e-4,        # 3.4e-5 / 2.8e-5
    'p1_all': 2.7e-4,           # 8.9e-5 / 4.2e-5
    'p1_valid': 3.3e-4,         # 1.1e-4 / 1.7e-5
    'p1_state': 4.3e-3,         # 1.4e-3 (relative L2 distance of the worst parameter tensor after 16 steps)
    'p1_moments': 3.5e-4,       # 5.8e-5 / 1.1e-4
    # the same 16 steps with the products as three-term bf16 splits (--f32_products x3).  Round 6 (eight-wave x3 recurrence kernels, gate non-linearities on
    # the transcendental unit, split-plane gate gradients): first step 1.1e-7, second 1.7e-5, all 16 2.8e-4 (round 4's kernels: 4.0e-6 / 2.5e-4).  From the
    # second step on the distance is not a measure of kernel accuracy: Adam's first update is lr * sign(g) wherever |g| >> eps, so WHICH near-zero gradients
    # come out positive decides O(lr) parameter moves -- any change of rounding reshuffles them (the reference's own 8- vs 3-thread runs: 4.2e-5 by step 16)
    'p1x3_first1': 1e-5, 'p1x3_first2': 5.0e-5, 'p1x3_all': 8.4e-4,
    'p3_first2': 1e-6,          # 2.3e-7  (p3 starts from the fixture's p1 state: no reference spread is comparable -- see ref_spread.json's note)
    'p3_first8': 8.1e-5,        # 2.7e-5  (K = 6, the over-segmented run; K = 4: 2.7e-6)
    'p3_all': 8.1e-5,           # 2.7e-5
    'p3_param_norms': 5.4e-6,   # 1.8e-6
    # the same p3 runs with --f32_products x3: 3 x measured (round 6 kernels; gpurun_out/traj_deviation.jsonl -> profiles/r6_traj_deviation.json): first step
    # 4.3e-7, first two 9.1e-6, first eight 2.0e-4, all 24 3.0e-4 on the step losses (K = 4; K = 6: 3.0e-7 / 6.7e-6 / 1.4e-5 / 3.1e-5) -- KL itself stays
    # below 3e-6 throughout, the reconstruction term carries the walk (see the note on Adam above; round 5's kernels: 5.3e-7 / 4.6e-6 / 2.8e-5 / 8.7e-5) --
    # and 5.1e-6 on the end-state parameter norms
    'p3x3_first2': 2.7e-5, 'p3x3_first8': 6.0e-4, 'p3x3_all': 9.0e-4, 'p3x3_param_norms': 1.6e-5,
}


def trainer_args(**over):
    """The namespace oracle/make_golden_traj.py gave the reference's trainers (+ this package's own switches at their defaults)."""
    a = dict(log_level='WARNING', seed=7529, num_gpus=1, mode='train', restore=False, restore_metric='ae_mse', log_train_freq=1000,
             log_valid_freq=1000, hours_from_admission=24, num_workers=0, batch_size=100, norm_method='minmax', aug_input=False,
             aug_std=0.1, scale=5.0, denoise=False, num_variables=6, num_timestamps=96, data_filter=False, evaluate_interpolation=False,
             ref_points=24, dropout=0.0, fake_detection=False, triple_margin=0.0, triple_pos_std=0.1, loss='ae_mse', aux_tasks={},
             unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.}, aux_pos_weights={}, optimizer='Adam', init_lr=0.003,
             min_lr=1e-6, lr_decay_mode='step', lr_decay_step_or_patience=1, lr_decay_rate=0.5, max_epochs=3, grad_clip=15.0,
             weight_decay_rate=0.0004, early_stopping=50, amp_bf16=False, hip_graph=None, no_hip_graph=True, host_loader=False)
    a.update(over)
    return SimpleNamespace(**a)


@pytest.fixture(scope='module')
def run_dir(tmp_path_factory):
    from deep_interpolation_clustering_amd import dataloader, synthetic
    base = tmp_path_factory.mktemp('dic_traj')
    synthetic.write_split(str(base), 1000, C=6, T=96, H=24.0, lam=50.0, G=4)      # the cohort the fixtures were generated on
    run = base / 'run'
    run.mkdir()
    old_cwd, old_base = os.getcwd(), dataloader.BASE_PATH
    os.chdir(run)
    dataloader.BASE_PATH = str(base)
    yield run
    os.chdir(old_cwd)
    dataloader.BASE_PATH = old_base


def make_loaders(args, dev, kind):
    from deep_interpolation_clustering_amd.dataloader import DataSet, DeviceLoader
    out = {}
    for cohort in ('training', 'validation', 'testing'):
        ds = DataSet(args, cohort)
        if kind == 'device':          # the HBM-resident ragged store; padded per-batch tensors rebuilt beside it (unshuffled loaders do)
            out[cohort] = DeviceLoader(ds, args.batch_size, False, dev, seed=1, shard=False)       # fixed batch order, as in the fixture run
        elif kind == 'ragged':        # ... training batches as bare handles into the store, what a shuffled training loader yields
            out[cohort] = DeviceLoader(ds, args.batch_size, False, dev, seed=1, shard=False, dense_samples=False if cohort == 'training' else None)
        elif kind == 'padded':        # the padded (N,4C,T) array resident instead
            out[cohort] = DeviceLoader(ds, args.batch_size, False, dev, seed=1, shard=False, ragged=False)
        else:
            out[cohort] = torch.utils.data.DataLoader(ds, batch_size=args.batch_size, num_workers=0, shuffle=False)
    return out


def spy_steps(trainer, keys):
    """Record the loss terms of every optimisation step the trainer takes (they stay device tensors inside the trainer)."""
    rec, inner = [], trainer.stepper.step

    def step(*a, **k):
        out = inner(*a, **k)
        rec.append([out[0][key].detach().clone() for key in keys])
        return out
    trainer.stepper.step = step
    return rec


def p1_state():
    t = load('traj_cfg1.npz')
    return t, {k[5:]: torch.tensor(v) for k, v in t.items() if k.startswith('p1sd/')}


def write_checkpoint(path, state, epoch, optimizer):
    """Upstream's layout (utils.py:141-145): {'epoch','state_dict','optimizer'} at <exp>/weight/<metric>/model.pth.tar."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({'epoch': epoch, 'state_dict': state, 'optimizer': optimizer.state_dict()}, path)


# ---------------------------------------------------------------------------------------------------------------- joint step, cfg shape
@pytest.mark.parametrize('K', [4, 8])
@pytest.mark.parametrize('use_lengths', [False, True])
def test_joint_step_cfg_shape_kmeans_centroids(K, use_lengths):
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    g = load(f'netstep_cfg_K{K}.npz')
    _, sd = p1_state()
    sd['cluster_assignment.cluster_centers'] = torch.tensor(g['centers'])
    args = trainer_args(loss='ae_mse_kl', cluster_number=K)
    dev = torch.device('cuda')
    net = Net(args, dev).to(dev)
    net.load_state_dict(sd, strict=True)
    net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args)
    x, ob = torch.tensor(g['x'], device=dev), torch.tensor(g['ob'], device=dev)
    mask = x[:, 6:12].contiguous()
    losses, gnorm, z = st.step(x, ob, mask, mask.sum(-1).to(torch.int32) if use_lengths else None)
    assert float(g['loss_kl']) > 0.05                                   # the p3 regime: KL is O(0.1), not the 1e-4 of Xavier centroids
    for k in ('loss', 'ae_mse', 'kl'):
        np.testing.assert_allclose(float(losses[k]), float(g['loss_' + k]), rtol=1e-5, atol=0, err_msg=k)
    np.testing.assert_allclose(float(gnorm), float(g['gnorm']), rtol=1e-4)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g['z'], rtol=1e-4, atol=2e-6)
    q = net.cluster_assignment(z.detach())
    assert (q.argmax(1).cpu().numpy() == g['q'].argmax(1)).all()
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


# ---------------------------------------------------------------------------------------------------------------- joint step, cfg4 shape
def _wide_net(g, dev):
    from deep_interpolation_clustering_amd.clustering_interp import Net
    args = trainer_args(loss='ae_mse_kl', cluster_number=16, num_variables=12, num_timestamps=288)
    net = Net(args, dev).to(dev)
    net.load_state_dict({k[4:]: torch.tensor(v) for k, v in g.items() if k.startswith('sd0/')}, strict=True)
    net.train()
    return args, net


@pytest.mark.parametrize('inp', ['padded', 'lengths', 'store'])
def test_joint_step_wide_shape_K16(inp):
    """BASELINE configs[3]'s shape: C = 12 (the encoder LSTM takes 3C = 36 features, clustering_interp.py:102-111), T = 288, R = 24,
    K = 16 -- one joint step against the reference's own (tests/golden/netstep_wide_K16.npz, oracle/make_golden_wide.py): loss / ae_mse /
    kl at rtol 1e-5 with no floor, argmax of q exact, gradient norm, latents, the state after the step."""
    from deep_interpolation_clustering_amd.ragged import RaggedBatch, RaggedStore
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    g = load('netstep_wide_K16.npz')
    dev = torch.device('cuda')
    args, net = _wide_net(g, dev)
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args)
    x, ob = torch.tensor(g['x'], device=dev), torch.tensor(g['ob'], device=dev)
    mask = x[:, 12:24].contiguous()
    if inp == 'store':
        rb = RaggedBatch(RaggedStore(g['x'], 12, dev), torch.arange(x.shape[0], device=dev))
        losses, gnorm, z = st.step(rb, None, None)
    else:
        losses, gnorm, z = st.step(x, ob, mask, mask.sum(-1).to(torch.int32) if inp == 'lengths' else None)
    assert float(g['loss_kl']) > 0.02                                   # k-means centroids: KL is well conditioned
    for k in ('loss', 'ae_mse', 'kl'):
        np.testing.assert_allclose(float(losses[k]), float(g['loss_' + k]), rtol=1e-5, atol=0, err_msg=k)
    np.testing.assert_allclose(float(gnorm), float(g['gnorm']), rtol=1e-4)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g['z'], rtol=1e-4, atol=2e-6)
    q = net.cluster_assignment(z.detach())
    assert (q.argmax(1).cpu().numpy() == g['q'].argmax(1)).all()
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


@pytest.mark.parametrize('B', [64, 4160])
def test_joint_step_wide_shape_bf16_tracks_f32(B):
    """The bf16 mode at C = 12: packed 64-wide encoder rows (36 features + the bias column), the fused-projection recurrence and the one-pass
    weight-gradient kernel at that width (B = 4160: the 64-row kernels, not a multiple of the tile sizes; B = 64: the 32-row ones) -- three
    steps follow the f32 mode, whose first step is pinned against the reference above."""
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    g = load('netstep_wide_K16.npz')
    dev = torch.device('cuda')
    if B == 64:
        x = torch.tensor(g['x'], device=dev)
    else:
        coh = synthetic.make_cohort(B, C=12, T=288, H=24.0, lam=200.0, G=16, seed=45)
        x = torch.tensor(synthetic.stacked_batch(coh)[0], device=dev)
    ob, ln = x[:, :12].contiguous(), x[:, 12:24].sum(-1).to(torch.int32)
    traj = {}
    for mode in ('bf16', 'f32'):
        args, net = _wide_net(g, dev)
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16 if mode == 'bf16' else None)
        out = []
        for _ in range(3):
            losses, gnorm, _ = st.step(x, ob, None, ln)
            out.append([float(losses[k].detach()) for k in ('loss', 'ae_mse', 'kl')] + [float(gnorm)])
        traj[mode] = np.array(out)
    np.testing.assert_allclose(traj['bf16'][:, :2], traj['f32'][:, :2], rtol=2e-2)
    np.testing.assert_allclose(traj['bf16'][:, 2], traj['f32'][:, 2], rtol=2e-2, atol=5e-4)      # (KL falls from 0.04 to ~2e-3 within three steps here)
    np.testing.assert_allclose(traj['bf16'][:, 3], traj['f32'][:, 3], rtol=5e-2)
    assert traj['f32'][2, 1] < traj['f32'][0, 1]


# ---------------------------------------------------------------------------------------------------------------- p1 trajectory
@pytest.mark.parametrize('kind', ['device', 'ragged', 'padded', 'host'])
def test_pretrain_trainer_follows_reference(run_dir, kind, tmp_path):
    from deep_interpolation_clustering_amd.pretrain_interp import Net
    from deep_interpolation_clustering_amd.pretrain_trainer import Trainer
    t, sd_end = p1_state()
    plain = load('netstep_plain.npz')
    args = trainer_args(host_loader=(kind == 'host'))
    dev = torch.device('cuda')
    net = Net(args, dev)
    net.load_state_dict({k[4:]: torch.tensor(v) for k, v in plain.items() if k.startswith('sd0/') and 'cluster' not in k}, strict=True)
    exp = str(tmp_path / 'Pretrain')
    tr = Trainer(args, net, make_loaders(args, dev, kind), exp, dev)
    rec = spy_steps(tr, ['ae_mse'])
    valid, lrs, inner = [], [], tr.aly_pred

    def aly(scope, md):
        valid.append(float(md['ae_mse']))
        r = inner(scope, md)
        lrs.append(tr.optimizer.param_groups[0]['lr'])
        return r
    tr.aly_pred = aly
    tr.train()
    got = np.array([[float(v) for v in row] for row in rec])[:, 0]
    ref = t['p1/train_ae_mse']
    assert got.shape == ref.shape == (16,)
    # two f32 implementations of one Adam(amsgrad) loop separate step by step (rounding noise on near-zero gradients becomes O(lr)
    # moves); the oracle itself is 3e-5 from the reference by step 5 (tests/test_oracle_golden.py)
    dev_state, dev_mom = {}, {}
    log_deviation(f'p1[{kind}]', step_loss_rel=np.abs(got - ref) / np.abs(ref), valid_rel=relmax(valid, t['p1/valid_batch_ae_mse'][:, 0]))
    np.testing.assert_allclose(got[:2], ref[:2], rtol=TOL['p1_first2'])
    np.testing.assert_allclose(got[:8], ref[:8], rtol=TOL['p1_first8'])
    np.testing.assert_allclose(got, ref, rtol=TOL['p1_all'])
    np.testing.assert_allclose(valid, t['p1/valid_batch_ae_mse'][:, 0], rtol=TOL['p1_valid'])          # eval mode: BatchNorm running statistics in use
    np.testing.assert_allclose(lrs, t['p1/lr_after_epoch'], rtol=1e-12)
    for k, v in net.state_dict().items():
        # (a bias in front of a training-mode BatchNorm has an exactly-zero true gradient and no effect on the function: what Adam
        #  makes of it is a random walk driven by rounding noise, in the reference too)
        if v.dtype.is_floating_point and v.numel() > 1 and k != 'rbf.compress_fc.module.model.0.bias':
            a, b = v.detach().cpu().numpy().astype(np.float64), sd_end[k].numpy().astype(np.float64)
            dev_state[k] = float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
            assert np.linalg.norm(a - b) <= TOL['p1_state'] * np.linalg.norm(b) + 1e-6, k
    bn = 'rbf.compress_fc.module.model.1.'
    np.testing.assert_allclose(net.state_dict()[bn + 'running_var'].cpu().numpy(), sd_end[bn + 'running_var'].numpy(), rtol=2e-3)
    assert int(net.state_dict()[bn + 'num_batches_tracked']) == int(sd_end[bn + 'num_batches_tracked']) == 16
    # amsgrad running maximum, step count
    for name, p in net.named_parameters():
        if name == 'rbf.compress_fc.module.model.0.bias':
            continue
        got_n = float(torch.linalg.vector_norm(tr.optimizer.state[p]['max_exp_avg_sq'].double()))
        dev_mom[name] = abs(got_n - float(t[f'p1opt/max_exp_avg_sq/{name}'])) / max(float(t[f'p1opt/max_exp_avg_sq/{name}']), 1e-30)
        np.testing.assert_allclose(got_n, float(t[f'p1opt/max_exp_avg_sq/{name}']), rtol=TOL['p1_moments'], atol=1e-12, err_msg=name)
    log_deviation(f'p1[{kind}]/end', state_rel_max=max(dev_state.values()), max_exp_avg_sq_rel_max=max(dev_mom.values()))
    assert float(next(iter(tr.optimizer.state.values()))['step']) == float(t['p1opt/step'])
    # the checkpoint: upstream's file layout and key names
    ck = torch.load(os.path.join(exp, 'weight', 'ae_mse', 'model.pth.tar'), map_location='cpu', weights_only=False)
    assert set(ck) == {'epoch', 'state_dict', 'optimizer'} and int(ck['epoch']) == int(t['p1/ckpt_epoch'])
    assert sorted(ck['state_dict'].keys()) == list(t['p1/ckpt_keys'])


def test_pretrain_trainer_on_split_products_follows_reference(run_dir, tmp_path):
    """The same 16 pretrain steps with the f32 step's products as three-term bf16 splits (--f32_products x3): products good to ~2^-17 instead of
    2^-24, so the trajectory separates from the reference's a little earlier than the exact mode's -- measured (profiles/r4_traj_deviation.json,
    key p1[x3]) and bounded at 3 x that."""
    from deep_interpolation_clustering_amd.pretrain_interp import Net
    from deep_interpolation_clustering_amd.pretrain_trainer import Trainer
    t, sd_end = p1_state()
    plain = load('netstep_plain.npz')
    args = trainer_args(f32_products='x3')
    dev = torch.device('cuda')
    net = Net(args, dev)
    net.load_state_dict({k[4:]: torch.tensor(v) for k, v in plain.items() if k.startswith('sd0/') and 'cluster' not in k}, strict=True)
    tr = Trainer(args, net, make_loaders(args, dev, 'ragged'), str(tmp_path / 'Pretrain'), dev)
    assert tr.stepper.precision == 'x3'
    rec = spy_steps(tr, ['ae_mse'])
    tr.train()
    got = np.array([[float(v) for v in row] for row in rec])[:, 0]
    ref = t['p1/train_ae_mse']
    log_deviation('p1[x3]', step_loss_rel=np.abs(got - ref) / np.abs(ref))
    np.testing.assert_allclose(got[:1], ref[:1], rtol=TOL['p1x3_first1'])          # the north-star bar, on the step that measures the kernels
    np.testing.assert_allclose(got[:2], ref[:2], rtol=TOL['p1x3_first2'])
    np.testing.assert_allclose(got, ref, rtol=TOL['p1x3_all'])


# ---------------------------------------------------------------------------------------------------------------- p3 trajectory
@pytest.mark.parametrize('K,tag,kind,products', [(4, 'p3', 'ragged', None), (6, 'p3k6', 'ragged', None), (4, 'p3', 'padded', None),
                                                 (4, 'p3', 'ragged', 'x3'), (6, 'p3k6', 'ragged', 'x3')])
def test_cluster_trainer_follows_reference(run_dir, K, tag, kind, products, tmp_path):
    """(products 'x3': --f32_products x3, every dense product of the step a three-term bf16 split on the matrix cores -- k-means initialisation from
    its latents, the 24 joint steps, the label deltas; tolerances of their own, 3 x measured: TOL['p3x3_*'].)"""
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.clustering_trainer import TrainerCluster
    from deep_interpolation_clustering_amd.utils import set_seed
    t, sd_p1 = p1_state()
    args = trainer_args(loss='ae_mse_kl', cluster_number=K, dc_restore_metric='ae_mse', init_cluster_center='kmeans',
                        stopping_delta=None, update_interval=1, max_epochs=4, f32_products=products)
    x3 = 'x3' if products == 'x3' else ''
    dev = torch.device('cuda')
    set_seed(args.seed)                                            # np.random.seed(7529): the k-means++ draws (p3:110)
    torch.manual_seed(7529)
    net = Net(args, dev)
    pre, exp = str(tmp_path / 'Pretrain'), str(tmp_path / 'Clustering')
    tr = TrainerCluster(args, net, make_loaders(args, dev, kind), exp, pre, dev)
    write_checkpoint(os.path.join(pre, 'weight', 'ae_mse', 'model.pth.tar'), sd_p1, 2, torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))]))
    assert tr.stepper.precision == products
    rec = spy_steps(tr, ['loss', 'ae_mse', 'kl'])
    seen = {}
    inner_init, inner_gpc = tr.init_cluster_center, tr.generate_pred_cluster

    def init_spy(c):
        seen['centers'] = c.detach().cpu().numpy().copy()
        return inner_init(c)

    def gpc_spy(scope, dl, prev, denoise=False):
        seen.setdefault('prev', np.asarray(prev).copy())
        r = inner_gpc(scope, dl, prev, denoise)
        seen.setdefault('delta', []).append(r[0])
        seen.setdefault('labels', []).append(np.asarray(r[1]).copy())
        return r
    tr.init_cluster_center, tr.generate_pred_cluster = init_spy, gpc_spy
    tr.train()
    # k-means initialisation (clustering_trainer.py:72-82): same restart wins, same cluster order, centres to f32 rounding
    np.testing.assert_allclose(seen['centers'], t[f'{tag}/kmeans_centers'], rtol=2e-4, atol=2e-5)
    assert (seen['prev'] == t[f'{tag}/valid_prev_labels']).all()
    got = np.array([[float(v) for v in row] for row in rec])
    ref = t[f'{tag}/train_losses']
    assert got.shape == ref.shape == (24, 3)
    log_deviation(f'p3[{tag},{kind}{"," + x3 if x3 else ""}]', step_loss_rel=(np.abs(got - ref) / np.abs(ref)).max(axis=1), kl_rel=np.abs(got[:, 2] - ref[:, 2]) / np.abs(ref[:, 2]),
                  centers_rel=float(np.abs(seen['centers'] - t[f'{tag}/kmeans_centers']).max() / np.abs(t[f'{tag}/kmeans_centers']).max()))
    np.testing.assert_allclose(got[:1], ref[:1], rtol=1e-5, atol=0)                            # the first joint step: north_star's 1e-5 on loss, ae_mse AND kl, no absolute floor, in either products mode
    np.testing.assert_allclose(got[:2], ref[:2], rtol=TOL[f'p3{x3}_first2'], atol=0)           # loss, ae_mse AND kl: no absolute floor
    np.testing.assert_allclose(got[:8], ref[:8], rtol=TOL[f'p3{x3}_first8'])
    np.testing.assert_allclose(got, ref, rtol=TOL[f'p3{x3}_all'])
    delta, ref_delta = np.array(seen['delta']), t[f'{tag}/delta']
    if K == 4:
        assert (delta == ref_delta).all() and all((a == b).all() for a, b in zip(seen['labels'], t[f'{tag}/valid_labels']))
    else:
        assert delta[0] == ref_delta[0] and (seen['labels'][0] == t[f'{tag}/valid_labels'][0]).all()
        assert ref_delta[1] > 0 and np.abs(delta - ref_delta).max() <= 0.03       # over-segmented: a few of 100 labels sit on a boundary
    assert tr.optimizer.param_groups[0]['lr'] == pytest.approx(float(t[f'{tag}/lr_end']), rel=1e-12)
    sd = net.state_dict()
    for key in ('sci.kernel', 'cci.kernel', 'rbf.kernel', 'cluster_assignment.cluster_centers'):
        np.testing.assert_allclose(sd[key].cpu().numpy(), t[f'{tag}sd/{key}'], rtol=5e-3, atol=5e-4, err_msg=key)
    norm_dev = {}
    for key in (k for k in t if k.startswith(f'{tag}sdn/')):
        gn = float(torch.linalg.vector_norm(sd[key.split('/', 1)[1]].double()))
        norm_dev[key] = abs(gn - float(t[key])) / float(t[key])
        np.testing.assert_allclose(gn, float(t[key]), rtol=TOL[f'p3{x3}_param_norms'], err_msg=key)
    log_deviation(f'p3[{tag},{kind}{"," + x3 if x3 else ""}]/end', param_norm_rel_max=max(norm_dev.values()))


# ---------------------------------------------------------------------------------------------------------------- feature dump
@pytest.mark.parametrize('kind', ['device', 'host', 'device-cpu_padded_ob', 'host-cpu_padded_ob', 'device-x3'])
def test_feature_dump_equals_reference(run_dir, kind, tmp_path):
    """TrainerCluster.eval('validation', generate_feat=True) from the reference's own p3 checkpoint: the dictionary np.save writes."""
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.clustering_trainer import TrainerCluster
    f = load('featdump_cfg1.npz')
    cpu_ob = kind.endswith('cpu_padded_ob')            # --cpu_padded_ob: the padded 'ob' slots as the reference's CPU run (the fixture) dumps them
    products = 'x3' if kind.endswith('-x3') else None  # --f32_products x3: the evaluation forward on three-term bf16 split products
    kind = kind.split('-')[0]
    args = trainer_args(loss='ae_mse_kl', cluster_number=4, dc_restore_metric='ae_mse', init_cluster_center='kmeans', mode='eval',
                        stopping_delta=None, update_interval=1, host_loader=(kind == 'host'), cpu_padded_ob=cpu_ob, f32_products=products)
    dev = torch.device('cuda')
    net = Net(args, dev)
    exp = str(tmp_path / 'Clustering')
    tr = TrainerCluster(args, net, make_loaders(args, dev, kind), exp, str(tmp_path / 'Pretrain'), dev)
    state = {k[3:]: torch.tensor(v) for k, v in f.items() if k.startswith('sd/')}
    assert set(state) == set(net.state_dict())                                        # a reference checkpoint loads key for key
    write_checkpoint(os.path.join(exp, 'weight', 'ae_mse', 'model.pth.tar'), state, int(f['ckpt_epoch']), tr.optimizer)
    tr.eval('validation', generate_feat=True, viz_feat=False, denoise=False)
    dump = np.load(os.path.join(exp, 'out_feat', 'ae_mse', 'validation.npy'), allow_pickle=True).item()
    assert sorted(dump.keys()) == list(f['keys'])
    assert tr.epoch == int(f['ckpt_epoch'])
    for k in f['keys']:
        got, ref = np.asarray(dump[k]), f[f'dump/{k}']
        assert got.shape == tuple(f[f'shape/{k}']), k
        assert str(got.dtype) == str(f[f'dtype/{k}']), (k, got.dtype)
        if k in ('encounter_id', 'padding_mask', 'timestamp', 'ae_mask'):
            np.testing.assert_array_equal(got, ref, err_msg=k)
        elif k == 'ob' and cpu_ob:                                                    # the whole array, padded slots included, as the fixture has it
            np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-4, err_msg=k)
        elif k == 'ob':                                                               # re_norm_data: back in physiologic units (f32 arithmetic)
            m = f['dump/padding_mask'] > 0
            np.testing.assert_allclose(got[m], ref[m], rtol=1e-6, atol=1e-4, err_msg=k)
            # padded slots: upstream dumps batch_sample['ob'] AFTER `ob *= padding_mask` (clustering_trainer.py:299-303), which edits the
            # loader's tensor in place only when `.to(device)` is the identity, i.e. on a CPU run like the one that wrote the fixture
            # (0 -> mid-range after re_norm_data); on the GPU runs upstream is made for the dump keeps the loader's -scale/2 (-> range minimum),
            # which is what this package writes
            lo = np.array([20, 5, 0, 24, 0, 0], np.float32)[None, :, None]              # info.MIN_MAX_VALUES minima
            mid = np.array([160, 115, 150, 34.5, 50, 30], np.float32)[None, :, None]
            np.testing.assert_allclose(ref[~m], np.broadcast_to(mid, ref.shape)[~m], rtol=1e-6)
            np.testing.assert_allclose(got[~m], np.broadcast_to(lo, got.shape)[~m], rtol=1e-6)
        elif k == 'rec_ob':
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-3, err_msg=k)          # units of mmHg / bpm: values are O(100)
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-6, err_msg=k)
    assert (dump['cluster_pred'].argmax(1) == f['dump/cluster_pred'].argmax(1)).all()


@pytest.mark.parametrize('kind', ['ragged', 'device', 'padded'])
def test_trainer_takes_the_fused_reconstruction_path(run_dir, tmp_path, monkeypatch, kind):
    """the trainers hand Stepper BOTH the padding mask and the prefix lengths; the step must still run the de-interpolation
    kernels that emit the reconstruction loss themselves (ops.rbf_rec_loss) -- what bench.py times -- not the masked-MSE pass."""
    from deep_interpolation_clustering_amd import ops
    from deep_interpolation_clustering_amd.pretrain_interp import Net
    from deep_interpolation_clustering_amd.pretrain_trainer import Trainer
    args = trainer_args(max_epochs=2)
    dev = torch.device('cuda')
    torch.manual_seed(1)
    net = Net(args, dev)
    tr = Trainer(args, net, make_loaders(args, dev, kind), str(tmp_path / 'P'), dev)
    calls = {'fused': 0, 'mse': 0}
    inner_f, inner_m = ops.rbf_rec_loss, ops.masked_mse
    monkeypatch.setattr(ops, 'rbf_rec_loss', lambda *a, **k: (calls.__setitem__('fused', calls['fused'] + 1), inner_f(*a, **k))[1])
    monkeypatch.setattr(ops, 'masked_mse', lambda *a, **k: (calls.__setitem__('mse', calls['mse'] + 1), inner_m(*a, **k))[1])
    tr.train_one_epoch(tr.train_dl, denoise=False)
    assert calls == {'fused': 8, 'mse': 0}, calls

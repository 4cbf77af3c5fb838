"""Fused clip + Adam(amsgrad) over the flat bucket (csrc/dic_optim.hip, flat_adam.py) against torch.optim.Adam +
torch.nn.utils.clip_grad_norm_ -- the tail of the reference's training step (pretrain_trainer.py:228-229, utils.py:83)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(dev):
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(37, 64), torch.nn.Tanh(), torch.nn.Linear(64, 5)).to(dev)


def test_flat_adam_matches_torch_adam_and_roundtrips_state():
    from deep_interpolation_clustering_amd import dist
    from deep_interpolation_clustering_amd.flat_adam import FlatAdam
    dev = torch.device('cuda')
    m1, m2 = _model(dev), _model(dev)
    flat = dist.FlatParams(m1)
    o1 = FlatAdam(m1.parameters(), lr=3e-3, weight_decay=4e-4).bind(flat)
    o2 = torch.optim.Adam(m2.parameters(), lr=3e-3, weight_decay=4e-4, amsgrad=True)
    x = torch.randn(200, 37, device=dev)
    y = torch.randn(200, 5, device=dev)

    def run(m, o, fl, steps, scale):
        for s in range(steps):
            if fl is not None:
                fl.zero_grad()
            else:
                o.zero_grad()
            ((m(x) - y) ** 2).mean().mul(scale * (1 + s)).backward()
            if fl is not None:
                total, coef = fl.clip_coef(0.5)
                o.step(grad_scale=coef)
            else:
                total = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.5)
                o.step()
        return float(total)

    t1, t2 = run(m1, o1, flat, 7, 3.0), run(m2, o2, None, 7, 3.0)       # large loss scale: the clip is active
    assert abs(t1 - t2) <= 1e-5 * abs(t2)
    for p, q in zip(m1.parameters(), m2.parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
    # gradients were scaled in place, as clip_grad_norm_ leaves them
    for p, q in zip(m1.parameters(), m2.parameters()):
        np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.cpu().numpy(), rtol=2e-5, atol=1e-7)

    # state_dict compatibility in both directions, then keep training: trajectories stay together
    sd1, sd2 = copy.deepcopy(o1.state_dict()), copy.deepcopy(o2.state_dict())
    assert set(sd1['state'][0]) == set(sd2['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq', 'max_exp_avg_sq'}
    m3, m4 = copy.deepcopy(m2), copy.deepcopy(m1)
    flat3 = dist.FlatParams(m3)
    o3 = FlatAdam(m3.parameters(), lr=3e-3, weight_decay=4e-4).bind(flat3)
    o3.load_state_dict(sd2)                                              # torch Adam -> FlatAdam
    o4 = torch.optim.Adam(m4.parameters(), lr=3e-3, weight_decay=4e-4, amsgrad=True)
    o4.load_state_dict(sd1)                                              # FlatAdam -> torch Adam
    run(m3, o3, flat3, 3, 0.01), run(m4, o4, None, 3, 0.01)              # small loss scale: clip inactive (coef = 1)
    run(m1, o1, flat, 3, 0.01), run(m2, o2, None, 3, 0.01)
    for name, a, b in (('torch->flat', m3, m2), ('flat->torch', m4, m1), ('flat vs torch', m1, m2)):
        for p, q in zip(a.parameters(), b.parameters()):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=5e-5, atol=5e-6, err_msg=name)


def test_flat_adam_requires_binding():
    from deep_interpolation_clustering_amd.flat_adam import FlatAdam
    m = _model(torch.device('cuda'))
    with pytest.raises(RuntimeError):
        FlatAdam(m.parameters(), lr=1e-3).step()


def test_captured_step_follows_the_scheduler_and_cache_is_bounded():
    """the fused optimiser reads lr from device memory, so ONE captured hipGraph serves every learning rate (no
    re-capture per scheduler step), the graph cache is bounded, and a graphed trajectory with a changing lr equals the eager one."""
    from types import SimpleNamespace
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={},
                           fake_detection=False, triple_margin=0.0, cluster_number=4, loss='ae_mse_kl', grad_clip=15.0,
                           unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
    dev = torch.device('cuda')
    coh = synthetic.make_cohort(384, seed=8)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
    traj = {}
    for graphs in (False, True):
        torch.manual_seed(4)
        net = Net(args, dev).to(dev)
        net.train()
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16, use_graphs=graphs)
        out = []
        for i in range(6):
            st.optimizer.param_groups[0]['lr'] = 3e-3 * (0.5 ** i)             # a scheduler step before every batch
            losses, gnorm, _ = st.step(X[:256], OB[:256], None, LEN[:256])
            out.append([float(losses['loss'].detach()), float(gnorm)])
        if graphs:
            assert len(st._graphs) == 1                                          # lr is not part of the graph key
            for b in (64, 96, 128, 160):                                         # four more shapes: the cache stays bounded
                st.step(X[:b], OB[:b], None, LEN[:b])
            assert len(st._graphs) <= st.MAX_GRAPHS
        traj[graphs] = np.array(out)
    np.testing.assert_allclose(traj[True], traj[False], rtol=2e-3)

#!/usr/bin/env python3
"""Golden vectors for the loss terms the plain / fake-detection fixtures do not reach (clustering_interp.py:209-247,
pretrain_interp.py): supervised auxiliary heads (masked future-vital MSE, weighted BCE tasks), and the triplet term.

TEST INFRASTRUCTURE ONLY; run in the build container: ``PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_sup.py``.
IMPORTS the reference (read-only) and writes tests/golden/netstep_{sup,triplet}.npz -- arrays only.  The private cohort's
labels are replaced by seeded synthetic ones; everything else is the reference's own code path:
  sup:      clustering_interp.Net, aux_tasks {future_vital, AKI_overall, ICU_24h}, fake detection, ae_mse_sup_fake_detect_kl
  triplet:  clustering_interp.Net, fake detection + triplet margin, ae_mse_fake_detect_triplet
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, import_reference, synth_stack          # noqa: E402


TRIPLE_MARGIN = 1.5      # large enough for the hinge to be active on part of the batch (asserted below)


def t(a):
    return torch.tensor(a)


PLAIN = dict(np.load(os.path.join(OUT, 'netstep_plain.npz')))       # same torch seed and shapes: shared layers start identically


def pack_step(pack, net, sd0, losses, grads, gnorm):
    for k, v in losses.items():
        pack['loss_' + k] = float(v)
    pack['gnorm'] = float(gnorm)
    for k, v in sd0.items():                 # only what differs from the 'plain' fixture's initial state (the test overlays it)
        if 'sd0/' + k in PLAIN and np.array_equal(PLAIN['sd0/' + k], v):
            continue
        pack['sd0/' + k] = v
    for k, v in net.state_dict().items():
        v = v.detach().numpy()
        if v.size <= 4096:
            pack['sd1/' + k] = v
        else:
            pack['sd1n/' + k] = np.float64(np.linalg.norm(v.astype(np.float64)))
    for k, v in grads.items():
        if v.size <= 4096:
            pack['g/' + k] = v
        else:
            pack['gn/' + k] = np.float64(np.linalg.norm(v.astype(np.float64)))


def main():
    ref = import_reference()
    rng = np.random.default_rng(424242)
    B, C, T, R, H, K = 12, 6, 48, 12, 24, 4          # C, T, R, K as in netstep_plain
    x = synth_stack(rng, B, C, T, H, 'ragged', 20)
    mask, ob = x[:, C:2 * C], x[:, :C].copy()
    fx = x.copy()
    fx[:, :C] = rng.uniform(-2.5, 2.5, (B, C, T)).astype(np.float32) * mask
    perm = rng.permutation(2 * B)
    fake_label = np.concatenate([np.ones(B), np.zeros(B)])[perm].astype(np.int64)
    weights_unsup = {'fake_detection': 1.0, 'triplet': 1.0, 'kl': 10.0}

    # ---- supervised heads + fake detection + kl on the clustering net
    aux_tasks = {'future_vital': 0.5, 'AKI_overall': 0.3, 'ICU_24h': 0.2}
    pos_w = {'AKI_overall': 2.0, 'ICU_24h': 1.5}
    args = SimpleNamespace(num_variables=C, num_timestamps=T, ref_points=R, hours_from_admission=H, dropout=0.0, aux_tasks=aux_tasks,
                           aux_pos_weights=pos_w, fake_detection=True, triple_margin=0.0, cluster_number=K)
    torch.manual_seed(7529)
    net = ref.cnet.Net(args, torch.device('cpu'))
    net.train()
    sd0 = {k: v.detach().clone().numpy() for k, v in net.state_dict().items()}
    labels = {'future_vital': rng.uniform(0, 1, (B, C)).astype(np.float32),
              'AKI_overall': rng.integers(0, 2, B).astype(np.float32), 'ICU_24h': rng.integers(0, 2, B).astype(np.float32)}
    fv_mask = (rng.uniform(0, 1, (B, C)) < 0.7).astype(np.float32)
    opt = torch.optim.Adam(net.parameters(), lr=3e-3, weight_decay=4e-4, amsgrad=True)
    opt.zero_grad()
    z, y, aux = net(t(x), t(fx), t(perm), None)
    rec = net.rec_loss(t(ob), y, t(mask))
    terms = net.sup_aux_loss(aux_tasks, {k: t(v) for k, v in labels.items()}, aux, t(fv_mask))
    terms.update(net.fake_det_loss(t(fake_label), aux['fake_det']))
    terms.update(net.kl_loss(aux['cluster_label'], aux['cluster_pred']))
    tasks = dict(aux_tasks)
    tasks.update(weights_unsup)
    losses = net.multi_task_loss(tasks, rec, terms)
    losses['loss'].backward()
    grads = {k: p.grad.detach().clone().numpy() for k, p in net.named_parameters()}
    gnorm = torch.nn.utils.clip_grad_norm_(net.parameters(), 15.0)
    opt.step()
    pack = dict(x=x, ob=ob, fake_x=fx, fake_perm_idx=perm, fake_label=fake_label, R=R, H=H, K=K, fv_mask=fv_mask,
                pos_w_AKI_overall=pos_w['AKI_overall'], pos_w_ICU_24h=pos_w['ICU_24h'], z=z.detach().numpy(),
                future_vital_pred=aux['future_vital'].detach().numpy())
    for k, v in labels.items():
        pack['label_' + k] = v
    for k, v in aux_tasks.items():
        pack['w_' + k] = v
    pack_step(pack, net, sd0, losses, grads, gnorm)
    np.savez_compressed(os.path.join(OUT, 'netstep_sup.npz'), **pack)
    print('sup', {k: float(v) for k, v in losses.items()}, float(gnorm))

    # ---- triplet + fake detection (only clustering_interp.Net implements the triplet branch, :171-180, :234-236)
    args = SimpleNamespace(num_variables=C, num_timestamps=T, ref_points=R, hours_from_admission=H, dropout=0.0, aux_tasks={},
                           aux_pos_weights={}, fake_detection=True, triple_margin=TRIPLE_MARGIN, cluster_number=K)
    torch.manual_seed(7529)
    net = ref.cnet.Net(args, torch.device('cpu'))
    net.train()
    sd0 = {k: v.detach().clone().numpy() for k, v in net.state_dict().items()}
    px = x.copy()
    px[:, :C] = (x[:, :C] + rng.normal(0, 0.3, (B, C, T)).astype(np.float32)) * mask
    px[:, 2 * C:3 * C] = (x[:, 2 * C:3 * C] + rng.normal(0, 0.01, (B, C, T)).astype(np.float32)) * mask
    opt = torch.optim.Adam(net.parameters(), lr=3e-3, weight_decay=4e-4, amsgrad=True)
    opt.zero_grad()
    z, y, aux = net(t(x), t(fx), t(perm), t(px))
    rec = net.rec_loss(t(ob), y, t(mask))
    terms = net.fake_det_loss(t(fake_label), aux['fake_det'])
    terms.update(net.triplet_loss(z, aux['positive'], aux['negative'], args.triple_margin))
    assert float(terms['triplet']) > 0.05, float(terms['triplet'])
    losses = net.multi_task_loss(weights_unsup, rec, terms)
    losses['loss'].backward()
    grads = {k: p.grad.detach().clone().numpy() for k, p in net.named_parameters() if p.grad is not None}
    gnorm = torch.nn.utils.clip_grad_norm_(net.parameters(), 15.0)
    opt.step()
    pack = dict(x=x, ob=ob, fake_x=fx, positive_x=px, fake_perm_idx=perm, fake_label=fake_label, R=R, H=H, K=K, margin=args.triple_margin,
                z=z.detach().numpy())
    pack_step(pack, net, sd0, losses, grads, gnorm)
    np.savez_compressed(os.path.join(OUT, 'netstep_triplet.npz'), **pack)
    print('triplet', {k: float(v) for k, v in losses.items()}, float(gnorm))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (read-only) on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the build container, where ``/root/reference``
exists:  ``PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py``.
Only arrays (inputs, parameters, outputs, gradients) are written; no reference
source or bytecode is copied.  The uninstalled ``tensorflow`` / ``warmup_scheduler``
/ ``tensorboardX`` modules are replaced by inert stand-ins (they touch TF seeding,
an unused scheduler mode and logging only -- none of the path's arithmetic).
"""
import importlib.machinery
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = os.environ.get('DIC_REFERENCE', '/root/reference')
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def _standin(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    tf = _standin('tensorflow')
    tf.random = SimpleNamespace(set_seed=lambda s: None)
    _standin('warmup_scheduler', GradualWarmupScheduler=object)

    class _Writer:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def add_embedding(self, *a, **k):
            pass

    _standin('tensorboardX', SummaryWriter=_Writer)
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import clustering_interp
    import dec
    import interpolation_layer
    import pretrain_interp
    import rbf
    return SimpleNamespace(il=interpolation_layer, dec=dec, rbf=rbf, cnet=clustering_interp, pnet=pretrain_interp)


def synth_stack(rng, B, C, T, H, mode='ragged', lam=None):
    """Stacked (B,4C,T) input.  mode 'ragged' = prefix masks with Poisson lengths
    (what p0/the trainers produce); 'random' = arbitrary 0/1 masks (upstream smoke test)."""
    val = rng.normal(0, 1.2, (B, C, T)).astype(np.float32)
    tim = np.zeros((B, C, T), np.float32)
    mask = np.zeros((B, C, T), np.float32)
    if mode == 'ragged':
        lam = lam or max(2, T // 2)
        n = np.clip(rng.poisson(lam, (B, C)), 1, T)
        for b in range(B):
            for c in range(C):
                k = n[b, c]
                tim[b, c, :k] = np.sort(rng.uniform(0, H, k)).astype(np.float32)
                mask[b, c, :k] = 1
    else:
        tim = rng.uniform(0, H, (B, C, T)).astype(np.float32)
        mask = rng.integers(0, 2, (B, C, T)).astype(np.float32)
        mask[:, :, 0] = 1            # keep n >= 1 unless a case overrides it
    hold = rng.integers(0, 2, (B, C, T)).astype(np.float32)
    val = val * mask
    return np.concatenate([val, mask, tim, hold], axis=1)


def t(a, grad=False):
    return torch.tensor(a, dtype=torch.float32, requires_grad=grad)


def gen_interp(ref, rng):
    cases = {
        'smoke_6_30_11_6': dict(B=10, C=6, T=30, R=11, H=6, mode='random'),
        'cfg_6_96_24_24': dict(B=8, C=6, T=96, R=24, H=24, mode='ragged', lam=50),
        'default_6_354_6_6': dict(B=4, C=6, T=354, R=6, H=6, mode='ragged', lam=60),
        'wide_12_288_24_24': dict(B=3, C=12, T=288, R=24, H=24, mode='ragged', lam=200),
        'edge_6_17_7_24': dict(B=5, C=6, T=17, R=7, H=24, mode='ragged', lam=6, edge=True),
    }
    out = {}
    for name, cs in cases.items():
        B, C, T, R, H = cs['B'], cs['C'], cs['T'], cs['R'], cs['H']
        x = synth_stack(rng, B, C, T, H, cs['mode'], cs.get('lam'))
        if cs.get('edge'):
            # encounter 1 channel 2: exactly one observation; encounter 3 channel 0: none (NaN case)
            x[1, C + 2, :] = 0; x[1, C + 2, 0] = 1; x[1, 2, 1:] = 0
            x[3, C + 0, :] = 0; x[3, 0, :] = 0
        sci = ref.il.SingleChannelInterp(R, H, C, T, torch.device('cpu'))
        cci = ref.il.CrossChannelInterp(C, T, torch.device('cpu'))
        with torch.no_grad():
            sci.kernel.copy_(t(rng.uniform(-0.5, 1.5, C)))
            cci.kernel.copy_(torch.eye(C) + t(rng.normal(0, 0.2, (C, C))))
        xt = t(x)
        s = sci(xt)
        o = cci(s)
        cot = t(rng.normal(0, 1, tuple(o.shape)))
        if cs.get('edge'):
            g_sci = np.zeros(C, np.float32); g_cci = np.zeros((C, C), np.float32)   # NaN case: no grads pinned
        else:
            (o * cot).sum().backward()
            g_sci, g_cci = sci.kernel.grad.numpy(), cci.kernel.grad.numpy()
        out[name] = dict(x=x, R=R, H=H, sci_kernel=sci.kernel.detach().numpy(), cci_kernel=cci.kernel.detach().numpy(),
                         sci_out=s.detach().numpy(), cci_out=o.detach().numpy(), cot=cot.numpy(),
                         g_sci=g_sci, g_cci=g_cci)
    for name, d in out.items():
        np.savez_compressed(os.path.join(OUT, f'interp_{name}.npz'), **d)


def gen_rbf(ref, rng):
    cases = {
        'smoke_6_30_11_6': dict(B=10, C=6, T=30, R=11, H=6, mode='random'),
        'cfg_6_96_24_24': dict(B=8, C=6, T=96, R=24, H=24, mode='ragged', lam=50),
        'default_6_354_6_6': dict(B=4, C=6, T=354, R=6, H=6, mode='ragged', lam=60),
        'wide_12_288_24_24': dict(B=3, C=12, T=288, R=24, H=24, mode='ragged', lam=200),
    }
    basis = ref.rbf.basis_func_dict()['gaussian']
    for name, cs in cases.items():
        B, C, T, R, H = cs['B'], cs['C'], cs['T'], cs['R'], cs['H']
        x = synth_stack(rng, B, C, T, H, cs['mode'], cs.get('lam'))
        layer = ref.rbf.RBF(H, R, 256, C, 0.0, basis, torch.device('cpu'))
        with torch.no_grad():
            layer.kernel.copy_(t(rng.uniform(-0.5, 1.5, C)))
        # isolate the de-interpolation: make compress_fc the identity on a (B,R,C) input
        layer.compress_fc = torch.nn.Identity()
        v = t(rng.normal(0, 1, (B, C, R)), grad=True)
        y = layer(v, t(x))                       # forward permutes (B,C,R)->(B,R,C)->identity->(B,C,R)
        ob = t(rng.normal(0, 1.2, (B, C, T)))
        mask = t(x[:, C:2 * C])
        # rec_loss through the reference Net method (unbound; uses no self state)
        loss = ref.cnet.Net.rec_loss(None, ob * mask, y, mask)['ae_mse']
        loss.backward()
        np.savez_compressed(os.path.join(OUT, f'rbf_{name}.npz'), x=x, R=R, H=H, kernel=layer.kernel.detach().numpy(),
                            v=v.detach().numpy(), y=y.detach().numpy(), ob=(ob * mask).numpy(), loss=loss.detach().numpy(),
                            g_v=v.grad.numpy(), g_kernel=layer.kernel.grad.numpy())


def gen_dec(ref, rng):
    for K in (2, 4, 8, 16, 20):
        B, D = 37, 256
        z = rng.uniform(-1, 1, (B, D)).astype(np.float32) * 0.4
        mu = (z[rng.choice(B, K, replace=False)] + rng.normal(0, 0.05, (K, D))).astype(np.float32)
        ca = ref.dec.ClusterAssignment(K, D, 1.0, cluster_centers=t(mu))
        zt = t(z, grad=True)
        q = ca(zt)
        p = ref.dec.target_distribution(q).detach()
        kl = ref.cnet.Net.kl_loss(None, p, q)['kl']
        kl.backward()
        np.savez_compressed(os.path.join(OUT, f'dec_K{K}.npz'), z=z, mu=mu, q=q.detach().numpy(), p=p.numpy(),
                            kl=kl.detach().numpy(), g_z=zt.grad.numpy(), g_mu=ca.cluster_centers.grad.numpy())


def gen_net_step(ref, rng):
    """One full clustering_interp.Net step (dropout 0): ae_mse + 10*kl (+ fake detection variant).

    The 'fake' fixture stores only the initial parameters that differ from the 'plain' one
    (same torch seed -> identical sci/cci/LSTM/rbf initialisation); the test overlays them."""
    sd0_plain = {}
    for name, fake in (('plain', False), ('fake', True)):
        B, C, T, R, H, K = 16, 6, 48, 12, 24, 4
        args = SimpleNamespace(num_variables=C, num_timestamps=T, ref_points=R, hours_from_admission=H, dropout=0.0,
                               aux_tasks={}, fake_detection=fake, triple_margin=0.0, cluster_number=K)
        torch.manual_seed(7529)
        net = ref.cnet.Net(args, torch.device('cpu'))
        net.train()
        sd0 = {k: v.detach().clone().numpy() for k, v in net.state_dict().items()}
        x = synth_stack(rng, B, C, T, H, 'ragged', 24)
        mask = x[:, C:2 * C]
        ob = x[:, :C].copy()
        xt, obt, mt = t(x), t(ob), t(mask)
        kw = {}
        if fake:
            fx = x.copy()
            fx[:, :C] = (rng.uniform(-2.5, 2.5, (B, C, T)).astype(np.float32)) * mask
            perm = rng.permutation(2 * B)
            label = np.concatenate([np.ones(B), np.zeros(B)])[perm].astype(np.int64)
            kw = dict(fake_x=fx, fake_perm_idx=perm, fake_label=label)
        opt = torch.optim.Adam(net.parameters(), lr=3e-3, weight_decay=4e-4, amsgrad=True)
        opt.zero_grad()
        if fake:
            z, y, aux = net(xt, t(kw['fake_x']), torch.tensor(kw['fake_perm_idx']), None)
        else:
            z, y, aux = net(xt, None, None, None)
        rec = net.rec_loss(obt, y, mt)
        extra = net.kl_loss(aux['cluster_label'], aux['cluster_pred'])
        if fake:
            extra.update(net.fake_det_loss(torch.tensor(kw['fake_label']), aux['fake_det']))
        losses = net.multi_task_loss({'fake_detection': 1.0, 'triplet': 1.0, 'kl': 10.0}, rec, extra)
        losses['loss'].backward()
        grads = {k: p.grad.detach().clone().numpy() for k, p in net.named_parameters()}
        gnorm = torch.nn.utils.clip_grad_norm_(net.parameters(), 15.0)
        opt.step()
        sd1 = {k: v.detach().numpy() for k, v in net.state_dict().items()}
        pack = dict(x=x, ob=ob, R=R, H=H, K=K, z=z.detach().numpy(), y=y.detach().numpy(),
                    q=aux['cluster_pred'].detach().numpy(), p=aux['cluster_label'].numpy(), gnorm=float(gnorm))
        for k, v in losses.items():
            pack['loss_' + k] = float(v)
        for k, v in kw.items():
            pack[k] = v
        for k, v in sd0.items():
            if name == 'plain':
                sd0_plain[k] = v
            elif k in sd0_plain and np.array_equal(sd0_plain[k], v):
                continue
            pack['sd0/' + k] = v
        # after-step parameters: small tensors in full, big ones as (norm, first 64 values)
        for k, v in sd1.items():
            if v.size <= 4096:
                pack['sd1/' + k] = v
            else:
                pack['sd1n/' + k] = np.float64(np.linalg.norm(v.astype(np.float64)))
                pack['sd1h/' + k] = v.reshape(-1)[:64]
        for k, v in grads.items():
            if v.size <= 4096:
                pack['g/' + k] = v
            else:
                pack['gn/' + k] = np.float64(np.linalg.norm(v.astype(np.float64)))
                pack['gh/' + k] = v.reshape(-1)[:64]
        np.savez_compressed(os.path.join(OUT, f'netstep_{name}.npz'), **pack)


def gen_kmeans(rng):
    """sklearn (third-party, version pinned in the fixture) KMeans with a fixed init array.
    X is regenerated in the test from the stored seed (oracle/synth.py: latent_blobs)."""
    import sklearn
    from sklearn.cluster import KMeans
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from oracle.synth import latent_blobs
    for K in (4, 16):
        seed, N, D = 1000 + K, 3000, 256
        X, _ = latent_blobs(seed, N, D, K)
        Xv, _ = latent_blobs(seed + 1, 500, D, K, centers_seed=seed)
        init_idx = rng.choice(N, K, replace=False)
        km = KMeans(n_clusters=K, init=X[init_idx].copy(), n_init=1).fit(X)
        np.savez_compressed(os.path.join(OUT, f'kmeans_K{K}.npz'), seed=seed, N=N, D=D, K=K, init_idx=init_idx,
                            labels=km.labels_.astype(np.int32), centers=km.cluster_centers_,
                            inertia=np.float64(km.inertia_), n_iter=km.n_iter_,
                            pred=km.predict(Xv).astype(np.int32), sklearn_version=sklearn.__version__)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    rng = np.random.default_rng(7529)
    gen_interp(ref, rng)
    gen_rbf(ref, rng)
    gen_dec(ref, rng)
    gen_net_step(ref, rng)
    gen_kmeans(rng)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print('golden fixtures written to', os.path.normpath(OUT), f'({total / 1e6:.2f} MB)')


if __name__ == '__main__':
    main()

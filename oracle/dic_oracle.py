"""CPU restatement (torch, any float dtype) of the reference hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Parity of this file is
PINNED: ``tests/test_oracle_golden.py`` checks every function below against
golden vectors produced by importing the reference itself
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``).

Each function cites the reference lines (relative to the upstream repository)
whose arithmetic it restates.  The restatement is written from the closed-form
math with broadcasting; it does not reproduce the
reference's op-by-op sequence of ``repeat`` temporaries.

Index convention: b encounter, c channel, t observation slot, r grid point.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

KAPPA = 10.0          # interpolation_layer.py:80 (high-pass bandwidth multiplier)
RBF_EPS = 1e-10       # rbf.py:107


def ref_grid(hours: float, ref_points: int, dtype, device=None) -> torch.Tensor:
    """interpolation_layer.py:41 / rbf.py:44: ``linspace(0, H, R)``."""
    return torch.linspace(0, hours, ref_points, dtype=dtype, device=device)


def split_planes(x: torch.Tensor, C: int):
    """interpolation_layer.py:26-30: planes of the stacked (B,4C,T) input."""
    return x[:, 0:C], x[:, C:2 * C], x[:, 2 * C:3 * C], x[:, 3 * C:4 * C]


# --------------------------------------------------------------------------- k1
def sci_forward(x: torch.Tensor, kernel: torch.Tensor, ref_points: int, hours: float) -> torch.Tensor:
    """SingleChannelInterp.forward, interpolation_layer.py:31-86.

    x (B,4C,T) -> (B,R,3C) = [y | w | y_trans] on the last axis.
    A channel with no observation gives NaN y/y_trans and -inf w, as upstream.
    """
    C = kernel.shape[0]
    val, mask, tim, _ = split_planes(x, C)
    ref = ref_grid(hours, ref_points, x.dtype, x.device)
    u = (tim[..., None] - ref) ** 2                        # (B,C,T,R)   :49
    alpha = F.softplus(kernel)[None, :, None, None]        # :51
    logm = torch.log(mask)[..., None]                      # :59 (log 0 = -inf)

    def smooth(scale):
        logit = -scale * alpha * u + logm
        lse = torch.logsumexp(logit, dim=2)                # (B,C,R)
        wt = torch.exp(logit - lse[:, :, None, :])
        return lse, (wt * val[..., None]).sum(dim=2)

    w, y = smooth(1.0)                                     # :59-64
    _, y_trans = smooth(KAPPA)                             # :80-83
    return torch.cat([y, w, y_trans], dim=1).permute(0, 2, 1)   # :84-85


# -------------------------------------------------------------------------- k1'
def cci_forward(s: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """CrossChannelInterp.forward, interpolation_layer.py:99-127.  (B,R,3C)->(B,R,3C)."""
    C = kernel.shape[0]
    y, w, y_trans = s[..., 0:C], s[..., C:2 * C], s[..., 2 * C:3 * C]      # (B,R,C)
    intensity = torch.exp(w)                                               # :104
    w_hat = torch.exp(w - torch.logsumexp(w, dim=2, keepdim=True))         # :108-110
    mean = y.mean(dim=1, keepdim=True)                                     # :111-112
    smooth = torch.matmul(w_hat * (y - mean), kernel) + mean               # :113
    return torch.cat([smooth, intensity, y_trans - smooth], dim=2)         # :122-126


def sci_cci_forward(x, sci_kernel, cci_kernel, ref_points, hours):
    return cci_forward(sci_forward(x, sci_kernel, ref_points, hours), cci_kernel)


# --------------------------------------------------------------------------- k2
def rbf_deinterp(v: torch.Tensor, raw_input: torch.Tensor, kernel: torch.Tensor,
                 ref_points: int, hours: float) -> torch.Tensor:
    """RBF.forward minus compress_fc, rbf.py:57-108 with ``gaussian`` (rbf.py:129-131).

    v (B,C,R) = compress_fc output, raw_input (B,4C,T) -> (B,C,T).
    """
    C = kernel.shape[0]
    _, mask, tim, _ = split_planes(raw_input, C)
    ref = ref_grid(hours, ref_points, raw_input.dtype, raw_input.device)
    dist = ((tim[..., None] - ref) ** 2) ** 0.5            # rbf.py:76 (|t - ref|)
    beta = F.softplus(kernel)[None, :, None, None]         # rbf.py:78
    phi = torch.exp(-beta * dist ** 2) * mask[..., None]   # rbf.py:95-96
    norm = phi.sum(dim=-1)                                 # rbf.py:97
    num = (phi * v[:, :, None, :]).sum(dim=-1)             # rbf.py:104-106
    return num / (norm + RBF_EPS) * mask                   # rbf.py:107


def rec_loss(org_ob, rec_ob, padding_mask):
    """Net.rec_loss, clustering_interp.py:197-203: masked SSE / #{mask == 1}."""
    diff = rec_ob * padding_mask - org_ob * padding_mask
    return (diff * diff).sum() / (padding_mask == 1.0).sum()


# --------------------------------------------------------------------------- k3
def dec_soft_assign(z: torch.Tensor, centers: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """ClusterAssignment.forward, dec.py:49-63."""
    d2 = ((z[:, None, :] - centers[None]) ** 2).sum(dim=2)     # :56
    num = (1.0 / (1.0 + d2 / alpha)) ** (float(alpha + 1) / 2)  # :57-60
    return num / num.sum(dim=1, keepdim=True)                   # :61


def dec_target(q: torch.Tensor) -> torch.Tensor:
    """target_distribution, dec.py:66-76 (f_j is a whole-batch column sum)."""
    weight = q ** 2 / q.sum(dim=0)
    return weight / weight.sum(dim=1, keepdim=True)


def kl_loss(p: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """Net.kl_loss, clustering_interp.py:205-207: kl_div(log q, p, 'batchmean')."""
    return F.kl_div(q.log(), p, reduction='batchmean')


# ---------------------------------------------------- closed-form backward (App. A)
def sci_cci_backward(x, sci_kernel, cci_kernel, ref_points, hours, grad_out):
    """Closed-form grads of <grad_out, cci(sci(x))> wrt (sci.kernel, cci.kernel).

    Restates SURVEY.md Appendix A k1/k1'; checked against autograd of the
    functions above in tests/test_oracle_golden.py.  Inputs carry no grad.
    """
    C = sci_kernel.shape[0]
    R = ref_points
    val, mask, tim, _ = split_planes(x, C)
    ref = ref_grid(hours, R, x.dtype, x.device)
    u = (tim[..., None] - ref) ** 2
    alpha = F.softplus(sci_kernel)[None, :, None, None]
    logm = torch.log(mask)[..., None]

    def stats(scale):
        logit = -scale * alpha * u + logm
        lse = torch.logsumexp(logit, dim=2)
        s = torch.exp(logit - lse[:, :, None, :])
        xb = val[..., None]
        return lse, (s * xb).sum(2), (s * u).sum(2), (s * xb * u).sum(2)

    w, y, eu1, exu1 = stats(1.0)
    _, yt, eu10, exu10 = stats(KAPPA)
    # to (B,R,C)
    yr, wr = y.permute(0, 2, 1), w.permute(0, 2, 1)
    g1, g2, g3 = grad_out[..., 0:C], grad_out[..., C:2 * C], grad_out[..., 2 * C:3 * C]
    w_hat = torch.softmax(wr, dim=2)
    mean = yr.mean(dim=1, keepdim=True)
    a = w_hat * (yr - mean)
    gs = g1 - g3
    g_K = torch.einsum('bri,brj->ij', a, gs)
    ga = gs @ cci_kernel.t()
    g_what = ga * (yr - mean)
    g_y = ga * w_hat + (gs.sum(1, keepdim=True) - (ga * w_hat).sum(1, keepdim=True)) / R
    g_w = g2 * torch.exp(wr) + w_hat * (g_what - (g_what * w_hat).sum(2, keepdim=True))
    g_yt = g3
    gy, gw, gyt = g_y.permute(0, 2, 1), g_w.permute(0, 2, 1), g_yt.permute(0, 2, 1)   # (B,C,R)
    g_alpha = (-gw * eu1 - gy * (exu1 - y * eu1) - KAPPA * gyt * (exu10 - yt * eu10)).sum(dim=(0, 2))
    return torch.sigmoid(sci_kernel) * g_alpha, g_K


def rbf_backward(v, raw_input, kernel, ref_points, hours, grad_y):
    """Closed-form grads of <grad_y, rbf_deinterp(...)> wrt (v, rbf.kernel)."""
    C = kernel.shape[0]
    _, mask, tim, _ = split_planes(raw_input, C)
    ref = ref_grid(hours, ref_points, raw_input.dtype, raw_input.device)
    u = (tim[..., None] - ref) ** 2
    beta = F.softplus(kernel)[None, :, None, None]
    phi = torch.exp(-beta * u) * mask[..., None]
    den = phi.sum(-1) + RBF_EPS
    S = (phi * v[:, :, None, :]).sum(-1)
    gm = grad_y * mask
    g_v = (gm[..., None] * phi / den[..., None]).sum(dim=2)
    inner = (-u * phi * v[:, :, None, :]).sum(-1) / den - S * (-u * phi).sum(-1) / den ** 2
    g_beta = (gm * inner).sum(dim=(0, 2))
    return g_v, torch.sigmoid(kernel) * g_beta


def dec_backward(z, centers, grad_q, alpha: float = 1.0):
    """Closed-form grads of <grad_q, dec_soft_assign(z, centers)> wrt (z, centers)."""
    diff = z[:, None, :] - centers[None]
    d2 = (diff ** 2).sum(2)
    t = 1.0 / (1.0 + d2 / alpha)
    n = t ** ((alpha + 1) / 2)
    s = n.sum(1, keepdim=True)
    q = n / s
    dn = -((alpha + 1) / (2 * alpha)) * n * t                   # dn/dd2
    coef = 2.0 * dn * (grad_q - (grad_q * q).sum(1, keepdim=True)) / s
    g_z = (coef[..., None] * diff).sum(1)
    g_c = -(coef[..., None] * diff).sum(0)
    return g_z, g_c


# ------------------------------------------------------------------ model (a4, a7)
class _Seq(nn.Module):
    """Holder giving the upstream ``<name>.model.<idx>`` parameter names."""

    def __init__(self, *layers):
        super().__init__()
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class _PerStep(nn.Module):
    """utils.py:202-224 TimeDistributed: fold (B,R,F) to (B*R,F) around ``module``."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        if x.dim() <= 2:
            return self.module(x)
        out = self.module(x.reshape(-1, x.shape[-1]))
        return out.reshape(x.shape[0], -1, out.shape[-1])


def _compress_fc(idim, odim, p):
    """rbf.py:111-125."""
    return _Seq(nn.Linear(idim, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Dropout(p), nn.Linear(128, odim))


def _head(idim, odim, p, tail=None):
    """clustering_interp.py:43-87 (AuxFc / FuturePredFc / FakeDetFc)."""
    layers = [nn.Linear(idim, 128), nn.BatchNorm1d(128), nn.Dropout(p), nn.Linear(128, odim)]
    if tail is not None:
        layers.append(tail)
    return _Seq(*layers)


class _Param(nn.Module):
    def __init__(self, name, value):
        super().__init__()
        self.register_parameter(name, nn.Parameter(value))


class _Lstm(nn.Module):
    def __init__(self, idim, hdim):
        super().__init__()
        self.lstm = nn.LSTM(idim, hdim, num_layers=1, dropout=0, bidirectional=True)


class _Rbf(nn.Module):
    def __init__(self, C, p):
        super().__init__()
        self.compress_fc = _PerStep(_compress_fc(256, C, p))
        self.kernel = nn.Parameter(torch.rand(C))


class OracleNet(nn.Module):
    """clustering_interp.Net (clustering_interp.py:89-247) / pretrain_interp.Net
    restated on the oracle ops, with upstream ``state_dict`` key names so golden
    checkpoints load directly.  ``clustering=False`` gives the pretrain variant.
    """

    def __init__(self, num_variables=6, ref_points=6, hours=6, cluster_number=4, dropout=0.0,
                 fake_detection=False, aux_tasks: Optional[Dict[str, float]] = None, clustering=True):
        super().__init__()
        C = num_variables
        self.C, self.R, self.H = C, ref_points, hours
        self.fake_detection = fake_detection
        self.aux_tasks = dict(aux_tasks or {})
        self.clustering = clustering
        self.sci = _Param('kernel', torch.rand(C))
        self.cci = _Param('kernel', torch.eye(C))
        self.encoder = _Lstm(3 * C, 128)
        self.decoder = _Lstm(256, 128)
        self.rbf = _Rbf(C, dropout)
        n_aux = len(self.aux_tasks)
        if 'future_vital' in self.aux_tasks:
            self.predict_future = _head(256, C, dropout, nn.Sigmoid())
            n_aux -= 1
        if n_aux > 0:
            self.aux_head = _head(256, n_aux, dropout)
        if fake_detection:
            self.fake_det_head = _head(256, 2, dropout, nn.LogSoftmax(dim=1))
        if clustering:
            centers = torch.zeros(cluster_number, 256)
            nn.init.xavier_uniform_(centers)
            self.cluster_assignment = _Param('cluster_centers', centers)

    def encode(self, x):
        feats = sci_cci_forward(x, self.sci.kernel, self.cci.kernel, self.R, self.H)
        context, (h, c) = self.encoder.lstm(feats.permute(1, 0, 2))
        return context, h, c, torch.cat([h[0], h[1]], dim=-1)

    def forward(self, x, fake_x=None, fake_perm_idx=None, positive_x=None):
        context, h, c, z = self.encode(x)
        dec_out, _ = self.decoder.lstm(F.relu(context), (h, c))          # clustering_interp.py:38-41
        v = self.rbf.compress_fc(dec_out.permute(1, 0, 2)).permute(0, 2, 1)   # (B,C,R)
        y = rbf_deinterp(v, x, self.rbf.kernel, self.R, self.H)
        aux = {}
        if 'future_vital' in self.aux_tasks:
            aux['future_vital'] = self.predict_future(z)
        rest = [t for t in self.aux_tasks if t != 'future_vital']
        if rest:
            pred = self.aux_head(z)
            for i, t in enumerate(rest):
                aux[t] = pred[:, i]
        if self.fake_detection:
            _, _, _, fz = self.encode(fake_x)
            aux['fake_det'] = self.fake_det_head(torch.cat([z, fz], dim=0)[fake_perm_idx])
        if self.clustering:
            q = dec_soft_assign(z, self.cluster_assignment.cluster_centers, 1.0)
            aux['cluster_pred'] = q
            aux['cluster_label'] = dec_target(q).detach()
        return z, y, aux


def joint_loss(net: OracleNet, x, ob, padding_mask, kl_weight=10.0, fake_x=None, fake_perm_idx=None,
               fake_label=None, fake_weight=1.0):
    """Loss of the north-star step: ae_mse + 10*kl (clustering_trainer.py:251-253,
    p3_clustering_main.py:85) and optionally + fake-detection NLL (:254-258)."""
    z, y, aux = net(x, fake_x, fake_perm_idx)
    terms = {'ae_mse': rec_loss(ob, y, padding_mask)}
    total = terms['ae_mse']
    if net.fake_detection and fake_label is not None:
        terms['fake_detection'] = F.nll_loss(aux['fake_det'], fake_label)
        total = total + fake_weight * terms['fake_detection']
    if net.clustering and kl_weight:
        terms['kl'] = kl_loss(aux['cluster_label'], aux['cluster_pred'])
        total = total + kl_weight * terms['kl']
    terms['loss'] = total
    return terms, z, y, aux


def make_optimizer(net, lr=3e-3, wd=4e-4):
    """utils.py:83: Adam(amsgrad=True) with L2 weight decay (p1:86,96)."""
    return torch.optim.Adam(net.parameters(), lr=lr, weight_decay=wd, amsgrad=True)


def train_step(net, opt, x, ob, padding_mask, kl_weight=10.0, grad_clip=15.0, **kw):
    """clustering_trainer.py:222-279: zero_grad, fwd, loss, bwd, clip 15, Adam."""
    opt.zero_grad()
    terms, z, _, _ = joint_loss(net, x, ob, padding_mask, kl_weight, **kw)
    terms['loss'].backward()
    gnorm = torch.nn.utils.clip_grad_norm_(net.parameters(), grad_clip)
    opt.step()
    return {k: float(v.detach()) for k, v in terms.items()}, float(gnorm), z.detach()

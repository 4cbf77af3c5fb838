"""Drop-in for the reference's ``rbf`` module (rbf.py:15-202): Gaussian RBF de-interpolation head.

``CompressFC`` (two small GEMMs + BatchNorm, rbf.py:111-125) stays on PyTorch-ROCm; the
(B,C,T,R)-shaped basis evaluation, normalisation and contraction run in ``csrc/dic_rbf.hip`` (k2).
"""
import torch
import torch.nn as nn

from . import ops
from .utils import TimeDistributed


class CompressFC(nn.Module):
    """Applied per (b, r) row through TimeDistributed: 256 -> 128 -> BN -> ReLU -> Dropout -> C."""

    def __init__(self, idim, odim, dropout):
        super().__init__()
        self.model = nn.Sequential(nn.Linear(idim, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Dropout(dropout),
                                   nn.Linear(128, odim))

    def forward(self, rec_input):
        last = self.model[4]
        fast = (rec_input.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16
                and last.in_features == ops.HEAD_IN and last.out_features in ops.HEAD_OUT and rec_input.dim() == 2)
        if not fast:
            if (rec_input.is_cuda and rec_input.dim() == 2 and rec_input.dtype == torch.float32 and not torch.is_autocast_enabled()
                    and ops.f32_products() == 'x3' and rec_input.shape[1] % 4 == 0):
                # f32 step, products as three-term bf16 splits on the matrix cores (csrc/dic_gemm.hip): the 256 -> 128 layer without a library GEMM
                first, bn = self.model[0], self.model[1]
                z = ops.mfma_linear(rec_input, first.weight, first.bias, bias_grad_is_zero=bn.training)
                if last.in_features == ops.HEAD_IN and last.out_features in ops.BNHEAD_OUT:
                    # BatchNorm -> ReLU -> Dropout -> Linear(128, C) as the streaming kernels of csrc/dic_bnhead.hip on f32 rows
                    return ops.bn_relu_head(z, bn, last, dropout=self.model[3])
                return self.model[4](self.model[3](self.model[2](bn(z))))
            return self.model(rec_input)
        # bf16 step: the C-wide output layer over all B*R rows is a 100 MB stream, not a GEMM (csrc/dic_head.hip)
        first = self.model[0]
        bn, drop = self.model[1], self.model[3]
        if ops.compress_fc_fused_ok(rec_input, first, bn, last):
            # the large-batch training step: the whole module as one autograd node (the 128-wide gradient is never materialised either)
            with torch.autocast('cuda', enabled=False):
                return ops.compress_fc(rec_input, first, bn, drop, last)
        with torch.autocast('cuda', enabled=False):
            # split-K weight gradient; the bias sits in front of BatchNorm: its gradient is identically 0 in training mode
            # (training mode: the layer's kernel also delivers the column sums the BatchNorm behind it needs)
            z, col_sums = ops.rows_linear(rec_input, first.weight, first.bias, bias_grad_is_zero=bn.training, with_stats=True)
            if not (bn.training and last.out_features in ops.BNHEAD_OUT):
                col_sums = None
        if last.out_features in ops.BNHEAD_OUT:
            # BatchNorm -> ReLU -> Dropout -> Linear in four streaming passes over z, the hidden activation never materialised
            with torch.autocast('cuda', enabled=False):
                return ops.bn_relu_head(z, bn, last, dropout=drop, col_sums=col_sums)
        hidden = drop(self.model[2](bn(z)))
        with torch.autocast('cuda', enabled=False):
            return ops.head_linear(hidden, last.weight, last.bias)


def gaussian(beta, alpha):
    """phi = exp(-beta * alpha^2) (rbf.py:129-131).  Kept for API parity (torch, tiny inputs); the
    model path evaluates this basis inside the HIP kernel."""
    return torch.exp(-beta * alpha.pow(2))


def basis_func_dict():
    """Only 'gaussian' is ever selected upstream (clustering_interp.py:116); the other upstream
    entries have a signature their caller cannot use and are not provided."""
    return {'gaussian': gaussian}


class RBF(nn.Module):
    def __init__(self, hours_look_ahead, ref_points, in_dim, out_dim, dropout, basis_func, device):
        super().__init__()
        if basis_func is not gaussian:
            raise NotImplementedError("only the 'gaussian' basis is on the reference path")
        self.ref_points, self.hours_look_ahead, self.device = ref_points, hours_look_ahead, device
        self.interp_t = torch.linspace(0, hours_look_ahead, ref_points, device=device)
        self.out_dim = self.num_variables = out_dim
        self.basis_func = basis_func
        self.compress_fc = TimeDistributed(CompressFC(in_dim, out_dim, dropout))
        self.kernel = nn.Parameter(torch.rand(out_dim, device=device), requires_grad=True)

    def forward(self, interp_data, raw_input, lengths=None, prefix_only=False, rec_target=None):
        """interp_data (B,256,R), raw_input (B,4C,T) -> (B,C,T) (rbf.py:57-108).
        ``rec_target`` (B,C,T) observations (training step with prefix ``lengths`` only): returns (reconstruction, masked MSE against
        it) from the fused kernels of ops.rbf_rec_loss instead."""
        native = interp_data.permute(2, 0, 1)                                     # (R,B,256)
        if native.is_contiguous():
            # the decoder emits (R,B,256); CompressFC is row-wise and its BatchNorm moments are order-invariant, so run
            # it on the rows as they lie in memory and permute the small (R,B,C) result instead of copying 256-wide rows
            v = self.compress_fc(native).permute(1, 2, 0)                         # (B,C,R)
        else:
            v = self.compress_fc(interp_data.permute(0, 2, 1)).permute(0, 2, 1)   # (B,C,R)
        if self.interp_t.device != v.device:
            self.interp_t = self.interp_t.to(v.device)
        if rec_target is not None:
            if lengths is None:
                raise ValueError('rec_target needs prefix lengths')
            return ops.rbf_rec_loss(v, raw_input, self.kernel, self.interp_t, lengths, rec_target)
        # training step with prefix lengths: the padded slots of the reconstruction are never read (rec_loss goes by the same lengths)
        return ops.rbf_deinterp(v, raw_input, self.kernel, self.interp_t, lengths, prefix_only=bool(prefix_only) and lengths is not None)

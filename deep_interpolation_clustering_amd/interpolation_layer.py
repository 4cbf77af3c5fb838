"""Drop-in for the reference's ``interpolation_layer`` module (interpolation_layer.py:12-127) on the
HIP kernels: same class names, constructor signatures, parameter names (``kernel``) and tensor
shapes.  The arithmetic lives in ``csrc/dic_interp.hip`` (k1 / k1').

``Net`` does not chain the two modules: it calls :func:`fused_forward`, which runs SCI with CCI as an
epilogue in one launch.  The stand-alone ``forward`` methods exist for API parity and use the
stand-alone kernels.
"""
import torch
import torch.nn as nn

from . import ops


class SingleChannelInterp(nn.Module):
    """interpolation_layer.py:12-86.  ``kernel`` (C,) ~ U(0,1) is the raw per-channel bandwidth; the
    kernels apply log(1+e^k).  ``activation`` is accepted and ignored, as upstream."""

    def __init__(self, ref_points, hours_look_ahead, d_dim, timestamp, device, activation="sigmoid"):
        super().__init__()
        self.ref_points, self.hours_look_ahead = ref_points, hours_look_ahead
        self.d_dim, self.timestamp = d_dim, timestamp
        self.device, self.activation = device, activation
        self.kernel = nn.Parameter(torch.rand(d_dim, device=device), requires_grad=True)
        self._grid = None

    def grid(self):
        dev = self.kernel.device
        if self._grid is None or self._grid.device != dev:
            self._grid = ops.ref_grid(self.hours_look_ahead, self.ref_points, dev)
        return self._grid

    def forward(self, x, lengths=None):
        """x (B,4C,T) planes [value, mask, time, hold-out] -> (B,R,3C) = [y | w | y_trans].
        ``lengths`` (B,C) int32, optional: rows are prefixes of that many valid slots (skips the mask plane)."""
        return ops.sci_only(x, self.kernel, self.grid(), lengths)


class CrossChannelInterp(nn.Module):
    """interpolation_layer.py:89-127.  ``kernel`` (C,C) starts as the identity."""

    def __init__(self, d_dim, timestamp, device, activation="sigmoid"):
        super().__init__()
        self.d_dim, self.timestamp = d_dim, timestamp
        self.device, self.activation = device, activation
        self.kernel = nn.Parameter(torch.eye(d_dim, d_dim, device=device), requires_grad=True)

    def forward(self, x, reconstruction=False):
        """(B,R,3C) -> (B,R,3C) = [smooth | intensity | transient]; ``reconstruction`` is unused upstream too."""
        self.output_dim = x.size(1)
        return ops.cci(x, self.kernel)


def fused_forward(sci: SingleChannelInterp, cci: CrossChannelInterp, x, lengths=None):
    """cci(sci(x)) in ONE kernel launch: the (B,R,3C) SCI tensor stays in LDS."""
    return ops.sci_cci(x, sci.kernel, cci.kernel, sci.grid(), lengths)

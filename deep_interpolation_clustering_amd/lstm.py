"""Fused bidirectional LSTM (hidden 128, one layer) for the encoder / decoder of ``Net``
(clustering_interp.py:14-41) on the persistent HIP recurrence kernels of ``csrc/dic_lstm.hip``.

``torch.nn.LSTM`` semantics and parameters (``weight_ih_l0[_reverse]``, ``weight_hh_l0[_reverse]``,
``bias_ih_l0[_reverse]``, ``bias_hh_l0[_reverse]``; gate order i,f,g,o) are kept -- the module still owns an
``nn.LSTM`` so ``state_dict`` keys do not change.  What changes is who executes it when the step runs
under bf16 autocast on the GPU:

  time-parallel GEMMs (hipBLASLt, bf16 in / f32 accumulate), issued here:
      gx = X.W_ih^T + b_ih + b_hh            (R*B x I) . (I x 8H)     [I >= 32: decoder; the encoder's 18-wide
                                                                       projection runs inside the recurrence kernel]
      dX = dG.W_ih,  dW_ih = dG^T.X,  dW_hh = dG^T.H_prev,  db = sum dG
  sequential recurrence (dic_lstm_fwd / dic_lstm_bwd): one workgroup per 64 batch rows and direction keeps
  h, c (dh, dc) on chip for all R steps with W_hh resident in registers.

In f32 (no autocast) the stock ``nn.LSTM`` (MIOpen) is used: that is the configuration the 1e-5 parity
tests run in.  The bf16 path is checked against an f32 emulation with the same rounding points.
"""
import torch

from . import _native as N
from .ops import splitk_tn

H = 128
PROJ_WIDTH = 32      # dic_lstm_fwd_proj's compiled input width


def fused_available(x, lstm):
    return (x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16
            and lstm.hidden_size == H and lstm.num_layers == 1 and lstm.bidirectional and lstm.bias
            and not lstm.batch_first and lstm.proj_size == 0)


class _BiLstm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, bias, h0, c0):
        R, B, I = x.shape
        bf = torch.bfloat16
        proj = I < PROJ_WIDTH                                      # narrow input (encoder): projection inside the recurrence
        Ip = PROJ_WIDTH if proj else (I + 15) // 16 * 16           # K of the projection padded to the MFMA step
        xb = x.to(bf)
        wihb = w_ih.reshape(8 * H, I).to(bf)
        if Ip != I:
            xb = torch.nn.functional.pad(xb, (0, Ip - I))
            wihb = torch.nn.functional.pad(wihb, (0, Ip - I))
        xb = xb.contiguous()
        if proj:
            # the bias rides along as a constant-one input column (a spare one exists because I < 32); gx is never formed
            xb[..., I].fill_(1.0)
            wihb[:, I] = bias.reshape(8 * H).to(bf)
            gx = None
        else:
            gx = torch.addmm(bias.reshape(8 * H).to(bf), xb.view(R * B, Ip), wihb.t())      # (R*B, 2*4*H)
        whhb = w_hh.to(bf).contiguous()                            # (2,4H,H)
        need = any(ctx.needs_input_grad)
        dev = x.device
        out = torch.empty((R, B, 2 * H), device=dev, dtype=bf)
        hn = torch.empty((2, B, H), device=dev, dtype=torch.float32)
        cn = torch.empty_like(hn)
        Bp = (B + 63) // 64 * 64                                   # kernel-native saved state is tiled by 64 rows
        gates = torch.empty((R, Bp, 2, 4, H), device=dev, dtype=bf) if need else None
        cs = torch.empty((R, Bp, 2, H), device=dev, dtype=bf) if need else None    # bf16 copy for the backward; c itself stays f32 on chip
        h0c = None if h0 is None else h0.float().contiguous()
        c0c = None if c0 is None else c0.float().contiguous()
        if proj:
            N.check(N.lib().dic_lstm_fwd_proj(N.ptr(xb), N.ptr(wihb), N.ptr(whhb), N.ptr(h0c), N.ptr(c0c), R, B, H, Ip, N.ptr(out),
                                              N.ptr(hn), N.ptr(cn), N.ptr(gates), N.ptr(cs), N.stream_of(x)), 'dic_lstm_fwd_proj')
        else:
            N.check(N.lib().dic_lstm_fwd(N.ptr(gx), N.ptr(whhb), N.ptr(h0c), N.ptr(c0c), R, B, H, N.ptr(out), N.ptr(hn),
                                         N.ptr(cn), N.ptr(gates), N.ptr(cs), N.stream_of(x)), 'dic_lstm_fwd')
        ctx.dims = (R, B, I, Ip)
        ctx.x_dtype = x.dtype
        ctx.has_init = h0 is not None
        ctx.save_for_backward(xb, wihb, whhb, gates, cs, out, h0c, c0c)
        return out, hn, cn

    @staticmethod
    def backward(ctx, dout, dhn, dcn):
        xb, wihb, whhb, gates, cs, out, h0c, c0c = ctx.saved_tensors
        R, B, I, Ip = ctx.dims
        bf = torch.bfloat16
        dev = out.device
        dgx = torch.empty((R, B, 2, 4, H), device=dev, dtype=bf)
        dh0 = torch.empty((2, B, H), device=dev, dtype=torch.float32)
        dc0 = torch.empty_like(dh0)
        whh_t = whhb.transpose(1, 2).contiguous()                  # (2,H,4H)
        doutb = None if dout is None else dout.to(bf).contiguous()
        dhnc = None if dhn is None else dhn.float().contiguous()
        dcnc = None if dcn is None else dcn.float().contiguous()
        Lb = N.lib()
        dbias = torch.empty((2, 4 * H), device=dev, dtype=torch.float32)         # summed inside the kernel, f32
        ws = torch.empty(max(16, Lb.dic_lstm_bwd_workspace(B)), device=dev, dtype=torch.uint8)
        N.check(Lb.dic_lstm_bwd(N.ptr(whh_t), N.ptr(gates), N.ptr(cs), N.ptr(c0c), N.ptr(doutb), N.ptr(dhnc), N.ptr(dcnc),
                                R, B, H, N.ptr(dgx), N.ptr(dh0), N.ptr(dc0), N.ptr(dbias), N.ptr(ws), ws.numel(),
                                N.stream_of(out)), 'dic_lstm_bwd')
        dg2 = dgx.view(R * B, 8 * H)
        dx = dw_ih = dw_hh = None
        if ctx.needs_input_grad[0]:
            dx = (dg2 @ wihb)[:, :I].reshape(R, B, I).to(ctx.x_dtype)
        if ctx.needs_input_grad[1]:
            # dW = dG^T.X has K = R*B (hundreds of thousands) and a tiny output: split-K bmm (ops.splitk_tn)
            dw_ih = splitk_tn(dg2, xb.view(R * B, Ip))[:, :I].reshape(2, 4 * H, I)
        if ctx.needs_input_grad[2]:
            # dW_hh[d] = sum_t dG_t[d]^T h_prev_t[d], h_prev = h_{t-1} (forward) / h_{t+1} (reverse): shifted strided views
            # of dG and out, one direction at a time -- no H_prev copy and none of the cross-direction blocks a single
            # (8H, 2H) product would compute (0.42 ms against 0.82 at R*B = 786k, scripts/gemm_probe2.py / gemm_probe3.py)
            o2 = out.view(R * B, 2 * H)
            if R > 1:
                fwd = splitk_tn(dg2[B:, :4 * H], o2[:-B, :H], chunks=(8192, 4096, 2048))
                rev = splitk_tn(dg2[:-B, 4 * H:], o2[B:, H:], chunks=(8192, 4096, 2048))
            else:
                fwd = rev = torch.zeros((4 * H, H), device=dev, dtype=torch.float32)
            if h0c is not None:
                fwd = fwd + (dg2[:B, :4 * H].t() @ h0c[0].to(bf)).float()
                rev = rev + (dg2[-B:, 4 * H:].t() @ h0c[1].to(bf)).float()
            dw_hh = torch.stack([fwd, rev])
        return dx, dw_ih, dw_hh, (dbias if ctx.needs_input_grad[3] else None), (dh0 if ctx.has_init else None), (dc0 if ctx.has_init else None)


def bilstm(x, lstm, h0=None, c0=None):
    """(out (R,B,2H) bf16, (h_n, c_n) (2,B,H) f32) = bidirectional LSTM of x (R,B,I) with ``lstm``'s parameters."""
    w_ih = torch.stack([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse])
    w_hh = torch.stack([lstm.weight_hh_l0, lstm.weight_hh_l0_reverse])
    bias = torch.stack([lstm.bias_ih_l0 + lstm.bias_hh_l0, lstm.bias_ih_l0_reverse + lstm.bias_hh_l0_reverse])
    with torch.autocast('cuda', enabled=False):
        out, hn, cn = _BiLstm.apply(x, w_ih.float(), w_hh.float(), bias.float(), h0, c0)
    return out, (hn, cn)

"""Differentiable operators of the hot path, each a thin ``torch.autograd.Function`` over one or
two launches of the HIP library (``_native``).  PyTorch supplies device memory, the current HIP
stream and autograd plumbing only; every value is produced by the gfx950 kernels.

Operator <-> reference map (upstream file:line):
  sci_cci / sci_only    SingleChannelInterp.forward  interpolation_layer.py:31-86
                        + CrossChannelInterp.forward interpolation_layer.py:99-127 (fused epilogue)
  cci                   CrossChannelInterp.forward on an arbitrary (B,R,3C) tensor
  rbf_deinterp          RBF.forward minus compress_fc rbf.py:57-108
  masked_mse            Net.rec_loss                  clustering_interp.py:197-203
  dec_soft_assign       ClusterAssignment.forward     dec.py:49-63
  dec_target            target_distribution           dec.py:66-76
  kl_batchmean          Net.kl_loss                   clustering_interp.py:205-207

Sharded (one process per GPU) execution: the batch-level statistics -- mask count and SSE of the
reconstruction loss, the DEC column sums f_j, the KL batch divisor -- are all-reduced over RCCL
(``dist.all_reduce_sum_``) so a sharded step equals the single-device step on the global batch.
"""
from __future__ import annotations

import os

import torch

from . import _native as N
from . import dist
from .ragged import is_ragged


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _lengths_arg(lengths, B, C, device):
    if lengths is None:
        return None
    if lengths.dtype != torch.int32 or not lengths.is_contiguous() or lengths.device != device:
        lengths = lengths.to(device=device, dtype=torch.int32).contiguous()
    if lengths.numel() != B * C:
        raise ValueError(f'lengths must have B*C={B * C} entries, got {tuple(lengths.shape)}')
    return lengths


def _store_ptrs(rb, hold=True):
    """(t_pk, v_pk[, hold_pk], row_off, enc_idx, lengths) pointers of a ragged.RaggedBatch for the dic_*_store entry points."""
    st = rb.store
    head = (N.ptr(st.t_pk), N.ptr(st.v_pk)) + ((N.ptr(st.hold_pk) if rb.denoise else None,) if hold else ())
    return head + (N.ptr(st.row_off), N.ptr(rb.idx), N.ptr(rb.lengths))


def ref_grid(hours, ref_points, device):
    """interpolation_layer.py:41 / rbf.py:44 -- the grid is data for the kernels."""
    return torch.linspace(0, hours, ref_points, device=device, dtype=torch.float32)


# ----------------------------------------------------------------------------------------- k1
# ------------------------------------------------------------------------------------------ small parameter gradients
# Inside step.Stepper's backward (grad_sink_session) the small parameter gradients these Functions produce -- bandwidths, the cross-channel
# matrix, centroids, CompressFC's six -- are not returned to autograd (one AccumulateGrad add launch each, ~5 us a piece) but queued and
# added into the parameters' slots of the flat gradient bucket by ONE dic_accumulate_many launch when the session ends (or when the
# sharded step starts the early all-reduce of the decoder-side gradients).
_SINK = {'on': False, 'pairs': []}
GRAD_SINKS = os.environ.get('DIC_GRAD_SINKS', '1') != '0'


def _sink(param, grad):
    """Inside a session: queue ``grad`` for ``param.grad`` and tell autograd nothing (None); else hand ``grad`` back unchanged."""
    if grad is None or not (_SINK['on'] and GRAD_SINKS):
        return grad
    g = getattr(param, 'grad', None)
    if (g is None or not param.is_leaf or g.dtype != torch.float32 or grad.dtype != torch.float32 or not g.is_contiguous()
            or g.device != grad.device or g.numel() != grad.numel() or not grad.is_cuda):
        return grad
    _SINK['pairs'].append((grad if grad.is_contiguous() else grad.contiguous(), g))
    param._dic_grad_written = True          # dist.FlatParams.active_mask: autograd's accumulate hook will not fire for this one
    return None


def sink_session_active():
    """True inside step.Stepper's backward (grad_sink_session): kernels may then add parameter gradients straight into ``.grad``."""
    return _SINK['on']


def flush_grad_sinks():
    pairs, _SINK['pairs'] = _SINK['pairs'], []
    if not pairs:
        return
    import ctypes as C
    # a parameter used by several nodes (the interpolation bandwidths: real, corrupted and positive branch) is queued once per node:
    # one launch per occurrence rank, so that no two workgroups of a launch add into the same destination (and the order is fixed)
    rounds, seen = [], {}
    for src, dst in pairs:
        k = seen.get(dst.data_ptr(), 0)
        seen[dst.data_ptr()] = k + 1
        if k == len(rounds):
            rounds.append([])
        rounds[k].append((src, dst))
    for rnd in rounds:
        n = (C.c_int * len(rnd))(*[p[0].numel() for p in rnd])
        src, dst = N.ptr_array([p[0] for p in rnd]), N.ptr_array([p[1] for p in rnd])
        N.check(N.lib().dic_accumulate_many(src, dst, n, len(rnd), N.stream_of(rnd[0][1])), 'dic_accumulate_many')


class grad_sink_session:
    """``with grad_sink_session(): loss.backward()`` -- see _sink."""

    def __enter__(self):
        _SINK['on'], _SINK['pairs'] = True, []
        return self

    def __exit__(self, *exc):
        _SINK['on'] = False
        if exc[0] is None:
            flush_grad_sinks()
        else:
            _SINK['pairs'] = []
        return False


class _SciCci(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sci_kernel, cci_kernel, grid, lengths):
        N.require_gpu(x, sci_kernel, grid)
        store = is_ragged(x)                      # a ragged.RaggedBatch: the rows are read in place from the encounter store
        if not store:
            x = N.f32c(x)
        B, C4, T = x.shape
        C = sci_kernel.numel()
        if C4 != 4 * C:
            raise ValueError(f'stacked input must be (B, 4*{C}, T), got {tuple(x.shape)}')
        R = grid.numel()
        lengths = None if store else _lengths_arg(lengths, B, C, x.device)
        sk = N.f32c(sci_kernel.detach())
        ck = None if cci_kernel is None else N.f32c(cci_kernel.detach())
        need_grad = any(ctx.needs_input_grad)
        out = torch.empty((B, R, 3 * C), device=x.device, dtype=torch.float32)
        saved = torch.empty((B, 7, C, R), device=x.device, dtype=torch.float32) if need_grad else None
        if store:
            N.check(N.lib().dic_sci_cci_fwd_store(*_store_ptrs(x), B, C, T, R, N.ptr(grid), N.ptr(sk), N.ptr(ck), N.ptr(out), N.ptr(saved),
                                                  None, 0, int(x.store.times_sorted), N.stream_of(out)), 'dic_sci_cci_fwd_store')
        else:
            N.check(N.lib().dic_sci_cci_fwd(N.ptr(x), N.ptr(lengths), B, C, T, R, N.ptr(grid), N.ptr(sk), N.ptr(ck),
                                            N.ptr(out), N.ptr(saved), N.stream_of(x)), 'dic_sci_cci_fwd')
        ctx.dims = (B, C, R)
        ctx.has_cci = ck is not None
        ctx.save_for_backward(saved, sk, ck)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        saved, sk, ck = ctx.saved_tensors
        B, C, R = ctx.dims
        g = N.f32c(grad_out)
        gs = torch.empty(C, device=g.device, dtype=torch.float32)
        gc = torch.empty((C, C), device=g.device, dtype=torch.float32) if ctx.has_cci else None
        L = N.lib()
        ws = _ws(L.dic_sci_cci_bwd_workspace(B, C, R), g.device)
        N.check(L.dic_sci_cci_bwd(N.ptr(g), N.ptr(saved), N.ptr(sk), N.ptr(ck), B, C, R, N.ptr(gs), N.ptr(gc), N.ptr(ws),
                                  ws.numel(), N.stream_of(g)), 'dic_sci_cci_bwd')
        return None, gs, gc, None, None


def sci_cci(x, sci_kernel, cci_kernel, grid, lengths=None):
    """Fused SCI + CCI: x (B,4C,T) -> (B,R,3C) = [smooth | intensity | transient]."""
    return _SciCci.apply(x, sci_kernel, cci_kernel, grid, lengths)


def sci_only(x, sci_kernel, grid, lengths=None):
    """SCI alone: x (B,4C,T) -> (B,R,3C) = [y | w | y_trans]."""
    return _SciCci.apply(x, sci_kernel, None, grid, lengths)


PACKED_WIDTH = 32      # width of the encoder LSTM's packed input rows for the reference's six vitals (dic_lstm_fwd_proj / dic_lstm_dw: one MFMA k-step pair)


def packed_width(features):
    """Width of the packed bf16 rows [features | 1 | 0...] for an LSTM of ``features`` inputs: 32 (3C = 18: the reference's six vitals) or 64
    (3C = 36: BASELINE configs[3]'s twelve channels; clustering_interp.py:102-111 sizes the encoder as 3 * num_variables); 0 = not packed."""
    return 32 if features < 32 else (64 if features < 64 else 0)


class _SciCciPacked(torch.autograd.Function):
    """cci(sci(x)) written straight in the encoder LSTM's input layout: (R,B,32) bf16 rows [3C features | 1 | 0...] (the constant
    one carries the LSTM bias).  Forward and backward exchange that layout with lstm.bilstm_packed -- no permute / cast / pad
    passes between the interpolation kernel and the recurrence kernel."""

    @staticmethod
    def forward(ctx, x, sci_kernel, cci_kernel, grid, lengths):
        N.require_gpu(x, sci_kernel, cci_kernel, grid)
        store = is_ragged(x)
        if not store:
            x = N.f32c(x)
        B, C4, T = x.shape
        C = sci_kernel.numel()
        if C4 != 4 * C:
            raise ValueError(f'stacked input must be (B, 4*{C}, T), got {tuple(x.shape)}')
        xw = packed_width(3 * C)
        if not xw:
            raise ValueError(f'packed rows hold at most 63 features, got 3C = {3 * C}')
        R = grid.numel()
        lengths = None if store else _lengths_arg(lengths, B, C, x.device)
        sk, ck = N.f32c(sci_kernel.detach()), N.f32c(cci_kernel.detach())
        need_grad = any(ctx.needs_input_grad)
        xenc = torch.empty((R, B, xw), device=x.device, dtype=torch.bfloat16)
        saved = torch.empty((B, 7, C, R), device=x.device, dtype=torch.float32) if need_grad else None
        if store:
            N.check(N.lib().dic_sci_cci_fwd_store(*_store_ptrs(x), B, C, T, R, N.ptr(grid), N.ptr(sk), N.ptr(ck), None, N.ptr(saved),
                                                  N.ptr(xenc), xw, int(x.store.times_sorted), N.stream_of(xenc)), 'dic_sci_cci_fwd_store')
        else:
            N.check(N.lib().dic_sci_cci_fwd_packed(N.ptr(x), N.ptr(lengths), B, C, T, R, N.ptr(grid), N.ptr(sk), N.ptr(ck), None,
                                                   N.ptr(saved), N.ptr(xenc), xw, N.stream_of(x)), 'dic_sci_cci_fwd_packed')
        ctx.dims = (B, C, R)
        ctx.xw = xw
        ctx.sink_params = (sci_kernel, cci_kernel)
        ctx.save_for_backward(saved, sk, ck)
        return xenc

    @staticmethod
    def backward(ctx, grad):
        saved, sk, ck = ctx.saved_tensors
        B, C, R = ctx.dims
        g = grad if grad.dtype == torch.bfloat16 else grad.to(torch.bfloat16)
        g = g if g.is_contiguous() else g.contiguous()
        gs = torch.empty(C, device=g.device, dtype=torch.float32)
        gc = torch.empty((C, C), device=g.device, dtype=torch.float32)
        L = N.lib()
        ws = _ws(L.dic_sci_cci_bwd_workspace(B, C, R), g.device)
        N.check(L.dic_sci_cci_bwd_packed(N.ptr(g), ctx.xw, N.ptr(saved), N.ptr(sk), N.ptr(ck), B, C, R, N.ptr(gs), N.ptr(gc),
                                         N.ptr(ws), ws.numel(), N.stream_of(g)), 'dic_sci_cci_bwd_packed')
        return None, _sink(ctx.sink_params[0], gs), _sink(ctx.sink_params[1], gc), None, None


def sci_cci_packed(x, sci_kernel, cci_kernel, grid, lengths=None):
    """Fused SCI + CCI: x (B,4C,T) -> (R,B,32 or 64) bf16 = [smooth | intensity | transient | 1 | 0...], time-major (3C < 64)."""
    return _SciCciPacked.apply(x, sci_kernel, cci_kernel, grid, lengths)


class _Cci(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, cci_kernel):
        N.require_gpu(s, cci_kernel)
        s = N.f32c(s)
        B, R, C3 = s.shape
        C = cci_kernel.shape[0]
        if C3 != 3 * C:
            raise ValueError(f'input must be (B,R,3*{C}), got {tuple(s.shape)}')
        ck = N.f32c(cci_kernel.detach())
        out = torch.empty_like(s)
        N.check(N.lib().dic_cci_fwd(N.ptr(s), N.ptr(ck), B, C, R, N.ptr(out), N.stream_of(s)), 'dic_cci_fwd')
        ctx.dims = (B, C, R)
        ctx.save_for_backward(s, ck)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        s, ck = ctx.saved_tensors
        B, C, R = ctx.dims
        g = N.f32c(grad_out)
        gs = torch.empty_like(s)
        gk = torch.empty((C, C), device=g.device, dtype=torch.float32)
        L = N.lib()
        ws = _ws(L.dic_cci_bwd_workspace(B, C, R), g.device)
        N.check(L.dic_cci_bwd(N.ptr(g), N.ptr(s), N.ptr(ck), B, C, R, N.ptr(gs), N.ptr(gk), N.ptr(ws), ws.numel(),
                              N.stream_of(g)), 'dic_cci_bwd')
        return gs, gk


def cci(s, cci_kernel):
    return _Cci.apply(s, cci_kernel)


# ----------------------------------------------------------------------------------------- k2
class _Rbf(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, raw_input, rbf_kernel, grid, lengths, prefix_only=False):
        N.require_gpu(v, raw_input, rbf_kernel, grid)
        store = is_ragged(raw_input)              # a ragged.RaggedBatch: time stamps read in place from the encounter store
        x = raw_input if store else N.f32c(raw_input)
        B, C4, T = x.shape
        C = rbf_kernel.numel()
        R = grid.numel()
        if C4 != 4 * C or tuple(v.shape) != (B, C, R):
            raise ValueError(f'rbf: raw_input {tuple(x.shape)} / v {tuple(v.shape)} do not match C={C}, R={R}')
        # v usually arrives as a permuted VIEW of the (R,B,C) rows the FC head wrote: the kernels read that layout as it lies
        tm = v.dtype == torch.float32 and not v.is_contiguous() and v.permute(2, 0, 1).is_contiguous()
        vb = v.permute(2, 0, 1) if tm else N.f32c(v)
        lengths = x.lengths if store else _lengths_arg(lengths, B, C, x.device)
        rk = N.f32c(rbf_kernel.detach())
        need_grad = any(ctx.needs_input_grad)
        y = torch.empty((B, C, T), device=x.device, dtype=torch.float32)
        norm = torch.empty_like(y) if need_grad else None
        if store:
            N.check(N.lib().dic_rbf_fwd_store(*_store_ptrs(x, hold=False), B, C, T, R, N.ptr(grid), N.ptr(rk), N.ptr(vb), int(tm), 0, N.ptr(y),
                                              N.ptr(norm), int(bool(prefix_only)), None, None, 0, N.stream_of(y)), 'dic_rbf_fwd_store')
        else:
            N.check(N.lib().dic_rbf_fwd(N.ptr(x), N.ptr(lengths), B, C, T, R, N.ptr(grid), N.ptr(rk), N.ptr(vb), int(tm), N.ptr(y),
                                        N.ptr(norm), int(bool(prefix_only)), N.stream_of(x)), 'dic_rbf_fwd')
        ctx.dims = (B, C, T, R, bool(tm))
        ctx.rb = x if store else None
        ctx.save_for_backward(None if store else x, lengths, grid, rk, vb, y, norm)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, lengths, grid, rk, vb, y, norm = ctx.saved_tensors
        B, C, T, R, tm = ctx.dims
        g = N.f32c(grad_y)
        gv = torch.empty((R, B, C) if tm else (B, C, R), device=g.device, dtype=torch.float32)
        gk = torch.empty(C, device=g.device, dtype=torch.float32)
        L = N.lib()
        ws = _ws(L.dic_rbf_bwd_workspace(B, C, T, R), g.device)
        if ctx.rb is not None:
            N.check(L.dic_rbf_bwd_store(*_store_ptrs(ctx.rb, hold=False), B, C, T, R, N.ptr(grid), N.ptr(rk), N.ptr(vb), int(tm), N.ptr(y),
                                        N.ptr(norm), N.ptr(g), None, None, N.ptr(gv), N.ptr(gk), N.ptr(ws), ws.numel(), N.stream_of(g)),
                    'dic_rbf_bwd_store')
        else:
            N.check(L.dic_rbf_bwd(N.ptr(x), N.ptr(lengths), B, C, T, R, N.ptr(grid), N.ptr(rk), N.ptr(vb), int(tm), N.ptr(y), N.ptr(norm),
                                  N.ptr(g), N.ptr(gv), N.ptr(gk), N.ptr(ws), ws.numel(), N.stream_of(g)), 'dic_rbf_bwd')
        return (gv.permute(1, 2, 0) if tm else gv), None, gk, None, None, None


def rbf_deinterp(v, raw_input, rbf_kernel, grid, lengths=None, prefix_only=False):
    """v (B,C,R) grid values -> (B,C,T) values at the observed time stamps of ``raw_input``.
    ``prefix_only`` (needs ``lengths``): only the first n slots of each row are written -- the padding, which upstream zeroes, is
    left uninitialised.  For callers that consume the result through ``masked_mse(..., lengths)`` alone (the training step)."""
    return _Rbf.apply(v if v.dtype == torch.float32 else v.float(), raw_input, rbf_kernel, grid, lengths, prefix_only)   # .float(): bf16 autocast producers


class _RbfRecLoss(torch.autograd.Function):
    """k2 and Net.rec_loss in one node (training step with prefix lengths): forward = dic_rbf_fwd_loss (the masked SSE comes out of the
    registers that hold y), backward = dic_rbf_bwd_loss (dL/dy formed on the fly from ob).  Returns (y, mse); y is handed out for
    callers that look at it, but a gradient arriving through y is not supported here (the step consumes it through mse alone)."""

    @staticmethod
    def forward(ctx, v, raw_input, rbf_kernel, grid, lengths, ob):
        store = is_ragged(raw_input)              # a ragged.RaggedBatch: time stamps AND observations read in place from the store (ob is it too)
        if store:
            if not is_ragged(ob) or ob.store is not raw_input.store:
                raise ValueError('rbf_rec_loss: with a RaggedBatch input the observations are the batch itself')
            N.require_gpu(v, rbf_kernel, grid)
            x, obc = raw_input, None
        else:
            N.require_gpu(v, raw_input, rbf_kernel, grid, ob)
            x, obc = N.f32c(raw_input), N.f32c(ob)
        B, C4, T = x.shape
        C, R = rbf_kernel.numel(), grid.numel()
        if C4 != 4 * C or tuple(v.shape) != (B, C, R) or (obc is not None and tuple(obc.shape) != (B, C, T)):
            raise ValueError(f'rbf_rec_loss: raw_input {tuple(x.shape)} / v {tuple(v.shape)} / ob do not match C={C}, R={R}')
        tm = v.dtype == torch.float32 and not v.is_contiguous() and v.permute(2, 0, 1).is_contiguous()
        vb = v.permute(2, 0, 1) if tm else N.f32c(v)
        lengths = x.lengths if store else _lengths_arg(lengths, B, C, x.device)
        rk = N.f32c(rbf_kernel.detach())
        y = torch.empty((B, C, T), device=x.device, dtype=torch.float32)
        norm = torch.empty_like(y)
        out2 = torch.empty(2, device=x.device, dtype=torch.float32)
        L = N.lib()
        ws = _ws(L.dic_rbf_fwd_loss_workspace(B, C, T, R), x.device)
        if store:
            N.check(L.dic_rbf_fwd_store(*_store_ptrs(x, hold=False), B, C, T, R, N.ptr(grid), N.ptr(rk), N.ptr(vb), int(tm), 1, N.ptr(y),
                                        N.ptr(norm), 1, N.ptr(out2), N.ptr(ws), ws.numel(), N.stream_of(y)), 'dic_rbf_fwd_store')
        else:
            N.check(L.dic_rbf_fwd_loss(N.ptr(x), N.ptr(lengths), B, C, T, R, N.ptr(grid), N.ptr(rk), N.ptr(vb), int(tm), N.ptr(obc), N.ptr(y),
                                       N.ptr(norm), 1, N.ptr(out2), N.ptr(ws), ws.numel(), N.stream_of(x)), 'dic_rbf_fwd_loss')
        # global SSE and global #valid slots.  Sharded: the pair does not pay a latency-bound exchange of its own -- it rides on the next small
        # all-reduce of the step (the KL sum + row count, or the first global-mean loss term: dist.deferred_sum_) and the mean is formed when
        # it lands; step.compute_losses resolves whatever is still pending before it hands the losses out (one packed buffer)
        if dist.is_sharded():
            mse = torch.full((), float('nan'), device=x.device, dtype=torch.float32)      # (NaN until the pair has landed: an early read shows)
            mse_out = mse.detach()
            dist.deferred_sum_(out2, then=lambda: torch.div(out2[0], out2[1], out=mse_out))
        else:
            mse = torch.empty((), device=x.device, dtype=torch.float32)
            torch.div(out2[0], out2[1], out=mse)
        ctx.dims = (B, C, T, R, bool(tm))
        ctx.sink_params = (rbf_kernel,)
        ctx.rb = x if store else None
        ctx.out2 = out2                     # (not through save_for_backward: the deferred reduction writes it in place after this forward returns)
        ctx.save_for_backward(None if store else x, lengths, grid, rk, vb, y, norm, obc)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(y)
        return y, mse

    @staticmethod
    def backward(ctx, grad_y, grad_mse):
        x, lengths, grid, rk, vb, y, norm, obc = ctx.saved_tensors
        out2 = ctx.out2
        B, C, T, R, tm = ctx.dims
        if grad_mse is None:
            return None, None, None, None, None, None
        gl = N.f32c(grad_mse.reshape(1))
        dev = y.device
        gv = torch.empty((R, B, C) if tm else (B, C, R), device=dev, dtype=torch.float32)
        gk = torch.empty(C, device=dev, dtype=torch.float32)
        L = N.lib()
        ws = _ws(L.dic_rbf_bwd_workspace(B, C, T, R), dev)
        if ctx.rb is not None:
            N.check(L.dic_rbf_bwd_store(*_store_ptrs(ctx.rb, hold=False), B, C, T, R, N.ptr(grid), N.ptr(rk), N.ptr(vb), int(tm), N.ptr(y),
                                        N.ptr(norm), None, N.ptr(out2), N.ptr(gl), N.ptr(gv), N.ptr(gk), N.ptr(ws), ws.numel(), N.stream_of(y)),
                    'dic_rbf_bwd_store')
        else:
            N.check(L.dic_rbf_bwd_loss(N.ptr(x), N.ptr(lengths), B, C, T, R, N.ptr(grid), N.ptr(rk), N.ptr(vb), int(tm), N.ptr(y), N.ptr(norm),
                                       N.ptr(obc), N.ptr(out2), N.ptr(gl), N.ptr(gv), N.ptr(gk), N.ptr(ws), ws.numel(), N.stream_of(x)),
                    'dic_rbf_bwd_loss')
        return (gv.permute(1, 2, 0) if tm else gv), None, _sink(ctx.sink_params[0], gk), None, None, None


def rbf_rec_loss(v, raw_input, rbf_kernel, grid, lengths, ob):
    """(y, mse) = (rbf_deinterp(v, ..., prefix_only=True), masked_mse(ob, y, lengths=lengths)) from one forward and one backward
    kernel.  ``y`` (B,C,T) holds the observed prefix of each row only and carries no gradient (it is consumed through ``mse``)."""
    return _RbfRecLoss.apply(v if v.dtype == torch.float32 else v.float(), raw_input, rbf_kernel, grid, lengths, ob)


class _MaskedMse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, org_ob, rec_ob, mask, lengths, prefix_only=False):
        N.require_gpu(org_ob, rec_ob)
        ob, rec = N.f32c(org_ob), N.f32c(rec_ob)
        B, C, T = rec.shape
        mask = None if mask is None else N.f32c(mask)
        lengths = _lengths_arg(lengths, B, C, rec.device)
        out2 = torch.empty(2, device=rec.device, dtype=torch.float32)
        L = N.lib()
        ws = _ws(L.dic_masked_sse_workspace(B, C, T), rec.device)
        N.check(L.dic_masked_sse_fwd(N.ptr(ob), N.ptr(rec), N.ptr(mask), N.ptr(lengths), B, C, T, N.ptr(out2), N.ptr(ws),
                                     ws.numel(), N.stream_of(rec)), 'dic_masked_sse_fwd')
        dist.all_reduce_sum_(out2)          # global SSE and global #{mask == 1}
        ctx.dims = (B, C, T)
        ctx.prefix_only = bool(prefix_only) and lengths is not None and mask is None
        ctx.save_for_backward(ob, rec, mask, lengths, out2)
        return out2[0] / out2[1]

    @staticmethod
    def backward(ctx, grad_loss):
        ob, rec, mask, lengths, out2 = ctx.saved_tensors
        B, C, T = ctx.dims
        gl = N.f32c(grad_loss.reshape(1))
        grad_rec = torch.empty_like(rec)
        N.check(N.lib().dic_masked_sse_bwd(N.ptr(ob), N.ptr(rec), N.ptr(mask), N.ptr(lengths), B, C, T, N.ptr(out2),
                                           N.ptr(gl), N.ptr(grad_rec), int(ctx.prefix_only), N.stream_of(rec)), 'dic_masked_sse_bwd')
        return None, grad_rec, None, None, None


def masked_mse(org_ob, rec_ob, padding_mask=None, lengths=None, prefix_only=False):
    """sum((rec*m - ob*m)^2) / #{m == 1} over the (global) batch.  ``prefix_only`` (lengths, no mask): the gradient wrt ``rec_ob`` is
    written for the first n slots of each row only (see rbf_deinterp)."""
    if padding_mask is None and lengths is None:
        raise ValueError('masked_mse needs a mask or prefix lengths')
    return _MaskedMse.apply(org_ob, rec_ob.float(), padding_mask, lengths, prefix_only)


# ----------------------------------------------------------------------------------------- k3
class _DecAssign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, centers, alpha, want_colsum):
        N.require_gpu(z, centers)
        z, mu = N.f32c(z), N.f32c(centers.detach())
        B, D = z.shape
        K = mu.shape[0]
        if mu.shape[1] != D:
            raise ValueError(f'embedding dim mismatch: z {tuple(z.shape)} vs centers {tuple(mu.shape)}')
        q = torch.empty((B, K), device=z.device, dtype=torch.float32)
        need_grad = any(ctx.needs_input_grad)
        ts = torch.empty_like(q) if need_grad else None
        colsum = torch.empty(K, device=z.device, dtype=torch.float32) if want_colsum else None
        L = N.lib()
        ws = _ws(L.dic_dec_fwd_workspace(B, D, K), z.device)
        N.check(L.dic_dec_fwd(N.ptr(z), N.ptr(mu), B, D, K, float(alpha), N.ptr(q), N.ptr(ts), N.ptr(colsum), N.ptr(ws),
                              ws.numel(), N.stream_of(z)), 'dic_dec_fwd')
        ctx.dims = (B, D, K, float(alpha))
        ctx.sink_params = (centers,)
        ctx.save_for_backward(z, mu, q, ts)
        if want_colsum:
            ctx.mark_non_differentiable(colsum)
            return q, colsum
        return q, None

    @staticmethod
    def backward(ctx, grad_q, _unused=None):
        z, mu, q, ts = ctx.saved_tensors
        B, D, K, alpha = ctx.dims
        g = N.f32c(grad_q)
        gz = torch.empty_like(z)
        gc = torch.empty_like(mu)
        L = N.lib()
        ws = _ws(L.dic_dec_bwd_workspace(B, D, K), g.device)
        N.check(L.dic_dec_bwd(N.ptr(z), N.ptr(mu), N.ptr(q), N.ptr(ts), N.ptr(g), B, D, K, alpha, N.ptr(gz), N.ptr(gc),
                              N.ptr(ws), ws.numel(), N.stream_of(g)), 'dic_dec_bwd')
        return gz, _sink(ctx.sink_params[0], gc), None, None


def dec_soft_assign(z, centers, alpha=1.0, return_colsum=False):
    """Student-t soft assignment q (B,K); optionally also the local column sums f_j = sum_i q_ij."""
    q, colsum = _DecAssign.apply(z.float(), centers, alpha, return_colsum)
    return (q, colsum) if return_colsum else q


def dec_target(q, colsum=None):
    """p = (q^2/f)/sum_j(q^2/f).  ``colsum`` = local column sums from ``dec_soft_assign`` (recomputed
    through a forward-only pass of the same kernel family when absent); all-reduced when sharded."""
    N.require_gpu(q)
    qd = N.f32c(q.detach())
    B, K = qd.shape
    if colsum is None:
        colsum = qd.sum(dim=0)      # K floats of host-side plumbing; hot callers pass the kernel's colsum
    if getattr(colsum, '_dic_deferred_sum', False):
        dist.resolve_sum_(colsum)         # queued by the caller right after the soft assignment (dist.deferred_sum_): it has usually travelled already
    else:
        colsum = colsum.clone() if dist.is_sharded() else colsum
        dist.all_reduce_sum_(colsum)
    p = torch.empty_like(qd)
    N.check(N.lib().dic_dec_target(N.ptr(qd), N.ptr(N.f32c(colsum)), B, K, N.ptr(p), N.stream_of(qd)), 'dic_dec_target')
    return p


class _KlBatchMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, q):
        N.require_gpu(p, q)
        pd, qd = N.f32c(p.detach()), N.f32c(q)
        B, K = qd.shape
        sharded = dist.is_sharded()
        kl = torch.empty(1, device=qd.device, dtype=torch.float32)
        gq = torch.empty_like(qd) if ctx.needs_input_grad[1] else None
        L = N.lib()
        ws = _ws(L.dic_dec_kl_workspace(B, K), qd.device)
        # 'batchmean' over the GLOBAL batch: single device -> the kernel divides by B; sharded -> it returns this rank's sum and the
        # division by the all-reduced row count happens here (the shards of a batch may differ by a row)
        N.check(L.dic_dec_kl(N.ptr(qd), N.ptr(pd), B, K, 1.0 if sharded else float(B), 1.0, N.ptr(kl), N.ptr(gq), N.ptr(ws), ws.numel(),
                             N.stream_of(qd)), 'dic_dec_kl')
        if sharded:
            both = torch.cat([kl, torch.full((1,), float(B), device=qd.device)])
            dist.all_reduce_sum_(both)
            kl = both[:1] / both[1]
            if gq is not None:
                gq = gq / both[1]
        ctx.save_for_backward(gq)
        return kl[0]

    @staticmethod
    def backward(ctx, grad_kl):
        (gq,) = ctx.saved_tensors
        return None, gq * grad_kl


def kl_batchmean(p, q):
    """F.kl_div(q.log(), p, reduction='batchmean') with the closed-form d/dq fused into the forward."""
    return _KlBatchMean.apply(p, q.float())


# ----------------------------------------------------------------------------------------- a4 tail
class _HeadLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, weight, bias):
        N.require_gpu(h, weight, bias)
        hb = h.to(torch.bfloat16).contiguous()
        n, k = hb.shape
        c = weight.shape[0]
        w, b = N.f32c(weight.detach()), N.f32c(bias.detach())
        v = torch.empty((n, c), device=hb.device, dtype=torch.float32)
        N.check(N.lib().dic_head_fwd(N.ptr(hb), N.ptr(w), N.ptr(b), n, k, c, N.ptr(v), N.stream_of(hb)), 'dic_head_fwd')
        ctx.save_for_backward(hb, w)
        ctx.h_dtype = h.dtype
        return v

    @staticmethod
    def backward(ctx, dv):
        hb, w = ctx.saved_tensors
        n, k = hb.shape
        c = w.shape[0]
        g = N.f32c(dv)
        dh = torch.empty_like(hb)
        dw, db = torch.empty_like(w), torch.empty(c, device=hb.device, dtype=torch.float32)
        L = N.lib()
        ws = _ws(L.dic_head_bwd_workspace(n, k, c), hb.device)
        N.check(L.dic_head_bwd(N.ptr(hb), N.ptr(w), N.ptr(g), n, k, c, N.ptr(dh), N.ptr(dw), N.ptr(db), N.ptr(ws), ws.numel(),
                               N.stream_of(hb)), 'dic_head_bwd')
        return dh.to(ctx.h_dtype), dw, db


HEAD_IN, HEAD_OUT = 128, (1, 2, 3, 4, 5, 6, 7, 8, 12, 16)


def head_linear(h, weight, bias):
    """Linear(128, C) for small C on (N,128) bf16 activations -> (N,C) f32 (CompressFC's output layer)."""
    return _HeadLinear.apply(h, weight, bias)


class _BnReluHead(torch.autograd.Function):
    """BatchNorm1d(128) -> ReLU -> Linear(128, C) over (N,128) bf16 rows (csrc/dic_bnhead.hip), with nn.BatchNorm1d's running
    statistics updated by the same moments kernel (training mode)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, weight, bias, eps, running_mean, running_var, nbt, momentum, training, relu, drop_p=0.0, col_sums=None):
        N.require_gpu(z, gamma, beta, weight, bias)
        # bf16 rows (the bf16 step) or, for an f32 pre-activation outside autocast, f32 rows through the `_f32` twins of the same kernels
        f32z = z.dtype == torch.float32 and not torch.is_autocast_enabled() and f32_products() == 'x3'
        zb = z.contiguous() if f32z else z.to(torch.bfloat16).contiguous()
        sfx = '_f32' if f32z else ''
        n, k = zb.shape
        c = weight.shape[0]
        L, dev, st = N.lib(), zb.device, N.stream_of(zb)
        if training:
            if col_sums is not None and col_sums.numel() == 2 * k + 1:
                sums = col_sums                   # came out of the kernel that produced z (rows_linear(..., with_stats=True))
            else:
                sums = torch.empty(2 * k + 1, device=dev, dtype=torch.float64)
                ws = _ws(L.dic_bn_colstats_workspace(n, k), dev)
                N.check(getattr(L, 'dic_bn_colstats' + sfx)(N.ptr(zb), n, k, N.ptr(sums), N.ptr(ws), ws.numel(), st), 'dic_bn_colstats')      # [sum z | sum z^2 | rows]
            dist.all_reduce_sum_(sums)            # the moments of the GLOBAL batch
            mean = torch.empty(k, device=dev, dtype=torch.float32)
            rstd, cnt = torch.empty_like(mean), torch.empty(1, device=dev, dtype=torch.float32)
            track = running_mean is not None
            N.check(L.dic_bn_moments(N.ptr(sums), k, float(eps), -1.0 if momentum is None else float(momentum),
                                     N.ptr(running_mean) if track else None, N.ptr(running_var) if track else None,
                                     N.ptr(nbt) if (track and nbt is not None) else None, N.ptr(mean), N.ptr(rstd), N.ptr(cnt), st), 'dic_bn_moments')
        else:
            mean = N.f32c(running_mean)
            rstd = torch.rsqrt(N.f32c(running_var) + eps)
            cnt = None
        g, bt, w, b = N.f32c(gamma.detach()), N.f32c(beta.detach()), N.f32c(weight.detach()), N.f32c(bias.detach())
        v = torch.empty((n, c), device=dev, dtype=torch.float32)
        rng = _dropout_rng(dev) if drop_p > 0 else None
        N.check(getattr(L, 'dic_bnhead_fwd' + sfx)(N.ptr(zb), N.ptr(mean), N.ptr(rstd), N.ptr(g), N.ptr(bt), N.ptr(w), N.ptr(b), n, k, c, int(relu), float(drop_p), N.ptr(rng), N.ptr(v), st),
                'dic_bnhead_fwd')
        ctx.save_for_backward(zb, mean, rstd, g, bt, w, cnt, rng)
        ctx.z_dtype, ctx.training, ctx.relu, ctx.drop_p, ctx.sfx = z.dtype, bool(training), int(relu), float(drop_p), sfx
        return v

    @staticmethod
    def backward(ctx, dv):
        zb, mean, rstd, g, bt, w, cnt, rng = ctx.saved_tensors
        n, k = zb.shape
        c = w.shape[0]
        L, dev, st = N.lib(), zb.device, N.stream_of(zb)
        gv = N.f32c(dv)
        sums = torch.empty((2 + c) * k + c, device=dev, dtype=torch.float32)
        ws = _ws(L.dic_bnhead_bwd_workspace(n, k, c), dev)
        N.check(getattr(L, 'dic_bnhead_bwd_reduce' + ctx.sfx)(N.ptr(zb), N.ptr(mean), N.ptr(rstd), N.ptr(g), N.ptr(bt), N.ptr(w), N.ptr(gv), n, k, c, ctx.relu, ctx.drop_p, N.ptr(rng),
                                                              N.ptr(sums), N.ptr(ws), ws.numel(), st), 'dic_bnhead_bwd_reduce')
        dbeta, dgamma = sums[:k], sums[k:2 * k]                 # this rank's share; the gradient all-reduce sums them
        dw, db = sums[2 * k:(2 + c) * k].view(c, k), sums[(2 + c) * k:]
        dz = None
        if ctx.needs_input_grad[0]:
            if ctx.training:
                red = sums[:2 * k]
                if dist.is_sharded():
                    red = red.clone()
                    dist.all_reduce_sum_(red)
                count = cnt                         # sums / global row count, divided in-kernel (shards may differ by a row)
            else:
                red = torch.zeros(2 * k, device=dev, dtype=torch.float32)
                count = None
            dz = torch.empty_like(zb)
            N.check(getattr(L, 'dic_bnhead_bwd_input' + ctx.sfx)(N.ptr(zb), N.ptr(mean), N.ptr(rstd), N.ptr(g), N.ptr(bt), N.ptr(w), N.ptr(gv), N.ptr(red),
                                                                 N.ptr(red[k:]), 1.0, N.ptr(count), n, k, c, ctx.relu, ctx.drop_p, N.ptr(rng), N.ptr(dz), st), 'dic_bnhead_bwd_input')
            dz = dz.to(ctx.z_dtype)
        return dz, dgamma, dbeta, dw, db, None, None, None, None, None, None, None, None, None


BNHEAD_OUT = (1, 2, 3, 4, 5, 6, 7, 8, 12)      # head widths dic_bnhead_* is compiled for (12: BASELINE configs[3]'s twelve channels)
_DROP_STATE = {}


def _dropout_rng(device):
    """(seed, call counter) for the in-kernel dropout mask, as a fresh 2-word device tensor for ONE forward/backward pair.  The
    per-device counter advances with an in-place device add, so a captured hipGraph draws a new mask on every replay."""
    key = (device.type, device.index)
    st = _DROP_STATE.get(key)
    if st is None:
        st = _DROP_STATE[key] = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
    st[1:] += 1
    return st.clone()


def dropout_state_snapshot(device):
    """(for step.Stepper's hipGraph warm-up) the dropout call counter of ``device``; ``dropout_state_restore`` puts it back, so the warm-up
    runs leave no trace in the mask sequence."""
    st = _DROP_STATE.get((device.type, device.index))
    return None if st is None else st.clone()


def dropout_state_restore(device, snap):
    key = (device.type, device.index)
    if snap is None:
        if key in _DROP_STATE:
            _DROP_STATE[key][1:] = 0          # created during the warm-up: back to the state a first call expects
    else:
        _DROP_STATE[key].copy_(snap)


def bn_relu_head(z, bn, linear, relu=True, dropout=None, col_sums=None):
    """``linear(relu(bn(z)))`` (``relu=False``: ``linear(bn(z))``) for an nn.BatchNorm1d(128) and an nn.Linear(128, C <= 8) on (N,128) bf16 rows, with the
    module semantics of BatchNorm1d (batch moments + running statistics in training mode, running statistics in
    eval mode) and moments over the GLOBAL batch when the batch is sharded over ranks (dist.GlobalBatchNorm1d).  ``dropout``: the
    nn.Dropout between the activation and ``linear`` (active when in training mode with p > 0; the mask is drawn in-kernel).
    ``col_sums``: this rank's [sum z | sum z^2 | rows] (f64) when the producer of ``z`` already has them (rows_linear(with_stats=True))."""
    training = bn.training or bn.running_mean is None
    drop_p = float(dropout.p) if (dropout is not None and dropout.training) else 0.0
    track = bn.training and bn.track_running_stats and bn.running_mean is not None
    return _BnReluHead.apply(z, bn.weight, bn.bias, linear.weight, linear.bias, bn.eps, bn.running_mean if (track or not training) else None,
                             bn.running_var if (track or not training) else None, bn.num_batches_tracked if track else None, bn.momentum,
                             training, relu, drop_p, col_sums if (training and col_sums is not None) else None)


def splitk_tn(a, b, chunks=(8192, 4096, 2048)):
    """a^T b in f32 for a (N,M), b (N,K) bf16 (rows may be strided views) with N in the hundreds of thousands and a small
    (M,K) output: the weight-gradient shape of everything applied per (time step, encounter) row.  Issued as a bmm over
    row chunks + an f32 sum of the chunk products: measured 800 TFLOP/s with 8192-row chunks, against 230 for the single
    GEMM (4 output tiles) and 330 for one chunk per time step (scripts/gemm_probe.py)."""
    n = a.shape[0]
    for c in chunks:
        if n % c == 0 and n >= 2 * c:
            return torch.sum(torch.bmm(a.unflatten(0, (n // c, c)).transpose(1, 2), b.unflatten(0, (n // c, c))), dim=0, dtype=torch.float32)
    c = chunks[0]
    if n >= 4 * c:            # row counts that are not a multiple of a chunk (batch 10 000, 75 000, ...): chunked body + a short tail
        m = n // c * c
        body = torch.sum(torch.bmm(a[:m].unflatten(0, (m // c, c)).transpose(1, 2), b[:m].unflatten(0, (m // c, c))), dim=0, dtype=torch.float32)
        return body + (a[m:].t() @ b[m:]).float()
    return (a.t() @ b).float()


# ----------------------------------------------------------------------------------------- row-streaming MFMA products (csrc/dic_gemm.hip)
# How the f32 step (no autocast) forms its dense products -- nn.LSTM's input projections and recurrent products, nn.Linear, and their gradients
# (clustering_interp.py:14-41, rbf.py:111-125):
#   'exact'  the exact-f32 MFMA recurrence (csrc/dic_lstm32.hip, v_mfma_f32_32x32x2_f32) and f32 library GEMMs: the 1e-5 parity configuration
#            of the test suite, every shape;
#   'x3'     operands split on the fly into bf16 hi + lo pieces, products as hi.hi + lo.hi + hi.lo on the bf16 matrix cores with f32
#            accumulation (dic_gemm_nt / dic_gemm_tn, the split recurrence kernels): every tensor stays f32, no library GEMM, the step's losses
#            within ~1e-6 of the reference (tests/test_gpu_traj.py) at several times the exact mode's throughput.
_F32_PRODUCTS = [os.environ.get('DIC_F32_PRODUCTS') or 'exact']


def f32_products():
    return _F32_PRODUCTS[0]


class f32_products_mode:
    """``with f32_products_mode('x3'): ...`` -- see above; step.Stepper(precision=...) wraps its forward and backward in it."""

    def __init__(self, mode):
        if mode not in ('exact', 'x3'):
            raise ValueError(f"f32 products mode must be 'exact' or 'x3', got {mode!r}")
        self.mode = mode

    def __enter__(self):
        self.prev, _F32_PRODUCTS[0] = _F32_PRODUCTS[0], self.mode
        return self

    def __exit__(self, *exc):
        _F32_PRODUCTS[0] = self.prev
        return False


def _dt(t):
    if t.dtype == torch.float32:
        return N.DTYPE_F32
    if t.dtype == torch.bfloat16:
        return N.DTYPE_BF16
    raise TypeError(f'dic_gemm: f32 or bf16 operands, got {t.dtype}')


def _rows(t):
    """(rows, columns) matrix view whose rows are contiguous (row stride arbitrary)."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f'dic_gemm: need a 2-D operand with contiguous rows, got shape {tuple(t.shape)} strides {t.stride()}')
    return t


def gemm_nt(a, w, bias=None, out_dtype=None, relu_a=False, out=None):
    """y (M,N) = act(a (M,K)) . w (N,K)^T (+ bias (N) f32) on the matrix cores (dic_gemm_nt): bf16 operands -> one MFMA per product; f32
    operands -> the three-term bf16 split with f32 accumulation.  Rows may be strided views (16-B aligned, strides multiples of 4 / 8)."""
    N.require_gpu(a, w)
    a, w = _rows(a), _rows(w)
    if a.dtype != w.dtype or a.shape[1] != w.shape[1]:
        raise ValueError(f'gemm_nt: a {tuple(a.shape)} {a.dtype} vs w {tuple(w.shape)} {w.dtype}')
    M, K = a.shape
    n = w.shape[0]
    y = torch.empty((M, n), device=a.device, dtype=out_dtype or a.dtype) if out is None else _rows(out)
    b = None if bias is None else N.f32c(bias)
    N.check(N.lib().dic_gemm_nt(_dt(a), _dt(y), N.ptr(a), a.stride(0), N.ptr(w), w.stride(0), N.ptr(b), M, n, K, N.ptr(y), y.stride(0), int(bool(relu_a)),
                                N.stream_of(a)), 'dic_gemm_nt')
    return y


def _planes(a):
    """A "split-plane" f32 operand: (2, M, K) bf16, plane 0 = hi = bf16(x), plane 1 = lo = bf16(x - hi), x = hi + lo to 2^-17 -- what the x3 recurrence
    kernels write (same bytes as f32; the products multiply the planes as they lie).  Rows contiguous, both planes with the same row stride."""
    if a.dim() != 3 or a.shape[0] != 2 or a.dtype != torch.bfloat16 or a.stride(2) != 1:
        raise ValueError(f'split planes: need a (2, M, K) bf16 tensor with contiguous rows, got {tuple(a.shape)} {a.dtype} strides {a.stride()}')
    return a


def split_planes(x):
    """(2, *x.shape) bf16: hi = bf16(x), lo = bf16(x - hi) of an f32 tensor (contiguous planes)."""
    xc = x.contiguous()
    hi = xc.to(torch.bfloat16)
    return torch.stack([hi, (xc - hi.float()).to(torch.bfloat16)])


def planes_to_f32(a):
    """The f32 tensor a split-plane operand stands for (tests, odd consumers)."""
    return a[0].float() + a[1].float()


def gemm_nt_planes(a, w, bias=None):
    """y (M,N) f32 = a . w (N,K)^T (+ bias) with ``a`` a split-plane operand (2, M, K) and w f32: the three-term product without converting a."""
    N.require_gpu(a, w)
    a, w = _planes(a), _rows(w)
    _, M, K = a.shape
    if w.dtype != torch.float32 or w.shape[1] != K:
        raise ValueError(f'gemm_nt_planes: a {tuple(a.shape)} vs w {tuple(w.shape)} {w.dtype}')
    n = w.shape[0]
    y = torch.empty((M, n), device=a.device, dtype=torch.float32)
    b = None if bias is None else N.f32c(bias)
    N.check(N.lib().dic_gemm_nt_planes(N.ptr(a), a.stride(0), a.stride(1), N.ptr(w), w.stride(0), N.ptr(b), M, n, K, N.ptr(y), y.stride(0), N.stream_of(a)),
            'dic_gemm_nt_planes')
    return y


X3_ROW_PROJ = os.environ.get('DIC_X3_ROW_PROJ', '1') != '0'      # (A/B switch: 0 = dic_gemm_nt for the 256-input projections of the x3 step too)


def x3_row_proj_ok(x, w):
    """f32 (N,256) contiguous rows times f32 (128 | k*256, 256): the resident-weight form of the three-term-split product (dic_x3_row_proj)."""
    return (X3_ROW_PROJ and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 256 and x.is_contiguous()
            and w.dim() == 2 and w.shape[1] == 256 and w.is_contiguous() and (w.shape[0] == 128 or w.shape[0] % 256 == 0))


def x3_row_proj(x, w, bias=None, relu_a=False):
    """y (N, Nout) f32 = act(x (N,256)) . w (Nout,256)^T + bias with every product a three-term bf16 split, the weights resident in registers."""
    N.require_gpu(x, w)
    y = torch.empty((x.shape[0], w.shape[0]), device=x.device, dtype=torch.float32)
    b = None if bias is None else N.f32c(bias)
    N.check(N.lib().dic_x3_row_proj(N.ptr(x), N.ptr(w), N.ptr(b), x.shape[0], 256, w.shape[0], N.ptr(y), int(bool(relu_a)), N.stream_of(x)), 'dic_x3_row_proj')
    return y


def gemm_tn_into(a, x, dst, kcols=None, accumulate=False, x2=None, dst2=None, relu_x=False):
    """dst (N,kcols) f32 (+)= a (M,N)^T . x (M,K)[:, :kcols] (dic_gemm_tn): the weight-gradient shape -- a reduction over hundreds of thousands of
    rows into a small matrix; row chunks in parallel, fixed-order f64 second stage (deterministic).  ``x2`` (M,K2), ``dst2`` (N,K2): a second
    product a^T . x2 from the same pass over ``a`` (dW_ih and dW_hh of one LSTM direction read the gate gradients once).  ``relu_x``: the first product
    runs on relu(x)."""
    N.require_gpu(a, x, dst)
    planes = a.dim() == 3                            # a split-plane operand (2, M, N) bf16 standing for f32 gate gradients: x / x2 are f32
    a, x, dst = (_planes(a) if planes else _rows(a)), _rows(x), _rows(dst)
    a_dtype = torch.float32 if planes else a.dtype
    if a_dtype != x.dtype or a.shape[-2] != x.shape[0] or dst.dtype != torch.float32:
        raise ValueError(f'gemm_tn: a {tuple(a.shape)} {a.dtype}, x {tuple(x.shape)} {x.dtype}, dst {dst.dtype}')
    M, n = a.shape[-2:]
    K = x.shape[1]
    kcols = K if kcols is None else int(kcols)
    if tuple(dst.shape) != (n, kcols):
        raise ValueError(f'gemm_tn: dst {tuple(dst.shape)} != ({n}, {kcols})')
    K2 = 0
    if x2 is not None:
        x2, dst2 = _rows(x2), _rows(dst2)
        K2 = x2.shape[1]
        if x2.dtype != a_dtype or x2.shape[0] != M or tuple(dst2.shape) != (n, K2) or dst2.dtype != torch.float32:
            raise ValueError(f'gemm_tn: x2 {tuple(x2.shape)} {x2.dtype}, dst2 {tuple(dst2.shape)} {dst2.dtype}')
    L = N.lib()
    ws = _ws(L.dic_gemm_tn_workspace(M, n, K, K2), a.device)
    if planes:
        N.check(L.dic_gemm_tn_planes(N.ptr(a), a.stride(0), a.stride(1), N.ptr(x), x.stride(0), M, n, K, N.ptr(dst), dst.stride(0), kcols,
                                     N.ptr(x2), x2.stride(0) if x2 is not None else 0, K2, N.ptr(dst2), dst2.stride(0) if dst2 is not None else 0,
                                     int(bool(accumulate)), int(bool(relu_x)), N.ptr(ws), ws.numel(), N.stream_of(a)), 'dic_gemm_tn_planes')
        return dst
    N.check(L.dic_gemm_tn(_dt(a), N.ptr(a), a.stride(0), N.ptr(x), x.stride(0), M, n, K, N.ptr(dst), dst.stride(0), kcols,
                          N.ptr(x2), x2.stride(0) if x2 is not None else 0, K2, N.ptr(dst2), dst2.stride(0) if dst2 is not None else 0,
                          int(bool(accumulate)), int(bool(relu_x)), N.ptr(ws), ws.numel(), N.stream_of(a)), 'dic_gemm_tn')
    return dst


class _MfmaLinear(torch.autograd.Function):
    """y = x W^T + b over (N, in) rows on dic_gemm_nt / dic_gemm_tn -- nn.Linear (rbf.py:111-125's first layer, the heads' first layers) without a
    library GEMM: f32 operands in the 'x3' mode of the f32 step, bf16 operands in the bf16 step's small-batch path."""

    @staticmethod
    def forward(ctx, x, weight, bias, bias_grad_is_zero):
        xc = x if (x.dim() == 2 and x.stride(1) == 1) else x.contiguous()
        wc = weight.detach().to(xc.dtype).contiguous()
        ctx.save_for_backward(xc, wc)
        ctx.zero_db, ctx.sink_params = bool(bias_grad_is_zero), (weight, bias)
        if x3_row_proj_ok(xc, wc):
            return x3_row_proj(xc, wc, bias.detach())
        return gemm_nt(xc, wc, bias.detach())

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        dyc = dy.to(xc.dtype).contiguous()
        dx = gemm_nt(dyc, wc.t().contiguous()) if ctx.needs_input_grad[0] else None
        dw = gemm_tn_into(dyc, xc, torch.empty(wc.shape, device=xc.device, dtype=torch.float32))
        db = torch.zeros(wc.shape[0], device=xc.device, dtype=torch.float32) if ctx.zero_db else torch.sum(dyc, dim=0, dtype=torch.float32)
        return dx, _sink(ctx.sink_params[0], dw), _sink(ctx.sink_params[1], db), None


def mfma_linear(x, weight, bias, bias_grad_is_zero=False):
    return _MfmaLinear.apply(x, weight, bias, bias_grad_is_zero)


FC_BWD_SHAPE = (128, 256)      # dic_fc_bwd's compiled Linear(256, 128)
FC_BWD_MIN_ROWS = int(os.environ.get('DIC_FC_BWD_MIN_ROWS') or 1024)          # (8192 until round 3: the one-node path also wins where the step is launch-bound --
                                                                            # B = 256, 6144 rows: 73 -> 63 launches, 0.708 -> 0.677 ms from the hipGraph; 0.66 -> 0.63 at B = 64)


class _RowsLinear(torch.autograd.Function):
    """y = x W^T + b on (N, I) bf16 rows with N in the hundreds of thousands (library GEMMs).  Only the weight
    gradient is special: dW = dy^T x has K = N and a tiny output (see splitk_tn)."""

    @staticmethod
    def forward(ctx, x, weight, bias, bias_grad_is_zero, with_stats=False):
        xb, wb = x.to(torch.bfloat16), weight.to(torch.bfloat16)
        ctx.save_for_backward(xb, wb)
        ctx.x_dtype, ctx.zero_db = x.dtype, bool(bias_grad_is_zero)
        n = xb.shape[0]
        if with_stats and xb.is_cuda and tuple(wb.shape) == FC_BWD_SHAPE and n >= FC_BWD_MIN_ROWS and xb.is_contiguous():
            # CompressFC's first layer in front of its BatchNorm: z and its column sums from one kernel (csrc/dic_rowproj.hip)
            L = N.lib()
            z = torch.empty((n, wb.shape[0]), device=xb.device, dtype=torch.bfloat16)
            sums = torch.empty(2 * wb.shape[0] + 1, device=xb.device, dtype=torch.float64)
            ws = _ws(L.dic_row_proj_stats_workspace(n, wb.shape[0]), xb.device)
            N.check(L.dic_row_proj_stats(N.ptr(xb), N.ptr(wb.contiguous()), N.ptr(bias.detach().to(torch.bfloat16)), n, wb.shape[1], wb.shape[0],
                                         N.ptr(z), N.ptr(sums), N.ptr(ws), ws.numel(), N.stream_of(xb)), 'dic_row_proj_stats')
            ctx.mark_non_differentiable(sums)
            return z, sums
        z = torch.addmm(bias.to(torch.bfloat16), xb, wb.t())
        if not with_stats:
            return z
        none = torch.empty(0, device=xb.device, dtype=torch.float64)
        ctx.mark_non_differentiable(none)
        return z, none

    @staticmethod
    def backward(ctx, dy, *unused):
        xb, wb = ctx.saved_tensors
        dyb = dy.to(torch.bfloat16).contiguous()
        if ctx.zero_db:
            db = torch.zeros(wb.shape[0], device=dyb.device, dtype=torch.float32)
        else:
            db = torch.sum(dyb, dim=0, dtype=torch.float32)
        n = dyb.shape[0]
        if dyb.is_cuda and tuple(wb.shape) == FC_BWD_SHAPE and n >= FC_BWD_MIN_ROWS and xb.is_contiguous():
            # CompressFC's first layer over all (time step, encounter) rows: dx and dW from one pass over dy (csrc/dic_fcgrad.hip)
            L = N.lib()
            dxb = torch.empty_like(xb) if ctx.needs_input_grad[0] else None
            dw = torch.empty(wb.shape, device=dyb.device, dtype=torch.float32)
            ws = _ws(L.dic_fc_bwd_workspace(n, wb.shape[1], wb.shape[0]), dyb.device)
            N.check(L.dic_fc_bwd(N.ptr(dyb), N.ptr(xb), N.ptr(wb.contiguous()), n, wb.shape[1], wb.shape[0], N.ptr(dxb), N.ptr(dw), N.ptr(ws),
                                 ws.numel(), N.stream_of(dyb)), 'dic_fc_bwd')
            return (None if dxb is None else dxb.to(ctx.x_dtype)), dw, db, None, None
        dx = (dyb @ wb).to(ctx.x_dtype) if ctx.needs_input_grad[0] else None
        return dx, splitk_tn(dyb, xb), db, None, None


COMPRESS_FUSED = os.environ.get('DIC_COMPRESS_FUSED', '1') != '0'       # (A/B switch: 0 = rows_linear + bn_relu_head as two autograd nodes)
COMPRESS_FUSED_OUT = 6           # head width dic_fc_bwd_bnhead is compiled for (the reference's six vitals)


class _CompressFC(torch.autograd.Function):
    """CompressFC (rbf.py:111-125) in training mode, Linear(256,128) -> BatchNorm1d(128) -> ReLU -> Dropout -> Linear(128,6) over the
    (N,256) bf16 decoder rows, as ONE autograd node.  Forward: the kernels of rows_linear(with_stats=True) and bn_relu_head
    (dic_row_proj_stats -> dic_bn_moments -> dic_bnhead_fwd).  Backward: dic_bnhead_bwd_reduce, then dic_fc_bwd_bnhead forms the
    gradient of the 128-wide pre-activation on its way into LDS instead of reading it -- dic_bnhead_bwd_input's pass over z and the
    (N,128) dZ tensor (a 201 MB write and read at B = 32768) are gone; same arithmetic, same bits."""

    @staticmethod
    def forward(ctx, x, w1, b1, gamma, beta, w2, b2, eps, running_mean, running_var, nbt, momentum, drop_p):
        N.require_gpu(x, w1, b1, gamma, beta, w2, b2)
        xb, wb = x.to(torch.bfloat16).contiguous(), w1.detach().to(torch.bfloat16).contiguous()
        n, k, c = xb.shape[0], wb.shape[0], w2.shape[0]
        L, dev, st = N.lib(), xb.device, N.stream_of(xb)
        z = torch.empty((n, k), device=dev, dtype=torch.bfloat16)
        sums = torch.empty(2 * k + 1, device=dev, dtype=torch.float64)
        ws = _ws(L.dic_row_proj_stats_workspace(n, k), dev)
        N.check(L.dic_row_proj_stats(N.ptr(xb), N.ptr(wb), N.ptr(b1.detach().to(torch.bfloat16)), n, wb.shape[1], k, N.ptr(z), N.ptr(sums), N.ptr(ws),
                                     ws.numel(), st), 'dic_row_proj_stats')
        dist.all_reduce_sum_(sums)                # the moments of the GLOBAL batch
        mean = torch.empty(k, device=dev, dtype=torch.float32)
        rstd, cnt = torch.empty_like(mean), torch.empty(1, device=dev, dtype=torch.float32)
        track = running_mean is not None
        N.check(L.dic_bn_moments(N.ptr(sums), k, float(eps), -1.0 if momentum is None else float(momentum),
                                 N.ptr(running_mean) if track else None, N.ptr(running_var) if track else None,
                                 N.ptr(nbt) if (track and nbt is not None) else None, N.ptr(mean), N.ptr(rstd), N.ptr(cnt), st), 'dic_bn_moments')
        g, bt, w2f, b2f = N.f32c(gamma.detach()), N.f32c(beta.detach()), N.f32c(w2.detach()), N.f32c(b2.detach())
        v = torch.empty((n, c), device=dev, dtype=torch.float32)
        rng = _dropout_rng(dev) if drop_p > 0 else None
        N.check(L.dic_bnhead_fwd(N.ptr(z), N.ptr(mean), N.ptr(rstd), N.ptr(g), N.ptr(bt), N.ptr(w2f), N.ptr(b2f), n, k, c, 1, float(drop_p), N.ptr(rng), N.ptr(v), st),
                'dic_bnhead_fwd')
        ctx.save_for_backward(xb, wb, z, mean, rstd, g, bt, w2f, cnt, rng)
        ctx.x_dtype, ctx.drop_p = x.dtype, float(drop_p)
        ctx.sink_params = (w1, b1, gamma, beta, w2, b2)
        return v

    @staticmethod
    def backward(ctx, dv):
        xb, wb, z, mean, rstd, g, bt, w2f, cnt, rng = ctx.saved_tensors
        n, k, c = xb.shape[0], wb.shape[0], w2f.shape[0]
        L, dev, st = N.lib(), xb.device, N.stream_of(xb)
        gv = N.f32c(dv)
        sums = torch.empty((2 + c) * k + c, device=dev, dtype=torch.float32)
        ws = _ws(L.dic_bnhead_bwd_workspace(n, k, c), dev)
        N.check(L.dic_bnhead_bwd_reduce(N.ptr(z), N.ptr(mean), N.ptr(rstd), N.ptr(g), N.ptr(bt), N.ptr(w2f), N.ptr(gv), n, k, c, 1, ctx.drop_p, N.ptr(rng),
                                        N.ptr(sums), N.ptr(ws), ws.numel(), st), 'dic_bnhead_bwd_reduce')
        dbeta, dgamma = sums[:k], sums[k:2 * k]                 # this rank's share; the gradient all-reduce sums them
        dw2, db2 = sums[2 * k:(2 + c) * k].view(c, k), sums[(2 + c) * k:]
        red = sums[:2 * k]
        if dist.is_sharded():
            red = red.clone()
            dist.all_reduce_sum_(red)
        dxb = torch.empty_like(xb) if ctx.needs_input_grad[0] else None
        dw1 = torch.empty(wb.shape, device=dev, dtype=torch.float32)
        ws2 = _ws(L.dic_fc_bwd_workspace(n, wb.shape[1], k), dev)
        N.check(L.dic_fc_bwd_bnhead(N.ptr(z), N.ptr(gv), N.ptr(mean), N.ptr(rstd), N.ptr(g), N.ptr(bt), N.ptr(w2f), N.ptr(red), N.ptr(red[k:]), N.ptr(cnt),
                                    c, 1, ctx.drop_p, N.ptr(rng), N.ptr(xb), N.ptr(wb), n, wb.shape[1], k, N.ptr(dxb), N.ptr(dw1), N.ptr(ws2), ws2.numel(), st),
                'dic_fc_bwd_bnhead')
        # (the bias of the first layer sits in front of a training-mode BatchNorm: its gradient is identically 0)
        sp = ctx.sink_params
        db1 = None if (_SINK['on'] and GRAD_SINKS and getattr(sp[1], 'grad', None) is not None) else torch.zeros(k, device=dev, dtype=torch.float32)
        if db1 is None:
            sp[1]._dic_grad_written = True          # (adding zeros: nothing to queue)
        return ((None if dxb is None else dxb.to(ctx.x_dtype)), _sink(sp[0], dw1), db1, _sink(sp[2], dgamma), _sink(sp[3], dbeta),
                _sink(sp[4], dw2), _sink(sp[5], db2), None, None, None, None, None, None)


def compress_fc_fused_ok(x, first, bn, last):
    """The one-node CompressFC applies to the training step of the large-batch path: (N >= FC_BWD_MIN_ROWS, 256) contiguous rows,
    Linear(256,128), a training-mode BatchNorm1d, the six-channel head."""
    return (COMPRESS_FUSED and x.is_cuda and x.dim() == 2 and x.is_contiguous() and x.shape[0] >= FC_BWD_MIN_ROWS
            and tuple(first.weight.shape) == FC_BWD_SHAPE and bn.training and last.out_features == COMPRESS_FUSED_OUT
            and last.in_features == first.out_features)


def compress_fc(x, first, bn, dropout, last):
    drop_p = float(dropout.p) if (dropout is not None and dropout.training) else 0.0
    track = bn.track_running_stats and bn.running_mean is not None
    return _CompressFC.apply(x, first.weight, first.bias, bn.weight, bn.bias, last.weight, last.bias, bn.eps, bn.running_mean if track else None,
                             bn.running_var if track else None, bn.num_batches_tracked if track else None, bn.momentum, drop_p)


def rows_linear(x, weight, bias, bias_grad_is_zero=False, with_stats=False):
    """``bias_grad_is_zero``: the caller knows d loss / d bias == 0 identically -- a bias in front of a training-mode
    BatchNorm, whose mean subtraction cancels it (sum over rows of the BatchNorm input gradient is 0) -- so the (N, out)
    column reduction is skipped.  (The reference computes that sum and gets rounding noise around 0.)
    ``with_stats``: returns (z, col_sums) -- col_sums = [sum z | sum z^2 | rows] in f64 for bn_relu_head(col_sums=...) when the
    producing kernel can deliver them (else None)."""
    if not with_stats:
        return _RowsLinear.apply(x, weight, bias, bias_grad_is_zero, False)
    z, sums = _RowsLinear.apply(x, weight, bias, bias_grad_is_zero, True)
    return z, (sums if sums.numel() else None)

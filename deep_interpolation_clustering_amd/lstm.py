"""Fused bidirectional LSTM (hidden 128, one layer) for the encoder / decoder of ``Net``
(clustering_interp.py:14-41) on the persistent HIP recurrence kernels of ``csrc/dic_lstm.hip``.

``torch.nn.LSTM`` semantics and parameters (``weight_ih_l0[_reverse]``, ``weight_hh_l0[_reverse]``,
``bias_ih_l0[_reverse]``, ``bias_hh_l0[_reverse]``; gate order i,f,g,o) are kept -- the module still owns an
``nn.LSTM`` so ``state_dict`` keys do not change.  What changes is who executes it when the step runs
under bf16 autocast on the GPU:

  dic_lstm_pack (one launch): the eight f32 parameters -> bf16 operands (W_ih padded, bias column / bias vector, W_hh, W_hh^T)
  input projection gx = X.W_ih^T + b_ih + b_hh   (R*B x I) . (I x 8H):
      encoder (18 features packed to 32): inside the recurrence kernel (dic_lstm_fwd_proj), gx never exists;
      decoder (I = 256): dic_row_proj, weights resident in registers (smaller batches / other widths: library addmm)
  sequential recurrence (dic_lstm_fwd / dic_lstm_bwd, dic_lstm_rec_* for batches up to SMALL_BATCH): one workgroup per 64 (32)
      batch rows and direction keeps h, c (dh, dc) on chip for all R steps with W_hh resident in registers; the encoder's forward
      also writes the rectified copy of its output the decoder reads, its backward applies that ReLU's mask
  weight gradients dW_ih, dW_hh of both directions from ONE pass over dG (dic_lstm_dw: encoder, with its dX fused;
      dic_lstm_dw_wide: decoder); the decoder's dX = dG.W_ih is the one library GEMM left
  Parameter gradients are written by the kernels straight into ``param.grad`` when those exist (the flat gradient bucket of
  ``dist.FlatParams``): no (2,4H,.) staging copies, no per-parameter AccumulateGrad add launches.

In f32 (no autocast: the configuration of the 1e-5 parity tests) the same structure runs with f32 tensors throughout and the
recurrence on exact-f32 MFMA (csrc/dic_lstm32.hip) instead of MIOpen's nn.LSTM; batches up to SMALL_BATCH use that file's
one-tile-per-workgroup bf16 kernels too.  The bf16 path is checked against an f32 emulation with the same rounding points, the
f32 path against nn.LSTM itself.
"""
import os

import torch

from . import _native as N
from . import ops as _ops
from .ops import packed_width, splitk_tn

H = 128
PROJ_WIDTHS = (32, 64)         # dic_lstm_fwd_proj's / dic_lstm_dw's compiled packed input widths (3C = 18 / 36 features + the bias column)
PARAM_NAMES = ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0',
               'weight_ih_l0_reverse', 'weight_hh_l0_reverse', 'bias_ih_l0_reverse', 'bias_hh_l0_reverse')


FWD_EIGHT_WAVES = os.environ.get('DIC_FWD_EIGHT_WAVES', '1') != '0'   # (A/B switch: 0 = four waves per workgroup in the encoder's fused-projection recurrence)
DEFER_RELU = os.environ.get('DIC_DEFER_RELU', '1') != '0'             # (A/B switch: 0 = the encoder writes a rectified copy of its output for the decoder)
GX_LANE_NATIVE = os.environ.get('DIC_GX_LANE_NATIVE', '1') != '0'     # (A/B switch: 0 = row-major gx between dic_row_proj and dic_lstm_fwd)
ROW_PROJ = os.environ.get('DIC_ROW_PROJ', '1') != '0'                  # (A/B switch: 0 = library GEMM for the decoder's input projection)
RELU_IN_KERNEL = os.environ.get('DIC_RELU_IN_KERNEL', '1') != '0'      # (A/B switch: 0 = rectify the encoder output with a torch pass)
WIDE_INPUT = 256               # dic_lstm_dw_wide's compiled input width (the decoder: 2H rectified encoder outputs)
X3_DX_TILE = os.environ.get('DIC_X3_DX_TILE', '1') != '0'              # (A/B switch: 0 = dic_gemm_nt_planes for the decoder's dX in the x3 step)
X3_DW = os.environ.get('DIC_X3_DW', '1') != '0'                        # (A/B switch: 0 = dic_gemm_tn_planes for the x3 step's LSTM weight gradients)
X3_REC_PROJ = os.environ.get('DIC_X3_REC_PROJ', '1') != '0'            # (A/B switch: 0 = dic_gemm_nt + gx for the encoder's forward in the x3 step)
REC_PROJ = os.environ.get('DIC_REC_PROJ', '1') != '0'                  # (A/B switch: 0 = dic_gemm_nt + dic_lstm_rec_fwd for the encoder's small-batch forward)
FWD_XPROJ = os.environ.get('DIC_FWD_XPROJ', '1') != '0'                # (A/B switch: 0 = dic_row_proj + dic_lstm_fwd for the decoder's large-batch forward: gx through HBM)
SMALL_BATCH = int(os.environ.get('DIC_SMALL_BATCH') or 4096)      # up to here the one-tile-per-workgroup kernels of csrc/dic_lstm32.hip beat the 64-row pipelined ones


def _lstm_ok(lstm):
    return (lstm.hidden_size == H and lstm.num_layers == 1 and lstm.bidirectional and lstm.bias
            and not lstm.batch_first and lstm.proj_size == 0)


def fused_available(x, lstm):
    """bf16 step: autocast(bf16) on the GPU."""
    return x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16 and _lstm_ok(lstm)


def f32_available(x, lstm):
    """f32 step (no autocast) on the GPU: the exact-f32 MFMA recurrence instead of MIOpen's nn.LSTM."""
    return (x.is_cuda and not torch.is_autocast_enabled() and x.dtype == torch.float32 and _lstm_ok(lstm)
            and all(p.dtype == torch.float32 for p in lstm.parameters()))


# The decoder's weight-gradient kernel has no consumer before the optimizer: inside step.Stepper's backward it CAN run on a side stream next
# to the encoder backward that follows on the main stream (side_stream_session joins at the end).  Round 2 measured that as a win (6.56 ->
# 6.46 ms); round 3's same-box A/B (3 x 60 steps each) has it LOSING by 0.03-0.07 ms (6.505 vs 6.44-6.48 ms): both kernels are bound by the
# same HBM, the encoder's lstm_bwd stretches from 0.85 to 1.77 ms while they share the chip, and nothing is hidden.  Off by default.
DW_SIDE_STREAM = os.environ.get('DIC_DW_SIDE_STREAM', '0') == '1'
# the decoder's large-batch input gradient dX = dG.W_ih: dic_lstm_dx_tile (csrc/dic_dxproj.hip; 256 x 256 macro-tiles, both operands through LDS-DMA rings)
DX_TILE_MIN_ROWS = 256
_SIDE = {'on': False, 'streams': {}, 'pending': [], 'keep': []}
RECORD_STREAM = os.environ.get('DIC_SIDE_RECORD_STREAM', '0') == '1'      # (experiment switch: the allocator-side alternative)


class side_stream_session:
    def __enter__(self):
        _SIDE['on'], _SIDE['pending'] = True, []
        return self

    def __exit__(self, *exc):
        _SIDE['on'] = False
        join_side_streams()
        return False


def side_work_pending():
    return bool(_SIDE['pending'])


def pending_side_stream(dev):
    """The side stream of ``dev`` if work of the current session is still queued on it, else None."""
    for s in _SIDE['pending']:
        if s.device == dev:
            return s
    return None


def join_side_streams():
    pending, _SIDE['pending'] = _SIDE['pending'], []
    for s in pending:
        torch.cuda.current_stream(s.device).wait_stream(s)
    _SIDE['keep'] = []                 # (from here on the main stream is ordered behind the side work: its operands may be freed and reused)


def _side_stream(dev):
    key = (dev.type, dev.index)
    if key not in _SIDE['streams']:
        _SIDE['streams'][key] = torch.cuda.Stream(device=dev)
    return _SIDE['streams'][key]


def _grad_sinks(params, needs):
    """The tensors the kernels write the parameter gradients into, and whether they ACCUMULATE there.
    When every parameter already owns a dense f32 ``.grad`` (the views of the flat bucket, zeroed at the start of the step) the
    kernels add into it and autograd is handed ``None``; otherwise fresh tensors are returned through autograd."""
    from . import ops
    # (only inside step.Stepper's own backward -- ops.grad_sink_session: any other caller with populated .grad, e.g. torch.autograd.grad or
    #  a backward with hooks of its own, gets ordinary gradient tensors through autograd and an untouched .grad)
    if ops.sink_session_active() and all(needs) and all(p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() for p in params):
        for p in params:
            p._dic_grad_written = True         # dist.FlatParams.active_mask: autograd's accumulate hook will not fire for these
        return [p.grad for p in params], True
    return [torch.empty_like(p, dtype=torch.float32, memory_format=torch.contiguous_format) for p in params], False


class _BiLstm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h0, c0, packed, bm, f32, relu, relu_in, *params):
        R, B, I_in = x.shape
        T = torch.float32 if f32 else torch.bfloat16
        code = N.DTYPE_F32 if f32 else N.DTYPE_BF16
        dev = x.device
        I = params[0].shape[1]
        small = f32 or B <= SMALL_BATCH                           # one 32-row tile per workgroup (csrc/dic_lstm32.hip)
        x3 = f32 and _ops.f32_products() == 'x3'                  # f32 tensors, products as three bf16 MFMAs (csrc/dic_gemm.hip)
        narrow = (not f32) and packed_width(I) > 0                 # bf16, narrow input (encoder): rows packed to 32 / 64 with the bias as a constant-one column
        proj = narrow and not small                                # ... and projected inside the 64-row recurrence kernel
        # x3, narrow input (encoder): projected INSIDE the eight-wave x3 recurrence kernel from f32 rows [features | 1 | 0...] (round 6: no gx tensor)
        x3proj = x3 and I + 1 <= 32 and X3_REC_PROJ
        # K of the projection (bf16: padded to the MFMA step; f32 'x3': to the 16-B vector loads of dic_gemm_nt / the in-kernel projection)
        Ip = (((I + 1 if x3proj else I) + 3) // 4 * 4 if x3 else I) if f32 else (packed_width(I) if narrow else (I + 15) // 16 * 16)
        need = any(ctx.needs_input_grad)
        pf = [N.f32c(p.detach()) for p in params]
        wih = torch.empty((8 * H, Ip), device=dev, dtype=T)
        whh = torch.empty((2, 4 * H, H), device=dev, dtype=T)
        whh_t = torch.empty((2, H, 4 * H), device=dev, dtype=T) if (need and (x3 or not f32)) else None
        bias = None if (narrow or x3proj) else torch.empty(8 * H, device=dev, dtype=T)
        L, st = N.lib(), N.stream_of(x)
        # (the decoder's large-batch input gradient runs on dic_lstm_dx_tile, whose weight operand is W_ih^T: packed in the same launch)
        dx_tile = need and ctx.needs_input_grad[0] and (not f32) and I == WIDE_INPUT and Ip == I and not small and R * B >= DX_TILE_MIN_ROWS
        wih_t = torch.empty((Ip, 8 * H), device=dev, dtype=T) if dx_tile else None
        N.check(L.dic_lstm_pack(code, N.ptr_array(pf), H, I, Ip, int(narrow or x3proj), N.ptr(wih), N.ptr(whh), N.ptr(whh_t), N.ptr(bias), N.ptr(wih_t), st), 'dic_lstm_pack')
        ctx.wih_t = wih_t
        if packed:                                                 # (R,B,32) bf16 rows [features | 1 | 0...] from ops.sci_cci_packed
            if not narrow or I_in != Ip or x.dtype != T:
                raise ValueError(f'packed input must be (R,B,{Ip}) bf16 for an LSTM of input size {I}')
            xb = x if x.is_contiguous() else x.contiguous()
        else:
            if I_in != I:
                raise ValueError(f'input width {I_in} does not match the LSTM input size {I}')
            xb = x.to(T)
            if Ip != I:
                xb = torch.nn.functional.pad(xb, (0, Ip - I))
            xb = xb.contiguous()
            if narrow or x3proj:
                xb[..., I].fill_(1.0)                              # the bias rides along as a constant-one input column
        # input_rectify: relu(x) is applied by the projection / weight-gradient kernels on the raw x (large decoder path), else here
        relu_kernel = bool(relu_in) and (((not small) and (not proj) and Ip == WIDE_INPUT and ROW_PROJ)
                                         or (x3 and Ip == WIDE_INPUT and _ops.X3_ROW_PROJ))          # (x3: dic_x3_row_proj / dic_gemm_tn rectify on load)
        if relu_in and not relu_kernel:
            xb = torch.relu(xb)
        out_ext = torch.empty((R + 2, B, 2 * H), device=dev, dtype=T)       # [h0 | h_1..h_R | h0]: every step's h_prev is a row above / below
        out = out_ext[1:R + 1]
        kernel_boundary = need                                     # the kernels write the boundary rows of the dW_hh products themselves (h0 into time
                                                                   # slot 0 [:H] / slot R+1 [H:]; the other halves are never read): no fill / copy launches
        hn = torch.empty((B, 2, H) if bm else (2, B, H), device=dev, dtype=torch.float32)
        cn = torch.empty_like(hn)
        h0c = None if h0 is None else N.f32c(h0)
        c0c = None if c0 is None else N.f32c(c0)
        gates = cs = None
        out_r = torch.empty((R, B, 2 * H), device=dev, dtype=T) if (relu == 1 and not small and RELU_IN_KERNEL) else None     # written by the 64-row kernels themselves
        if small:
            if need:
                Bp = (B + 31) // 32 * 32                           # kernel-native saved state is tiled by 32 rows
                gates = torch.empty((R, Bp, 2, 4, H), device=dev, dtype=T)
                cs = torch.empty((R + 1, Bp, 2, H), device=dev, dtype=T)     # (time slot R: c0, written by the forward)
            # input projection gx (R*B, 2*4*H) of all steps: hand-written MFMA kernels (no library GEMM) except in the exact-f32 parity mode
            gx = None
            if (narrow and REC_PROJ) or x3proj:
                pass                                                             # (projected inside the recurrence kernel below: no gx)
            elif narrow:
                gx = _ops.gemm_nt(xb.view(R * B, Ip), wih)                       # (the bias rides in the constant-one column)
            elif (not f32) and Ip == WIDE_INPUT and ROW_PROJ:
                gx = torch.empty((R * B, 8 * H), device=dev, dtype=T)           # decoder: weights resident in registers (csrc/dic_rowproj.hip)
                N.check(L.dic_row_proj(N.ptr(xb), N.ptr(wih), N.ptr(bias), R * B, Ip, 8 * H, N.ptr(gx), 0, 0, st), 'dic_row_proj')
            elif x3 and _ops.x3_row_proj_ok(xb.view(R * B, Ip), wih):
                gx = _ops.x3_row_proj(xb.view(R * B, Ip), wih, bias, relu_a=relu_kernel)      # decoder, x3: weights split once, resident in registers
            elif x3 or not f32:
                gx = _ops.gemm_nt(xb.view(R * B, Ip), wih, bias.float(), relu_a=relu_kernel)
            else:
                gx = torch.addmm(bias, xb.view(R * B, Ip), wih.t())
            if gx is None and x3proj:
                N.check(L.dic_lstm_rec_fwd_proj_x3(N.ptr(xb), N.ptr(wih), Ip, N.ptr(whh), N.ptr(h0c), N.ptr(c0c), R, B, H, N.ptr(out), N.ptr(hn), N.ptr(cn),
                                                   N.ptr(gates), N.ptr(cs), int(bm) | (2 if kernel_boundary else 0), st), 'dic_lstm_rec_fwd_proj_x3')
            elif gx is None:
                N.check(L.dic_lstm_rec_fwd_proj(N.ptr(xb), N.ptr(wih), N.ptr(whh), N.ptr(h0c), N.ptr(c0c), R, B, H, Ip, N.ptr(out), N.ptr(hn), N.ptr(cn),
                                                N.ptr(gates), N.ptr(cs), int(bm) | (2 if kernel_boundary else 0), st), 'dic_lstm_rec_fwd_proj')
            else:
                N.check(L.dic_lstm_rec_fwd(N.DTYPE_F32X3 if x3 else code, N.ptr(gx), N.ptr(whh), N.ptr(h0c), N.ptr(c0c), R, B, H, N.ptr(out), N.ptr(hn), N.ptr(cn),
                                           N.ptr(gates), N.ptr(cs), int(bm) | (2 if kernel_boundary else 0), st), 'dic_lstm_rec_fwd')
        else:
            if need:
                Bp = (B + 63) // 64 * 64                           # kernel-native saved state is tiled by 64 rows
                gates = torch.empty((R, Bp, 2, 4, H), device=dev, dtype=T)
                cs = torch.empty((R, Bp, 2, H), device=dev, dtype=T)    # bf16 copy for the backward; c itself stays f32 on chip
            if proj:
                N.check(L.dic_lstm_fwd_proj(N.ptr(xb), N.ptr(wih), N.ptr(whh), N.ptr(h0c), N.ptr(c0c), R, B, H, Ip, N.ptr(out), N.ptr(out_r),
                                            N.ptr(hn), N.ptr(cn), N.ptr(gates), N.ptr(cs), int(bm), int(kernel_boundary), int(FWD_EIGHT_WAVES), st), 'dic_lstm_fwd_proj')
            else:
                native = 0
                if Ip == WIDE_INPUT and ROW_PROJ and FWD_XPROJ:
                    # decoder: the input projection INSIDE the recurrence kernel (csrc/dic_lstm32.hip, lstm_fwdx_kernel: W_ih in registers + LDS next to
                    # W_hh at one wave per SIMD) -- the (R*B, 8H) gx tensor, its 1.6 GB write and read-back at B = 32 768, does not exist
                    N.check(L.dic_lstm_fwd_xproj(N.ptr(xb), N.ptr(wih), N.ptr(whh), N.ptr(bias), N.ptr(h0c), N.ptr(c0c), R, B, H, Ip, N.ptr(out), N.ptr(out_r), N.ptr(hn),
                                                 N.ptr(cn), N.ptr(gates), N.ptr(cs), int(bm) | (2 if kernel_boundary else 0), int(relu_kernel), st), 'dic_lstm_fwd_xproj')
                    gx = None
                elif Ip == WIDE_INPUT and ROW_PROJ:
                    # decoder: the input projection with the weights resident in registers (csrc/dic_rowproj.hip); batches that tile by 64
                    # rows get gx in the order of the recurrence kernel's accumulators (no LDS staging of the gx tile over there)
                    native = B if (GX_LANE_NATIVE and B % 64 == 0) else 0
                    gx = torch.empty((R * B, 8 * H), device=dev, dtype=T)
                    N.check(L.dic_row_proj(N.ptr(xb), N.ptr(wih), N.ptr(bias), R * B, Ip, 8 * H, N.ptr(gx), native, int(relu_kernel), st), 'dic_row_proj')
                else:
                    gx = _ops.gemm_nt(xb.view(R * B, Ip), wih, bias.float())
                if gx is not None:
                    N.check(L.dic_lstm_fwd(N.ptr(gx), (2 if FWD_EIGHT_WAVES else 1) if native > 0 else 0, N.ptr(whh), N.ptr(h0c), N.ptr(c0c), R, B, H, N.ptr(out), N.ptr(out_r), N.ptr(hn),
                                       N.ptr(cn), N.ptr(gates), N.ptr(cs), int(bm), int(kernel_boundary), st), 'dic_lstm_fwd')
        ctx.dims = (R, B, I, Ip, narrow, small, bool(packed), bool(bm), bool(f32), bool(relu))
        ctx.x3 = x3
        ctx.x_relu_in_kernel = relu_kernel
        ctx.x_dtype = x.dtype
        ctx.has_init = h0 is not None
        ctx.params = params
        ctx.save_for_backward(xb, wih, whh if (f32 and not x3) else whh_t, gates, cs, out_ext, h0c, c0c)
        ctx.set_materialize_grads(False)       # an output nobody differentiates (c_n, often h_n or out) arrives as None, not as a zero tensor
        # relu: the caller consumes relu(out) only (the decoder's input, clustering_interp.py:38-41).  The raw rows stay in out_ext for the
        # weight-gradient products (the 64-row kernels write the rectified copy next to them); the backward kernel applies the ReLU
        # mask itself (sign of tanh(c_t)), so no mask pass and no saved copy
        # relu == 2 (deferred): the raw rows are handed out and the CONSUMER rectifies them on load (bilstm(input_rectify=True) of the
        # decoder); the backward of the ReLU is applied here all the same, so the pair behaves like relu between the two LSTMs
        if relu == 1:
            return (out_r if out_r is not None else torch.relu(out)), hn, cn
        return out, hn, cn

    @staticmethod
    def backward(ctx, dout, dhn, dcn):
        xb, wih, whh_b, gates, cs, out_ext, h0c, c0c = ctx.saved_tensors
        R, B, I, Ip, narrow, small, packed, bm, f32, relu = ctx.dims
        params = ctx.params
        T = torch.float32 if f32 else torch.bfloat16
        code = N.DTYPE_F32 if f32 else N.DTYPE_BF16
        dev = out_ext.device
        # x3: the gate gradients leave the recurrence kernel as SPLIT PLANES (2, R*B, 8H) bf16 -- hi / lo with dG = hi + lo, the same bytes as f32 -- which the
        # products below multiply as they lie (csrc/dic_lstm32.hip lstm_rec_bwd8x3_kernel; dic_gemm_tn_planes / dic_gemm_nt_planes)
        dgx = torch.empty((2, R * B, 8 * H), device=dev, dtype=torch.bfloat16) if ctx.x3 else torch.empty((R, B, 2, 4, H), device=dev, dtype=T)
        dh0 = torch.empty((B, 2, H) if bm else (2, B, H), device=dev, dtype=torch.float32)
        dc0 = torch.empty_like(dh0)
        doutb = None if dout is None else (dout if dout.dtype == T else dout.to(T)).contiguous()
        dhnc = None if dhn is None else N.f32c(dhn)
        dcnc = None if dcn is None else N.f32c(dcn)
        Lb, st = N.lib(), N.stream_of(out_ext)
        dbias = torch.empty((2, 4 * H), device=dev, dtype=torch.float32)         # summed inside the kernel, f32
        if small:
            ws = torch.empty(max(16, Lb.dic_lstm_rec_bwd_workspace(B)), device=dev, dtype=torch.uint8)
            N.check(Lb.dic_lstm_rec_bwd(N.DTYPE_F32X3 if ctx.x3 else code, N.ptr(whh_b), int(ctx.x3 or not f32), N.ptr(gates), N.ptr(cs), N.ptr(doutb), N.ptr(dhnc), N.ptr(dcnc),
                                        R, B, H, N.ptr(dgx), N.ptr(dh0), N.ptr(dc0), N.ptr(dbias), N.ptr(ws), ws.numel(), int(bm), int(relu), st), 'dic_lstm_rec_bwd')
        else:
            ws = torch.empty(max(16, Lb.dic_lstm_bwd_workspace(B)), device=dev, dtype=torch.uint8)
            N.check(Lb.dic_lstm_bwd(N.ptr(whh_b), N.ptr(gates), N.ptr(cs), N.ptr(c0c), N.ptr(doutb), N.ptr(dhnc), N.ptr(dcnc),
                                    R, B, H, N.ptr(dgx), N.ptr(dh0), N.ptr(dc0), N.ptr(dbias), N.ptr(ws), ws.numel(), int(bm), int(relu), st), 'dic_lstm_bwd')
        dg2 = dgx if ctx.x3 else dgx.view(R * B, 8 * H)
        use_dw = narrow and R * B >= 32                              # one-pass weight-gradient kernel (csrc/dic_lstmgrad.hip; it tiles the R*B rows by 32)
        fuse_dx = use_dw and Ip == 32 and I <= 19 and any(ctx.needs_input_grad[8:])       # ... which then also forms dX = dG.W_ih per direction (32-wide rows)
        x3 = ctx.x3
        dxp = torch.empty((2, R * B, Ip), device=dev, dtype=T) if (fuse_dx and ctx.needs_input_grad[0]) else None
        dx = None
        # x3, narrow input: dX rides along with the weight gradients (four partial tensors, one per direction and half of its gate rows: dic_lstm_dw_x3)
        x3_dw = x3 and X3_DW and R * B >= 16 and (I == WIDE_INPUT or Ip <= 32) and any(ctx.needs_input_grad[8:])
        dxp4 = torch.empty((4, R * B, Ip), device=dev, dtype=torch.float32) if (x3_dw and I != WIDE_INPUT and ctx.needs_input_grad[0]) else None
        if ctx.needs_input_grad[0] and dxp is None and dxp4 is None:
            if ctx.wih_t is not None:
                # decoder: dX = dG . W_ih on 256 x 256 macro-tiles, dG and W_ih^T (256, 1024: k contiguous like the rows of dG; written by the forward's
                # dic_lstm_pack) streamed through LDS (csrc/dic_dxproj.hip; until round 4: a library GEMM)
                dx = torch.empty((R * B, Ip), device=dev, dtype=T)
                N.check(Lb.dic_lstm_dx_tile(N.ptr(dg2), N.ptr(ctx.wih_t), R * B, 8 * H, Ip, N.ptr(dx), st), 'dic_lstm_dx_tile')
            elif f32 and not x3:
                dx = dg2 @ wih                                       # (R*B, Ip): the exact-f32 parity mode
            elif x3 and X3_DX_TILE and I == WIDE_INPUT and Ip == I and R * B >= DX_TILE_MIN_ROWS:
                # decoder, x3: dX = dG . W_ih on 256 x 256 macro-tiles from the split planes of dG and of W_ih^T, both streamed through LDS by DMA
                # (csrc/dic_dxproj.hip dx_tile_x3_kernel; round 6 -- until then dic_gemm_nt on the f32 operand)
                wt_pl = _ops.split_planes(wih.t())                   # (2, 256, 1024) bf16: a few launches on 1 MB
                dx = torch.empty((R * B, Ip), device=dev, dtype=torch.float32)
                N.check(Lb.dic_lstm_dx_tile_x3(N.ptr(dg2), dg2.stride(0), N.ptr(wt_pl), wt_pl.stride(0), R * B, 8 * H, Ip, N.ptr(dx), st), 'dic_lstm_dx_tile_x3')
            elif x3:
                dx = _ops.gemm_nt_planes(dg2, wih.t().contiguous())  # dX = dG . W_ih from the split planes of dG (W_ih^T: a (Ip, 8H) copy of the packed weights)
            else:
                dx = _ops.gemm_nt(dg2, wih.t().contiguous())         # dX = dG . W_ih on dic_gemm_nt (W_ih^T: a (Ip, 8H) copy of the packed weights)
            if packed:
                dx = dx.view(R, B, Ip)                               # consumed in this layout by ops._SciCciPacked.backward
            else:
                dx = (dx[:, :I] if Ip != I else dx).reshape(R, B, I).to(ctx.x_dtype)
        needs = ctx.needs_input_grad[8:]
        grads = [None] * 8
        if any(needs):
            sinks, accumulate = _grad_sinks(params, needs)
            gp = N.ptr_array(sinks)
            if use_dw:
                # dW_ih and dW_hh of both directions (and the per-direction dX) from one pass over dG
                ws2 = torch.empty(max(16, Lb.dic_lstm_dw_workspace(R, B)), device=dev, dtype=torch.uint8)
                N.check(Lb.dic_lstm_dw(N.ptr(dgx), N.ptr(out_ext), N.ptr(xb), N.ptr(wih) if dxp is not None else None, N.ptr(dxp), R, B, H, I, Ip, gp,
                                       int(accumulate), N.ptr(ws2), ws2.numel(), st), 'dic_lstm_dw')
                if dxp is not None:
                    dx = dxp[0] + dxp[1]                             # (columns >= 19 of the partials are never written, nor read downstream)
                    dx = dx.view(R, B, Ip) if packed else dx[:, :I].reshape(R, B, I).to(ctx.x_dtype)
                N.check(Lb.dic_lstm_unpack_grads(None, 0, None, N.ptr(dbias), H, I, gp, int(accumulate), st), 'dic_lstm_unpack_grads')
            elif (not f32) and I == WIDE_INPUT and Ip == I and R * B >= 32:
                # decoder: dW_ih and dW_hh of both directions from one pass over dG (csrc/dic_lstmgrad.hip, lstm_dw_wide_kernel)
                def weight_grads(stream):
                    ws2 = torch.empty(max(16, Lb.dic_lstm_dw_wide_workspace(R, B)), device=dev, dtype=torch.uint8)
                    N.check(Lb.dic_lstm_dw_wide(N.ptr(dgx), N.ptr(out_ext), N.ptr(xb), int(ctx.x_relu_in_kernel), R, B, H, I, gp, int(accumulate), N.ptr(ws2),
                                                ws2.numel(), stream), 'dic_lstm_dw_wide')
                    N.check(Lb.dic_lstm_unpack_grads(None, 0, None, N.ptr(dbias), H, I, gp, int(accumulate), stream), 'dic_lstm_unpack_grads')
                if _SIDE['on'] and DW_SIDE_STREAM and accumulate and not torch.cuda.is_current_stream_capturing():
                    side = _side_stream(dev)
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):
                        weight_grads(N.stream_of(dgx))
                    # the operands were allocated on the main stream and are read over there: instead of record_stream (deferred frees
                    # the caching allocator has to poll for, and a 1.6 GB block it cannot hand out meanwhile) the session keeps them
                    # alive until the main stream has waited for the side stream
                    if RECORD_STREAM:
                        for t_ in (dgx, out_ext, xb, dbias):
                            t_.record_stream(side)
                    _SIDE['pending'].append(side)
                    _SIDE['keep'].append((dgx, out_ext, xb, dbias, wih, sinks))
                else:
                    weight_grads(st)
            elif x3_dw:
                # x3: dW_ih and dW_hh of both directions (and the narrow input's dX) from ONE pass over the split planes of dG (csrc/dic_lstmgrad.hip,
                # lstm_dwx3_kernel: the planes by LDS-DMA as they lie, h_prev / x f32 through registers; round 6 -- until then five dic_gemm_tn launches and a
                # dic_gemm_nt on f32 operands)
                ws2 = torch.empty(max(16, Lb.dic_lstm_dw_x3_workspace(R, B, I)), device=dev, dtype=torch.uint8)
                N.check(Lb.dic_lstm_dw_x3(N.ptr(dgx), dgx.stride(0), N.ptr(out_ext), N.ptr(xb), Ip, int(ctx.x_relu_in_kernel), N.ptr(wih) if dxp4 is not None else None,
                                          N.ptr(dxp4), R, B, H, I, gp, int(accumulate), N.ptr(ws2), ws2.numel(), st), 'dic_lstm_dw_x3')
                if dxp4 is not None:
                    dx = dxp4.sum(0)[:, :I].reshape(R, B, I).to(ctx.x_dtype)
                N.check(Lb.dic_lstm_unpack_grads(None, 0, None, N.ptr(dbias), H, I, gp, int(accumulate), st), 'dic_lstm_unpack_grads')
            elif x3 or not f32:
                # dic_gemm_tn: dW_ih[d] = dG[d]^T . x and dW_hh[d] = dG[d]^T . h_prev[d] straight into the parameter gradients; h_prev = row-shifted
                # views of the extended output buffer (see below)
                oe = out_ext.view((R + 2) * B, 2 * H)
                xv = xb.view(R * B, Ip)
                for d in range(2):              # (both products of a direction from ONE pass over its gate gradients)
                    hp = oe[:R * B, :H] if d == 0 else oe[2 * B:, H:]
                    _ops.gemm_tn_into(dg2[..., 4 * H * d:4 * H * (d + 1)], xv, sinks[4 * d], kcols=I, accumulate=accumulate, x2=hp, dst2=sinks[4 * d + 1],
                                      relu_x=ctx.x_relu_in_kernel)
                N.check(Lb.dic_lstm_unpack_grads(None, 0, None, N.ptr(dbias), H, I, gp, int(accumulate), st), 'dic_lstm_unpack_grads')
            else:
                # dW = dG^T.X has K = R*B (hundreds of thousands) and a tiny output: split-K bmm (ops.splitk_tn).
                # dW_hh[d] = sum_t dG_t[d]^T h_prev_t[d], h_prev = h_{t-1} (forward) / h_{t+1} (reverse): row-shifted views of the
                # extended output buffer whose first / last time slot holds h0 (zeros without one; filled by the forward) -- no H_prev copy, no separate
                # products for the boundary step, none of the cross-direction blocks a single (8H, 2H) product would compute
                oe = out_ext.view((R + 2) * B, 2 * H)
                dw_ih = splitk_tn(dg2, xb.view(R * B, Ip))           # (8H, Ip) f32
                dw_hh = torch.empty((2, 4 * H, H), device=dev, dtype=torch.float32)
                dw_hh[0] = splitk_tn(dg2[:, :4 * H], oe[:R * B, :H])
                dw_hh[1] = splitk_tn(dg2[:, 4 * H:], oe[2 * B:, H:])
                N.check(Lb.dic_lstm_unpack_grads(N.ptr(dw_ih), Ip, N.ptr(dw_hh), N.ptr(dbias), H, I, gp, int(accumulate), st), 'dic_lstm_unpack_grads')
            if not accumulate:
                grads = [g if n else None for g, n in zip(sinks, needs)]
        return (dx, (dh0 if ctx.has_init else None), (dc0 if ctx.has_init else None), None, None, None, None, None, *grads)


def _params(lstm):
    return [getattr(lstm, n) for n in PARAM_NAMES]


def deferred_relu_ok(x_or_batch):
    """The F.relu between encoder and decoder (clustering_interp.py:38-41) can be left to the decoder's kernels -- its input projection and
    its weight-gradient kernel rectify the raw encoder output on load -- on the large-batch bf16 path: no rectified copy is written."""
    B = x_or_batch if isinstance(x_or_batch, int) else x_or_batch.shape[1]
    if not torch.is_autocast_enabled():          # f32 step: only the x3 products rectify on load (dic_x3_row_proj, dic_gemm_tn)
        return DEFER_RELU and RELU_IN_KERNEL and _ops.f32_products() == 'x3' and _ops.X3_ROW_PROJ
    return DEFER_RELU and ROW_PROJ and RELU_IN_KERNEL and B > SMALL_BATCH and torch.get_autocast_dtype('cuda') == torch.bfloat16


def bilstm(x, lstm, h0=None, c0=None, batch_major_state=False, rectified_out=False, input_rectify=False):
    """(out (R,B,2H), (h_n, c_n) (2,B,H) f32) = bidirectional LSTM of x (R,B,I) with ``lstm``'s parameters: bf16 operands under
    autocast(bf16) (out is bf16), exact f32 otherwise (x f32, no autocast; out is f32).
    ``batch_major_state``: h0, c0, h_n, c_n are (B,2,H) instead -- h_n.view(B, 2H) is then the concatenated latent
    [h_fwd | h_rev] of clustering_interp.py:139, and feeds the next LSTM as it lies.
    ``rectified_out``: out is relu(out) (what the decoder feeds on); the ReLU's backward is applied inside the recurrence kernel.
    ``rectified_out='deferred'``: out stays raw, the consumer promises to rectify it on load (``input_rectify=True`` of the next
    bilstm) and to hand back the gradient w.r.t. relu(out); the ReLU's backward is still applied here.
    ``input_rectify``: the LSTM runs on relu(x); x's gradient is returned WITHOUT the ReLU mask (its producer applies it: 'deferred')."""
    f32 = x.dtype == torch.float32 and not torch.is_autocast_enabled()
    with torch.autocast('cuda', enabled=False):
        out, hn, cn = _BiLstm.apply(x, h0, c0, False, batch_major_state, f32, _relu_mode(rectified_out), bool(input_rectify), *_params(lstm))
    return out, (hn, cn)


def _relu_mode(rectified_out):
    return 2 if rectified_out == 'deferred' else int(bool(rectified_out))


def bilstm_packed(xenc, lstm, h0=None, c0=None, batch_major_state=False, rectified_out=False):
    """The same for an input already in the recurrence kernel's layout: xenc (R,B,32) bf16 rows [features | 1 | 0...]
    (``ops.sci_cci_packed``; 64-wide rows for 32 <= 3C < 64); its gradient comes back in that layout too."""
    with torch.autocast('cuda', enabled=False):
        out, hn, cn = _BiLstm.apply(xenc, h0, c0, True, batch_major_state, False, _relu_mode(rectified_out), False, *_params(lstm))
    return out, (hn, cn)

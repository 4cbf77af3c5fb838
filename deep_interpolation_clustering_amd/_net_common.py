"""Shared body of the two ``Net`` variants (clustering_interp.py:14-247, pretrain_interp.py:14-215).

Layer inventory and ``state_dict`` keys are upstream's: ``sci.kernel``, ``cci.kernel``,
``encoder.lstm.*``, ``decoder.lstm.*``, ``rbf.kernel``, ``rbf.compress_fc.module.model.{0,1,4}.*``,
optional ``predict_future / aux_head / fake_det_head`` ``.model.*`` and (clustering only)
``cluster_assignment.cluster_centers`` -- checkpoints are interchangeable with the reference's.

What runs where: interpolation (k1), de-interpolation (k2) with the reconstruction loss, DEC soft assignment, target and KL (k3), the
bi-LSTM recurrences, their input projections and weight gradients, and CompressFC / the heads' BatchNorm tails are HIP kernels
(csrc/); what is left to the libraries is the decoder LSTM's input-gradient GEMM, the heads' small first layers, and -- off the GPU,
or for LSTM shapes the kernels are not compiled for -- torch.nn.LSTM itself.  Differences from upstream that do not change results: at small batch sizes the fake / positive
branches share ONE encoder call with the real batch (rows are independent in sci, cci and the LSTM),
and BatchNorm uses global-batch moments when the batch is sharded over ranks.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dist, ops
from . import lstm as fused_lstm
from .dec import ClusterAssignment, target_distribution
from .interpolation_layer import CrossChannelInterp, SingleChannelInterp, fused_forward
from .ragged import is_ragged
from .rbf import RBF, basis_func_dict
from .utils import logger


SEPARATE_ENCODER_ROWS = 65536      # (batch x grid points) from which the fake / positive branches get their own encoder call


class EncoderRNN(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, dropout, bidirectional, device):
        super().__init__()
        self.device, self.hidden_size, self.num_layers = device, hidden_size, num_layers
        self.num_directions = 2 if bidirectional else 1
        self.lstm = nn.LSTM(input_size, hidden_size, num_layers=num_layers, dropout=dropout, bidirectional=bidirectional)

    def forward(self, x, packed=False, batch_major_state=False, rectified_out=False):
        """``batch_major_state`` (fused path only): hidden / cell_state come back as (B,2,H) -- see lstm.bilstm.
        ``rectified_out`` (fused path only): output is relu(output), which is all the decoder reads of it (DecoderRNN.forward);
        ``'deferred'``: the output stays raw and the decoder's kernels rectify it on load (lstm.deferred_relu_ok), the ReLU's backward
        is still applied by the encoder's recurrence kernel."""
        if packed:                                           # (R,B,32) bf16 rows straight from ops.sci_cci_packed
            output, (hidden, cell_state) = fused_lstm.bilstm_packed(x, self.lstm, batch_major_state=batch_major_state, rectified_out=rectified_out)
        elif fused_lstm.fused_available(x, self.lstm) or fused_lstm.f32_available(x, self.lstm):      # on the GPU: persistent HIP recurrence
            output, (hidden, cell_state) = fused_lstm.bilstm(x, self.lstm, batch_major_state=batch_major_state, rectified_out=rectified_out)
        else:
            output, (hidden, cell_state) = self.lstm(x)
        return output, hidden, cell_state


class DecoderRNN(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, dropout, bidirectional, device):
        super().__init__()
        self.device, self.hidden_size = device, hidden_size
        self.lstm = nn.LSTM(input_size, hidden_size, num_layers=num_layers, dropout=dropout, bidirectional=bidirectional)

    def forward(self, x, hidden, context, batch_major_state=False, rectified=False):
        defer = rectified == 'deferred'                                         # raw encoder output, rectified on load by the decoder's kernels
        if not rectified:                                                       # (the fused encoder hands its output over rectified)
            x = F.relu(x)                                                       # clustering_interp.py:38-41
        if fused_lstm.fused_available(x, self.lstm) or fused_lstm.f32_available(x, self.lstm):
            x, (hidden, cell_state) = fused_lstm.bilstm(x, self.lstm, hidden, context, batch_major_state=batch_major_state, input_rectify=defer)
        else:
            if defer:
                x = F.relu(x)
            x, (hidden, cell_state) = self.lstm(x, (hidden, context))
        return x, (hidden, cell_state)


class _Head(nn.Module):
    """Linear(256,128) -> BN -> Dropout -> Linear(128,odim) [-> tail]; clustering_interp.py:43-87."""

    def __init__(self, idim, odim, dropout, tail=None):
        super().__init__()
        layers = [nn.Linear(idim, 128), nn.BatchNorm1d(128), nn.Dropout(dropout), nn.Linear(128, odim)]
        if tail is not None:
            layers.append(tail)
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        first, bn, drop, last = self.model[0], self.model[1], self.model[2], self.model[3]
        fast = (x.is_cuda and x.dim() == 2 and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16
                and last.in_features == ops.HEAD_IN and last.out_features in ops.BNHEAD_OUT)
        if not fast:
            return self.model(x)
        # bf16 step: BatchNorm -> Linear(128, <= 8) as the streaming kernels of csrc/dic_bnhead.hip (no ReLU in these heads);
        # the library picks a 256 x 16 macro-tile for the 2-wide output layer otherwise (0.33 ms for 65 536 rows)
        with torch.autocast('cuda', enabled=False):
            z = ops.rows_linear(x, first.weight, first.bias, bias_grad_is_zero=bn.training)
            out = ops.bn_relu_head(z, bn, last, relu=False, dropout=drop)
        for tail in self.model[4:]:
            out = tail(out)
        return out


class AuxFc(_Head):
    def __init__(self, idim, odim, dropout):
        super().__init__(idim, odim, dropout)


class FuturePredFc(_Head):
    def __init__(self, idim, odim, dropout):
        super().__init__(idim, odim, dropout, nn.Sigmoid())


class FakeDetFc(_Head):
    def __init__(self, idim, odim, dropout):
        super().__init__(idim, odim, dropout, nn.LogSoftmax(dim=1))


class NetBase(nn.Module):
    clustering = False
    internal_step = False      # set by step.Stepper around its own forward + loss + backward (see forward / rec_loss)
    on_decoder_side_grads = None      # callable (set by step.Stepper when sharded): every gradient behind the encoder output is complete
    rec_target = None                 # (set by step.Stepper around its own step, prefix lengths only) the observations: k2 then also returns the
    _fused_rec = None                 #   reconstruction loss (ops.rbf_rec_loss); (reconstruction, mse) waits here for rec_loss

    def __init__(self, args, device):
        super().__init__()
        self.device, self.args = device, args
        self.num_variables, self.num_timestamps = args.num_variables, args.num_timestamps
        self.nhidden, self.nlstm, self.bidirectional = 128, 1, True
        self.dim_enc_hidden = self.nlstm * self.nhidden * (2 if self.bidirectional else 1)
        self.dim_dec_out = self.nhidden * (2 if self.bidirectional else 1)
        C = self.num_variables
        self.sci = SingleChannelInterp(args.ref_points, args.hours_from_admission, C, self.num_timestamps, device)
        self.cci = CrossChannelInterp(C, self.num_timestamps, device)
        self.encoder = EncoderRNN(3 * C, self.nhidden, self.nlstm, 0, self.bidirectional, device)
        self.decoder = DecoderRNN(self.nhidden * 2, self.nhidden, self.nlstm, 0, self.bidirectional, device)
        self.rbf = RBF(hours_look_ahead=args.hours_from_admission, ref_points=args.ref_points, in_dim=self.dim_dec_out,
                       out_dim=C, dropout=args.dropout, basis_func=basis_func_dict()['gaussian'], device=device)
        n_aux = len(args.aux_tasks)
        if 'future_vital' in args.aux_tasks:
            self.predict_future = FuturePredFc(self.dim_enc_hidden, C, args.dropout)
            n_aux -= 1
        if n_aux > 0:
            self.aux_head = AuxFc(self.dim_enc_hidden, n_aux, args.dropout)
        if args.fake_detection:
            self.fake_det_head = FakeDetFc(self.dim_enc_hidden, 2, args.dropout)
        if self.clustering:
            self.cluster_assignment = ClusterAssignment(args.cluster_number, self.dim_enc_hidden, 1.0)
        dist.convert_batchnorm_(self)

    # ------------------------------------------------------------------------------ forward
    @staticmethod
    def _latent(hidden, batch_major):
        """z = cat(h_fwd, h_rev) (clustering_interp.py:139): a view when the states are batch-major."""
        return hidden.reshape(hidden.size(0), -1) if batch_major else torch.cat([h for h in hidden], dim=-1)

    def _packed_path(self, x):
        """bf16 step on the GPU with 3C < 64 features: the interpolation kernel writes the encoder LSTM's input rows itself."""
        return ops.packed_width(3 * self.num_variables) > 0 and fused_lstm.fused_available(x, self.encoder.lstm)

    def _interp(self, x, lengths=None, packed=False):
        """Time-major interpolated features: (R,B,3C) f32 (a permuted view of upstream's (B,R,3C)), or the packed (R,B,32) bf16 rows."""
        if packed:
            return ops.sci_cci_packed(x, self.sci.kernel, self.cci.kernel, self.sci.grid(), lengths)
        return fused_forward(self.sci, self.cci, x, lengths).permute(1, 0, 2)     # one launch

    def forward(self, x, fake_x=None, fake_perm_idx=None, positive_x=None, lengths=None):
        """x (B,4C,T) -> (cat_hidden (B,256), rec (B,C,T), aux_pred_dict); clustering_interp.py:134-189.
        ``lengths`` (B,C) int32 is an optional side channel: prefix lengths of the padding mask.
        ``x`` may be a ``ragged.RaggedBatch``: the interpolation / de-interpolation kernels then read the encounters' observed samples
        in place from the device-resident ragged store (no padded (B,4C,T) tensor exists on that path)."""
        args = self.args
        B = x.size(0)
        if is_ragged(x):
            lengths = x.lengths
        packed = self._packed_path(x)
        feats = [self._interp(x, lengths, packed)]
        want_fake = bool(args.fake_detection)
        want_pos = self.clustering and args.triple_margin != 0. and want_fake
        if want_fake:
            feats.append(self._interp(fake_x, lengths, packed))
        if want_pos:
            feats.append(self._interp(positive_x, lengths, packed))
        # fused recurrence (bf16 step): final states in batch-major (B,2,H) layout -- the latent z = [h_fwd | h_rev] is then a VIEW of
        # h_n, and the decoder takes (h_n, c_n) as they lie: no cat / slice / copy kernels between encoder, DEC head and decoder
        bm = fused_lstm.fused_available(feats[0], self.encoder.lstm) or fused_lstm.f32_available(feats[0], self.encoder.lstm)
        # the F.relu between encoder and decoder: large bf16 batches leave it to the decoder's kernels (no rectified copy), else the encoder
        # hands its output over rectified
        rect = ('deferred' if fused_lstm.deferred_relu_ok(B) else True) if bm else False
        if len(feats) > 1 and B * feats[0].size(0) >= SEPARATE_ENCODER_ROWS:
            # large batches: one encoder call per branch.  Each already fills the chip, and the shared call would cost three
            # 100-MB-class copies (stacking the inputs, slicing the real half of the context and of the final states back out)
            outs = [self.encoder(f, packed, bm, rect if i == 0 else False) for i, f in enumerate(feats)]      # (only the real branch's output feeds the decoder)
            context, hidden, cell = outs[0]
            z_all = torch.cat([self._latent(o[1], bm) for o in outs], dim=0)                     # (nB, 256)
        else:
            seq = feats[0] if len(feats) == 1 else torch.cat(feats, dim=1)                        # (R, nB, .)
            context, hidden, cell = self.encoder(seq, packed, bm, rect)
            z_all = self._latent(hidden, bm)                                                      # (nB, 256)
            if len(feats) > 1:
                context = context[:, :B]
                hidden, cell = (hidden[:B], cell[:B]) if bm else (hidden[:, :B].contiguous(), cell[:, :B].contiguous())
        cat_hidden = z_all if z_all.size(0) == B else z_all[:B]        # (no slice node -- and no zero-filled slice_backward -- without extra branches)
        q = colsum = None
        if self.clustering:
            # the soft assignment needs the latent only: computed here, its column sums f_j (a batch-level statistic: one exchange when the batch
            # is sharded over ranks) ride on the next small all-reduce -- CompressFC's BatchNorm moments -- instead of paying their own
            q = self.cluster_assignment(cat_hidden)
            _, colsum = self.cluster_assignment.last_colsum
            if dist.is_sharded():
                colsum = colsum.clone()
                colsum._dic_deferred_sum = True
                dist.deferred_sum_(colsum)
        if self.on_decoder_side_grads is not None and context.requires_grad:
            cb = self.on_decoder_side_grads
            context.register_hook(lambda g: cb())           # fires when the backward has passed the decoder, its head and the latent heads
        y, _ = self.decoder(context, hidden, cell, bm, rect) if bm else self.decoder(context, hidden, cell)
        # (B,C,T).  Inside step.Stepper's optimisation step (`internal_step`: the reconstruction is consumed by rec_loss alone and never
        # handed out) only the observed slots are materialised; every other caller gets zeros in the padding, as upstream's `* mask`
        if self.internal_step and self.rec_target is not None and lengths is not None:
            y, mse = self.rbf(y.permute(1, 2, 0), x, lengths, rec_target=self.rec_target)
            self._fused_rec = (y, mse)
        else:
            y = self.rbf(y.permute(1, 2, 0), x, lengths, prefix_only=self.internal_step)

        aux = dict()
        n_aux = len(args.aux_tasks)
        if 'future_vital' in args.aux_tasks:
            aux['future_vital'] = self.predict_future(cat_hidden)
            n_aux -= 1
        if n_aux > 0:
            pred = self.aux_head(cat_hidden)
            for i, task in enumerate(t for t in args.aux_tasks.keys() if t != 'future_vital'):
                aux[task] = pred[:, i]
        if want_fake:
            fake_hidden = z_all[B:2 * B]
            aux['fake_det'] = self.fake_det_head(torch.cat([cat_hidden, fake_hidden], dim=0)[fake_perm_idx])
        if want_pos:
            aux['positive'] = z_all[2 * B:3 * B]
            aux['negative'] = z_all[B:2 * B]
        if self.clustering:
            aux['cluster_pred'] = q
            aux['cluster_label'] = target_distribution(q, colsum).detach()
        return cat_hidden, y, aux

    # ------------------------------------------------------------------------------ losses
    def rec_loss(self, org_ob, rec_ob, padding_mask, lengths=None):
        """Masked SSE / #observed over the global batch (clustering_interp.py:197-203), one HIP reduction."""
        fused, self._fused_rec = self._fused_rec, None
        if fused is not None and fused[0] is rec_ob and padding_mask is None and org_ob is self.rec_target:
            return {'loss': fused[1], 'ae_mse': fused[1]}          # came out of the de-interpolation kernel (forward)
        if is_ragged(org_ob):                                      # (evaluation passes: the observations as a padded tensor after all)
            lengths, org_ob = org_ob.lengths, org_ob.ob_dense()
        mse = ops.masked_mse(org_ob, rec_ob, padding_mask, lengths, prefix_only=self.internal_step and padding_mask is None)
        return {'loss': mse, 'ae_mse': mse}

    def sup_aux_loss(self, aux_tasks, aux_label_dict, aux_pred_dict, future_vital_mask=None):
        out = dict()
        if 'future_vital' in aux_tasks:
            m = future_vital_mask
            sse = F.mse_loss(aux_pred_dict['future_vital'] * m, aux_label_dict['future_vital'] * m, reduction='sum')
            cnt = (m == 1.0).sum().to(sse.dtype)
            out['future_vital'] = sse / dist.all_reduce_sum_(cnt.clone())
        for task in aux_tasks:
            if task == 'future_vital':
                continue
            pw = torch.tensor(self.args.aux_pos_weights[task]).to(self.device)
            pred = aux_pred_dict[task]
            out[task] = dist.global_mean(F.binary_cross_entropy_with_logits(pred, aux_label_dict[task], pos_weight=pw, reduction='sum'),
                                         pred.numel())             # mean over the GLOBAL batch (ranks' shards may differ by a row)
        return out

    def fake_det_loss(self, label, pred):
        return {'fake_detection': dist.global_mean(F.nll_loss(pred, label, reduction='sum'), label.numel())}

    def multi_task_loss(self, aux_tasks, rec_loss_dict, aux_loss_dict):
        """ae_mse + sum_k w_k * loss_k (clustering_interp.py:239-247)."""
        loss = rec_loss_dict['ae_mse']
        for name, value in aux_loss_dict.items():
            loss = loss + aux_tasks[name] * value
            # (upstream logs the value here, clustering_interp.py:243: formatting a device tensor is a host sync per term)
        rec_loss_dict['loss'] = loss
        rec_loss_dict.update(aux_loss_dict)
        return rec_loss_dict

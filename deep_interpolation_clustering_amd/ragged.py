"""The ragged, device-resident encounter store and the batch handle the kernels read it through.

The reference pads every (encounter, channel) row to the cohort's longest one (p0_data_process.py:35-70) and stacks four such planes
into ``feed_data (N,4C,T)`` (dataloader.py:54-79): at ~50 observations in 96 slots half of every row is padding, and a batch is a
gathered copy of it.  ``RaggedStore`` keeps, per row, the observed samples only -- packed time stamps, packed (rescaled) values and
packed hold-out flags behind one ``row_off`` table -- and a batch is just ``RaggedBatch(store, idx)``: the interpolation and
de-interpolation kernels (``dic_sci_cci_fwd_store``, ``dic_rbf_fwd_store``, ``dic_rbf_bwd_store``) read the rows of a shuffled batch IN
PLACE through ``idx``; no gather pass and no padded ``(B,4C,T)`` tensor exist on that path.  The dense planes can be rebuilt exactly
(``dense_rows``) for the consumers that want them (feature dumps, the corrupted / augmented copies, the module-level API).
"""
from __future__ import annotations

import numpy as np
import torch

PAD_TAIL = 128      # readable elements behind the last row (the kernels' clamped look-ahead reads stay in bounds)


class RaggedStore:
    def __init__(self, feed_data, C, device):
        """``feed_data`` (N,4C,T) as DataSet builds it (values already rescaled); rows must be PREFIX-masked with constant padding
        values per plane and channel (what p0 writes) -- ``RaggedStore.fits`` tells."""
        fd = np.asarray(feed_data)
        N, C4, T = fd.shape
        if C4 != 4 * C:
            raise ValueError(f"stacked planes must be (N, 4*{C}, T), got {tuple(fd.shape)}")
        mask = fd[:, C:2 * C] != 0
        lengths = mask.sum(-1).astype(np.int32)                       # (N,C)
        row_off = np.zeros(N * C + 1, np.int64)
        np.cumsum(lengths.reshape(-1), out=row_off[1:])
        total = int(row_off[-1])
        # are the time stamps of every row non-decreasing?  (p0 writes them in charting order; the synthetic cohorts sort them.)  One host pass
        # here lets k1 find each grid point's nearest sample by bisection instead of a pass over the row (dic_sci_cci_fwd_store(times_sorted=1))
        tpl = fd[:, 2 * C:3 * C]
        self.times_sorted = bool(np.all((np.diff(tpl, axis=-1) >= 0) | ~mask[..., 1:]))

        def pack(plane, dtype):
            out = np.zeros(total + PAD_TAIL, dtype)
            out[:total] = plane[mask]                                  # row-major (n, c, slot) order = the rows back to back
            return torch.as_tensor(out, device=device)
        self.N, self.C, self.T = N, C, T
        self.t_pk = pack(fd[:, 2 * C:3 * C], np.float32)
        self.v_pk = pack(fd[:, 0:C] * mask, np.float32)                # the trainers' `ob *= padding_mask` (a no-op on observed slots)
        self.hold_pk = pack(fd[:, 3 * C:4 * C], np.uint8)
        self.row_off = torch.as_tensor(row_off, device=device)
        self.lengths = torch.as_tensor(lengths, device=device)         # (N,C) int32
        # what the padding of each plane holds (p0 leaves 0 everywhere, then min-max normalises the whole value array: a constant per
        # channel): kept so that dense_rows() reproduces feed_data bit for bit
        pad = ~mask
        self.pad_value = torch.as_tensor(np.array([fd[:, c][pad[:, c]][0] if pad[:, c].any() else 0.0 for c in range(C)], np.float32), device=device)
        self.has_pad = torch.as_tensor(np.array([bool(pad[:, c].any()) for c in range(C)]), device=device)      # (a channel with no padded slot SAYS nothing about the constant: concat)
        self.device = device

    @classmethod
    def from_device(cls, x, C):
        """The same store from a stacked ``(N,4C,T)`` tensor that already lives on the device (prefix masks; padding of the value plane
        constant per channel): packed with device ops, nothing crosses PCIe.  Cohorts too large to pad on the host (BASELINE configs[3]:
        300 000 x 48 x 288 f32 = 16.6 GB) are built chunk by chunk this way and joined with ``concat``."""
        N, C4, T = x.shape
        if C4 != 4 * C:
            raise ValueError(f"stacked planes must be (N, 4*{C}, T), got {tuple(x.shape)}")
        self = object.__new__(cls)
        dev = x.device
        mask = x[:, C:2 * C] != 0
        lengths = mask.sum(-1).to(torch.int32)
        # the host path's preconditions (fits()), checked here too: a mask that is not a prefix would be packed silently and wrongly
        if not bool((mask == (torch.arange(T, device=dev)[None, None] < lengths[..., None])).all()):
            raise ValueError('RaggedStore.from_device: the masks are not prefix masks')
        row_off = torch.zeros(N * C + 1, dtype=torch.int64, device=dev)
        torch.cumsum(lengths.reshape(-1), 0, out=row_off[1:])
        total = int(row_off[-1])
        tpl = x[:, 2 * C:3 * C]
        self.times_sorted = bool((((tpl[..., 1:] - tpl[..., :-1]) >= 0) | ~mask[..., 1:]).all())

        def pack(plane, dtype):
            out = torch.zeros(total + PAD_TAIL, dtype=dtype, device=dev)
            out[:total] = plane[mask].to(dtype)
            return out
        self.N, self.C, self.T = N, C, T
        self.t_pk = pack(tpl, torch.float32)
        self.v_pk = pack(x[:, 0:C] * mask, torch.float32)
        self.hold_pk = pack(x[:, 3 * C:4 * C], torch.uint8)
        self.row_off, self.lengths = row_off, lengths
        pad = ~mask
        pv = torch.zeros(C, dtype=torch.float32, device=dev)
        has = torch.zeros(C, dtype=torch.bool, device=dev)
        for c in range(C):
            if bool(pad[:, c].any()):
                vals = x[:, c][pad[:, c]]
                if not bool((vals == vals[0]).all()):
                    raise ValueError(f'RaggedStore.from_device: the padding of channel {c} is not one constant')
                pv[c], has[c] = vals[0], True
        self.pad_value, self.has_pad = pv, has
        self.device = dev
        return self

    @classmethod
    def concat(cls, stores):
        """One store holding the encounters of ``stores`` back to back (same C, T, device)."""
        s0 = stores[0]
        if any(s.C != s0.C or s.T != s0.T or torch.device(s.device) != torch.device(s0.device) for s in stores):
            raise ValueError('RaggedStore.concat: the stores differ in C, T or device')
        # the padding constant of a channel is whatever the stores that HAVE padded slots in it agree on (a chunk whose rows are all full-length
        # in some channel knows nothing about it and must not veto the others)
        pad_value, has_pad = s0.pad_value.clone(), s0.has_pad.clone()
        for s in stores[1:]:
            both = has_pad & s.has_pad
            if bool((pad_value[both] != s.pad_value[both]).any()):
                raise ValueError('RaggedStore.concat: the stores differ in their padding constants')
            new = s.has_pad & ~has_pad
            pad_value[new] = s.pad_value[new]
            has_pad |= s.has_pad
        self = object.__new__(cls)
        self.N, self.C, self.T, self.device = sum(s.N for s in stores), s0.C, s0.T, s0.device
        self.times_sorted = all(s.times_sorted for s in stores)
        tail = torch.zeros(PAD_TAIL, device=s0.t_pk.device)

        def join(name):
            parts = [getattr(s, name)[:getattr(s, name).numel() - PAD_TAIL] for s in stores]
            return torch.cat(parts + [tail.to(parts[0].dtype)])
        self.t_pk, self.v_pk, self.hold_pk = join('t_pk'), join('v_pk'), join('hold_pk')
        offs, base = [], 0
        for s in stores:
            offs.append(s.row_off[:-1] + base)
            base += int(s.row_off[-1])
        self.row_off = torch.cat(offs + [torch.tensor([base], dtype=torch.int64, device=s0.row_off.device)])
        self.lengths = torch.cat([s.lengths for s in stores])
        self.pad_value, self.has_pad = pad_value, has_pad
        return self

    @staticmethod
    def fits(feed_data, C):
        """True when ``feed_data`` can be stored ragged without loss: prefix masks, binary mask values, zero time / hold-out and one
        constant value per channel in the padding."""
        fd = np.asarray(feed_data)
        T = fd.shape[-1]
        m = fd[:, C:2 * C]
        if not np.isin(m, (0, 1)).all():
            return False
        n = m.sum(-1)
        if not (m == (np.arange(T)[None, None] < n[..., None])).all():
            return False
        pad = m == 0
        if (fd[:, 2 * C:3 * C][pad] != 0).any() or (fd[:, 3 * C:4 * C][pad] != 0).any():
            return False
        hold = fd[:, 3 * C:4 * C]
        if not np.isin(hold, (0, 1)).all():
            return False
        for c in range(C):
            pv = fd[:, c][pad[:, c]]
            if pv.size and (pv != pv[0]).any():
                return False
        return True

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in (self.t_pk, self.v_pk, self.hold_pk, self.row_off, self.lengths))

    def dense_rows(self, idx, masked_values=False):
        """feed_data[idx] rebuilt from the packed rows: (b,4C,T) f32 planes [value | mask | time | hold-out].  ``masked_values``: the
        value plane times the mask (0 in the padding) instead of the stored padding constant."""
        C, T = self.C, self.T
        idx = idx.to(self.device, torch.int64)
        lens = self.lengths.index_select(0, idx)                                    # (b,C)
        off = self.row_off[:-1].view(self.N, C).index_select(0, idx)                # (b,C)
        ar = torch.arange(T, device=self.device)
        m = ar < lens[..., None]                                                    # (b,C,T)
        pos = (off[..., None] + ar)[m]                                              # packed positions, rows back to back
        out = torch.zeros((idx.numel(), 4 * C, T), dtype=torch.float32, device=self.device)
        val = out[:, 0:C]
        if not masked_values:
            val += self.pad_value[None, :, None]
        val[m] = self.v_pk[pos]
        out[:, C:2 * C] = m.to(torch.float32)
        out[:, 2 * C:3 * C][m] = self.t_pk[pos]
        out[:, 3 * C:4 * C][m] = self.hold_pk[pos].to(torch.float32)
        return out


class RaggedBatch:
    """``B`` encounters of a ``RaggedStore`` named by ``idx`` -- what the model's forward takes in place of the stacked ``(B,4C,T)``
    tensor (it quacks like one where the callers look: ``size / shape / device / is_cuda / dtype``), and what ``rec_loss`` takes in
    place of the ``(B,C,T)`` observations.  ``denoise``: the model input is value x hold-out flag (pretrain_trainer.py:139-141)."""
    is_ragged = True

    def __init__(self, store: RaggedStore, idx, lengths=None, denoise=False):
        self.store = store
        self.idx = idx if (idx.dtype == torch.int32 and idx.is_contiguous()) else idx.to(torch.int32).contiguous()
        self.lengths = lengths if lengths is not None else store.lengths.index_select(0, self.idx.to(torch.int64))
        if self.lengths.dtype != torch.int32 or not self.lengths.is_contiguous():
            self.lengths = self.lengths.to(torch.int32).contiguous()
        self.denoise = bool(denoise)

    # ---- the corner of the tensor interface the model code touches
    @property
    def shape(self):
        return torch.Size((self.idx.numel(), 4 * self.store.C, self.store.T))

    def size(self, k=None):
        return self.shape if k is None else self.shape[k]

    @property
    def device(self):
        return self.store.device

    @property
    def is_cuda(self):
        return torch.device(self.store.device).type == 'cuda'

    dtype = torch.float32

    # ---- Stepper's hipGraph path keeps static copies of its inputs
    def clone(self):
        return RaggedBatch(self.store, self.idx.clone(), self.lengths.clone(), self.denoise)

    def copy_(self, other, non_blocking=False):
        if other.store is not self.store or other.denoise != self.denoise:      # (a captured step replays against ONE store / denoise setting)
            raise ValueError('RaggedBatch.copy_: the source batch belongs to another store or has another denoise setting')
        self.idx.copy_(other.idx, non_blocking=non_blocking)
        self.lengths.copy_(other.lengths, non_blocking=non_blocking)
        return self

    def with_denoise(self, denoise):
        return self if bool(denoise) == self.denoise else RaggedBatch(self.store, self.idx, self.lengths, denoise)

    # ---- dense views (rebuilt on demand)
    def dense(self):
        """The stacked input the trainers build (pretrain_trainer.py:132-143): [ob*mask (x hold-out when denoising) | mask | time | hold-out]."""
        x = self.store.dense_rows(self.idx, masked_values=True)
        if self.denoise:
            C = self.store.C
            x[:, 0:C] *= x[:, 3 * C:4 * C]
        return x

    def ob_dense(self):
        C = self.store.C
        return self.store.dense_rows(self.idx, masked_values=True)[:, 0:C].contiguous()


def is_ragged(x):
    return getattr(x, 'is_ragged', False)

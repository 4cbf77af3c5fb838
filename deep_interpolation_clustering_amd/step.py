"""The joint training step shared by both trainers and ``bench.py``.

One step = what pretrain_trainer.py:189-229 / clustering_trainer.py:222-279 do per batch:
zero_grad -> Net.forward -> loss switch on ``args.loss`` -> backward -> [gradient all-reduce over
RCCL when sharded] -> clip_grad_norm_(grad_clip) -> optimizer.step.  Loss terms stay on the device
(the reference's per-term ``.item()`` is a host sync per batch); callers read them when they log.
"""
import contextlib
import os
import sys

import torch

from . import dist, ops
from . import lstm as fused_lstm
from .ragged import is_ragged

LOSS_NAMES = ('ae_mse', 'ae_mse_sup', 'ae_mse_fake_detect', 'ae_mse_fake_detect_triplet', 'ae_mse_sup_fake_detect',
              'ae_mse_kl', 'ae_mse_fake_detect_kl', 'ae_mse_sup_kl', 'ae_mse_sup_fake_detect_kl')


def compute_losses(model, args, hidden, rec_ob, aux_pred, ob, padding_mask, lengths=None, aux_label_dict=None,
                   future_vital_mask=None, fake_det_label=None):
    """The loss switch of the trainers (pretrain_trainer.py:196-221, clustering_trainer.py:227-272).  Sharded: batch statistics queued to ride
    on a later exchange (the fused reconstruction SSE + count: ops.rbf_rec_loss) have landed BEFORE any term is combined or handed out --
    whether or not one of the terms happened to carry them.  (A switch that raises resolves nothing: a collective issued while one rank
    unwinds would meet no partner; Stepper drops the riders.)"""
    rec = model.rec_loss(ob, rec_ob, padding_mask, lengths)
    name = args.loss
    if name not in LOSS_NAMES:
        raise NotImplementedError(name)
    tasks, terms = {}, {}
    if '_sup' in name:
        tasks.update(args.aux_tasks)
        terms.update(model.sup_aux_loss(args.aux_tasks, aux_label_dict or {}, aux_pred, future_vital_mask))
    if 'fake_detect' in name:
        tasks.update(args.unsup_aux_tasks)
        terms.update(model.fake_det_loss(fake_det_label, aux_pred['fake_det']))
    if name.endswith('triplet'):
        terms.update(model.triplet_loss(hidden, aux_pred['positive'], aux_pred['negative'], args.triple_margin))
    if name.endswith('_kl'):
        tasks.update(args.unsup_aux_tasks)
        terms.update(model.kl_loss(aux_pred['cluster_label'], aux_pred['cluster_pred']))
    dist.resolve_all_()
    if name == 'ae_mse':
        return rec
    return model.multi_task_loss(tasks, rec, terms)


class Stepper:
    """Owns the flat parameter/gradient buckets and runs one optimisation step.

    ``autocast_dtype`` (e.g. torch.bfloat16) wraps the forward in torch.autocast so the bi-LSTMs and the
    FC heads run on bf16 MFMA; the HIP kernels always compute in f32."""

    def __init__(self, model, optimizer_factory, args, autocast_dtype=None, use_graphs=False, precision=None):
        """``precision`` (f32 step only, i.e. ``autocast_dtype=None``): how the dense products are formed -- 'exact' (exact-f32 MFMA recurrence +
        f32 library GEMMs) or 'x3' (every product as a three-term bf16 split on the matrix cores with f32 accumulation, all tensors f32:
        csrc/dic_gemm.hip; no library GEMM).  None = the process default (ops.f32_products(), env DIC_F32_PRODUCTS, 'exact')."""
        self.model, self.args = model, args
        if precision is not None and autocast_dtype is not None:
            raise ValueError('precision= selects the f32 step\'s products; it does not combine with autocast_dtype')
        self.precision = precision
        self.flat = dist.FlatParams(model)          # must precede the optimizer: it re-homes parameter storage
        self.flat.broadcast_(0)
        self.optimizer = optimizer_factory(model)
        self._fused_tail = hasattr(self.optimizer, 'bind')       # flat_adam.FlatAdam: clip scale + Adam in one kernel
        if self._fused_tail:
            self.optimizer.bind(self.flat)
        self.autocast_dtype = autocast_dtype
        if dist.is_sharded() and hasattr(model, 'decoder'):       # overlap the larger piece of the gradient all-reduce with the encoder backward
            self.flat.set_split(next(model.decoder.parameters()))
            model.on_decoder_side_grads = self.flat.begin_tail_reduce
        # hipGraph capture of the whole step (single-GPU): at the reference's batch size (256) the ~250 launches of a
        # step are launch-bound (2.5 ms); one graph replay runs them back to back.
        # 'auto': graphs for batches up to AUTO_GRAPH_BATCH encounters (0.89 against 1.5 ms per step at the reference's B = 256)
        # Sharded (one process per GPU): RCCL collectives are stream operations and capture with the kernels around them, so on the `nccl`
        # backend 'auto' captures the sharded step too for per-rank batches up to AUTO_GRAPH_BATCH -- what a strong-scaled batch of a few
        # thousand encounters per rank needs, where the step is launch-bound (4 096 rows: 1.53 ms eager for 1.10 ms of kernels); rehearsed
        # on RCCL with one rank (tests/test_gpu_dist.py).  DIC_SHARDED_GRAPHS=0 opts out.  gloo (CPU-side collectives) cannot be captured.
        # Replay-or-capture is decided from rank-invariant facts only (_agreed_key: the caller's global_rows), and a capture counts only when
        # it succeeded on EVERY rank (one MIN all-reduce per new key; otherwise all ranks fall back to the eager step together):
        # tests/test_dist_gloo.py::test_sharded_capture_is_agreed_between_ranks.
        self.auto_graphs = use_graphs == 'auto'
        want = self.auto_graphs or bool(use_graphs)
        if dist.is_sharded():
            allowed = (use_graphs is True) or (self.auto_graphs and os.environ.get('DIC_SHARDED_GRAPHS', '1') != '0')
            want = want and allowed and dist.graph_capturable()
        self.use_graphs = want
        self._graphs = {}
        self._store_ordinal = {}                 # id(ragged store) -> order of first appearance (rank-invariant: _agreed_key)
        self._sharded_capture_off = False        # set on EVERY rank together when a capture failed on any (_step_graphed)

    def _ctx(self):
        """Around the model's forward: autocast (bf16 step) or the f32 step's products mode."""
        if self.autocast_dtype is None:
            return self._mode()
        return torch.autocast('cuda', dtype=self.autocast_dtype)

    def _mode(self):
        """The f32 step's products mode ('exact' / 'x3'): entered around the WHOLE step -- forward, the loss operators and the backward --
        so that an operator consulting ops.f32_products() anywhere on that path sees this Stepper's mode, not the process default."""
        if self.autocast_dtype is None and self.precision is not None:
            return ops.f32_products_mode(self.precision)
        return contextlib.nullcontext()

    def forward_loss(self, x, ob, padding_mask, lengths=None, fake_x=None, fake_perm_idx=None, positive_x=None,
                     aux_label_dict=None, future_vital_mask=None, fake_det_label=None):
        if is_ragged(x):                  # ragged.RaggedBatch: lengths come with it; the observations are the batch itself unless given
            lengths, ob, padding_mask = x.lengths, (x if ob is None else ob), None
        if lengths is not None:
            padding_mask = None           # prefix lengths SAY what the mask is (the trainers hand over both): the kernels never read the plane
        with self._ctx():
            hidden, rec_ob, aux_pred = self.model(x, fake_x, fake_perm_idx, positive_x, lengths=lengths)
        losses = compute_losses(self.model, self.args, hidden, rec_ob, aux_pred, ob, padding_mask, lengths,
                                aux_label_dict, future_vital_mask, fake_det_label)
        return losses, hidden, rec_ob, aux_pred

    # ------------------------------------------------------------------------------ hipGraph path
    MAX_GRAPHS = 3        # captured steps kept (full batch, the short last batch, ...): least recently used first out

    def _graph_key(self, tensors):
        # (the fused optimiser reads lr / betas / eps / weight decay from device memory: they are not part of the key)
        lrs = () if self._fused_tail else tuple(float(g['lr']) for g in self.optimizer.param_groups)
        # (a ragged batch is a handle into ONE store, read in place by the captured launches, with its denoise flag baked in: both belong to the key)
        ragged = tuple((k, id(v.store), bool(v.denoise)) for k, v in tensors.items() if is_ragged(v))
        return tuple((k, tuple(v.shape), v.dtype) for k, v in tensors.items()) + (lrs, self.model.training, ragged)

    def _agreed_key(self, tensors, global_rows):
        """The cache key of a SHARDED step: built only from what every rank knows to be the same -- the GLOBAL batch's row count (the caller's:
        the loaders shard a batch of m rows as a pure function of (m, rank), so m fixes every rank's shard), the argument names / dtypes /
        trailing shapes, the learning rates, the training flag, and each ragged store by the ORDER in which this Stepper first met it.  A
        rank-local key (this rank's shard shape) is not one: a tail batch of 255 rows over 8 ranks gives rank 0 a new shape (31 rows: two eager
        warm-up steps + capture) and the others a cached one (32 rows: one replay), and the collectives pair up across different steps."""
        lrs = () if self._fused_tail else tuple(float(g['lr']) for g in self.optimizer.param_groups)
        ragged = []
        for k, v in tensors.items():
            if is_ragged(v):
                ragged.append((k, self._store_ordinal.setdefault(id(v.store), len(self._store_ordinal)), bool(v.denoise)))
        return ('sharded', int(global_rows)) + tuple((k, tuple(v.shape[1:]), v.dtype) for k, v in tensors.items()) + (lrs, self.model.training, tuple(ragged))

    def _snapshot(self, device):
        """Training state a warm-up step changes: parameters, buffers, optimiser state, the device RNG stream, the in-kernel dropout's call counter."""
        snap = {'flat': self.flat.flat.clone(), 'buf': [b.clone() for b in self.model.buffers()],
                'opt': {p: {k: v.clone() for k, v in st.items() if torch.is_tensor(v)} for p, st in self.optimizer.state.items()}}
        if device.type == 'cuda':
            snap['rng'] = torch.cuda.get_rng_state(device)               # dropout / randperm draws of the warm-up must not shift the stream
            snap['drop'] = ops.dropout_state_snapshot(device)            # ... nor the in-kernel dropout's own call counter (not CUDA RNG state)
        return snap

    def _restore(self, snap, device):
        with torch.no_grad():
            self.flat.flat.copy_(snap['flat'])
            for b, sb in zip(self.model.buffers(), snap['buf']):
                b.copy_(sb)
            for p_, st in self.optimizer.state.items():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        v.copy_(snap['opt'][p_][k]) if p_ in snap['opt'] and k in snap['opt'][p_] else v.zero_()
        if device.type == 'cuda':
            torch.cuda.set_rng_state(snap['rng'], device)
            ops.dropout_state_restore(device, snap['drop'])

    def _warm_up(self, run, device):
        """Two eager steps outside capture (lazy initialisation, hipFuncSetAttribute, allocator; sharded: every rank runs them on the same call,
        so their collectives pair up) that leave no trace: capturing executes nothing, the first replay IS the calling step."""
        snap = self._snapshot(device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                run()
        torch.cuda.current_stream().wait_stream(side)
        self._restore(snap, device)

    def _capture(self, run):
        """Record one step: (an object with .replay() / .reset(), the step's outputs -- written in place by every replay)."""
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = run()
        return graph, out

    def _step_graphed(self, tensors, global_rows=None):
        sharded = dist.is_sharded()
        key = self._agreed_key(tensors, global_rows) if sharded else self._graph_key(tensors)
        entry = self._graphs.get(key)
        if entry is not None and sharded and entry[3] != self._graph_key(tensors):
            # same global batch, another local shape / store: the caller's sharding is not a function of the global batch -- replaying here while
            # another rank captures would pair collectives of different steps.  Loud, on the rank that can see it.
            raise RuntimeError('Stepper: two sharded batches with the same global signature differ on this rank; pass a global_rows that '
                               'identifies the global batch (or use_graphs=False)')
        if entry is None:
            static = {k: v.clone() for k, v in tensors.items()}
            device = static['x'].device

            def run():
                return self._step_eager(static.get('x'), static.get('ob'), static.get('padding_mask'), static.get('lengths'),
                                        **{k: static[k] for k in static if k not in ('x', 'ob', 'padding_mask', 'lengths')})
            self._warm_up(run, device)                                   # (an exception here is an eager step's: it propagates, as without graphs)
            graph = out = None
            try:
                graph, out = self._capture(run)
                ok = True
            except Exception as e:                                       # noqa: BLE001 -- whatever failed, the fallback is the same
                if not sharded:
                    raise
                ok, graph, out = False, None, None
                self._capture_cleanup()
                print(f'[step] rank {dist.rank()}: capture of the sharded step failed ({type(e).__name__}: {e})', file=sys.stderr, flush=True)
            if sharded and not dist.all_agree(ok, device):
                # capture executes no collective, so the ranks are still in step; a rank whose capture succeeded drops its graph and ALL ranks run
                # this and every later step eagerly -- decided once, together, logged once
                if graph is not None:
                    graph.reset()
                self._sharded_capture_off = True
                if dist.rank() == 0:
                    print('[step] hipGraph capture of the sharded step did not succeed on every rank: all ranks run the eager step from here on',
                          file=sys.stderr, flush=True)
                return False, None
            while len(self._graphs) >= self.MAX_GRAPHS:                  # bound the memory held by captured steps (private pools, static inputs)
                old_graph, _, _, _ = self._graphs.pop(next(iter(self._graphs)))          # (sharded: the keys, hence the eviction order, are the same on every rank)
                old_graph.reset()
            entry = self._graphs[key] = (graph, static, out, self._graph_key(tensors))
        else:
            self._graphs[key] = self._graphs.pop(key)                    # most recently used last
        graph, static, out, _ = entry
        if self._fused_tail:
            self.optimizer.sync_hyper()
        for k, v in tensors.items():
            static[k].copy_(v, non_blocking=True)
        graph.replay()
        return True, out

    def _capture_cleanup(self):
        """After a capture that raised: nothing recorded may leak into the eager steps that follow."""
        dist.drop_riders()
        self.flat._tail_work = None              # (a handle created while recording never ran)
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    AUTO_GRAPH_BATCH = 8192

    def _graphable(self, x):
        return x.is_cuda

    def step(self, x, ob, padding_mask, lengths=None, global_rows=None, **kw):
        """One optimisation step.  Returns (loss terms, gradient norm, latents): device tensors, valid until the next call (a replayed
        hipGraph writes them in place; a graph evicted from the small cache frees their memory -- clone what must outlive the step).
        With ``lengths`` (prefix masks, all the loaders produce) ``padding_mask`` is redundant and dropped: the de-interpolation kernels
        then emit the reconstruction loss themselves.  ``global_rows``: rows of the GLOBAL batch this rank's ``x`` is a shard of (the loaders'
        samples carry it) -- a sharded step is replayed from a hipGraph only with it, because the replay-or-capture decision must fall the same
        way on every rank (``_agreed_key``); without it a sharded step runs eagerly."""
        if is_ragged(x):
            lengths = padding_mask = None                # (both travel inside x)
        if lengths is not None:
            padding_mask = None
        sharded = dist.is_sharded()
        if (self.use_graphs and self._graphable(x) and (not self.auto_graphs or x.shape[0] <= self.AUTO_GRAPH_BATCH)
                and not (sharded and (global_rows is None or self._sharded_capture_off))):
            tensors = {'x': x} if ob is None else {'x': x, 'ob': ob}
            if padding_mask is not None:
                tensors['padding_mask'] = padding_mask
            if lengths is not None:
                tensors['lengths'] = lengths
            extra = {k: v for k, v in kw.items() if v is not None and not (isinstance(v, dict) and not v)}
            if all(torch.is_tensor(v) for v in extra.values()):
                tensors.update(extra)
                replayed, out = self._step_graphed(tensors, global_rows)
                if replayed:
                    return out
        return self._step_eager(x, ob, padding_mask, lengths, **kw)

    def _step_eager(self, x, ob, padding_mask, lengths=None, **kw):
        self.flat.zero_grad()
        if is_ragged(x):
            lengths, ob = x.lengths, (x if ob is None else ob)
        self.model.internal_step = True          # the reconstruction stays inside this step: its padded slots need not be written
        # ... and with prefix lengths its loss comes out of the de-interpolation kernels themselves (ops.rbf_rec_loss)
        self.model.rec_target = ob if (padding_mask is None and lengths is not None and ob.is_cuda) else None
        try:
            with self._mode():
                losses, hidden, _, _ = self.forward_loss(x, ob, padding_mask, lengths, **kw)
                # the small parameter gradients: one add launch for all of them at the end; the decoder's weight-gradient kernel on a side stream
                with ops.grad_sink_session(), fused_lstm.side_stream_session():
                    losses['loss'].backward()
        finally:
            dist.drop_riders()
            self.model.internal_step = False
            self.model.rec_target = None
            self.model._fused_rec = None
        self.flat.all_reduce_grads()
        if self._fused_tail:
            gnorm, coef = self.flat.clip_coef(self.args.grad_clip)
            self.optimizer.step(grad_scale=coef)
        else:
            gnorm = self.flat.clip_grad_norm_(self.args.grad_clip)
            self.optimizer.step()
        return losses, gnorm, hidden

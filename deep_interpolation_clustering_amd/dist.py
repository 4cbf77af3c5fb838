"""One-process-per-GPU data parallelism over RCCL/xGMI (replaces the reference's single-process
``torch.nn.DataParallel``, pretrain_trainer.py:21 / clustering_trainer.py:25).

Encounters are sharded by rank; parameters and optimizer state are replicated.  Per joint step the
exchanges are: the small forward statistics that couple the batch (reconstruction
SSE + mask count, DEC column sums, KL sum, BatchNorm moments) and ONE flat gradient bucket.  All
are sum all-reduces; with ``backend='nccl'`` they run on RCCL, with ``'gloo'`` on CPU (tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as td


# DIC_DIST_SINGLE_RANK=1: a world of ONE rank counts as sharded, so that every collective of the sharded step / k-means / bench
# runs on the real backend (RCCL on a 1-GPU box: tests/test_gpu_dist.py) -- a rehearsal switch, never set in production
_SINGLE_RANK_REHEARSAL = os.environ.get('DIC_DIST_SINGLE_RANK') == '1'


def graph_capturable() -> bool:
    """True when the collectives of a sharded step can be recorded into a hipGraph: RCCL enqueues them on HIP streams (gloo does not)."""
    return is_sharded() and td.get_backend() == 'nccl'


def is_sharded() -> bool:
    return td.is_available() and td.is_initialized() and (td.get_world_size() > 1 or _SINGLE_RANK_REHEARSAL)


def world_size() -> int:
    return td.get_world_size() if (td.is_available() and td.is_initialized()) else 1


def rank() -> int:
    return td.get_rank() if (td.is_available() and td.is_initialized()) else 0


# Small statistics that are not needed at once ride on the NEXT small sum all-reduce of the step instead of paying a latency-bound exchange of
# their own ("pack all small statistics into one buffer per step"): the DEC column sums f_j are known right after the encoder
# but consumed only by the target distribution at the end of the forward -- they travel with the BatchNorm moments of CompressFC.
_RIDERS = []
RIDER_MAX_CARRIER = 4096          # elements: only small exchanges carry riders (never the gradient bucket: the concatenation would copy it)


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum all-reduce; a no-op when not sharded.  Pending riders (``deferred_sum_``) of a dtype this buffer can carry exactly are
    appended, reduced in the same collective and written back in place."""
    if not is_sharded():
        return t
    riders = []
    if _RIDERS and t.numel() <= RIDER_MAX_CARRIER and t.is_contiguous():
        for entry in list(_RIDERS):
            r = entry[0]
            if r.device == t.device and (r.dtype == t.dtype or (t.dtype == torch.float64 and r.dtype == torch.float32)):
                riders.append(entry)
                _RIDERS.remove(entry)
    if not riders:
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return t
    buf = torch.cat([t.reshape(-1)] + [r.reshape(-1).to(t.dtype) for r, _ in riders])
    td.all_reduce(buf, op=td.ReduceOp.SUM)
    o = t.numel()
    t.copy_(buf[:o].view_as(t))
    for r, then in riders:
        r.copy_(buf[o:o + r.numel()].view_as(r))          # (f32 riders on an f64 carrier: summed in f64, rounded once)
        o += r.numel()
        if then is not None:
            then()
    return t


def deferred_sum_(t: torch.Tensor, then=None) -> torch.Tensor:
    """Queue ``t`` for an in-place sum all-reduce that rides on the next small ``all_reduce_sum_``; ``resolve_sum_(t)`` (or ``resolve_all_()``)
    before reading it.  ``then``: called right after ``t`` holds its global sum (whatever carried it) -- for values derived from it."""
    if is_sharded() and not any(r is t for r, _ in _RIDERS):
        _RIDERS.append((t, then))
    elif not is_sharded() and then is not None:
        then()
    return t


def resolve_sum_(t: torch.Tensor) -> torch.Tensor:
    """``t`` holds its global sum after this call: a rider that found no carrier gets a collective of its own."""
    for i, (r, then) in enumerate(_RIDERS):
        if r is t:
            del _RIDERS[i]
            if is_sharded():
                td.all_reduce(t, op=td.ReduceOp.SUM)
            if then is not None:
                then()
            break
    return t


def resolve_all_():
    """Every pending rider holds its global sum after this call: the ones still waiting travel together in ONE collective per dtype."""
    while _RIDERS:
        first = _RIDERS[0][0]
        if sum(1 for r, _ in _RIDERS if r.dtype == first.dtype and r.device == first.device) > 1 and first.numel() <= RIDER_MAX_CARRIER:
            r, then = _RIDERS.pop(0)
            all_reduce_sum_(r)                  # carries the others of its dtype
            if then is not None:
                then()
        else:
            resolve_sum_(first)


def drop_riders():
    """(step.Stepper, after a forward that raised: nothing may ride into the next step)"""
    del _RIDERS[:]


def all_gather_rows(t: torch.Tensor) -> torch.Tensor:
    """(world, *t.shape): every rank's ``t`` (equal shapes), one all-gather (RCCL / gloo); ``t[None]`` when not sharded."""
    if not is_sharded():
        return t[None]
    t = t.contiguous()
    out = torch.empty((world_size() * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)      # (the concatenated form: gloo takes no other)
    td.all_gather_into_tensor(out, t)
    return out.view((world_size(),) + tuple(t.shape))


class _GlobalMean(torch.autograd.Function):
    """sum over ranks of ``local_sum`` / sum over ranks of ``local_count``: the value every rank reports is the global-batch mean,
    and d/d local_sum = 1 / global count, so the SUM of the ranks' gradients (FlatParams.all_reduce_grads) is the gradient of that
    mean -- also when the shards differ in size."""

    @staticmethod
    def forward(ctx, local_sum, local_count):
        # (the count as a device scalar WITHOUT a host-to-device copy: a fill kernel is capturable in a hipGraph, a pageable copy is not)
        cnt = (local_count.to(device=local_sum.device, dtype=torch.float32).reshape(()) if torch.is_tensor(local_count)
               else torch.full((), float(local_count), dtype=torch.float32, device=local_sum.device))
        both = torch.stack([local_sum.detach().float().reshape(()), cnt])
        all_reduce_sum_(both)
        ctx.save_for_backward(both[1])
        return (both[0] / both[1]).to(local_sum.dtype)

    @staticmethod
    def backward(ctx, grad):
        (count,) = ctx.saved_tensors
        return grad / count.to(grad.dtype), None


def global_mean(local_sum: torch.Tensor, local_count) -> torch.Tensor:
    """Mean over the GLOBAL batch of a quantity each rank holds as (sum over its rows, number of its rows)."""
    if not is_sharded():
        return local_sum / local_count
    return _GlobalMean.apply(local_sum, local_count)


def all_agree(ok: bool, device=None) -> bool:
    """True on every rank iff ``ok`` was true on every rank (one MIN all-reduce of one int; not sharded: ``ok``).  A host sync -- for decisions
    taken once (step.Stepper: did the capture of the sharded step succeed everywhere?), not per step."""
    if not is_sharded():
        return bool(ok)
    on_dev = td.get_backend() == 'nccl'
    flag = torch.full((1,), 1 if ok else 0, dtype=torch.int32, device=(device if on_dev and device is not None else 'cpu'))
    td.all_reduce(flag, op=td.ReduceOp.MIN)
    return bool(int(flag.item()))


def barrier():
    """Rank barrier (no-op when not sharded): rank 0 writes checkpoints / feature files that the other ranks read."""
    if is_sharded():
        td.barrier()


def init_from_env(backend=None):
    """Initialise the default process group from torchrun-style env vars (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size, local_rank)."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    rk = int(os.environ.get('RANK', '0'))
    lr = int(os.environ.get('LOCAL_RANK', '0'))
    if (ws > 1 or _SINGLE_RANK_REHEARSAL) and not td.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:     # DIC_DIST_BACKEND=gloo lets several ranks share ONE GPU (rehearsal of the N>1 path on a 1-GPU box)
            backend = os.environ.get('DIC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(lr)
            td.init_process_group(backend, rank=rk, world_size=ws, device_id=torch.device('cuda', lr))
        else:
            td.init_process_group(backend, rank=rk, world_size=ws)
    return rk, ws, lr


def launch_ranks(n_ranks, module, argv):
    """Start ``n_ranks`` ranks of ``python -m <module> <argv>`` through ``torch.distributed.run`` (one process per GPU, 127.0.0.1 rendezvous on a
    free port) as CHILD processes and return their exit code.  The caller must not have touched the GPU: it only waits (replacing a process
    that has initialised HIP takes the machine down on this pool, and a parent holding the card would be one process too many on it)."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={int(n_ranks)}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), '-m', module] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL across processes needs it on this driver
    return subprocess.run(cmd, env=env).returncode


def ranks_for_num_gpus(num_gpus, module, argv):
    """The reference's multi-GPU entry is ``python p1_pretrain_main.py --num_gpus N`` -> ``DataParallel(device_ids=range(N))`` in ONE process
    (p1_pretrain_main.py:27,118; pretrain_trainer.py:21; clustering_trainer.py:25).  Here N GPUs are N processes, so a driver started that way
    starts its own ranks: returns the children's exit code when this process was the launcher (the caller exits with it), None when this
    process is a rank (or a single-GPU / CPU run) and should go on.  A launcher's WORLD_SIZE that contradicts an explicit ``--num_gpus N``
    (N > 1) is an error; under a launcher with the default ``--num_gpus 1`` the world size is the launcher's."""
    import sys
    ws = os.environ.get('WORLD_SIZE')
    if ws is None:
        if num_gpus > 1:
            print(f'[dist] --num_gpus {num_gpus}: starting {num_gpus} ranks of {module} through torch.distributed.run', file=sys.stderr, flush=True)
            return launch_ranks(num_gpus, module, sys.argv[1:] if argv is None else argv)
        return None
    if num_gpus > 1 and int(ws) != num_gpus:
        raise SystemExit(f'--num_gpus {num_gpus} but WORLD_SIZE={ws}: start {module} either plainly with --num_gpus N (it launches its own ranks) '
                         f'or under torch.distributed.run with --nproc-per-node equal to --num_gpus')
    return None


def shard_bounds(n, rk=None, ws=None):
    """Contiguous shard [lo, hi) of n items for this rank (rank r gets rows [r*n/P, (r+1)*n/P))."""
    rk = rank() if rk is None else rk
    ws = world_size() if ws is None else ws
    return (n * rk) // ws, (n * (rk + 1)) // ws


class FlatParams:
    """Re-homes a module's parameters (and their grads) into ONE flat f32 buffer each, so the
    per-step gradient exchange is a single contiguous all-reduce (2.33 MB at K=4) with no packing
    copies, and gradient clipping is one norm over one buffer."""

    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError('module has no trainable parameters')
        dev, total = self.params[0].device, sum(p.numel() for p in self.params)
        self.flat = torch.empty(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self._attach(copy=True)
        # which parameters a backward pass actually reached: torch.optim skips parameters whose .grad is None (no weight decay,
        # no state), and with pre-allocated gradient views "None" has to be observed instead -- a hook per parameter flags it
        self._reached = [False] * len(self.params)
        self._mask_key, self._mask = None, None
        self._norm_ws = None
        self._split, self._split_index, self._tail_work = None, 0, None
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(_param):
            self._reached[i] = True
        return hook

    def active_mask(self):
        """Per-element byte mask of the parameters the last backward reached, or None when it reached all of them.  Cached per
        reach pattern (it is a property of the loss configuration, constant from step to step)."""
        # (kernels that add a parameter gradient straight into its .grad view -- lstm._grad_sinks -- bypass autograd's
        # AccumulateGrad and therefore the hook: they leave the mark `_dic_grad_written` on the parameter instead)
        key = tuple(r or getattr(p, '_dic_grad_written', False) for r, p in zip(self._reached, self.params))
        if all(key) or not any(key):          # nothing recorded (e.g. a replayed hipGraph) counts as 'all', the common case
            return None
        if key != self._mask_key:
            m = torch.zeros(self.flat.numel(), dtype=torch.uint8)
            o = 0
            for p, on in zip(self.params, key):
                n = p.numel()
                if on:
                    m[o:o + n] = 1
                o += n
            self._mask_key, self._mask = key, m.to(self.flat.device)
        return self._mask

    def _attach(self, copy=False):
        o = 0
        for p in self.params:
            n = p.numel()
            if copy:
                self.flat[o:o + n].copy_(p.data.reshape(-1))
            if copy or p.data.data_ptr() != self.flat[o:o + n].data_ptr():
                p.data = self.flat[o:o + n].view_as(p.data)
            if p.grad is None or p.grad.data_ptr() != self.grad[o:o + n].data_ptr():
                p.grad = self.grad[o:o + n].view_as(p.data)
            o += n

    def zero_grad(self):
        if self._tail_work is not None:      # a backward that raised after the early all-reduce started: finish it before the bucket is reused
            self._tail_work.wait()
            self._tail_work = None
        self.grad.zero_()
        self._attach()          # re-attach in case something replaced .data / .grad
        self._reached = [False] * len(self.params)
        for p in self.params:
            p._dic_grad_written = False

    # ---- gradient exchange.  The bucket is reduced in two pieces so that the larger one travels during the rest of the backward:
    # everything registered from `split_at` on (decoder LSTM, de-interpolation head, auxiliary heads, centroids: 74 % of the
    # bucket) is complete when the backward reaches the encoder -- begin_tail_reduce() is called there (a tensor hook the model
    # places on the encoder output) -- and its all-reduce overlaps the encoder LSTM / interpolation backward.
    def set_split(self, first_tail_param):
        """``first_tail_param``: the first parameter (in registration order) of the piece that finishes early."""
        o = 0
        for i, p in enumerate(self.params):
            if p is first_tail_param:
                self._split, self._split_index = o, i
                return
            o += p.numel()
        raise ValueError('set_split: not a parameter of this bucket')

    def begin_tail_reduce(self):
        """Start the asynchronous all-reduce of grad[split:] -- only if every parameter of that piece has its gradient already."""
        if not is_sharded() or self._split is None or self._tail_work is not None:
            return
        from . import lstm, ops
        ops.flush_grad_sinks()          # gradients queued by the decoder-side kernels go into the bucket before its tail leaves
        done = all(r or getattr(p, '_dic_grad_written', False) for r, p in zip(self._reached[self._split_index:], self.params[self._split_index:]))
        if not (done and self._split < self.grad.numel()):
            return
        # the decoder's weight gradients may still be forming on the side stream (bf16 step, lstm.DW_SIDE_STREAM): the collective is then
        # enqueued BEHIND that stream (which first catches up with everything the main stream has launched so far), so the encoder backward
        # on the main stream overlaps both the weight-gradient kernel and the all-reduce
        side = lstm.pending_side_stream(self.grad.device) if self.grad.is_cuda else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(self.grad.device))
            with torch.cuda.stream(side):
                self._tail_work = td.all_reduce(self.grad[self._split:], op=td.ReduceOp.SUM, async_op=True)
        else:
            self._tail_work = td.all_reduce(self.grad[self._split:], op=td.ReduceOp.SUM, async_op=True)

    def all_reduce_grads(self):
        """Sum over ranks: each rank's loss is already normalised by GLOBAL batch statistics, so
        the sum (not the mean) of rank gradients is the gradient of the global-batch loss."""
        if self._tail_work is not None:
            if self._split > 0:
                all_reduce_sum_(self.grad[:self._split])
            self._tail_work.wait()
            self._tail_work = None
        else:
            all_reduce_sum_(self.grad)

    def clip_coef(self, max_norm: float):
        """(total norm, clip coefficient) of torch.nn.utils.clip_grad_norm_ (pretrain_trainer.py:228), both on the device."""
        if self.grad.is_cuda:          # one reduction kernel + one scalar kernel (csrc/dic_optim.hip) instead of five torch launches
            from . import _native as N
            L = N.lib()
            if self._norm_ws is None:
                self._norm_ws = torch.empty(max(16, L.dic_grad_norm_workspace(self.grad.numel())), dtype=torch.uint8, device=self.grad.device)
            out2 = torch.empty(2, device=self.grad.device, dtype=torch.float32)
            N.check(L.dic_grad_norm_clip(N.ptr(self.grad), self.grad.numel(), float(max_norm), N.ptr(out2), N.ptr(self._norm_ws),
                                         self._norm_ws.numel(), N.stream_of(self.grad)), 'dic_grad_norm_clip')
            return out2[0], out2[1]
        total = torch.linalg.vector_norm(self.grad)
        return total, torch.clamp(max_norm / (total + 1e-6), max=1.0)

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ semantics on the flat bucket."""
        total, coef = self.clip_coef(max_norm)
        self.grad.mul_(coef)
        return total

    def broadcast_(self, src: int = 0):
        if is_sharded():
            td.broadcast(self.flat, src=src)


# ---------------------------------------------------------------------------------------------
# BatchNorm over the GLOBAL batch.  The reference's heads (rbf.py:118, clustering_interp.py:50,65,80)
# normalise with whole-batch moments on one device; when the batch is sharded each rank must use the
# same moments or the sharded step stops being the single-device step.
class _SyncBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        F_ = x.shape[1]
        stats = torch.empty(2 * F_ + 1, device=x.device, dtype=torch.float32)
        xf = x.float()
        stats[:F_] = xf.sum(0)
        stats[F_:2 * F_] = (xf * xf).sum(0)
        stats[2 * F_] = x.shape[0]
        all_reduce_sum_(stats)
        n = stats[2 * F_]
        mean = stats[:F_] / n
        var = (stats[F_:2 * F_] / n - mean * mean).clamp_min_(0)
        rstd = torch.rsqrt(var + eps)
        xhat = (xf - mean) * rstd
        ctx.save_for_backward(xhat, rstd, weight, n)
        y = xhat * weight.float() + bias.float()
        ctx.mark_non_differentiable(mean, var, n)
        return y.to(x.dtype), mean, var, n

    @staticmethod
    def backward(ctx, dy, _m, _v, _n):
        xhat, rstd, weight, n = ctx.saved_tensors
        dyf = dy.float()
        F_ = dyf.shape[1]
        red = torch.empty(2 * F_, device=dy.device, dtype=torch.float32)
        red[:F_] = dyf.sum(0)
        red[F_:] = (dyf * xhat).sum(0)
        local = red.clone()
        all_reduce_sum_(red)
        dx = (weight.float() * rstd / n) * (n * dyf - red[:F_] - xhat * red[F_:])
        return dx.to(dy.dtype), local[F_:].to(weight.dtype), local[:F_].to(weight.dtype), None


class GlobalBatchNorm1d(torch.nn.BatchNorm1d):
    """nn.BatchNorm1d whose training-mode moments span all ranks; identical module state / keys."""

    def forward(self, x):
        if not (self.training and is_sharded()):
            return super().forward(x)
        y, mean, var, n = _SyncBNFn.apply(x, self.weight, self.bias, self.eps)
        if self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                mom = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                self.running_mean.mul_(1 - mom).add_(mean.to(self.running_mean.dtype), alpha=mom)
                unbiased = var * (n / (n - 1).clamp_min(1))
                self.running_var.mul_(1 - mom).add_(unbiased.to(self.running_var.dtype), alpha=mom)
        return y


def convert_batchnorm_(module: torch.nn.Module) -> torch.nn.Module:
    """Re-class every nn.BatchNorm1d in ``module`` as GlobalBatchNorm1d (in place; same parameters)."""
    for m in module.modules():
        if type(m) is torch.nn.BatchNorm1d:
            m.__class__ = GlobalBatchNorm1d
    return module

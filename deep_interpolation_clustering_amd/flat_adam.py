"""Adam(amsgrad, L2 weight decay) of the reference (utils.py:76-83: ``optim.Adam(..., amsgrad=True)``) over the ONE flat
parameter / gradient bucket of ``dist.FlatParams``: gradient clipping scale + moment updates + parameter update are a
single HIP kernel (csrc/dic_optim.hip) instead of torch's multi-tensor Adam over 25 small tensors.

Drop-in for ``torch.optim.Adam``: same constructor defaults, same ``param_groups`` keys (lr schedulers work), same
``state_dict`` layout (per parameter ``step, exp_avg, exp_avg_sq, max_exp_avg_sq``), so checkpoints written by either
load into the other.  The per-parameter state tensors are views into three flat buffers.  Parameters that the backward pass
did not reach are skipped like ``torch.optim`` skips ``grad is None`` (``FlatParams.active_mask``); one difference remains:
the step count is shared, so a parameter that starts receiving gradients later uses the global count in its bias correction.
"""
import torch

from . import _native as N


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=True, maximize=False, foreach=None,
                        capturable=True, differentiable=False, fused=None, decoupled_weight_decay=False)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError('FlatAdam: one parameter group (the flat bucket has one set of hyper-parameters)')
        self._flat = None

    # ---- binding to the flat buckets (done by step.Stepper right after construction)
    def bind(self, flat):
        self._flat = flat
        n, dev = flat.flat.numel(), flat.flat.device
        self._m = torch.zeros(n, device=dev, dtype=torch.float32)
        self._v = torch.zeros_like(self._m)
        self._vmax = torch.zeros_like(self._m)
        self._step = torch.zeros((), device=dev, dtype=torch.float32)
        # [lr, beta1, beta2, eps, weight_decay] on the device: the kernel reads them there, so a captured hipGraph of the step keeps
        # following the learning-rate scheduler (sync_hyper uploads a changed value OUTSIDE the captured region)
        self._hyper = torch.zeros(5, device=dev, dtype=torch.float32)
        self._hyper_host = None
        self.sync_hyper()
        self._views()
        return self

    def sync_hyper(self):
        g = self.param_groups[0]
        now = (float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay']))
        if now != self._hyper_host:
            self._hyper.copy_(torch.tensor(now, dtype=torch.float32))
            self._hyper_host = now

    def _views(self):
        o = 0
        for p in self._flat.params:
            k = p.numel()
            self.state[p] = {'step': self._step, 'exp_avg': self._m[o:o + k].view_as(p), 'exp_avg_sq': self._v[o:o + k].view_as(p),
                             'max_exp_avg_sq': self._vmax[o:o + k].view_as(p)}
            o += k

    def state_dict(self):
        sd = super().state_dict()
        # one independent step tensor per parameter, as optim.Adam keeps them (a shared one would be incremented once per parameter by
        # its foreach path after loading).  The per-parameter dicts super() returns ARE the live ones: they are copied, not edited --
        # editing them left a frozen step count in self.state, and every later checkpoint carried that stale count
        sd['state'] = {k: ({**st, 'step': st['step'].clone()} if 'step' in st else dict(st)) for k, st in sd['state'].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)               # validates, casts and copies into fresh per-parameter tensors
        if self._flat is None:
            return
        loaded = {p: dict(st) for p, st in self.state.items()}
        self._views()                                     # back to views of the flat buffers, then take the loaded values
        with torch.no_grad():
            for p, st in loaded.items():
                if p not in self.state:
                    continue
                for k in ('exp_avg', 'exp_avg_sq', 'max_exp_avg_sq'):
                    if k in st:
                        self.state[p][k].copy_(st[k])
                if 'step' in st:
                    self._step.copy_(torch.as_tensor(st['step'], dtype=torch.float32))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """``grad_scale``: optional device scalar multiplied into the gradients first (the clip_grad_norm_ coefficient)."""
        if self._flat is None:
            raise RuntimeError('FlatAdam.bind(FlatParams) first: the optimiser works on the flat buckets only')
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        f = self._flat
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        self._step += 1
        N.check(N.lib().dic_adam_amsgrad_step(N.ptr(f.flat), N.ptr(f.grad), N.ptr(self._m), N.ptr(self._v), N.ptr(self._vmax), f.flat.numel(),
                                              float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                                              float(g['weight_decay']), N.ptr(self._step), N.ptr(grad_scale), N.ptr(f.active_mask()),
                                              N.ptr(self._hyper), N.stream_of(f.flat)),
                'dic_adam_amsgrad_step')
        return loss

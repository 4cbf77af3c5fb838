"""world_size-2 gloo tests (CPU) of the data-parallel plumbing in ``dist``: the flat gradient bucket,
global-batch BatchNorm and the sharding helpers make a sharded step equal the single-process step."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as td
    from deep_interpolation_clustering_amd import dist
    dist.init_from_env('gloo')
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(16, 128), torch.nn.BatchNorm1d(128), torch.nn.ReLU(), torch.nn.Linear(128, 3))
        dist.convert_batchnorm_(net)
        flat = dist.FlatParams(net)
        flat.broadcast_(0)
        g = torch.Generator().manual_seed(5)
        X, Y = torch.randn(64, 16, generator=g), torch.randn(64, 3, generator=g)
        lo, hi = dist.shard_bounds(64)
        opt = torch.optim.Adam(net.parameters(), lr=3e-3, weight_decay=4e-4, amsgrad=True)
        for _ in range(3):
            flat.zero_grad()
            # loss normalised by the GLOBAL count -> SUM of rank gradients is the global gradient
            loss = ((net(X[lo:hi]) - Y[lo:hi]) ** 2).sum() / 64
            loss.backward()
            flat.all_reduce_grads()
            gn = flat.clip_grad_norm_(15.0)
            opt.step()
        stat = torch.tensor([float(loss.detach())])
        dist.all_reduce_sum_(stat)
        torch.save({'flat': flat.flat.clone(), 'gn': float(gn), 'loss': float(stat), 'rm': net[1].running_mean.clone(),
                    'rv': net[1].running_var.clone(), 'bounds': (lo, hi)}, os.path.join(out, f'r{rank}.pt'))
    finally:
        td.destroy_process_group()


def _single():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 128), torch.nn.BatchNorm1d(128), torch.nn.ReLU(), torch.nn.Linear(128, 3))
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(64, 16, generator=g), torch.randn(64, 3, generator=g)
    opt = torch.optim.Adam(net.parameters(), lr=3e-3, weight_decay=4e-4, amsgrad=True)
    for _ in range(3):
        opt.zero_grad()
        loss = ((net(X) - Y) ** 2).sum() / 64
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 15.0)
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    return flat, float(gn), float(loss.detach()), net[1].running_mean, net[1].running_var


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_step_equals_single_process(tmp_path, world):
    """Two ranks with equal shards, and three with shards of 21 / 21 / 22 rows (a global batch the world size does not divide)."""
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ranks = [torch.load(tmp_path / f'r{k}.pt') for k in range(world)]
    r0, r1 = ranks[0], ranks[1]
    assert [r['bounds'] for r in ranks] == [((64 * k) // world, (64 * (k + 1)) // world) for k in range(world)]
    for r in ranks[1:]:
        assert torch.equal(r0['flat'], r['flat'])                     # replicas stay bit-identical
    flat, gn, loss, rm, rv = _single()
    live = np.ones(flat.numel(), bool)
    live[16 * 128:16 * 128 + 128] = False      # Linear bias feeding BatchNorm: true gradient 0, Adam amplifies rounding noise
    np.testing.assert_allclose(r0['flat'].numpy()[live], flat.numpy()[live], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(r0['gn'], gn, rtol=1e-5)
    np.testing.assert_allclose(r0['loss'], loss, rtol=1e-5)           # sum of the two shard losses
    np.testing.assert_allclose(r0['rm'].numpy(), rm.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r0['rv'].numpy(), rv.numpy(), rtol=1e-4, atol=1e-6)


def test_shard_bounds_cover_everything():
    from deep_interpolation_clustering_amd import dist
    for n in (1, 7, 64, 75000):
        for ws in (1, 2, 3, 8):
            b = [dist.shard_bounds(n, r, ws) for r in range(ws)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(ws - 1))


def test_flat_params_views_survive_state_dict_load():
    from deep_interpolation_clustering_amd import dist
    net = torch.nn.Linear(4, 3)
    flat = dist.FlatParams(net)
    net.load_state_dict({'weight': torch.ones(3, 4), 'bias': torch.zeros(3)})
    assert float(flat.flat.sum()) == 12.0 and net.weight.data_ptr() == flat.flat.data_ptr()
    net(torch.ones(2, 4)).sum().backward()
    assert float(flat.grad.abs().sum()) > 0 and net.weight.grad.data_ptr() == flat.grad.data_ptr()


def _mean_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as td
    from deep_interpolation_clustering_amd import dist
    dist.init_from_env('gloo')
    try:
        g = torch.Generator().manual_seed(3)
        X, w = torch.randn(7, 5, generator=g), torch.randn(5, generator=g).requires_grad_()
        lo, hi = (0, 3) if rank == 0 else (3, 7)                     # uneven shards of the global batch of 7
        loss = dist.global_mean(((X[lo:hi] @ w) ** 2).sum(), hi - lo)
        loss.backward()
        grad = w.grad.clone()
        dist.all_reduce_sum_(grad)                                   # what FlatParams.all_reduce_grads does
        torch.save({'loss': float(loss.detach()), 'grad': grad}, os.path.join(out, f'm{rank}.pt'))
    finally:
        td.destroy_process_group()


def test_global_mean_with_uneven_shards(tmp_path):
    """dist.global_mean: every rank reports the global-batch mean, and the sum of the ranks' gradients is its gradient, with shards
    of 3 and 4 rows (a global batch that does not divide by the world size)."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_mean_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'm0.pt'), torch.load(tmp_path / 'm1.pt')
    g = torch.Generator().manual_seed(3)
    X, w = torch.randn(7, 5, generator=g), torch.randn(5, generator=g).requires_grad_()
    ref = ((X @ w) ** 2).mean()
    ref.backward()
    assert r0['loss'] == r1['loss']
    np.testing.assert_allclose(r0['loss'], float(ref.detach()), rtol=1e-6)
    np.testing.assert_allclose(r0['grad'].numpy(), w.grad.numpy(), rtol=1e-5, atol=1e-7)
    assert torch.equal(r0['grad'], r1['grad'])


def _rider_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as td
    from deep_interpolation_clustering_amd import dist
    dist.init_from_env('gloo')
    try:
        calls, inner = [], td.all_reduce

        def counting(t, *a, **k):
            calls.append(int(t.numel()))
            return inner(t, *a, **k)
        td.all_reduce = counting
        res = {}
        # an f32 rider on an f64 carrier: one collective, both reduced in place (the DEC column sums on CompressFC's BatchNorm moments)
        colsum = torch.tensor([1.0 + rank, 2.0, 0.5 * rank, 4.0])
        dist.deferred_sum_(colsum)
        sums = torch.arange(257, dtype=torch.float64) * (rank + 1)
        dist.all_reduce_sum_(sums)
        n_after_carrier = len(calls)
        dist.resolve_sum_(colsum)                                        # already reduced: no second collective
        res['colsum'], res['sums_tail'], res['calls_a'] = colsum.clone(), sums[-3:].clone(), (n_after_carrier, len(calls), calls[:])
        # a rider that meets no carrier gets a collective of its own when it is resolved
        lone = torch.tensor([float(rank)])
        dist.deferred_sum_(lone)
        dist.resolve_sum_(lone)
        res['lone'], res['calls_b'] = lone.clone(), len(calls)
        # the gradient bucket (large) never carries riders; an f64 rider does not ride on an f32 carrier
        r64 = torch.tensor([1.5 * (rank + 1)], dtype=torch.float64)
        dist.deferred_sum_(r64)
        big = torch.ones(10000)
        dist.all_reduce_sum_(big)
        small32 = torch.ones(4)
        dist.all_reduce_sum_(small32)
        res['still_pending'] = float(r64)                                # untouched so far
        dist.resolve_sum_(r64)
        res['r64'], res['calls_c'] = float(r64), calls[len(calls) - 3:]
        td.all_reduce = inner
        torch.save(res, os.path.join(out, f'rider{rank}.pt'))
    finally:
        td.destroy_process_group()


def test_deferred_sums_ride_on_the_next_small_all_reduce(tmp_path):
    """dist.deferred_sum_ / resolve_sum_: small statistics that are needed later travel with the next small exchange (one
    packed buffer per dependency level), exactly, and never with the gradient bucket."""
    port = 29800 + (os.getpid() % 1000)
    mp.spawn(_rider_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f'rider{r}.pt', weights_only=False) for r in (0, 1))
    for r in (r0, r1):
        assert torch.equal(r['colsum'], torch.tensor([3.0, 4.0, 0.5, 8.0]))
        assert torch.equal(r['sums_tail'], torch.tensor([254.0, 255.0, 256.0], dtype=torch.float64) * 3)
        assert r['calls_a'][0] == 1 and r['calls_a'][1] == 1 and r['calls_a'][2] == [261]          # 257 + 4 in ONE collective, none at resolve
        assert float(r['lone']) == 1.0 and r['calls_b'] == 2
        assert r['still_pending'] in (1.5, 3.0) and r['r64'] == 4.5 and r['calls_c'] == [10000, 4, 1]


# ---------------------------------------------------------------------------------------------------------------------------------------
# step.Stepper's sharded hipGraph path: replay-or-capture must fall the same way on every rank, and a capture counts only when it
# succeeded everywhere.  gloo cannot be captured, so the CUDA-specific pieces (_warm_up, _capture) are replaced by host stand-ins that keep
# their CONTRACT -- warm-up = two eager steps that leave no trace, capture = records without executing (no collective), replay = one step --
# and the agreement logic around them is the product's own.
class _TinyNet(torch.nn.Module):
    """The Net surface Stepper drives (forward -> (hidden, rec_ob, aux_pred), rec_loss), with a global-batch BatchNorm (two exchanges per
    step) and a reconstruction loss normalised over the global batch (a third)."""

    def __init__(self):
        super().__init__()
        self.body = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 8))

    def forward(self, x, fake_x=None, fake_perm_idx=None, positive_x=None, lengths=None):
        return x, self.body(x), {}

    def rec_loss(self, ob, rec_ob, padding_mask, lengths=None):
        from deep_interpolation_clustering_amd import dist
        mse = dist.global_mean(((rec_ob - ob) ** 2).sum(), ob.numel())
        return {'loss': mse, 'ae_mse': mse}


class _FakeGraph:
    def __init__(self, run, out):
        self.run, self.out, self.was_reset = run, out, False

    def replay(self):
        self.out['last'] = self.run()

    def reset(self):
        self.was_reset = True


def _capture_worker(rank, world, port, out, scenario):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import types
    import torch.distributed as td
    from deep_interpolation_clustering_amd import dist
    from deep_interpolation_clustering_amd.step import Stepper
    dist.init_from_env('gloo')
    try:
        log = {'captures': 0, 'replays_or_eager': 0, 'collectives': []}

        class HostStepper(Stepper):
            def _graphable(self, x):
                return True

            def _warm_up(self, run, device):
                snap = self._snapshot(device)
                for _ in range(2):
                    run()
                self._restore(snap, device)

            def _capture(self, run):
                log['captures'] += 1
                if scenario == 'capture_raises_on_rank_1' and rank == 1:
                    raise RuntimeError('injected: capture failed on this rank')
                return _FakeGraph(run, {}), {}

        inner = td.all_reduce

        def counting(t, *a, **k):
            log['collectives'].append(int(t.numel()))
            return inner(t, *a, **k)
        td.all_reduce = counting
        dist.graph_capturable = lambda: True              # (what RCCL answers)

        def make(use_graphs):
            torch.manual_seed(0)
            net = dist.convert_batchnorm_(_TinyNet())
            args = types.SimpleNamespace(loss='ae_mse', grad_clip=15.0, aux_tasks={}, unsup_aux_tasks={})
            return Stepper.__new__(HostStepper if use_graphs else Stepper), net, args

        def run_steps(use_graphs):
            obj, net, args = make(use_graphs)
            obj.__init__(net, lambda m: torch.optim.Adam(m.parameters(), lr=3e-3, weight_decay=4e-4, amsgrad=True), args, use_graphs=use_graphs)
            g = torch.Generator().manual_seed(11)
            X = torch.randn(300, 8, generator=g)
            # global batches of 100, 99 (a tail the world does not divide: rank 0 gets 49 rows -- a NEW local shape --, rank 1 gets 50 -- the
            # shape it already holds a graph for), 100 again
            per_step = []
            for lo, m in ((0, 100), (100, 99), (200, 100)):
                a, b = dist.shard_bounds(m)
                x = X[lo + a:lo + b]
                n0 = len(log['collectives'])
                obj.step(x, x.clone(), None, None, global_rows=m)
                per_step.append(len(log['collectives']) - n0)
            return obj, per_step

        eager, _ = run_steps(False)
        n_eager_caps = log['captures']
        st, per_step = run_steps(True)
        td.all_reduce = inner
        torch.save({'eager': eager.flat.flat.clone(), 'graphed': st.flat.flat.clone(), 'captures': log['captures'] - n_eager_caps,
                    'per_step': per_step, 'off': st._sharded_capture_off, 'cached': len(st._graphs), 'use_graphs': st.use_graphs},
                   os.path.join(out, f'cap{rank}.pt'))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize('scenario', ['uneven_tail', 'capture_raises_on_rank_1'])
def test_sharded_capture_is_agreed_between_ranks(tmp_path, scenario):
    """(ADVICE r5 / VERDICT r5 #3.)  Three sharded steps through Stepper's graph path on two ranks -- global batches of 100, 99, 100 rows.
    'uneven_tail': the 99-row batch gives rank 0 a new shard shape and rank 1 a cached one; keyed on the GLOBAL batch both ranks warm up and
    capture together (same number of collectives per step on both ranks), and the trajectory is the eager one.  'capture_raises_on_rank_1':
    the capture raises on one rank only; both ranks agree (one MIN all-reduce), drop to the eager step together, finish the same three
    steps with identical parameters."""
    port = 30100 + (os.getpid() % 1000) + (0 if scenario == 'uneven_tail' else 7)
    mp.spawn(_capture_worker, args=(2, port, str(tmp_path), scenario), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f'cap{r}.pt', weights_only=False) for r in (0, 1))
    assert r0['use_graphs'] and r1['use_graphs']
    assert r0['per_step'] == r1['per_step']                               # the ranks issued the same collectives on every step
    assert torch.equal(r0['graphed'], r1['graphed'])                      # replicas stay bit-identical
    assert torch.equal(r0['graphed'], r0['eager'])                        # and on the eager trajectory (the stand-in replay IS an eager step)
    if scenario == 'uneven_tail':
        assert r0['captures'] == r1['captures'] == 2 and r0['cached'] == r1['cached'] == 2 and not r0['off'] and not r1['off']
        # a new key = 2 warm-up steps + the flag + 1 replay, a cached key = 1 replay
        assert r0['per_step'][0] == r0['per_step'][1] == 3 * r0['per_step'][2] + 1
    else:
        assert r0['captures'] == r1['captures'] == 1 and r0['off'] and r1['off'] and r0['cached'] == r1['cached'] == 0
        assert r0['per_step'][1] == r0['per_step'][2]                     # eager from the second step on, on both ranks


def _resolve_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import types
    import torch.distributed as td
    from deep_interpolation_clustering_amd import dist, step
    dist.init_from_env('gloo')
    try:
        calls, inner = [], td.all_reduce

        def counting(t, *a, **k):
            calls.append(int(t.numel()))
            return inner(t, *a, **k)
        td.all_reduce = counting
        res = {}
        # resolve_all_: the riders still waiting travel together, one collective per dtype, each `then` runs once its rider has landed
        a, b, c = torch.tensor([1.0 + rank, 2.0]), torch.tensor([10.0 * (rank + 1)]), torch.tensor([0.5], dtype=torch.float64)
        derived = {}
        dist.deferred_sum_(a, then=lambda: derived.__setitem__('mean', float(a[0] / a[1])))
        dist.deferred_sum_(b, then=lambda: derived.__setitem__('b', float(b)))
        dist.deferred_sum_(c)
        dist.resolve_all_()
        res['a'], res['b'], res['c'], res['derived'], res['calls'] = a.clone(), float(b), float(c), dict(derived), calls[:]
        dist.resolve_all_()                                               # nothing pending: no collective
        res['calls_after'] = len(calls)

        # a loss configuration in which NO term carries the queued reconstruction pair (plain 'ae_mse'): compute_losses resolves it before
        # the value is handed out; the value is NaN until then (what ops._RbfRecLoss does)
        class Net:
            def rec_loss(self, ob, rec_ob, padding_mask, lengths=None):
                pair = torch.tensor([4.0 * (rank + 1), 2.0])              # (SSE, count) of this rank
                mse = torch.full((), float('nan'))
                dist.deferred_sum_(pair, then=lambda: torch.div(pair[0], pair[1], out=mse))
                res['before'] = float(mse)
                return {'loss': mse, 'ae_mse': mse}
        n0 = len(calls)
        losses = step.compute_losses(Net(), types.SimpleNamespace(loss='ae_mse'), None, None, {}, None, None)
        res['mse'], res['n_resolve'] = float(losses['loss']), len(calls) - n0

        # a loss switch that raises issues no collective on its way out (the other rank would not be there to meet it)
        class Bad(Net):
            pass
        n0 = len(calls)
        try:
            step.compute_losses(Bad(), types.SimpleNamespace(loss='no_such_loss'), None, None, {}, None, None)
        except NotImplementedError:
            res['raised'] = True
        res['n_on_error'] = len(calls) - n0
        dist.drop_riders()
        td.all_reduce = inner
        torch.save(res, os.path.join(out, f'res{rank}.pt'))
    finally:
        td.destroy_process_group()


def test_resolve_all_and_completion_callbacks(tmp_path):
    """  dist.resolve_all_ / deferred_sum_(then=...) directly, a loss configuration with no carrier, and no collective while an
    exception unwinds."""
    port = 30500 + (os.getpid() % 1000)
    mp.spawn(_resolve_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in (torch.load(tmp_path / f'res{k}.pt', weights_only=False) for k in (0, 1)):
        assert torch.equal(r['a'], torch.tensor([3.0, 4.0])) and r['b'] == 30.0 and r['c'] == 1.0
        assert r['derived'] == {'mean': 0.75, 'b': 30.0}
        assert r['calls'] == [3, 1] and r['calls_after'] == 2            # the two f32 riders in ONE collective, the f64 one alone
        assert np.isnan(r['before']) and r['mse'] == 3.0 and r['n_resolve'] == 1
        assert r['raised'] and r['n_on_error'] == 0

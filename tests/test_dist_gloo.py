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

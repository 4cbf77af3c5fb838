#!/usr/bin/env python3
"""A/B of the k2 forward (DIC_RBF_FWD_ROW = 0 tile kernel / 1 row-per-wave kernel) and backward variants (csrc/dic_rbf.hip) on the store path with the fused loss -- what the timed step launches -- at BASELINE configs[1]
(C=6, T=96, ~50 obs, B=32768) and configs[3] (C=12, T=288, ~200 obs, B=8192): DIC_RBF_BWD_SLOT = 0 (wave-per-encounter kernel where it applies, else the
tile kernel), 1 (slots-on-lanes kernel where the wave kernel does not apply), 2 (slots-on-lanes everywhere).  Prints time per launch and the largest
difference of dL/dv and dL/dkernel against mode 0 (the variants add the slots of a row in different orders).  usage: python3 scripts/k2_bwd_ab.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from deep_interpolation_clustering_amd import _native as N  # noqa: E402
from deep_interpolation_clustering_amd import synthetic  # noqa: E402
from deep_interpolation_clustering_amd.ragged import RaggedStore  # noqa: E402

L = N.lib()
P = N.ptr
dev = torch.device('cuda', 0)
f32 = dict(device=dev, dtype=torch.float32)
for name, (C, T, lam, B) in {'cfg2': (6, 96, 50.0, 32768), 'cfg4': (12, 288, 200.0, 8192), 'default': (6, 354, 50.0, 4096)}.items():
    R = 6 if name == 'default' else 24
    coh = synthetic.make_cohort(B, C=C, T=T, H=bench.H, lam=lam, G=4, seed=4)
    x_np, _, n = synthetic.stacked_batch(coh)
    lengths = torch.tensor(n, device=dev, dtype=torch.int32)
    stor = RaggedStore(x_np, C, dev)
    perm = torch.randperm(B, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).to(torch.int32)
    lperm = lengths.index_select(0, perm.to(torch.int64)).contiguous()
    torch.manual_seed(3)
    grid = torch.linspace(0, bench.H, R, **f32)
    rk = torch.randn(C, **f32) * 0.3
    v = torch.randn((B, C, R), **f32)
    y, norm = torch.empty((B, C, T), **f32), torch.empty((B, C, T), **f32)
    out2, gl = torch.empty(2, **f32), torch.ones(1, **f32)
    st = N.stream_of(v)
    wsf = torch.empty(max(16, L.dic_rbf_fwd_loss_workspace(B, C, T, R)), dtype=torch.uint8, device=dev)
    N.check(L.dic_rbf_fwd_store(P(stor.t_pk), P(stor.v_pk), P(stor.row_off), P(perm), P(lperm), B, C, T, R, P(grid), P(rk), P(v), 0, 1, P(y), P(norm), 1,
                                P(out2), P(wsf), wsf.numel(), st), 'fwd')
    fres = {}
    for mode in ('0', '1'):
        os.environ['DIC_RBF_FWD_ROW'] = mode
        y.fill_(-7.0); norm.fill_(-7.0)
        ffn = lambda: L.dic_rbf_fwd_store(P(stor.t_pk), P(stor.v_pk), P(stor.row_off), P(perm), P(lperm), B, C, T, R, P(grid), P(rk), P(v), 0, 1, P(y), P(norm), 1,
                                          P(out2), P(wsf), wsf.numel(), st)
        assert ffn() == 0
        fres[mode] = (y.clone(), norm.clone(), out2.clone(), bench.time_kernel(ffn, 20))
    print('%-8s forward: tile kernel %8.1f us, row-per-wave kernel %8.1f us   y equal %s  norm equal %s  sse rel diff %.2e' % (
        name, fres['0'][3] * 1e3, fres['1'][3] * 1e3, bool(torch.equal(fres['0'][0], fres['1'][0])), bool(torch.equal(fres['0'][1], fres['1'][1])),
        float(abs(fres['0'][2][0] - fres['1'][2][0]) / fres['0'][2][0])))
    res = {}
    for mode in ('0', '1', '2'):
        os.environ['DIC_RBF_BWD_SLOT'] = mode
        ws = torch.empty(max(16, L.dic_rbf_bwd_workspace(B, C, T, R)), dtype=torch.uint8, device=dev)
        gv, gk = torch.zeros((B, C, R), **f32), torch.zeros(C, **f32)
        fn = lambda: L.dic_rbf_bwd_store(P(stor.t_pk), P(stor.v_pk), P(stor.row_off), P(perm), P(lperm), B, C, T, R, P(grid), P(rk), P(v), 0, P(y), P(norm),
                                         None, P(out2), P(gl), P(gv), P(gk), P(ws), ws.numel(), st)
        assert fn() == 0
        ms = bench.time_kernel(fn, 20)
        res[mode] = (gv.clone(), gk.clone(), ms)
    g0, k0, _ = res['0']
    for mode in ('0', '1', '2'):
        g, k, ms = res[mode]
        print('%-8s mode %s: %8.1f us   max|dv - dv0| / max|dv0| %.2e   max|dk - dk0| / max|dk0| %.2e' % (
            name, mode, ms * 1e3, float((g - g0).abs().max() / g0.abs().max()), float((k - k0).abs().max() / k0.abs().max())))

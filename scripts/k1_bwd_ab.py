#!/usr/bin/env python3
"""A/B of the k1 backward (csrc/dic_interp.hip) at the reference shape (C=6, R=24, packed bf16 gradient rows, B=32768): DIC_K1_BWD_LANES=0 (tile kernel) against the
grid-points-on-lanes kernel at several workgroups-per-CU settings (DIC_K1_BWD_LANE_WGS).  usage: python3 scripts/k1_bwd_ab.py [B]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from deep_interpolation_clustering_amd import _native as N  # noqa: E402

L, P = N.lib(), N.ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
C, R, dev = 6, 24, 'cuda'
torch.manual_seed(0)
saved = torch.randn(B, 7, C, R, device=dev)
saved[:, 3:] = saved[:, 3:].abs()
sk, ck = torch.randn(C, device=dev) * 0.5, torch.randn(C, C, device=dev) * 0.4
gp = torch.zeros(R, B, 32, device=dev, dtype=torch.bfloat16)
gp[:, :, :3 * C] = torch.randn(R, B, 3 * C, device=dev).to(torch.bfloat16)
ref = None
for lanes, wgs in (('0', '3'), ('1', '1'), ('1', '2'), ('1', '3'), ('1', '4'), ('1', '6'), ('1', '8')):
    os.environ['DIC_K1_BWD_LANES'], os.environ['DIC_K1_BWD_LANE_WGS'] = lanes, wgs
    ws = torch.empty(max(16, L.dic_sci_cci_bwd_workspace(B, C, R)), dtype=torch.uint8, device=dev)
    gs, gc = torch.zeros(C, device=dev), torch.zeros(C, C, device=dev)
    fn = lambda: L.dic_sci_cci_bwd_packed(P(gp), 32, P(saved), P(sk), P(ck), B, C, R, P(gs), P(gc), P(ws), ws.numel(), N.stream_of(gp))
    assert fn() == 0
    ms = bench.time_kernel(fn, 30)
    if ref is None:
        ref = (gs.clone(), gc.clone())
    print('lanes %s  workgroups/CU %s: %7.1f us   max rel diff vs tile kernel: %.2e %.2e' % (
        lanes, wgs, ms * 1e3, float((gs - ref[0]).abs().max() / ref[0].abs().max()), float((gc - ref[1]).abs().max() / ref[1].abs().max())))

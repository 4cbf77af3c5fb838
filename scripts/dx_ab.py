#!/usr/bin/env python3
"""The decoder's input gradient dX = dG . W_ih on the same operands (N = 24 x B rows, K = 1024, 256 columns, bf16): dic_lstm_dx_tile (256 x 256
macro-tiles; the product's only form in the library from round 6 on) beside torch's library GEMM as a yardstick.  usage: python3 scripts/dx_ab.py [B]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from deep_interpolation_clustering_amd import _native as N  # noqa: E402

L, P = N.lib(), N.ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
rows = 24 * B
dev, bf = 'cuda', torch.bfloat16
torch.manual_seed(0)
dg = (torch.randn(rows, 1024, device=dev) * 0.3).to(bf)
w = (torch.randn(1024, 256, device=dev) * 0.06).to(bf)
wt = w.t().contiguous()
dx = torch.empty(rows, 256, device=dev, dtype=bf)
st = N.stream_of(dg)


def lib():
    return dg @ w


def tile():
    N.check(L.dic_lstm_dx_tile(P(dg), P(wt), rows, 1024, 256, P(dx), st), 'dx_tile')


def tile_with_transpose():
    t = w.t().contiguous()
    N.check(L.dic_lstm_dx_tile(P(dg), P(t), rows, 1024, 256, P(dx), st), 'dx_tile')


ref = lib()
tile()
torch.cuda.synchronize()
print('max |tile - lib| = %.3e (max |lib| %.3e)' % (float((dx.float() - ref.float()).abs().max()), float(ref.float().abs().max())))
gb = (rows * 1024 * 2 + rows * 256 * 2) / 1e9
for name, fn in (('library GEMM', lib), ('dx_tile', tile), ('dx_tile + W^T copy', tile_with_transpose), ('library GEMM', lib), ('dx_tile', tile)):
    ms = bench.time_kernel(fn, 20)
    print('%-20s %8.1f us   %.2f TB/s algorithmic   %.0f TFLOP/s' % (name, ms * 1e3, gb / ms, 2.0 * rows * 1024 * 256 / ms / 1e9))

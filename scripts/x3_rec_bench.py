#!/usr/bin/env python3
"""The x3 f32 recurrence kernels alone at the headline shape (R = 24, H = 128; B rows): forward with gx from HBM (decoder), forward with the narrow
input projected in the kernel (encoder), backward.  usage: python3 scripts/x3_rec_bench.py [B]
With DIC_AB_LIB=<second libdic_hip.so> (scripts/two_lib_build.sh) both builds are timed alternately in this one process (boxes drift by several per cent)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from deep_interpolation_clustering_amd import _native as N  # noqa: E402

L, P = N.lib(), N.ptr
LIBS = [('this build', L)]
if os.environ.get('DIC_AB_LIB'):
    LB = C.CDLL(os.path.abspath(os.environ['DIC_AB_LIB']))
    for name, (res, args) in N.SIGNATURES.items():
        if hasattr(LB, name):
            fn = getattr(LB, name)
            fn.restype, fn.argtypes = res, args
    LIBS.append(('DIC_AB_LIB', LB))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
R, H = 24, 128
dev = 'cuda'
torch.manual_seed(0)
f = dict(device=dev, dtype=torch.float32)
gx = torch.randn(R, B, 2, 4, H, **f) * 0.5
whh = torch.randn(2, 4 * H, H, **f) * 0.08
whh_t = whh.transpose(1, 2).contiguous()
x = torch.randn(R, B, 20, **f)
x[..., 18] = 1.0
x[..., 19] = 0.0
wih = torch.randn(8 * H, 20, **f) * 0.2
Bp = (B + 31) // 32 * 32
out_ext = torch.empty(R + 2, B, 2 * H, **f)
out = out_ext[1:R + 1]
hn, cn = torch.empty(B, 2, H, **f), torch.empty(B, 2, H, **f)
gates, cs = torch.empty(R, Bp, 2, 4, H, **f), torch.empty(R + 1, Bp, 2, H, **f)
dout = torch.randn(R, B, 2 * H, **f) * 0.1
dgx = torch.empty(2, R * B, 8 * H, device=dev, dtype=torch.bfloat16)
dh0, dc0, dbias = torch.empty(B, 2, H, **f), torch.empty(B, 2, H, **f), torch.empty(2, 4 * H, **f)
ws = torch.empty(max(16, L.dic_lstm_rec_bwd_workspace(B)), device=dev, dtype=torch.uint8)
st = N.stream_of(gx)


def fwd(L=L):
    N.check(L.dic_lstm_rec_fwd(N.DTYPE_F32X3, P(gx), P(whh), None, None, R, B, H, P(out), P(hn), P(cn), P(gates), P(cs), 3, st), 'fwd')


def fwd_proj(L=L):
    N.check(L.dic_lstm_rec_fwd_proj_x3(P(x), P(wih), 20, P(whh), None, None, R, B, H, P(out), P(hn), P(cn), P(gates), P(cs), 3, st), 'fwd_proj')


def bwd(L=L):
    N.check(L.dic_lstm_rec_bwd(N.DTYPE_F32X3, P(whh_t), 1, P(gates), P(cs), P(dout), None, None, R, B, H, P(dgx), P(dh0), P(dc0), P(dbias), P(ws), ws.numel(), 1, 1, st), 'bwd')


rows = R * B * 2
for name, fn, nbytes in (('rec_fwd8x3 (gx in)', fwd, rows * (2048 + 2048 + 512 + 512)), ('rec_fwd8x3 (proj)', fwd_proj, rows * (2048 + 512 + 512) + R * B * 80),
                         ('rec_bwd8x3', bwd, rows * (2048 + 512 + 512 + 2048))):
    for rnd in range(3 if len(LIBS) > 1 else 1):
        for tag, lib in LIBS:
            f = (lambda lib=lib: fn(lib))
            f()
            torch.cuda.synchronize()
            ms = bench.time_kernel(f, 10)
            print('%-22s %-12s %8.1f us   %.2f TB/s algorithmic (%.2f GB)   %.3f of the 8 TB/s peak' % (name, tag, ms * 1e3, nbytes / ms / 1e9, nbytes / 1e9, nbytes / ms / 1e9 / 8.0),
                  flush=True)

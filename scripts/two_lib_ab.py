#!/usr/bin/env python3
"""Interleaved A/B of one entry point between TWO builds of libdic_hip.so in one process (boxes drift by several per cent within a run, so builds are
timed alternately, several rounds): DIC_AB_LIB=<second .so> python3 scripts/two_lib_ab.py {fwd_proj|fwd_xproj|bwd|dx} [B] [rounds]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from deep_interpolation_clustering_amd import _native as N  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else 'fwd_proj'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
LA = N.lib()
LB = C.CDLL(os.path.abspath(os.environ['DIC_AB_LIB']))
for name, (res, args) in N.SIGNATURES.items():
    if hasattr(LB, name):
        fn = getattr(LB, name)
        fn.restype, fn.argtypes = res, args
R, H = 24, 128
dev, bf, P = 'cuda', torch.bfloat16, N.ptr
torch.manual_seed(0)
Bp = (B + 63) // 64 * 64
out = torch.empty(R + 2, B, 2 * H, device=dev, dtype=bf)
hn, cn = torch.empty(2, B, H, device=dev), torch.empty(2, B, H, device=dev)
gates, cs = torch.empty(R, Bp, 2, 4, H, device=dev, dtype=bf), torch.empty(R + 1, Bp, 2, H, device=dev, dtype=bf)
whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf)
st = N.stream_of(out)
if what == 'fwd_proj' or what == 'bwd':
    x = torch.randn(R, B, 32, device=dev).to(bf)
    wih = (torch.randn(2, 4 * H, 32, device=dev) * 0.1).to(bf)

    def call(L):
        return lambda: N.check(L.dic_lstm_fwd_proj(P(x), P(wih), P(whh), None, None, R, B, H, 32, P(out[1]), None, P(hn), P(cn), P(gates), P(cs), 0, 0, 1, st), what)
else:
    x = (torch.randn(R, B, 256, device=dev) * 0.5).to(bf)
    wih = (torch.randn(8 * H, 256, device=dev) * 0.06).to(bf)
    bias = (torch.randn(8 * H, device=dev) * 0.1).to(bf)

    def call(L):
        return lambda: N.check(L.dic_lstm_fwd_xproj(P(x), P(wih), P(whh), P(bias), None, None, R, B, H, 256, P(out[1]), None, P(hn), P(cn), P(gates), P(cs), 0, 1, st), what)
if what == 'dx':
    rows = R * B
    dg = (torch.randn(rows, 1024, device=dev) * 0.3).to(bf)
    wt = (torch.randn(256, 1024, device=dev) * 0.06).to(bf)
    dxo = torch.empty(rows, 256, device=dev, dtype=bf)

    def call(L):
        return lambda: N.check(L.dic_lstm_dx_tile(P(dg), P(wt), rows, 1024, 256, P(dxo), st), what)
    chk = lambda: (dxo.float().abs().sum().item(),)
elif what == 'bwd':
    # the 64-row backward on the state a forward saved: lstm_bwd8 (dic_lstm_bwd)
    x = torch.randn(R, B, 32, device=dev).to(bf)
    wih = (torch.randn(2, 4 * H, 32, device=dev) * 0.1).to(bf)
    N.check(LA.dic_lstm_fwd_proj(P(x), P(wih), P(whh), None, None, R, B, H, 32, P(out[1]), None, P(hn), P(cn), P(gates), P(cs), 0, 0, 1, st), 'fwd')
    whh_t = whh.transpose(1, 2).contiguous()
    dout = (torch.randn(R, B, 2 * H, device=dev) * 0.1).to(bf)
    dgx = torch.empty(R, B, 2, 4, H, device=dev, dtype=bf)
    dh0, dc0, db = torch.empty(2, B, H, device=dev), torch.empty(2, B, H, device=dev), torch.empty(2, 4 * H, device=dev)
    ws = torch.empty(max(16, LA.dic_lstm_bwd_workspace(B)), dtype=torch.uint8, device=dev)

    def call(L):
        return lambda: N.check(L.dic_lstm_bwd(P(whh_t), P(gates), P(cs), None, P(dout), None, None, R, B, H, P(dgx), P(dh0), P(dc0), P(db), P(ws), ws.numel(), 0, 0, st), what)
    chk = lambda: (dgx.float().abs().sum().item(), dh0.abs().sum().item(), dc0.abs().sum().item())
elif True:
    chk = lambda: (out[1:R + 1].float().abs().sum().item(), gates.float().abs().sum().item(), cs[:R].float().abs().sum().item())
a, b = call(LA), call(LB)
a(); sa = chk()
b(); sb = chk()
print('checksums A', sa, 'B', sb)
ta, tb = [], []
for _ in range(rounds):
    ta.append(bench.time_kernel(a, 10) * 1e3)
    tb.append(bench.time_kernel(b, 10) * 1e3)
print(what, 'A (in-tree lib) us:', ' '.join('%.1f' % t for t in ta), ' median %.1f' % sorted(ta)[len(ta) // 2])
print(what, 'B (DIC_AB_LIB)  us:', ' '.join('%.1f' % t for t in tb), ' median %.1f' % sorted(tb)[len(tb) // 2])

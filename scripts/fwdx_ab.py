#!/usr/bin/env python3
"""The decoder's forward two ways on the same inputs (B = 32768, R = 24, bf16): dic_row_proj + dic_lstm_fwd (gx through HBM) against dic_lstm_fwd_xproj (the
input projection inside the recurrence kernel).  Prints times and the largest differences.  usage: python3 scripts/fwdx_ab.py [B]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from deep_interpolation_clustering_amd import _native as N  # noqa: E402

L, P = N.lib(), N.ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
R, H, I = 24, 128, 256
dev, bf = 'cuda', torch.bfloat16
torch.manual_seed(0)
x = (torch.randn(R, B, I, device=dev) * 0.5).to(bf)
wih = (torch.randn(2 * 4 * H, I, device=dev) * 0.06).to(bf)
whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf)
bias = (torch.randn(2 * 4 * H, device=dev) * 0.1).to(bf)
h0, c0 = torch.randn(2, B, H, device=dev) * 0.1, torch.randn(2, B, H, device=dev) * 0.1
Bp = (B + 63) // 64 * 64
st = N.stream_of(x)


def outputs():
    return dict(out=torch.zeros(R, B, 2 * H, device=dev, dtype=bf), hn=torch.zeros(2, B, H, device=dev), cn=torch.zeros(2, B, H, device=dev),
                gates=torch.zeros(R, Bp, 2, 4, H, device=dev, dtype=bf), cs=torch.zeros(R, Bp, 2, H, device=dev, dtype=bf))


a, b = outputs(), outputs()
gx = torch.empty(R * B, 8 * H, device=dev, dtype=bf)
native = B if B % 64 == 0 else 0


def old():
    N.check(L.dic_row_proj(P(x), P(wih), P(bias), R * B, I, 8 * H, P(gx), native, 1, st), 'row_proj')
    N.check(L.dic_lstm_fwd(P(gx), 2 if native else 0, P(whh), P(h0), P(c0), R, B, H, P(a['out']), None, P(a['hn']), P(a['cn']), P(a['gates']), P(a['cs']), 0, 0, st), 'lstm_fwd')


def new():
    N.check(L.dic_lstm_fwd_xproj(P(x), P(wih), P(whh), P(bias), P(h0), P(c0), R, B, H, I, P(b['out']), None, P(b['hn']), P(b['cn']), P(b['gates']), P(b['cs']), 0, 1, st), 'fwd_xproj')


old(); new()
torch.cuda.synchronize()
for k in a:
    d = (a[k].float() - b[k].float()).abs()
    print('%-6s max|a-b| %.3e  mean %.3e  max|a| %.3e' % (k, float(d.max()), float(d.mean()), float(a[k].float().abs().max())))
print('row_proj + lstm_fwd: %.1f us    lstm_fwd_xproj: %.1f us' % (bench.time_kernel(old, 10) * 1e3, bench.time_kernel(new, 10) * 1e3))
if os.environ.get('FWDX_NOSAVE'):
    def nosave():
        N.check(L.dic_lstm_fwd_xproj(P(x), P(wih), P(whh), P(bias), P(h0), P(c0), R, B, H, I, P(b['out']), None, P(b['hn']), P(b['cn']), None, None, 0, 1, st), 'fwd_xproj')
    print('lstm_fwd_xproj without the saved state: %.1f us' % (bench.time_kernel(nosave, 10) * 1e3))

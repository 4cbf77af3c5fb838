"""Forward recurrence kernels (decoder: lane-native gx; encoder: fused projection) against the number of resident workgroups, as
scripts/lstm_bwd_scale.py does for the backward.  Usage: python3 scripts/lstm_fwd_scale.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from deep_interpolation_clustering_amd import _native as N
if len(sys.argv) > 1:                 # A/B: another build of the library
    N.LIB_PATH = os.path.abspath(sys.argv[1])
L = N.lib()
R, H = 24, 128
dev, bf = torch.device('cuda'), torch.bfloat16
torch.manual_seed(0)
P = N.ptr


def timed(fn, it=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


for B in (1024, 2048, 4096, 8192, 16384, 32768):
    gx = (torch.randn(R * B, 8 * H, device=dev) * 0.5).to(bf)
    x = torch.randn(R, B, 32, device=dev).to(bf); wih = (torch.randn(2, 4 * H, 32, device=dev) * 0.1).to(bf)
    whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf)
    out = torch.empty(R, B, 2 * H, device=dev, dtype=bf); gates = torch.empty(R, B, 2, 4, H, device=dev, dtype=bf)
    cs = torch.empty(R, B, 2, H, device=dev, dtype=bf); hn = torch.empty(2, B, H, device=dev); cn = torch.empty(2, B, H, device=dev)
    st = N.stream_of(x)
    t1 = timed(lambda: N.check(L.dic_lstm_fwd(P(gx), 2, P(whh), None, None, R, B, H, P(out), None, P(hn), P(cn), P(gates), P(cs), 0, 0, st), 'fwd'))
    t2 = timed(lambda: N.check(L.dic_lstm_fwd_proj(P(x), P(wih), P(whh), None, None, R, B, H, 32, P(out), None, P(hn), P(cn), P(gates), P(cs), 0, 0, 1, st), 'proj'))
    wgs = B // 64 * 2
    ps = max(1.0, wgs / 256)
    print('B %6d  workgroups %5d  decoder fwd %7.1f us (%6.1f per pass, %.2f TB/s)   encoder fwd %7.1f us (%6.1f per pass, %.2f TB/s)' %
          (B, wgs, t1, t1 / ps, 2560.0 * 2 * R * B / t1 / 1e6, t2, t2 / ps, 1568.0 * 2 * R * B / t2 / 1e6), flush=True)

"""lstm_bwd time against the number of resident workgroups (B / 64 x 2 directions; 256 = one per CU): separates the per-workgroup chain
from what the workgroups cost each other in the memory system.  Usage: python3 scripts/lstm_bwd_scale.py"""
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
for B in (1024, 2048, 4096, 8192, 16384, 32768):
    whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf); whh_t = whh.transpose(1, 2).contiguous()
    gates = torch.rand(R, B, 2, 4, H, device=dev).to(bf); cs = torch.randn(R, B, 2, H, device=dev).to(bf)
    dout = (torch.randn(R, B, 2 * H, device=dev) * 0.1).to(bf)
    dgx = torch.empty(R, B, 2, 4, H, device=dev, dtype=bf); dh0 = torch.empty(2, B, H, device=dev); dc0 = torch.empty(2, B, H, device=dev)
    db = torch.empty(2, 4 * H, device=dev); ws = torch.empty(max(16, L.dic_lstm_bwd_workspace(B)), dtype=torch.uint8, device=dev)
    st = N.stream_of(dout)
    fn = lambda: N.check(L.dic_lstm_bwd(P(whh_t), P(gates), P(cs), None, P(dout), None, None, R, B, H, P(dgx), P(dh0), P(dc0), P(db), P(ws), ws.numel(), 0, 0, st), 'bwd')
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    wgs = B // 64 * 2
    print('B %6d  workgroups %5d  passes %.2f  %7.1f us  per pass %6.1f us  %.2f TB/s' % (B, wgs, wgs / 256, us, us / max(1.0, wgs / 256), 2560.0 * 2 * R * B / us / 1e6), flush=True)

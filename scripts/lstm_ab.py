"""Timing of the four recurrence launches of a step (B = 32768, R = 24): decoder fwd (lane-native gx), encoder fwd, bwd."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from deep_interpolation_clustering_amd import _native as N
if len(sys.argv) > 1:                 # A/B: another build of the library (scripts: csrc/ab/lib?.so)
    N.LIB_PATH = os.path.abspath(sys.argv[1])
L = N.lib()
R, B, H = 24, 32768, 128
dev, bf = torch.device('cuda'), torch.bfloat16
torch.manual_seed(0)
gx = (torch.randn(R * B, 8 * H, device=dev) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)).to(bf)
x = torch.randn(R, B, 32, device=dev).to(bf); wih = (torch.randn(2, 4 * H, 32, device=dev) * 0.1).to(bf)
whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf); whh_t = whh.transpose(1, 2).contiguous()
out = torch.empty(R, B, 2 * H, device=dev, dtype=bf); outr = torch.empty_like(out); gates = torch.empty(R, B, 2, 4, H, device=dev, dtype=bf)
cs = torch.empty(R, B, 2, H, device=dev, dtype=bf); hn = torch.empty(2, B, H, device=dev); cn = torch.empty(2, B, H, device=dev)
dout = (torch.randn(R, B, 2 * H, device=dev) * 0.1).to(bf)
dgx = torch.empty(R, B, 2, 4, H, device=dev, dtype=bf); dh0 = torch.empty(2, B, H, device=dev); dc0 = torch.empty(2, B, H, device=dev)
db = torch.empty(2, 4 * H, device=dev); ws = torch.empty(max(16, L.dic_lstm_bwd_workspace(B)), dtype=torch.uint8, device=dev)
P, st = N.ptr, N.stream_of(x)


def timed(fn, it=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


for rep in range(2):
    t1 = timed(lambda: N.check(L.dic_lstm_fwd(P(gx), 1 + int(os.environ.get('DIC_FWD8', '1')), P(whh), None, None, R, B, H, P(out), None, P(hn), P(cn), P(gates), P(cs), 0, 0, st), 'fwd'))
    t0 = timed(lambda: N.check(L.dic_lstm_fwd(P(gx), 0, P(whh), None, None, R, B, H, P(out), None, P(hn), P(cn), P(gates), P(cs), 0, 0, st), 'fwd'))
    t2 = timed(lambda: N.check(L.dic_lstm_fwd_proj(P(x), P(wih), P(whh), None, None, R, B, H, 32, P(out), P(outr), P(hn), P(cn), P(gates), P(cs), 0, 0, int(os.environ.get('DIC_FWD8', '1')), st), 'proj'))
    t3 = timed(lambda: N.check(L.dic_lstm_bwd(P(whh_t), P(gates), P(cs), None, P(dout), None, None, R, B, H, P(dgx), P(dh0), P(dc0), P(db), P(ws), ws.numel(), 0, 0, st), 'bwd'))
    print('fwd(native) %6.1f  fwd(rows) %6.1f  fwd_proj %6.1f  bwd %6.1f us   checks %.1f %.1f' % (t1, t0, t2, t3, out.float().abs().sum().item(), dgx.float().abs().sum().item()), flush=True)

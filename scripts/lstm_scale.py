"""Is the recurrence kernel bound per workgroup or by the shared HBM?  One launch with B/64 x 2 workgroups (one per CU up to B = 8192):
the same per-workgroup work on a quarter / half / all of the chip."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from deep_interpolation_clustering_amd import _native as N
L = N.lib()
R, H = 24, 128
dev, bf = torch.device('cuda'), torch.bfloat16
P = N.ptr
for B in (1024, 2048, 4096, 8192, 16384, 32768):
    torch.manual_seed(0)
    gx = (torch.randn(R * B, 8 * H, device=dev) * 0.5).to(bf)
    whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf); whh_t = whh.transpose(1, 2).contiguous()
    out = torch.empty(R, B, 2 * H, device=dev, dtype=bf); gates = torch.empty(R, B, 2, 4, H, device=dev, dtype=bf)
    cs = torch.empty(R, B, 2, H, device=dev, dtype=bf); hn = torch.empty(2, B, H, device=dev); cn = torch.empty(2, B, H, device=dev)
    dout = (torch.randn(R, B, 2 * H, device=dev) * 0.1).to(bf)
    dgx = torch.empty(R, B, 2, 4, H, device=dev, dtype=bf); dh0 = torch.empty(2, B, H, device=dev); dc0 = torch.empty(2, B, H, device=dev)
    db = torch.empty(2, 4 * H, device=dev); ws = torch.empty(max(16, L.dic_lstm_bwd_workspace(B)), dtype=torch.uint8, device=dev)
    st = N.stream_of(gx)

    def timed(fn, it=20):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / it * 1e3
    tf = timed(lambda: N.check(L.dic_lstm_fwd(P(gx), 0, P(whh), None, None, R, B, H, P(out), None, P(hn), P(cn), P(gates), P(cs), 0, 0, st), 'fwd'))
    x32 = torch.randn(R, B, 32, device=dev).to(bf); wih32 = (torch.randn(2, 4 * H, 32, device=dev) * 0.1).to(bf)
    tp = timed(lambda: N.check(L.dic_lstm_fwd_proj(P(x32), P(wih32), P(whh), None, None, R, B, H, 32, P(out), None, P(hn), P(cn), P(gates), P(cs), 0, 0, int(os.environ.get('DIC_FWD8', '1')), st), 'proj'))
    tb = timed(lambda: N.check(L.dic_lstm_bwd(P(whh_t), P(gates), P(cs), None, P(dout), None, None, R, B, H, P(dgx), P(dh0), P(dc0), P(db), P(ws), ws.numel(), 0, 0, st), 'bwd'))
    wgs = B // 64 * 2
    print('B %6d  workgroups %5d (%.2f per CU)  fwd %7.1f us  fwd_proj %7.1f us  bwd %7.1f us   per round of 256: fwd %6.1f proj %6.1f bwd %6.1f' % (B, wgs, wgs / 256, tf, tp, tb, tf / max(1, wgs / 256), tp / max(1, wgs / 256), tb / max(1, wgs / 256)), flush=True)

"""A/B: the decoder's gx between dic_row_proj and dic_lstm_fwd, row-major (LDS-staged tile) against lane-native (B = 32768, R = 24)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from deep_interpolation_clustering_amd import _native as N
if os.environ.get('DIC_AB_LIB'):
    N.LIB_PATH = os.path.abspath(os.environ['DIC_AB_LIB'])
L = N.lib()
R, B, H = 24, int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 128
dev, bf = torch.device('cuda'), torch.bfloat16
x = (torch.randn(R * B, 256, device=dev) * 0.5).clamp_min(0).to(bf)
wih = (torch.randn(8 * H, 256, device=dev) * 0.06).to(bf); bias = (torch.randn(8 * H, device=dev) * 0.1).to(bf)
whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf)
gx = torch.empty(R * B, 8 * H, device=dev, dtype=bf)
out = torch.empty(R, B, 2 * H, device=dev, dtype=bf); gates = torch.empty(R, B, 2, 4, H, device=dev, dtype=bf)
cs = torch.empty(R, B, 2, H, device=dev, dtype=bf); hn = torch.empty(2, B, H, device=dev); cn = torch.empty(2, B, H, device=dev)
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


res = {}
for native in (0, B, 0, B):
    tp = timed(lambda: N.check(L.dic_row_proj(P(x), P(wih), P(bias), R * B, 256, 8 * H, P(gx), native, 0, st), 'row_proj'))
    tf = timed(lambda: N.check(L.dic_lstm_fwd(P(gx), int(native > 0), P(whh), None, None, R, B, H, P(out), None, P(hn), P(cn), P(gates), P(cs), 0, 0, st), 'lstm_fwd'))
    res[native] = out.float().abs().sum().item()
    print('%-11s row_proj %7.1f us   lstm_fwd %7.1f us   sum %7.1f' % ('lane-native' if native else 'row-major', tp, tf, tp + tf), flush=True)
print('checksums', res)

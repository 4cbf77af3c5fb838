"""Time dic_lstm_dw_wide (decoder one-pass weight gradients) alone.  usage: python scripts/dww_timing.py [B] [R]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from deep_interpolation_clustering_amd import _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
R = int(sys.argv[2]) if len(sys.argv) > 2 else 24
H, I = 128, 256
dev, bf = torch.device('cuda'), torch.bfloat16
dg = (torch.randn(R, B, 2, 4 * H, device=dev) * 0.3).to(bf)
out_ext = (torch.randn(R + 2, B, 2 * H, device=dev) * 0.5).to(bf)
x = torch.randn(R, B, I, device=dev).clamp_min(0).to(bf)
grads = [torch.zeros(4 * H, I, device=dev), torch.zeros(4 * H, H, device=dev), torch.zeros(4 * H, device=dev), torch.zeros(4 * H, device=dev)] * 2
grads = [g.clone() for g in grads]
L = N.lib()
ws = torch.empty(L.dic_lstm_dw_wide_workspace(R, B), dtype=torch.uint8, device=dev)
gp, st = N.ptr_array(grads), N.stream_of(dg)
call = lambda: L.dic_lstm_dw_wide(N.ptr(dg), N.ptr(out_ext), N.ptr(x), 0, R, B, H, I, gp, 0, N.ptr(ws), ws.numel(), st)
for _ in range(3): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n): call()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
gb = (dg.numel() + x.numel() + out_ext.numel()) * 2 / 1e9
print('lstm_dw_wide (+finalize) %.1f us   %.2f GB once-through -> %.2f TB/s' % (ms * 1e3, gb, gb / ms))

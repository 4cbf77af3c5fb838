"""Experiment: per-phase cycle stamps of sci_cci_fwd workgroups (library built with -DDIC_K1_EXP_TIMING)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from deep_interpolation_clustering_amd import _native as N, synthetic
from deep_interpolation_clustering_amd.clustering_interp import Net
B = 32768
dev = torch.device('cuda')
coh = synthetic.make_cohort(B, seed=7)
x_np, ob_np, n = synthetic.stacked_batch(coh)
x, lens = torch.tensor(x_np, device=dev), torch.tensor(n, device=dev)
net = Net(bench.make_args(4), dev).to(dev)
L, P = N.lib(), N.ptr
grid = net.sci.grid(); sk, ck = net.sci.kernel.detach(), net.cci.kernel.detach()
out = torch.empty((B, bench.R, 3 * bench.C), device=dev); saved = torch.empty((B, 7, bench.C, bench.R), device=dev)
for _ in range(3):
    L.dic_sci_cci_fwd(P(x), P(lens), B, bench.C, bench.T, bench.R, P(grid), P(sk), P(ck), P(out), P(saved), N.stream_of(x))
torch.cuda.synchronize()
buf = np.zeros((64, 8), dtype=np.uint64)
fn = L.dic_k1_debug_stamps; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
assert fn(buf.ctypes.data) == 0
t = buf[:, :5].astype(np.int64)
t = t[t[:, 0] > 0]
d = np.diff(t, axis=1)
print('%d sampled workgroups; cycles per phase (median / 90th percentile):' % len(t))
for i, name in enumerate(['1 lengths + parameters', '2 stage rows into LDS', '3 streaming passes', '4 epilogue + stores']):
    print('  %-26s %8.0f %8.0f' % (name, np.median(d[:, i]), np.percentile(d[:, i], 90)))
print('  workgroup lifetime         %8.0f' % np.median(t[:, 4] - t[:, 0]))

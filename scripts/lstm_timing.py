"""Experiment: per-phase cycle stamps of one lstm_fwd workgroup (library built with -DDIC_LSTM_EXP_TIMING)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from deep_interpolation_clustering_amd import _native as N
L = N.lib()
proj = len(sys.argv) > 1 and sys.argv[1] == 'proj'
R, B, H = 24, 32768, 128
dev, bf = torch.device('cuda'), torch.bfloat16
gx = (torch.randn(R, B, 2, 4, H, device=dev) * 0.5).to(bf)
x = torch.randn(R, B, 32, device=dev).to(bf); wih = (torch.randn(2, 4 * H, 32, device=dev) * 0.1).to(bf)
whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf)
out = torch.empty(R, B, 2 * H, device=dev, dtype=bf); gates = torch.empty(R, B, 2, 4, H, device=dev, dtype=bf)
cs = torch.empty(R, B, 2, H, device=dev, dtype=bf); hn = torch.empty(2, B, H, device=dev); cn = torch.empty(2, B, H, device=dev)
P, st = N.ptr, N.stream_of(gx)
for _ in range(3):
    if proj: L.dic_lstm_fwd_proj(P(x), P(wih), P(whh), None, None, R, B, H, 32, P(out), None, P(hn), P(cn), P(gates), P(cs), 0, 0, 0, st)
    else: L.dic_lstm_fwd(P(gx), 0, P(whh), None, None, R, B, H, P(out), None, P(hn), P(cn), P(gates), P(cs), 0, 0, st)
torch.cuda.synchronize()
buf = np.zeros((2, 32, 8), dtype=np.uint64)
fn = L.dic_lstm_debug_stamps; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
assert fn(buf.ctypes.data) == 0
t = buf[0, :R, :5].astype(np.int64)
d = np.diff(t, axis=1)                                  # phases within a step
nxt = t[1:, 0] - t[:-1, 4]                              # end barrier -> next step start
print('cycles (shader clock) per phase, median over steps 2..%d of workgroup 7 / wave 0:' % (R - 1))
names = ['LDS init phase', 'half 0: MFMA+math+stores', 'half 1: MFMA+math+stores', 'closing barrier']
for i, n in enumerate(names):
    print('  %-22s %8.0f' % (n, np.median(d[2:, i])))
print('  %-22s %8.0f' % ('loop overhead', np.median(nxt[2:])))
print('  step total             %8.0f' % np.median(t[3:, 0] - t[2:-1, 0]))
if not proj:
    # backward kernel on the state the forward just saved
    whh_t = whh.transpose(1, 2).contiguous(); dout = (torch.randn(R, B, 2 * H, device=dev) * 0.1).to(bf)
    dgx = torch.empty(R, B, 2, 4, H, device=dev, dtype=bf); dh0 = torch.empty(2, B, H, device=dev); dc0 = torch.empty(2, B, H, device=dev)
    db = torch.empty(2, 4 * H, device=dev); ws = torch.empty(max(16, L.dic_lstm_bwd_workspace(B)), dtype=torch.uint8, device=dev)
    for _ in range(3):
        L.dic_lstm_bwd(P(whh_t), P(gates), P(cs), None, P(dout), None, None, R, B, H, P(dgx), P(dh0), P(dc0), P(db), P(ws), ws.numel(), 0, 0, st)
    torch.cuda.synchronize()
    assert fn(buf.ctypes.data) == 0
    t = buf[1, :R, :6].astype(np.int64)
    d = np.diff(t, axis=1)
    print('lstm_bwd, cycles per phase:')
    for i, n in enumerate(['barrier (half 0 published)', 'phase X: MFMA+stores half 0 || math half 1', 'barrier (half 1 published)', 'phase Y: MFMA+stores half 1 || math half 0 (next step)', '-']):
        print('  %-34s %8.0f' % (n, np.median(d[2:, i])))
    print('  step total                         %8.0f' % np.median(t[3:, 0] - t[2:-1, 0]))

#!/usr/bin/env python3
"""Experiment: per-phase cycle stamps of one lstm_dwx3 workgroup (decoder shape), every wave -- a library built with -DDIC_DWX3_EXP_TIMING, e.g. as a second
library:  bash scripts/two_lib_build.sh dic_lstmgrad.hip - "-DDIC_DWX3_EXP_TIMING" build/ab/libdic_dwx3_stamps.so
          DIC_LIB_PATH=build/ab/libdic_dwx3_stamps.so python3 scripts/dwx3_timing.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from deep_interpolation_clustering_amd import _native as N
L, P = N.lib(), N.ptr
R, B, H, I = 24, 32768, 128, 256
dev = 'cuda'
torch.manual_seed(0)
dgx = (torch.randn(2, R * B, 8 * H, device=dev) * 0.1).to(torch.bfloat16)
out_ext = torch.randn(R + 2, B, 2 * H, device=dev) * 0.5
xb = torch.randn(R * B, I, device=dev) * 0.5
shapes = [(4 * H, I), (4 * H, H), (4 * H,), (4 * H,)] * 2
sinks = [torch.zeros(s, device=dev) for s in shapes]
gp = N.ptr_array(sinks)
ws = torch.empty(max(16, L.dic_lstm_dw_x3_workspace(R, B, I)), device=dev, dtype=torch.uint8)
st = N.stream_of(dgx)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(4):
    if it == 1:
        e0.record()
    N.check(L.dic_lstm_dw_x3(P(dgx), dgx.stride(0), P(out_ext), P(xb), I, 1, None, None, R, B, H, I, gp, 0, P(ws), ws.numel(), st), 'dic_lstm_dw_x3')
e1.record()
torch.cuda.synchronize()
print('lstm_dwx3<256>: %.1f us per launch' % (e0.elapsed_time(e1) / 3 * 1e3))
if hasattr(L, 'dic_dwx3_debug_stamps'):
    buf = np.zeros((8, 32, 8), dtype=np.uint64)
    fn = L.dic_dwx3_debug_stamps; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
    assert fn(buf.ctypes.data) == 0
    t = buf[:, :, :6].astype(np.int64)
    names = ['counted wait (loads / DMA of this tile)', 'barrier', 'split h / x -> LDS images', 'requests issued (h / x, DMA)', 'fragments + 36 MFMAs issued']
    print('cycles per phase, median over tiles 8..39 of workgroup (5, 0); columns = waves 0..7')
    d = np.diff(t, axis=2)
    for i, n in enumerate(names):
        print('  %-42s' % n, ' '.join('%6.0f' % np.median(d[w, :, i]) for w in range(8)))
    print('  %-42s' % 'tile total', ' '.join('%6.0f' % np.median(t[w, 1:, 0] - t[w, :-1, 0]) for w in range(8)))

#!/usr/bin/env python3
"""Experiment: per-phase cycle stamps of one lstm_fwdx8 workgroup, every wave (library built with -DDIC_FWDX_EXP_TIMING):
    make -C deep_interpolation_clustering_amd/csrc CXXFLAGS="... -DDIC_FWDX_EXP_TIMING"; python3 scripts/fwdx_timing.py [nosave]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from deep_interpolation_clustering_amd import _native as N
# the stamped build may be a second library (scripts/two_lib_build.sh dic_lstm32.hip - "-DDIC_FWDX_EXP_TIMING" <out.so>): the tree stays as it is
if os.environ.get('DIC_AB_LIB'):
    N.LIB_PATH = os.path.abspath(os.environ['DIC_AB_LIB'])
L, P = N.lib(), N.ptr
nosave = len(sys.argv) > 1 and sys.argv[1] == 'nosave'
R, B, H, I = 24, 32768, 128, 256
dev, bf = 'cuda', torch.bfloat16
torch.manual_seed(0)
x = (torch.randn(R, B, I, device=dev) * 0.5).to(bf)
wih = (torch.randn(8 * H, I, device=dev) * 0.06).to(bf); whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf)
bias = (torch.randn(8 * H, device=dev) * 0.1).to(bf)
Bp = (B + 63) // 64 * 64
out = torch.empty(R, B, 2 * H, device=dev, dtype=bf); hn = torch.empty(2, B, H, device=dev); cn = torch.empty(2, B, H, device=dev)
gates = torch.empty(R, Bp, 2, 4, H, device=dev, dtype=bf); cs = torch.empty(R, Bp, 2, H, device=dev, dtype=bf)
st = N.stream_of(x)
for _ in range(3):
    N.check(L.dic_lstm_fwd_xproj(P(x), P(wih), P(whh), P(bias), None, None, R, B, H, I, P(out), None, P(hn), P(cn), None if nosave else P(gates),
                                 None if nosave else P(cs), 0, 1, st), 'fwd_xproj')
torch.cuda.synchronize()
buf = np.zeros((8, 32, 8), dtype=np.uint64)
fn = L.dic_fwdx_debug_stamps; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
assert fn(buf.ctypes.data) == 0
t = buf[:, :R, :6].astype(np.int64)
names = ['bias + 32 projection MFMAs issued', '16 recurrent MFMAs issued', 'x tile staged / next loads issued', 'gate arithmetic + stores issued', 'closing barrier']
print('cycles per phase (%s), median over steps 3..%d, workgroup (7, 0); rows = waves 0..7' % ('no saved state' if nosave else 'with saved state', R - 1))
d = np.diff(t, axis=2)
for i, n in enumerate(names):
    print('  %-36s' % n, ' '.join('%6.0f' % np.median(d[w, 3:, i]) for w in range(8)))
print('  %-36s' % 'step total', ' '.join('%6.0f' % np.median(t[w, 4:, 0] - t[w, 3:-1, 0]) for w in range(8)))
print('  %-36s' % 'start skew vs wave 0', ' '.join('%6.0f' % np.median(t[w, 3:, 0] - t[0, 3:, 0]) for w in range(8)))

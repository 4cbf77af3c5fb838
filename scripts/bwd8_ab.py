"""A/B of the 64-row bf16 backward recurrence: eight waves per workgroup against four (DIC_BWD_EIGHT_WAVES=0), same inputs, odd shapes.
python3 scripts/bwd8_ab.py run <out.pt>  (one mode per process)  |  python3 scripts/bwd8_ab.py cmp a.pt b.pt"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

if sys.argv[1] == 'cmp':
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = 0
    for key in a:
        for k in a[key]:
            x, y = a[key][k].float(), b[key][k].float()
            eq = bool(torch.equal(a[key][k], b[key][k]))
            d = (x - y).abs().max().item()
            tol = 1e-5 * max(1.0, x.abs().max().item()) if k == 'db' else 0.0
            if d > tol:
                bad += 1
            print('%-14s %-4s max|a-b| %.3e  max|a| %.3e  equal %s' % (key, k, d, x.abs().max().item(), eq))
    sys.exit(1 if bad else 0)

from deep_interpolation_clustering_amd import _native as N
L = N.lib()
H = 128
dev, bf = torch.device('cuda'), torch.bfloat16
P = N.ptr
res = {}
for R, B, relu, with_c0 in ((1, 64, 0, 0), (2, 100, 1, 1), (3, 4160, 0, 0), (24, 8192 + 40, 1, 1), (5, 31, 0, 1)):
    torch.manual_seed(R * 100000 + B)
    Bp = (B + 63) // 64 * 64
    whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf); whh_t = whh.transpose(1, 2).contiguous()
    gates = torch.rand(R, Bp, 2, 4, H, device=dev).to(bf); cs = torch.randn(R, Bp, 2, H, device=dev).to(bf)
    c0 = (torch.randn(2, B, H, device=dev) * 0.3) if with_c0 else None
    dout = (torch.randn(R, B, 2 * H, device=dev) * 0.1).to(bf); dhn = torch.randn(2, B, H, device=dev) * 0.1; dcn = torch.randn(2, B, H, device=dev) * 0.1
    dgx = torch.zeros(R, B, 2, 4, H, device=dev, dtype=bf); dh0 = torch.zeros(2, B, H, device=dev); dc0 = torch.zeros(2, B, H, device=dev)
    db = torch.zeros(2, 4 * H, device=dev); ws = torch.empty(max(16, L.dic_lstm_bwd_workspace(B)), dtype=torch.uint8, device=dev)
    N.check(L.dic_lstm_bwd(P(whh_t), P(gates), P(cs), P(c0), P(dout), P(dhn), P(dcn), R, B, H, P(dgx), P(dh0), P(dc0), P(db), P(ws), ws.numel(), 0, relu,
                           N.stream_of(dout)), 'bwd')
    torch.cuda.synchronize()
    res['R%d_B%d' % (R, B)] = dict(dgx=dgx.cpu(), dh0=dh0.cpu(), dc0=dc0.cpu(), db=db.cpu())
torch.save(res, sys.argv[2])

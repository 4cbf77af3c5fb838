"""A/B of the small-batch bf16 recurrence: eight waves per 32-row tile against four (DIC_REC_EIGHT_WAVES=0), same inputs.
python3 scripts/rec8_ab.py run <out.pt>   (one mode per process: the switch is read once)  |  python3 scripts/rec8_ab.py cmp a.pt b.pt"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

if sys.argv[1] == 'cmp':
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for B in a:
        for k in a[B]:
            if k.startswith('t_'):
                continue
            x, y = a[B][k].float(), b[B][k].float()
            d = (x - y).abs().max().item()
            print('%-10s %-6s max|a-b| %.3e  max|a| %.3e  equal %s' % (B, k, d, x.abs().max().item(), bool(torch.equal(a[B][k], b[B][k]))))
        print('%-10s  fwd %.1f vs %.1f us   bwd %.1f vs %.1f us' % (B, a[B]['t_fwd'], b[B]['t_fwd'], a[B]['t_bwd'], b[B]['t_bwd']))
    sys.exit(0)

from deep_interpolation_clustering_amd import _native as N
L = N.lib()
H = 128
dev, bf = torch.device('cuda'), torch.bfloat16
P = N.ptr
res = {}
SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ['REC_AB_SHAPES'].split(',')] if os.environ.get('REC_AB_SHAPES') else ((24, 256), (24, 300), (24, 1024), (24, 4096), (1, 40), (2, 96))
for R, B in SHAPES:
    torch.manual_seed(B)
    gx = (torch.randn(R, B, 2, 4, H, device=dev) * 0.5).to(bf)
    whh = (torch.randn(2, 4 * H, H, device=dev) * 0.08).to(bf); whh_t = whh.transpose(1, 2).contiguous()
    h0 = torch.randn(2, B, H, device=dev) * 0.1; c0 = torch.randn(2, B, H, device=dev) * 0.1
    Bp = (B + 31) // 32 * 32
    out = torch.zeros(R, B, 2 * H, device=dev, dtype=bf); hn = torch.zeros(2, B, H, device=dev); cn = torch.zeros(2, B, H, device=dev)
    gates = torch.zeros(R, Bp, 2, 4, H, device=dev, dtype=bf); cs = torch.zeros(R + 1, Bp, 2, H, device=dev, dtype=bf)
    st = N.stream_of(gx)
    fwd = lambda: N.check(L.dic_lstm_rec_fwd(N.DTYPE_BF16, P(gx), P(whh), P(h0), P(c0), R, B, H, P(out), P(hn), P(cn), P(gates), P(cs), 0, st), 'fwd')
    dout = (torch.randn(R, B, 2 * H, device=dev) * 0.1).to(bf); dhn = torch.randn(2, B, H, device=dev) * 0.1; dcn = torch.randn(2, B, H, device=dev) * 0.1
    dgx = torch.zeros(R, B, 2, 4, H, device=dev, dtype=bf); dh0 = torch.zeros(2, B, H, device=dev); dc0 = torch.zeros(2, B, H, device=dev)
    db = torch.zeros(2, 4 * H, device=dev); ws = torch.empty(max(16, L.dic_lstm_rec_bwd_workspace(B)), dtype=torch.uint8, device=dev)
    bwd = lambda: N.check(L.dic_lstm_rec_bwd(N.DTYPE_BF16, P(whh_t), 1, P(gates), P(cs), P(dout), P(dhn), P(dcn), R, B, H, P(dgx), P(dh0), P(dc0), P(db),
                                             P(ws), ws.numel(), 0, 1, st), 'bwd')

    def timed(fn, it=30):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / it * 1e3
    tf = timed(fwd); tb = timed(bwd)
    res['R%d_B%d' % (R, B)] = dict(out=out.cpu(), hn=hn.cpu(), cn=cn.cpu(), gates=gates.cpu(), cs=cs.cpu(), dgx=dgx.cpu(), dh0=dh0.cpu(), dc0=dc0.cpu(), db=db.cpu(), t_fwd=tf, t_bwd=tb)
torch.save(res, sys.argv[2])

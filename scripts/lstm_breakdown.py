import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
H, R, B = 128, 24, int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev, bf = torch.device('cuda'), torch.bfloat16
def tm(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, I in (('enc', 18), ('dec', 256)):
    Ip = (I + 15) // 16 * 16
    x = torch.randn(R, B, I, device=dev)
    xb = torch.nn.functional.pad(x.to(bf), (0, Ip - I)).contiguous()
    wih = (torch.randn(8 * H, Ip, device=dev) * 0.05).to(bf)
    bias = torch.zeros(8 * H, device=dev, dtype=bf)
    dg2 = (torch.randn(R * B, 8 * H, device=dev) * 0.01).to(bf)
    out = torch.randn(R, B, 2 * H, device=dev).to(bf)
    hprev = torch.empty((R, B, 2, H), device=dev, dtype=bf)
    dg3 = dg2.view(R, B, 8 * H).transpose(1, 2)
    print(name, 'to_bf16+pad   %.3f' % tm(lambda: torch.nn.functional.pad(x.to(bf), (0, Ip - I)).contiguous()))
    print(name, 'gx addmm      %.3f' % tm(lambda: torch.addmm(bias, xb.view(R * B, Ip), wih.t())))
    print(name, 'dx mm         %.3f' % tm(lambda: (dg2 @ wih)))
    print(name, 'dx slice+cast %.3f' % tm(lambda: (dg2 @ wih)[:, :I].reshape(R, B, I).to(torch.float32)))
    print(name, 'dw_ih bmm     %.3f' % tm(lambda: torch.bmm(dg3, xb).float().sum(0)))
    def mk():
        o4 = out.view(R, B, 2, H)
        hprev[1:, :, 0] = o4[:-1, :, 0]; hprev[:-1, :, 1] = o4[1:, :, 1]
        hprev[0, :, 0].zero_(); hprev[R - 1, :, 1].zero_()
    print(name, 'hprev build   %.3f' % tm(mk))
    print(name, 'dw_hh bmm     %.3f' % tm(lambda: torch.bmm(dg3, hprev.view(R, B, 2 * H)).float().sum(0)))
    print(name, 'bias sum      %.3f' % tm(lambda: torch.sum(dg2, dim=0, dtype=torch.float32)))
    print(name, 'whh_t         %.3f' % tm(lambda: wih.view(2, 4 * H, Ip).transpose(1, 2).contiguous()))

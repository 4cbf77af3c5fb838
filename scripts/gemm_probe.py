import torch, sys
R, B, H = 24, int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 128
dev, bf = torch.device('cuda'), torch.bfloat16
def tm(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
dg = (torch.randn(R, B, 8 * H, device=dev) * 0.01).to(bf)
x = torch.randn(R, B, 256, device=dev).to(bf)
dgT = dg.transpose(1, 2).contiguous()
xT = x.transpose(1, 2).contiguous()
gf = 2 * R * B * 1024 * 256 / 1e9
for name, fn in [
    ('TN view bmm(dg^T view, x)      ', lambda: torch.bmm(dg.transpose(1, 2), x)),
    ('NN bmm(dgT contig, x)          ', lambda: torch.bmm(dgT, x)),
    ('NT bmm(dgT contig, xT^T view)  ', lambda: torch.bmm(dgT, xT.transpose(1, 2))),
    ('transpose copy dg              ', lambda: dg.transpose(1, 2).contiguous()),
    ('single mm TN (RB flattened)    ', lambda: dg.view(R * B, -1).t() @ x.view(R * B, -1)),
    ('chunk 8192: bmm TN             ', lambda: torch.bmm(dg.view(R * B // 8192, 8192, -1).transpose(1, 2), x.view(R * B // 8192, 8192, -1))),
    ('chunk 65536: bmm TN            ', lambda: torch.bmm(dg.view(R * B // 65536, 65536, -1).transpose(1, 2), x.view(R * B // 65536, 65536, -1))),
    ('f32-out baddbmm TN             ', lambda: torch.bmm(dg.transpose(1, 2), x).float().sum(0)),
]:
    ms = tm(fn)
    print(f'{name} {ms:7.3f} ms  {gf / ms:8.1f} TFLOP/s' if 'copy' not in name else f'{name} {ms:7.3f} ms')
print('--- chunk sweep, dW_ih (1024 x 256) and per-direction dW_hh (512 x 128)')
hp = torch.randn(R, B, 2, H, device=dev).to(bf)
dg5 = dg.view(R, B, 2, 4 * H)
for c in (1024, 2048, 4096, 8192, 16384):
    n = R * B // c
    ms = tm(lambda: torch.bmm(dg.view(n, c, -1).transpose(1, 2), x.view(n, c, -1)).float().sum(0))
    d0 = dg5[:, :, 0].reshape(R * B, 4 * H)    # contiguous copy just for the probe
    h0 = hp[:, :, 0].reshape(R * B, H)
    ms2 = tm(lambda: torch.bmm(d0.view(n, c, -1).transpose(1, 2), h0.view(n, c, -1)).float().sum(0))
    ms3 = tm(lambda: torch.bmm(dg.view(n, c, -1).transpose(1, 2), hp.view(n, c, 2 * H)).float().sum(0))
    print(f'chunk {c:6d}: dW_ih {ms:6.3f} ms ({gf / ms:6.0f} TF/s)   dW_hh one dir {ms2:6.3f} ms   dW_hh full 8Hx2H {ms3:6.3f} ms')

"""dW_hh per direction from shifted strided views of dG / out (no H_prev copy) vs the full 8Hx2H product."""
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
out = torch.randn(R, B, 2 * H, device=dev).to(bf)
def full():
    hprev = torch.empty((R, B, 2, H), device=dev, dtype=bf)
    o4 = out.view(R, B, 2, H)
    hprev[1:, :, 0] = o4[:-1, :, 0]; hprev[:-1, :, 1] = o4[1:, :, 1]
    hprev[0, :, 0].zero_(); hprev[R - 1, :, 1].zero_()
    n = R * B // 8192
    f = torch.bmm(dg.view(n, 8192, -1).transpose(1, 2), hprev.view(n, 8192, 2 * H)).float().sum(0)
    return torch.stack([f[:4 * H, :H], f[4 * H:, H:]])
def perdir(c):
    a0 = dg[1:].view(-1, 8 * H)[:, :4 * H]; b0 = out[:-1].view(-1, 2 * H)[:, :H]
    a1 = dg[:-1].view(-1, 8 * H)[:, 4 * H:]; b1 = out[1:].view(-1, 2 * H)[:, H:]
    n = a0.shape[0] // c
    r0 = torch.bmm(a0.unflatten(0, (n, c)).transpose(1, 2), b0.unflatten(0, (n, c))).float().sum(0)
    r1 = torch.bmm(a1.unflatten(0, (n, c)).transpose(1, 2), b1.unflatten(0, (n, c))).float().sum(0)
    return torch.stack([r0, r1])
ref = full()
print('full 8Hx2H + hprev copy: %.3f ms' % tm(full))
for c in (2048, 4096, 8192, 16384):
    if ((R - 1) * B) % c: continue
    got = perdir(c)
    err = float((got - ref).abs().max() / ref.abs().max())
    print('per-direction chunk %5d: %.3f ms   rel err vs full %.2e' % (c, tm(lambda: perdir(c)), err))
print('--- f32 output from bmm')
a = dg.view(R * B // 8192, 8192, -1).transpose(1, 2); b = out.view(R * B // 8192, 8192, -1)
ref64 = torch.bmm(a.double(), b.double()).sum(0)
r_bf = torch.bmm(a, b).float().sum(0)
print('bf16-out chunks: %.3f ms  rel err vs f64 %.2e' % (tm(lambda: torch.bmm(a, b).float().sum(0)), float((r_bf - ref64).abs().max() / ref64.abs().max())))
try:
    r32 = torch.bmm(a, b, out_dtype=torch.float32).sum(0)
    print('f32-out chunks:  %.3f ms  rel err vs f64 %.2e' % (tm(lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0)), float((r32 - ref64).abs().max() / ref64.abs().max())))
except Exception as e:
    print('out_dtype unsupported:', repr(e)[:200])

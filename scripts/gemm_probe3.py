"""Per-direction dW_hh (512 x 128 output, K = 23*B rows of strided views) under TunableOp: chunk size and operand order."""
import os, sys
os.environ['PYTORCH_TUNABLEOP_ENABLED'] = '1'
os.environ['PYTORCH_TUNABLEOP_TUNING'] = '1'
os.environ['PYTORCH_TUNABLEOP_FILENAME'] = '/tmp/probe3.csv'
os.environ['PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS'] = '20'
import torch
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
dg = (torch.randn(R * B, 8 * H, device=dev) * 0.01).to(bf)
out = torch.randn(R * B, 2 * H, device=dev).to(bf)
a0, b0 = dg[B:, :4 * H], out[:-B, :H]
for c in (1024, 2048, 4096, 8192, 16384):
    n = a0.shape[0]
    if n % c: continue
    A, Bm = a0.unflatten(0, (n // c, c)), b0.unflatten(0, (n // c, c))
    t1 = tm(lambda: torch.bmm(A.transpose(1, 2), Bm).float().sum(0))
    t2 = tm(lambda: torch.bmm(Bm.transpose(1, 2), A).float().sum(0))
    print('chunk %5d: a^T b %.3f ms   b^T a %.3f ms' % (c, t1, t2), flush=True)
# both directions in one bmm over a (R*B, 1024) x (R*B, 256) product restricted to diagonal blocks is not expressible; full product:
hp = torch.randn(R * B, 2 * H, device=dev).to(bf)
for c in (4096, 8192):
    n = R * B
    t = tm(lambda: torch.bmm(dg.unflatten(0, (n // c, c)).transpose(1, 2), hp.unflatten(0, (n // c, c))).float().sum(0))
    print('full 8Hx2H chunk %5d: %.3f ms' % (c, t), flush=True)

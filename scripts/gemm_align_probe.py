"""Is the decoder's input-projection GEMM (786432 x 256 . 256 x 1024 -> bf16) sensitive to where its operands sit?
Times torch.addmm with the output (and the input) placed at different offsets inside one arena.  usage: python scripts/gemm_align_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
dev, bf = torch.device('cuda'), torch.bfloat16
M, K, N = 786432, 256, 1024
arena = torch.empty(6 * 1024 ** 3, dtype=torch.uint8, device=dev)
w = (torch.randn(N, K, device=dev) * 0.05).to(bf)
b = torch.randn(N, device=dev).to(bf)


def view(off, rows, cols):
    return arena[off:off + rows * cols * 2].view(bf).view(rows, cols)


def timed(x, out):
    for _ in range(2):
        torch.addmm(b, x, w.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.addmm(b, x, w.t(), out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


xs = M * K * 2
x = view(0, M, K); x.copy_(torch.randn(M, K, device=dev).clamp_min(0).to(bf))
for off in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 16 << 20, (16 << 20) + 65536, 64 << 20, 256 << 20, 1 << 30):
    base = ((xs + (2 << 20) - 1) // (2 << 20)) * (2 << 20)          # first 2-MiB boundary behind x
    out = view(base + off, M, N)
    print('out at x_end_aligned + %10d B: %7.1f us' % (off, timed(x, out) * 1e3))

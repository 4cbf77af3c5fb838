#!/usr/bin/env python3
"""Time dic_gemm_nt / dic_gemm_tn on the shapes of the f32 'x3' step at the headline batch (R*B = 24 x 32768 rows): per launch, the bf16 MFMA work
(3 x the useful flops) against the 2.5 PF/s peak and the operand bytes against the 8 TB/s HBM peak.  usage: python3 scripts/gemm_x3_shapes.py [B]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

from deep_interpolation_clustering_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
M = 24 * B
dev = torch.device('cuda', 0)


def timeit(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


print(f'M = {M}')
for name, K, N in (('gx enc (x.W_ih^T)', 20, 1024), ('gx dec', 256, 1024), ('fc fwd', 256, 128), ('dX dec (dG.W_ih)', 1024, 256),
                   ('dX enc', 1024, 20), ('dX fc', 128, 256)):
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.1
    y = torch.empty(M, N, device=dev)
    ms = timeit(lambda: ops.gemm_nt(a, w, out=y))
    fl, by = 2.0 * M * K * N, 4.0 * (M * K + M * N)
    print(f'gemm_nt {name:20s} K={K:5d} N={N:5d}: {ms:7.3f} ms  {3 * fl / ms / 1e9:7.1f} TF bf16 ({3 * fl / ms / 1e9 / 2500:.3f} of peak)  {by / ms / 1e6:7.1f} GB/s min-traffic')
    del a, w, y
for name, N in (('gx dec (x3_row_proj)', 1024), ('fc fwd (x3_row_proj)', 128)):
    a = torch.randn(M, 256, device=dev)
    w = torch.randn(N, 256, device=dev) * 0.1
    ms = timeit(lambda: ops.x3_row_proj(a, w))
    fl, by = 2.0 * M * 256 * N, 4.0 * (M * 256 + M * N)
    print(f'x3_row_proj {name:20s} N={N:5d}: {ms:7.3f} ms  {3 * fl / ms / 1e9:7.1f} TF bf16 ({3 * fl / ms / 1e9 / 2500:.3f} of peak)  {by / ms / 1e6:7.1f} GB/s min-traffic')
    del a, w
for name, N, K, K2 in (('dW enc dir', 512, 20, 128), ('dW dec dir', 512, 256, 128), ('dW fc', 128, 256, 0)):
    a = torch.randn(M, 2 * N if N == 512 else N, device=dev)[:, :N]
    x = torch.randn(M, K, device=dev)
    h = torch.randn(M, K2, device=dev) if K2 else None
    d1, d2 = torch.empty(N, K, device=dev), (torch.empty(N, K2, device=dev) if K2 else None)
    ms = timeit(lambda: ops.gemm_tn_into(a, x, d1, x2=h, dst2=d2))
    fl, by = 2.0 * M * N * (K + K2), 4.0 * M * (N + K + K2)
    print(f'gemm_tn {name:20s} N={N:5d} K={K}+{K2}: {ms:7.3f} ms  {3 * fl / ms / 1e9:7.1f} TF bf16 ({3 * fl / ms / 1e9 / 2500:.3f} of peak)  {by / ms / 1e6:7.1f} GB/s min-traffic')
    del a, x, h, d1, d2

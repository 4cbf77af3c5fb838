#!/usr/bin/env python3
"""Launch each hand-written kernel a few times at the bench shape: the target of
`rocprofv3 --kernel-trace --stats` / `--pmc` runs.
Usage: python3 scripts/kbench.py [batch] [iters] [K] [C T lam]     (cfg4 stress: 8192 10 16 12 288 200)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from deep_interpolation_clustering_amd import synthetic  # noqa: E402
from deep_interpolation_clustering_amd.clustering_interp import Net  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
if len(sys.argv) > 6:
    bench.C, bench.T, bench.LAM = int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
dev = torch.device('cuda', 0)
coh = synthetic.make_cohort(B, C=bench.C, T=bench.T, H=bench.H, lam=bench.LAM, G=K, seed=7)
x_np, ob_np, n = synthetic.stacked_batch(coh)
x, ob, lens = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
net = Net(bench.make_args(K), dev).to(dev)
table = bench.kernel_table(net, x, ob, lens, K, iters, with_lstm='nolstm' not in sys.argv,
                           input_path='store' if 'store' in sys.argv else ('dense' if 'dense' in sys.argv else 'both'))
# ('store' / 'dense' on the command line: the PMC passes cannot tell the two input paths of k1 / k2 apart by kernel name -- launch one of them only)
for k, v in table.items():
    print(f'{k:16s} {v["ms"] * 1e3:9.1f} us  {v["GBps"]:8.1f} GB/s  {100 * v["frac_hbm_peak"]:5.1f}% of HBM peak')

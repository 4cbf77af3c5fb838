#!/usr/bin/env python3
"""Kernel list of one bf16 joint step at a small batch (default: the reference's own 256, p1_pretrain_main.py:43): launches per step and GPU time
per kernel, from the tracer behind torch.profiler.  usage: python3 scripts/small_batch_trace.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from deep_interpolation_clustering_amd import synthetic  # noqa: E402
from deep_interpolation_clustering_amd.clustering_interp import Net  # noqa: E402
from deep_interpolation_clustering_amd.step import Stepper  # noqa: E402
from deep_interpolation_clustering_amd.utils import pytorch_optimizer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda', 0)
coh = synthetic.make_cohort(B, seed=5)
x_np, ob_np, n = synthetic.stacked_batch(coh)
x, ob, ln = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
args = bench.make_args(4)
torch.manual_seed(1)
net = Net(args, dev).to(dev)
net.train()
st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16, use_graphs=False)
for _ in range(5):
    st.step(x, ob, None, ln)
kernels, groups = bench.step_trace(lambda i: st.step(x, ob, None, ln), 0, 3)
print(groups)
for k, v in kernels.items():
    print('%6.2f x %8.2f us  %s' % (v['launches_per_step'], v['us_per_launch'], k[:110]))
print('launches', sum(v['launches_per_step'] for v in kernels.values()), 'kernel us', sum(v['ms_per_step'] for v in kernels.values()) * 1e3)

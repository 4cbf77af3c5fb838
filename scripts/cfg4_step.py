#!/usr/bin/env python3
"""The joint step at BASELINE configs[3]'s shape (C=12, T=288, ~200 obs/channel, K=16, batch 8192, bf16 mode, ragged store, shuffled index) a few
times: the target of `rocprofv3 --kernel-trace --stats` (profiles/r4_cfg4_step_kernel_stats.csv).  usage: python3 scripts/cfg4_step.py [steps] [batch]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from deep_interpolation_clustering_amd import synthetic  # noqa: E402
from deep_interpolation_clustering_amd.clustering_interp import Net  # noqa: E402
from deep_interpolation_clustering_amd.ragged import RaggedBatch, RaggedStore  # noqa: E402
from deep_interpolation_clustering_amd.step import Stepper  # noqa: E402
from deep_interpolation_clustering_amd.utils import pytorch_optimizer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device('cuda', 0)
C4, T4, K4 = 12, 288, 16
coh = synthetic.make_cohort(batch, C=C4, T=T4, H=bench.H, lam=200.0, G=K4, seed=4)
x_np, _, _ = synthetic.stacked_batch(coh)
a = bench.make_args(K4)
a.num_variables, a.num_timestamps = C4, T4
torch.manual_seed(1234)
net = Net(a, dev).to(dev)
net.train()
st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), a, autocast_dtype=torch.bfloat16)
store = RaggedStore(x_np, C4, dev)
rb = RaggedBatch(store, torch.randperm(batch, device=dev, generator=torch.Generator(device=dev).manual_seed(2)).to(torch.int32))
for _ in range(3):
    st.step(rb, None, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    losses, _, _ = st.step(rb, None, None)
e1.record()
torch.cuda.synchronize()
print(f'cfg4 step: {e0.elapsed_time(e1) / steps:.3f} ms at batch {batch}, loss {float(losses["loss"]):.5f}')

#!/usr/bin/env python3
"""Which torch operators (not the package's HIP kernels) launch GPU work inside one bf16 joint step, with their input shapes: python3 scripts/torch_glue_trace.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from deep_interpolation_clustering_amd import synthetic  # noqa: E402
from deep_interpolation_clustering_amd.clustering_interp import Net  # noqa: E402
from deep_interpolation_clustering_amd.ragged import RaggedBatch, RaggedStore  # noqa: E402
from deep_interpolation_clustering_amd.step import Stepper  # noqa: E402
from deep_interpolation_clustering_amd.utils import pytorch_optimizer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device('cuda', 0)
coh = synthetic.make_cohort(B, seed=5)
x_np, _, _ = synthetic.stacked_batch(coh)
args = bench.make_args(4)
torch.manual_seed(1)
net = Net(args, dev).to(dev)
net.train()
st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16, use_graphs=False)
store = RaggedStore(x_np, 6, dev)
rb = RaggedBatch(store, torch.randperm(B, device=dev).to(torch.int32))
for _ in range(4):
    st.step(rb, None, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    st.step(rb, None, None)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    dt = getattr(ev, 'self_device_time_total', None)
    if dt is None:
        dt = getattr(ev, 'self_cuda_time_total', 0)
    if dt > 0 and ev.key.startswith('aten::'):
        rows.append((dt, ev.count, ev.key, str(ev.input_shapes)[:150]))
for dt, n, k, sh in sorted(rows, reverse=True)[:30]:
    print('%8.1f us  x%-3d %-28s %s' % (dt, n, k, sh))

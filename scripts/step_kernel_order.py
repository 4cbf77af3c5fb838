#!/usr/bin/env python3
"""The kernels of ONE eager joint step in launch order (name, us), at a given batch -- where the launches of a small-batch step come from.
Usage: python3 scripts/step_kernel_order.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from torch.autograd import DeviceType  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from deep_interpolation_clustering_amd import synthetic  # noqa: E402
from deep_interpolation_clustering_amd.clustering_interp import Net  # noqa: E402
from deep_interpolation_clustering_amd.ragged import RaggedBatch, RaggedStore  # noqa: E402
from deep_interpolation_clustering_amd.step import Stepper  # noqa: E402
from deep_interpolation_clustering_amd.utils import pytorch_optimizer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda', 0)
args = bench.make_args(4)
coh = synthetic.make_cohort(4 * B, C=bench.C, T=bench.T, H=bench.H, lam=bench.LAM, G=4, seed=synthetic.SEED)
x_np, ob_np, len_np = synthetic.stacked_batch(coh)
store = RaggedStore(x_np, bench.C, dev)
LEN = torch.tensor(len_np, device=dev)
IDX = torch.arange(4 * B, device=dev, dtype=torch.int32)
torch.manual_seed(1234)
net = Net(args, dev).to(dev)
net.train()
st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16, use_graphs=False)


def one(i):
    lo = (i % 4) * B
    return st.step(RaggedBatch(store, IDX[lo:lo + B], LEN[lo:lo + B]), None, None, LEN[lo:lo + B])


for i in range(6):
    one(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    one(7)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == DeviceType.CUDA]
evs.sort(key=lambda e: e.time_range.start)
for n, e in enumerate(evs):
    print('%3d %7.1f us  %s' % (n, float(getattr(e, 'device_time_total', 0.0) or 0.0), e.name[:150]))

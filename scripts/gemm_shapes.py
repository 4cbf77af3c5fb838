"""Which library GEMM is which: aten op, input shapes and device time of every mm / addmm / bmm of one joint step.
usage: python scripts/gemm_shapes.py [B]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from deep_interpolation_clustering_amd import synthetic
from deep_interpolation_clustering_amd.clustering_interp import Net
from deep_interpolation_clustering_amd.step import Stepper
from deep_interpolation_clustering_amd.utils import pytorch_optimizer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device('cuda')
coh = synthetic.make_cohort(B, seed=3)
x_np, ob_np, n = synthetic.stacked_batch(coh)
X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
net = Net(bench.make_args(4), dev).to(dev); net.train()
st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), bench.make_args(4), autocast_dtype=torch.bfloat16, use_graphs=False)
for _ in range(5): st.step(X, OB, None, LEN)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3): st.step(X, OB, None, LEN)
    torch.cuda.synchronize()
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::baddbmm', 'aten::matmul', 'aten::linear'):
        print('%-12s x%-3d %9.1f us  %s' % (e.key, e.count, e.device_time_total / max(e.count, 1), e.input_shapes))

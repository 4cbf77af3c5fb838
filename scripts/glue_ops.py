import os, sys
sys.path.insert(0, '/root/repo')
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from deep_interpolation_clustering_amd import synthetic
from deep_interpolation_clustering_amd.clustering_interp import Net
from deep_interpolation_clustering_amd.step import Stepper
from deep_interpolation_clustering_amd.utils import pytorch_optimizer
B = 32768
dev = torch.device('cuda')
coh = synthetic.make_cohort(B, seed=3)
x_np, ob_np, n = synthetic.stacked_batch(coh)
X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
net = Net(bench.make_args(4), dev).to(dev); net.train()
st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), bench.make_args(4), autocast_dtype=torch.bfloat16, use_graphs=False)
for _ in range(4): st.step(X, OB, None, LEN)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    st.step(X, OB, None, LEN)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.device_time_total > 0 and e.key.startswith('aten::') and e.key not in ('aten::mm', 'aten::addmm'):
        rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:110]))
for r in sorted(rows, reverse=True)[:40]:
    print('%8.1f us x%-3d %-28s %s' % r)

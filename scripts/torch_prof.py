import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
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
st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), bench.make_args(4), autocast_dtype=torch.bfloat16)
for _ in range(3): st.step(X, OB, None, LEN)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): st.step(X, OB, None, LEN)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=45, max_name_column_width=60))

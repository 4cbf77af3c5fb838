"""Per-kernel GPU time of the joint step at a small batch (default 256).  usage: python scripts/trace_small.py [B]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from deep_interpolation_clustering_amd import synthetic
from deep_interpolation_clustering_amd.clustering_interp import Net
from deep_interpolation_clustering_amd.step import Stepper
from deep_interpolation_clustering_amd.utils import pytorch_optimizer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda')
coh = synthetic.make_cohort(B, seed=3)
x_np, ob_np, n = synthetic.stacked_batch(coh)
X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
net = Net(bench.make_args(4), dev).to(dev); net.train()
st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), bench.make_args(4), autocast_dtype=torch.bfloat16)
for _ in range(5): st.step(X, OB, None, LEN)
kernels, groups = bench.step_trace(lambda i: st.step(X, OB, None, LEN), 0, 5)
print(groups)
tot = 0
for k, v in kernels.items():
    tot += v['ms_per_step']
    print('%8.2f us x%-5.1f = %8.1f us/step  %s' % (v['us_per_launch'], v['launches_per_step'], v['ms_per_step'] * 1e3, k[:100]))
print('total kernel time %.1f us/step' % (tot * 1e3))

"""Which aten ops / autograd nodes launch the torch glue kernels of one joint step.  usage: python scripts/op_prof.py [B]"""
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
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(N): st.step(X, OB, None, LEN)
    torch.cuda.synchronize()
# ops that directly launch kernels: self device time > 0
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    sd = getattr(e, 'self_device_time_total', 0)
    if sd > 0 and e.key.startswith('aten::'):
        rows.append((sd / N, e.count / N, e.key, [str(e.input_shapes)[:110]]))
rows.sort(reverse=True)
tot = 0
for sd, cnt, key, stack in rows[:90]:
    tot += sd
    print('%8.1f us/step x%-5.1f %-28s %s' % (sd, cnt, key, ' <- '.join(s.split('/')[-1] for s in stack)))
print('total aten self device time/step: %.1f us' % sum(r[0] for r in rows))

"""Which torch (non-dic) ops run inside one joint step at the bench batch: op name, input shapes, device time (torch.profiler)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from types import SimpleNamespace
from deep_interpolation_clustering_amd import synthetic
from deep_interpolation_clustering_amd.clustering_interp import Net
from deep_interpolation_clustering_amd.step import Stepper
from deep_interpolation_clustering_amd.utils import pytorch_optimizer
from torch.profiler import ProfilerActivity, profile
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
args = SimpleNamespace(num_variables=6, num_timestamps=96, ref_points=24, hours_from_admission=24, dropout=0.0, aux_tasks={}, fake_detection=False,
                       triple_margin=0.0, cluster_number=4, loss='ae_mse_kl', grad_clip=15.0, unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.})
dev = torch.device('cuda')
coh = synthetic.make_cohort(B, seed=1)
x_np, ob_np, n = synthetic.stacked_batch(coh)
X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
torch.manual_seed(0)
net = Net(args, dev).to(dev); net.train()
st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16)
for _ in range(3):
    st.step(X, OB, None, LEN)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    st.step(X, OB, None, LEN)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, 'device_time_total', 0) or getattr(e, 'cuda_time_total', 0)
    if t > 0 and e.key.startswith('aten::'):
        rows.append((t, e.count, e.key, str(e.input_shapes)[:110]))
for t, c, k, s in sorted(rows, reverse=True)[:40]:
    print('%8.1f us  x%-3d %-28s %s' % (t, c, k, s))

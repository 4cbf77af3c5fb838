import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from deep_interpolation_clustering_amd import synthetic, lstm as L
from deep_interpolation_clustering_amd.clustering_interp import Net
from deep_interpolation_clustering_amd.step import Stepper
from deep_interpolation_clustering_amd.utils import pytorch_optimizer
B, steps = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda')
coh = synthetic.make_cohort(B * 4, seed=3)
x_np, ob_np, n = synthetic.stacked_batch(coh)
X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
orig = L.fused_available
for mode in ('f32', 'bf16_miopen', 'bf16_fused'):
    L.fused_available = orig if mode == 'bf16_fused' else (lambda x, l: False)
    torch.manual_seed(11)
    net = Net(bench.make_args(4), dev).to(dev); net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), bench.make_args(4),
                 autocast_dtype=None if mode == 'f32' else torch.bfloat16)
    out = []
    for i in range(steps):
        lo = (i % 4) * B
        losses, gn, _ = st.step(X[lo:lo + B], OB[lo:lo + B], None, LEN[lo:lo + B])
        if i % max(1, steps // 12) == 0 or i == steps - 1:
            out.append('%d:%.4f/%.3f/g%.2f' % (i, float(losses['loss'].detach()), float(losses['kl'].detach()), float(gn)))
    print(mode, ' '.join(out))

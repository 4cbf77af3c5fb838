"""Run the bench step under PyTorch TunableOp to pick the fastest hipBLASLt / rocBLAS solution per GEMM shape, for
per-GPU batches 8192 / 16384 / 32768, and store the table as deep_interpolation_clustering_amd/tuned_gemm_gfx950.csv
(read back by tuned.enable()).  Run on an MI355X: ``python scripts/tune_gemms.py``."""
import glob, os, shutil, sys, time
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, root)
out_dir = os.path.join(root, 'gpurun_out')
os.makedirs(out_dir, exist_ok=True)
os.environ['PYTORCH_TUNABLEOP_ENABLED'] = '1'
os.environ['PYTORCH_TUNABLEOP_TUNING'] = '1'
os.environ['PYTORCH_TUNABLEOP_FILENAME'] = os.path.join(out_dir, 'tunableop_results.csv')
os.environ.setdefault('PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS', '20')
os.environ.setdefault('PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS', '5')
import torch
import bench
from deep_interpolation_clustering_amd import synthetic
from deep_interpolation_clustering_amd.clustering_interp import Net
from deep_interpolation_clustering_amd.step import Stepper
from deep_interpolation_clustering_amd.utils import pytorch_optimizer
dev = torch.device('cuda')
for B in [int(a) for a in sys.argv[1:]] or [256, 2048, 8192, 16384, 32768]:
    coh = synthetic.make_cohort(B, seed=3)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(n, device=dev)
    # the fake-detection variant of the step first (its extra shapes: the detection head on 2B rows), then the headline step
    args_f = bench.make_args(4, True)
    net = Net(args_f, dev).to(dev); net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args_f, autocast_dtype=torch.bfloat16)
    perm = torch.randperm(2 * B, device=dev)
    lab = torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)])[perm].to(torch.int64)
    st.step(X, OB, None, LEN, fake_x=X.clone(), fake_perm_idx=perm, fake_det_label=lab); torch.cuda.synchronize()
    del st, net
    net = Net(bench.make_args(4), dev).to(dev); net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), bench.make_args(4), autocast_dtype=torch.bfloat16)
    t0 = time.time()
    st.step(X, OB, None, LEN); torch.cuda.synchronize()
    print('B=%d: tuning step took %.1f s' % (B, time.time() - t0), flush=True)
    for _ in range(3): st.step(X, OB, None, LEN)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20): st.step(X, OB, None, LEN)
    torch.cuda.synchronize()
    print('B=%d tuned: %.3f ms/step' % (B, (time.time() - t0) / 20 * 1e3), flush=True)
    del st, net, X, OB, LEN
import torch.cuda.tunable as T
print('results:', len(T.get_results()))
import atexit
def _copy():
    for f in glob.glob(os.path.join(out_dir, 'tunableop_results*.csv')):
        shutil.copy(f, os.path.join(out_dir, 'tuned_gemm_gfx950.csv'))
atexit.register(_copy)       # registered after torch's own exit hook was set up at import: runs BEFORE it (LIFO) -> also copy in the shell

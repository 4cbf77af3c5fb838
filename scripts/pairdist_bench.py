"""Time the all-pairs statistics pass (csrc/dic_pairdist.hip) on 75 000 x 256 latents and the indices built on it."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from deep_interpolation_clustering_amd.synthetic import latent_blobs
from deep_interpolation_clustering_amd import cluster_stats as cs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 75000
for K in (2, 4, 16):
    x, lab = latent_blobs(11, n, 256, K)
    xd = torch.tensor(x, device='cuda')
    cs.pair_stats(xd, lab); torch.cuda.synchronize()
    t0 = time.perf_counter(); st = cs.pair_stats(xd, lab); torch.cuda.synchronize(); t1 = time.perf_counter()
    sil = cs.silhouette_score(xd, lab, st); dunn = cs.dunn_index(xd, lab, st); v1 = cs.inertia_v1(xd, lab, st)
    ch = cs.calinski_harabasz_score(xd, lab); db = cs.davies_bouldin_score(xd, lab); torch.cuda.synchronize(); t2 = time.perf_counter()
    pairs = float(n) * n
    print('N=%d K=%d: pair pass %.1f ms (%.2f T pair-features/s, %.1f TFLOP/s f32 at 3 flop each); all indices +%.1f ms; sil %.4f dunn %.4f v1 %.4f ch %.1f db %.4f'
          % (n, K, 1e3 * (t1 - t0), pairs * 256 / (t1 - t0) / 1e12, 3 * pairs * 256 / (t1 - t0) / 1e12, 1e3 * (t2 - t1), sil, dunn, v1, ch, db))
if n <= 20000:
    from sklearn import metrics
    t0 = time.perf_counter(); s = metrics.silhouette_score(x, lab); t1 = time.perf_counter()
    print('sklearn silhouette_score on the host: %.2f s (%.4f)' % (t1 - t0, s))

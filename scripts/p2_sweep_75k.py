"""BASELINE configs[4]: p2_clustering_optK sweep K = 2..20 on 75 000 x 256 latents (elbow + gap statistic with n_init = 10,
gap_b = 10 reference sets, silhouette / Davies-Bouldin / Calinski-Harabasz per K), timed end to end on one GPU."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from deep_interpolation_clustering_amd.synthetic import latent_blobs
from deep_interpolation_clustering_amd import p2_clustering_optK as p2

n = int(sys.argv[1]) if len(sys.argv) > 1 else 75000
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 20
run = tempfile.mkdtemp(prefix='dic_p2_')
os.chdir(run)
folder = os.path.join(run, 'Results', 'Pretrain', 'out_feat', 'ae_mse')
os.makedirs(folder)
for cohort, (m, seed) in {'training': (n, 1), 'validation': (n // 8, 2), 'testing': (n // 8, 3)}.items():
    X, _ = latent_blobs(seed, m, 256, 4, centers_seed=99)
    np.save(os.path.join(folder, cohort + '.npy'), {'encounter_id': np.arange(m), 'hidden': X, 'ob': np.zeros((m, 1, 1), np.float32),
                                                    'padding_mask': np.ones((m, 1, 1), np.float32)})
a = p2.get_arguments(['--k_max', str(kmax), '--n_init', '10', '--gap_b', '10'])
a.restore_metric = ['ae_mse']
t0 = time.perf_counter()
res = p2.main(a)['ae_mse']
el = time.perf_counter() - t0
print(res['gap_sts'][['k', 'gap', 'Sihouette', 'Davies-Bouldin_Index', 'Calinski-Harabasz']].to_string(index=False))
print('p2 sweep K=2..%d on %d x 256 latents: %.1f s (elbow + gap statistic, 10 reference sets, 3 indices per K)' % (kmax, n, el))

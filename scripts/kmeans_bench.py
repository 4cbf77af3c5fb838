#!/usr/bin/env python3
"""k-means (row a10 / BASELINE config 5) timing on the MI355X next to scikit-learn on the host cores.
Usage: python3 scripts/kmeans_bench.py [N] [--sklearn]      (DIC_AB_LIB=<second libdic_hip.so>: the Lloyd iteration of both builds, timed alternately)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from deep_interpolation_clustering_amd import _native as N  # noqa: E402
from deep_interpolation_clustering_amd.kmeans import KMeans, lloyd  # noqa: E402
from deep_interpolation_clustering_amd.synthetic import latent_blobs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 75000
X, _ = latent_blobs(2024, n, 256, 4, spread=0.35, noise=0.3)
Xd = torch.tensor(X, device='cuda')


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


print(f'N={n} D=256')
for K, n_init in ((4, 20), (8, 20), (16, 10)):
    np.random.seed(7529)
    dt, km = timed(lambda: KMeans(n_clusters=K, n_init=n_init).fit(Xd))
    print(f'HIP  KMeans(K={K}, n_init={n_init}).fit: {dt * 1e3:8.1f} ms  inertia {km.inertia_:.1f}  n_iter {km.n_iter_}')
# one Lloyd iteration, all restarts in one launch: algorithmic bytes = n_runs * N * (4D + 4)
LIBS = [('this build', N.lib())]
if os.environ.get('DIC_AB_LIB'):
    LB = C.CDLL(os.path.abspath(os.environ['DIC_AB_LIB']))
    for name, (res, args) in N.SIGNATURES.items():
        if hasattr(LB, name):
            fn = getattr(LB, name)
            fn.restype, fn.argtypes = res, args
    LIBS.append(('DIC_AB_LIB', LB))
for (lib_name, L), (K, runs) in [(lb, kr) for kr in ((4, 1), (4, 10), (4, 20), (8, 10), (16, 10), (16, 20), (20, 10)) for lb in LIBS * (2 if len(LIBS) > 1 else 1)]:
    Xc = Xd - Xd.mean(0)
    xn = (Xc * Xc).sum(1)
    cent = Xc[torch.randint(0, n, (runs, K), device='cuda')].contiguous()
    labels = torch.full((runs, n), -1, dtype=torch.int32, device='cuda')
    status = torch.zeros((runs, 8), device='cuda')
    status[:, 7] = 1e9
    ws = torch.empty(L.dic_kmeans_workspace(n, 256, K, runs), dtype=torch.uint8, device='cuda')
    st = N.stream_of(Xc)
    f = lambda: L.dic_kmeans_lloyd_iter(N.ptr(Xc), N.ptr(xn), n, 256, K, runs, N.ptr(cent), N.ptr(labels), N.ptr(status), N.ptr(ws), ws.numel(), st)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gb = runs * n * (4 * 256 + 4) / 1e9
    print(f'[{lib_name}] lloyd_iter K={K} restarts={runs}: {ms * 1e3:8.1f} us  {gb / ms * 1e3:8.1f} GB/s algorithmic (data cache-resident: {n * 1024 / 1e6:.0f} MB)')
if '--sklearn' in sys.argv:
    from sklearn.cluster import KMeans as SK
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=16):
        for K, n_init in ((4, 20), (16, 10)):
            np.random.seed(7529)
            t0 = time.perf_counter()
            sk = SK(n_clusters=K, n_init=n_init).fit(X)
            print(f'sklearn KMeans(K={K}, n_init={n_init}).fit (<=16 threads): {(time.perf_counter() - t0) * 1e3:8.1f} ms  inertia {sk.inertia_:.1f}')

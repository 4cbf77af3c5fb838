#!/usr/bin/env python3
"""The k-means kernels on BASELINE configs[4]'s own data (75 000 x 256 latents, bench.record_cfg5's generator), launched the way
bench.py times them -- state reset, Lloyd iteration 1, Lloyd iteration 2 -- a few times: the target of
`rocprofv3 --kernel-trace --stats` and `--pmc FETCH_SIZE / WRITE_SIZE` passes (scripts/profile_round.sh).
Usage: python3 scripts/kmeans_prof.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

from deep_interpolation_clustering_amd import _native as N  # noqa: E402
from deep_interpolation_clustering_amd.synthetic import latent_blobs  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda', 0)
n = 75000
X, _ = latent_blobs(2024, n, 256, 4, spread=0.35, noise=0.3)
Xd = torch.tensor(X, device=dev)
Xc = Xd - Xd.mean(0)
xn = (Xc * Xc).sum(1)
L = N.lib()
torch.manual_seed(0)
for Kk, runs in ((4, 20), (16, 10)):
    cent0 = Xc[torch.randint(0, n, (runs, Kk), device=dev)].contiguous()
    cent = cent0.clone()
    labels = torch.full((runs, n), -1, dtype=torch.int32, device=dev)
    status = torch.zeros((runs, 8), device=dev)
    ws = torch.empty(L.dic_kmeans_workspace(n, 256, Kk, runs), dtype=torch.uint8, device=dev)
    st = N.stream_of(Xc)
    for _ in range(reps):
        cent.copy_(cent0)
        labels.fill_(-1)
        status.zero_()
        status[:, 7] = 1e9
        for _ in range(2):
            N.check(L.dic_kmeans_lloyd_iter(N.ptr(Xc), N.ptr(xn), n, 256, Kk, runs, N.ptr(cent), N.ptr(labels), N.ptr(status), N.ptr(ws), ws.numel(), st), 'lloyd')
    torch.cuda.synchronize()
    print(f'K={Kk} x {runs} restarts: {reps} x 2 Lloyd iterations, active at the end: {int((status[:, 0] == 0).sum())}')

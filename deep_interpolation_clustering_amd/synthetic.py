"""Synthetic irregular vitals in the reference's on-disk layout.

The hospital data behind p0_data_process.py is private; this generator produces cohort pickles with
the same keys and shapes (``feat, padding_mask, time_step, drop_mask, encounter_id`` --
p0_data_process.py:35-70,95-117; dataloader.py:64-68) so p1/p3 run unchanged, and the in-memory
batches ``bench.py`` times.

Per encounter e and channel c: n = clip(Poisson(lam), 1, T) observations at sorted U(0,H) hours in
the first n slots; values follow one of G latent phenotypes,
``clip(mu[g,c] + 0.10 sin(2 pi t/24 + phi_e) + 0.05 N(0,1), 0, 1)`` (already min-max normalised, like
p0's output); ``padding_mask`` marks the first n slots; ``drop_mask`` holds out 20 % of the observed
points of a row when that is more than one point (p0 ``hold_out``).
"""
import os
import pickle

import numpy as np

from .info import COHORTS

SEED = 7529      # echoes p1_pretrain_main.py:26


def make_cohort(n_enc, C=6, T=96, H=24.0, lam=50.0, G=4, seed=SEED, id_offset=0):
    """dict(feat (N,C,T) f64 in [0,1], padding_mask int8, time_step f64, drop_mask int8, encounter_id list)
    + 'lengths' (N,C) int32 and 'phenotype' (N,) (extras; ignored by the reference loader)."""
    rng = np.random.default_rng(seed)
    mu = np.random.default_rng(SEED).uniform(0.2, 0.8, (G, C))      # phenotype means are fixed across cohorts
    n = np.clip(rng.poisson(lam, (n_enc, C)), 1, T).astype(np.int32)
    slot = np.arange(T)[None, None, :]
    mask = slot < n[..., None]
    t = rng.uniform(0, H, (n_enc, C, T)).astype(np.float32)
    t = np.sort(np.where(mask, t, np.inf), axis=-1)                 # valid times first, ascending
    t = np.where(mask, t, 0.0)
    g = rng.integers(0, G, n_enc)
    phi = rng.uniform(0, 2 * np.pi, n_enc)
    val = mu[g][:, :, None] + 0.10 * np.sin(2 * np.pi * t / 24.0 + phi[:, None, None]) \
        + 0.05 * rng.standard_normal((n_enc, C, T))
    val = np.clip(val, 0.0, 1.0) * mask
    n_drop = (0.20 * n).astype(np.int64)
    n_drop = np.where(n_drop > 1, n_drop, 0)                        # p0: only when int(0.2*count) > 1
    order = np.argsort(np.where(mask, rng.random((n_enc, C, T)), 2.0), axis=-1)     # random order of the valid slots
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.broadcast_to(slot, order.shape).copy(), axis=-1)
    drop = (mask & ~(rank < n_drop[..., None])).astype(np.int8)
    return dict(feat=val.astype(np.float64), padding_mask=mask.astype(np.int8), time_step=t.astype(np.float64),
                drop_mask=drop, encounter_id=list(range(id_offset, id_offset + n_enc)), lengths=n,
                phenotype=g.astype(np.int32))


def write_split(base_path, n_total, split=(0.8, 0.1, 0.1), **kw):
    """Write {training,validation,testing}.pickle under <base_path>/Data/model_data/split_processed/."""
    out = os.path.join(base_path, 'Data', 'model_data', 'split_processed')
    os.makedirs(out, exist_ok=True)
    seed = kw.pop('seed', SEED)
    sizes = [int(round(n_total * f)) for f in split]
    sizes[0] = n_total - sizes[1] - sizes[2]
    off = 0
    for i, (cohort, n) in enumerate(zip(COHORTS, sizes)):
        d = make_cohort(n, seed=seed + i, id_offset=off, **kw)
        with open(os.path.join(out, f'{cohort}.pickle'), 'wb') as f:
            pickle.dump(d, f, protocol=4)
        off += n
    return out


def stacked_batch(cohort, scale=5.0, denoise=False):
    """(N,4C,T) f32 stacked input exactly as the trainers build it (dataloader.py:64-79,
    pretrain_trainer.py:132-143): [ob*mask (x drop_mask if denoise) | mask | time | drop_mask], values rescaled
    to +-scale/2.  Returns (stacked, ob*mask, lengths)."""
    feat = cohort['feat'].astype(np.float32)
    if scale != 0:
        feat = scale * feat - scale / 2
    mask = cohort['padding_mask'].astype(np.float32)
    ob = feat * mask
    drop = cohort['drop_mask'].astype(np.float32)
    first = ob * drop if denoise else ob
    x = np.concatenate([first, mask, cohort['time_step'].astype(np.float32), drop], axis=1)
    return x, ob, cohort['lengths'].astype(np.int32)


def latent_blobs(seed, N, D, K, centers_seed=None, spread=0.35, noise=0.25):
    """(N,D) f32 Gaussian-mixture 'latents' with K components; returns (X, component) -- the inputs of the k-means / K-sweep
    measurements (bench.py cfg5, scripts/).  Same draws as the test suite's generator (oracle/synth.py keeps its own copy: nothing
    outside tests / the CPU baseline touches oracle/)."""
    crng = np.random.default_rng(seed if centers_seed is None else centers_seed)
    cent = crng.normal(0, spread, (K, D)).astype(np.float32)
    rng = np.random.default_rng([seed, 17])
    comp = rng.integers(0, K, N)
    X = (cent[comp] + rng.normal(0, noise, (N, D))).astype(np.float32)
    return X, comp


def device_cohort_store(n_enc, C, T, H, lam, G, seed, device, chunk=16384, scale=5.0):
    """A whole synthetic cohort as a ``RaggedStore`` resident on ``device`` WITHOUT ever holding its padded planes on the host: the same
    generative rule as ``make_cohort`` / ``stacked_batch`` (Poisson lengths, sorted U(0,H) times in a prefix, phenotype mean + diurnal sine +
    noise clipped to [0,1] and rescaled to +-scale/2, 20 % hold-out flags), drawn chunk by chunk from a torch generator on the device
    (other draws than ``make_cohort``'s NumPy stream, same distribution).  BASELINE configs[3] -- 300 000 encounters x 12 channels x ~200
    observations -- is 16.6 GB padded and 6.5 GB as a store.  Returns (store, phenotype (N,) int64 on the device)."""
    import torch

    from .ragged import RaggedStore
    g = torch.Generator(device=device).manual_seed(int(seed))
    mu = torch.as_tensor(np.random.default_rng(SEED).uniform(0.2, 0.8, (G, C)), dtype=torch.float32, device=device)
    stores, pheno = [], []
    for lo in range(0, n_enc, chunk):
        b = min(chunk, n_enc - lo)
        n = torch.poisson(torch.full((b, C), float(lam), device=device), generator=g).clamp_(1, T)
        mask = torch.arange(T, device=device)[None, None, :] < n[..., None]
        t = torch.rand((b, C, T), device=device, generator=g) * H
        t = torch.where(mask, t, torch.full_like(t, 2 * H)).sort(dim=-1).values * mask
        ph = torch.randint(0, G, (b,), device=device, generator=g)
        phi = torch.rand((b, 1, 1), device=device, generator=g) * (2 * np.pi)
        val = mu[ph][:, :, None] + 0.10 * torch.sin(2 * np.pi * t / 24.0 + phi) + 0.05 * torch.randn((b, C, T), device=device, generator=g)
        val = (scale * val.clamp_(0.0, 1.0) - scale / 2) * mask
        hold = ((torch.rand((b, C, T), device=device, generator=g) >= 0.2) & mask).float()
        stores.append(RaggedStore.from_device(torch.cat([val, mask.float(), t, hold], dim=1), C))
        pheno.append(ph)
        del val, mask, t, hold
    return (stores[0] if len(stores) == 1 else RaggedStore.concat(stores)), torch.cat(pheno)

"""Seeded synthetic inputs shared by the golden generator and the tests.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.
"""
import numpy as np


def latent_blobs(seed, N, D, K, centers_seed=None, spread=0.35, noise=0.25):
    """(N,D) f32 Gaussian-mixture 'latents' with K components; returns (X, component)."""
    crng = np.random.default_rng(seed if centers_seed is None else centers_seed)
    cent = crng.normal(0, spread, (K, D)).astype(np.float32)
    rng = np.random.default_rng([seed, 17])
    comp = rng.integers(0, K, N)
    X = (cent[comp] + rng.normal(0, noise, (N, D))).astype(np.float32)
    return X, comp


def vitals_stack(seed, B, C, T, H, lam, dtype=np.float32):
    """Ragged (prefix-mask) stacked input (B,4C,T): planes value*mask, mask, time, holdout.
    Poisson(lam) observations per channel, clipped to [1,T]; sorted uniform times."""
    rng = np.random.default_rng(seed)
    n = np.clip(rng.poisson(lam, (B, C)), 1, T)
    slot = np.arange(T)[None, None, :]
    mask = (slot < n[..., None]).astype(dtype)
    tim = np.sort(rng.uniform(0, H, (B, C, T)).astype(dtype) * mask + (1 - mask) * 1e9, axis=-1)
    tim = np.where(mask > 0, tim, 0).astype(dtype)
    val = (rng.normal(0, 1.2, (B, C, T)).astype(dtype)) * mask
    hold = (rng.uniform(0, 1, (B, C, T)) > 0.2).astype(dtype)
    return np.concatenate([val, mask, tim, hold], axis=1), n.astype(np.int32)

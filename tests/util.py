"""Shared helpers for the tests."""
import numpy as np


def pca_basis_32(latent_dim=6, n=32, seed=7):
    """The PCA basis of tests/golden/model_32.npz, regenerated from its seed (make_golden.py)."""
    rs = np.random.RandomState(seed)
    vec = rs.normal(0, 0.02 / np.sqrt(latent_dim), (latent_dim, 3 * n ** 3)).astype(np.float32)
    mean = rs.normal(0, 0.002, (3 * n ** 3,)).astype(np.float32)
    return vec, mean

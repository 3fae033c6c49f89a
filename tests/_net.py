"""Shared helpers for the net-level parity tests."""
import numpy as np

from oracle import uresnet_np as O


def make_inputs(dims, ncls, N, seed, sparse=True):
    """Synthetic batch in the reference feed format (flat fp32 arrays, lib/ssnet.py:30-32)."""
    rng = np.random.default_rng(seed)
    dsz, lsz = int(np.prod(dims)), int(np.prod(dims[:-1]))
    data = rng.uniform(0, 4, (N, dsz))
    if sparse:
        data *= rng.uniform(0, 1, (N, dsz)) > 0.7
    label = rng.integers(0, ncls, (N, lsz)).astype(np.float32)
    weight = rng.uniform(0.5, 1.5, (N, lsz))
    weight /= weight.sum(axis=1, keepdims=True)  # lib/ssnet_trainval.py:173
    return data.astype(np.float32), label, weight.astype(np.float32)


def oracle_params(dims, base, ncls, seed=11, beta_scale=0.2, dtype=np.float64):
    return O.init_params(len(dims) - 1, dims[-1], base, ncls, seed=seed, dtype=dtype, beta_scale=beta_scale)


def max_rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

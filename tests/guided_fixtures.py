"""Synthetic per-primitive radiosity grids for the guided-sampling tests (the reference fills these with its
radiosity pre-pass, which is out of scope; any non-negative grid is a valid input)."""
import numpy as np


def synthetic_radiosity_grids(n_prims, seed=0, empty_every=5):
    """(n_prims, 256, 3) float32.  Row v = polar band (rows 0..7 upper hemisphere), column u = azimuth.
    Every `empty_every`-th primitive gets an all-zero grid (-> invalid record -> cosine fallback), one gets a grid
    with empty rows and cells below 1e-8 (-> the 1e-6 pdf floor and the uniform-row CDF branch)."""
    rng = np.random.default_rng(seed)
    v = (np.arange(16) + 0.5) / 8 * (np.pi / 2)
    u = (np.arange(16) + 0.5) / 16 * (2 * np.pi)
    g = np.zeros((n_prims, 16, 16, 3), np.float32)
    for p in range(n_prims):
        if empty_every and p % empty_every == empty_every - 1:
            continue
        lobe = np.clip(np.cos(v), 0, None)[:, None] * (1.2 + np.sin(u + 0.37 * p))[None, :]
        lobe = lobe * rng.uniform(0.5, 1.5, (16, 16))
        lobe[8:] = rng.uniform(0, 1, (8, 16))          # lower-hemisphere cells are ignored by the CDFs
        col = rng.uniform(0.2, 1.0, 3)
        g[p] = (lobe[:, :, None] * col[None, None, :]).astype(np.float32)
    if n_prims > 2:
        g[1, 2:5] = 0.0                                # empty rows
        g[1, 0, :8] = 1e-9                             # cells under the 1e-8 threshold
    return g.reshape(n_prims, 256, 3)

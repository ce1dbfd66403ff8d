"""Synthetic SPE10-like permeability/porosity fields.

The raw SPE10 files (spe_perm.dat, spe_phi.dat) are absent from the reference checkout
(/root/reference/.MISSING_LARGE_BLOBS), so BASELINE configs 2-5 run on this generator unless
real slices are supplied (see create_SPE10_slice.py).  Output arrays have the same shape and
units as the reference's ``slice_*.npy`` files: (Nx, Ny[, Nz]), permeability in mm^2.
"""
import numpy as np
from scipy.ndimage import uniform_filter

MD_TO_MM2 = 9.869233e-10      # data/create_SPE10_slice.py:42 of the reference


DEAD_FRACTION = 0.025         # share of zero-porosity ("rock only") cells, SURVEY.md 8d


def synthetic_spe10(Nx, Ny, Nz=None, seed=10, dead=True):
    """log10(Kx[mD]) ~ N(1, 1.5^2) box-filtered 3x5(x1) and rescaled to keep sigma; Ky = Kx;
    clip to [1e-3, 2e4] mD; phi = clip(0.2 + 0.08 z, 0, 0.5) with 2.5 % of cells set to 0;
    3-D: Kz = 0.1 Kx with 30 % of cells further scaled by 1e-3 (shale-like).
    dead=False leaves the zero-porosity cells out (the random numbers are still drawn, so every other field is
    unchanged): a refined model punches them at the FINE resolution instead (kill_cells)."""
    rng = np.random.default_rng(seed)
    shape = (Nx, Ny) if Nz is None else (Nx, Ny, Nz)
    g = rng.standard_normal(shape)
    size = (3, 5) if Nz is None else (3, 5, 1)
    z = uniform_filter(g, size=size, mode="nearest")
    z = (z - z.mean())/z.std()
    logk = 1.0 + 1.5*z
    kx_md = np.clip(10.0**logk, 1e-3, 2e4)
    Kx = kx_md*MD_TO_MM2
    Ky = Kx.copy()
    phi = np.clip(0.2 + 0.08*z, 0.0, 0.5)
    mask = rng.random(shape) < DEAD_FRACTION
    if dead:
        phi[mask] = 0.0
    out = {"phi": phi, "perm_x": Kx, "perm_y": Ky}
    if Nz is not None:
        Kz = 0.1*Kx
        Kz[rng.random(shape) < 0.30] *= 1e-3
        out["perm_z"] = Kz
    return out


def upsample(fields, factor):
    """Piecewise-constant refinement (config 5: 60x220x85 -> 240x880x340)."""
    out = {}
    for k, v in fields.items():
        for ax in range(v.ndim):
            v = np.repeat(v, factor, axis=ax)
        out[k] = v
    return out


def kill_cells(phi, seed):
    """Zero porosity in 2.5 % of the cells of an already refined field, in place: isolated dead CELLS at the fine
    resolution, as the unrefined SPE10-like model has them (config 5; a mask applied before the x4 refinement turns
    each dead cell into a 4x4x4 dead block, in which S_o has no accumulation term anywhere and the time loop's
    saturation guard halves dt at every step).  Drawn plane by plane so that the 72 M-cell box needs no 0.6 GB scratch."""
    rng = np.random.default_rng(seed)
    for i in range(phi.shape[0]):
        phi[i][rng.random(phi.shape[1:]) < DEAD_FRACTION] = 0.0
    return phi

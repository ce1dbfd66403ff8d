"""SPE10 ingest: cut (Nx,Ny[,Nz]) windows out of the raw SPE10 .dat files and save slice_*.npy.

Same on-disk contract as /root/reference/data/create_SPE10_slice.py:10-71 and
create_SPE10_slice2D.py:11-60: ``spe_phi.dat`` holds 60*220*85 values, x fastest, then y, then
z with the TOP layer first; ``spe_perm.dat`` holds three such blocks (Kx, Ky, Kz) in mD.
Output arrays are indexed [i, j, k] with k increasing upwards (z flipped, :31), permeability
converted mD -> mm^2 (x 9.869233e-10, :42).  Vectorised; no Firedrake import needed.
"""
import os

import numpy as np

from .synthetic_spe10 import MD_TO_MM2

NX, NY, NZ = 60, 220, 85


def _window(flat, Nx, Ny, Nz, x_shift, y_shift, z_shift):
    full = flat.reshape(NZ, NY, NX)                        # [kk, j, i], kk = 0 is the top layer
    w = full[z_shift:z_shift + Nz, y_shift:y_shift + Ny, x_shift:x_shift + Nx]
    return np.ascontiguousarray(w.transpose(2, 1, 0)[:, :, ::-1])   # [i, j, Nz-1-kk]


def create_SPE10_slice(Nx, Ny, Nz, x_shift=0, y_shift=0, z_shift=0, dirname=None, perm_factor=1.0):
    dirname = dirname or os.path.dirname(__file__)
    phi = np.loadtxt(os.path.join(dirname, "spe_phi.dat")).reshape(-1)
    np.save(os.path.join(dirname, "slice_phi.npy"), _window(phi, Nx, Ny, Nz, x_shift, y_shift, z_shift))
    perm = np.loadtxt(os.path.join(dirname, "spe_perm.dat")).reshape(3, NX*NY*NZ)*MD_TO_MM2*perm_factor
    for c, name in enumerate(("x", "y", "z")):
        np.save(os.path.join(dirname, "slice_perm_%s.npy" % name),
                _window(perm[c], Nx, Ny, Nz, x_shift, y_shift, z_shift))


def create_SPE10_slice2D(Nx, Ny, x_shift=0, y_shift=0, z_shift=0, dirname=None, perm_factor=1.0):
    """One horizontal layer (create_SPE10_slice2D.py:11-60): arrays of shape (Nx, Ny)."""
    dirname = dirname or os.path.dirname(__file__)
    phi = np.loadtxt(os.path.join(dirname, "spe_phi.dat")).reshape(-1)
    np.save(os.path.join(dirname, "slice_phi.npy"), _window(phi, Nx, Ny, 1, x_shift, y_shift, z_shift)[:, :, 0])
    perm = np.loadtxt(os.path.join(dirname, "spe_perm.dat")).reshape(3, NX*NY*NZ)*MD_TO_MM2*perm_factor
    for c, name in enumerate(("x", "y")):
        np.save(os.path.join(dirname, "slice_perm_%s.npy" % name),
                _window(perm[c], Nx, Ny, 1, x_shift, y_shift, z_shift)[:, :, 0])


def _vertical_window(flat, n_h, Nz, horizontal, x_shift, y_shift, z_shift):
    """Vertical section [h, kk] at fixed y (horizontal='x') or fixed x ('y').  NB: unlike the 3-D maker the reference's
    vertical slice makers do NOT flip z (create_SPE10_slicexz.py:24,42,51 / :69,86,94): kk = 0 stays the top layer."""
    full = flat.reshape(NZ, NY, NX)
    if horizontal == "x":
        w = full[z_shift:z_shift + Nz, y_shift, x_shift:x_shift + n_h]
    else:
        w = full[z_shift:z_shift + Nz, y_shift:y_shift + n_h, x_shift]
    return np.ascontiguousarray(w.T)


def _vertical_slice(n_h, Nz, horizontal, comps, x_shift, y_shift, z_shift, dirname, perm_factor):
    dirname = dirname or os.path.dirname(__file__)
    phi = np.loadtxt(os.path.join(dirname, "spe_phi.dat")).reshape(-1)
    np.save(os.path.join(dirname, "slice_phi.npy"), _vertical_window(phi, n_h, Nz, horizontal, x_shift, y_shift, z_shift))
    perm = np.loadtxt(os.path.join(dirname, "spe_perm.dat")).reshape(3, NX*NY*NZ)*MD_TO_MM2*perm_factor
    for c, name in zip(comps, ("x", "y")):       # the model's (x, y) = (horizontal, vertical) directions of the section
        np.save(os.path.join(dirname, "slice_perm_%s.npy" % name),
                _vertical_window(perm[c], n_h, Nz, horizontal, x_shift, y_shift, z_shift))


def create_SPE10_slicexz(Nx, Nz, x_shift=0, y_shift=0, z_shift=0, dirname=None, perm_factor=1.0):
    """x-z section at y = y_shift (create_SPE10_slicexz.py:9-54): arrays (Nx, Nz); perm_x = Kx, perm_y = Kz."""
    _vertical_slice(Nx, Nz, "x", (0, 2), x_shift, y_shift, z_shift, dirname, perm_factor)


def create_SPE10_sliceyz(Ny, Nz, x_shift=0, y_shift=0, z_shift=0, dirname=None, perm_factor=1.0):
    """y-z section at x = x_shift (create_SPE10_slicexz.py:56-99): arrays (Ny, Nz); perm_x = Ky, perm_y = Kz."""
    _vertical_slice(Ny, Nz, "y", (1, 2), x_shift, y_shift, z_shift, dirname, perm_factor)

"""Cell picking (mirror of /root/reference/thermalporous/utils.py:7-25) and DG0 delta helpers.

``GetNodeClosestToCoordinate`` scans dofs in order with a strict ``<`` (:14-18), i.e. the FIRST
dof at minimal distance wins.  Firedrake's dof numbering is not reproducible here, so ties
(common: e.g. ``test0`` wells at x=2 with Dx=2 are equidistant from centres 1 and 3) are broken
by this build's documented order: lowest flat index c = i + Nx*(j + Ny*k) (x fastest).
MATLAB export helpers (:27-49) are debug aids outside the hot path.
"""
import numpy as np


def cell_index(geo, i, j, k=0):
    return i + geo.Nx*(j + geo.Ny*k)


def GetNodeClosestToCoordinate(geo, coord):
    cc = geo.cell_centres()
    d2 = sum((c - w)**2 for c, w in zip(cc, coord))
    # flat order x fastest -> transpose so that ravel() runs i fastest
    flat = np.transpose(d2, tuple(range(d2.ndim))[::-1]).ravel()
    return int(np.argmin(np.sqrt(flat)))     # argmin returns the first minimum


class Delta():
    """Sparse DG0 'delta' function: ``cells`` (flat indices) with ``values`` (the DG0 nodal values,
    normalised so that sum(values*|E|) = 1, wellcase.py:117-122)."""

    def __init__(self, cells, values, vol):
        self.cells = np.asarray(cells, dtype=np.int64)
        self.values = np.asarray(values, dtype=float)
        self.vol = vol

    @property
    def weights(self):
        return self.values*self.vol

    def dense(self, geo):
        out = np.zeros(geo.Nx*geo.Ny*geo.Nz)
        np.add.at(out, self.cells, self.values)
        return out


def cell_volume(geo):
    return geo.Dx*geo.Dy*(geo.Dz if geo.dim == 3 else 1.0)


def well_delta(geo, w):
    """1/|E| at the nearest cell centre (wellcase.py:157-169)."""
    node = GetNodeClosestToCoordinate(geo, w)
    vol = cell_volume(geo)
    return Delta([node], [1.0/vol], vol)


def well_circle(geo, w, radius, height=None):
    """Bump exp(-1/(r^2-d^2)) for d<r sampled at cell centres, normalised by its integral;
    3-D adds |z-zw| < height; falls back to well_delta when no centre is inside
    (wellcase.py:110-123,141-155)."""
    cc = geo.cell_centres()
    d2 = (cc[0] - w[0])**2 + (cc[1] - w[1])**2
    inside = d2 < radius**2
    if geo.dim == 3:
        inside = inside & (np.abs(cc[2] - w[2]) < height)
    vol = cell_volume(geo)
    if not inside.any():
        return well_delta(geo, w)
    idx = np.nonzero(inside)
    vals = np.exp(-(1.0/(-d2[idx] + radius**2)))
    normalise = vals.sum()*vol
    if normalise == 0:
        return well_delta(geo, w)
    cells = cell_index(geo, *idx) if geo.dim == 3 else cell_index(geo, idx[0], idx[1])
    order = np.argsort(cells)
    return Delta(cells[order], (vals/normalise)[order], vol)

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


def _window(geo, w, half):
    """Index ranges [lo, hi) per axis of the cells within `half[a]` cells of the cell containing the point w, and the
    cell-centre coordinates of that window (same expressions as geo.cell_centres(), so the values are bit-identical).
    Everything a well needs lies in such a window; evaluating the reference's whole-grid expressions on it instead of on
    every cell makes well set-up O(1) per well (42 wells on the 71.8 M-cell box of config 5: minutes -> milliseconds)."""
    N = (geo.Nx, geo.Ny, geo.Nz)[:geo.dim]
    D = (geo.Dx, geo.Dy, getattr(geo, "Dz", 1.0))[:geo.dim]
    lo, hi, axes = [], [], []
    for a in range(geo.dim):
        i = min(max(int(np.floor(w[a]/D[a])), 0), N[a] - 1)
        lo.append(max(i - half[a], 0))
        hi.append(min(i + half[a] + 1, N[a]))
        axes.append((np.arange(lo[a], hi[a]) + 0.5)*D[a])
    return lo, hi, np.meshgrid(*axes, indexing="ij")


def GetNodeClosestToCoordinate(geo, coord):
    lo, hi, cc = _window(geo, coord, [2]*geo.dim)
    d2 = sum((c - w)**2 for c, w in zip(cc, coord))
    # first minimum in flat order (x fastest): transpose so that ravel() runs i fastest, as the whole-grid scan did; a
    # centre outside the +-2 window is farther away than every centre inside it by more than a cell size -- never a tie
    flat = np.transpose(d2, tuple(range(d2.ndim))[::-1]).ravel()
    m = int(np.argmin(np.sqrt(flat)))
    shape = [h - l for l, h in zip(lo, hi)]
    idx, rem = [], m
    for a in range(geo.dim):
        idx.append(lo[a] + rem % shape[a])
        rem //= shape[a]
    return int(cell_index(geo, *idx))


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
    (wellcase.py:110-123,141-155).  Evaluated on the window of cells that can lie inside (see _window)."""
    half = [int(np.ceil(radius/geo.Dx)) + 2, int(np.ceil(radius/geo.Dy)) + 2]
    if geo.dim == 3:
        half.append(int(np.ceil(height/geo.Dz)) + 2)
    lo, hi, cc = _window(geo, w, half)
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
    gidx = [ix + l for ix, l in zip(idx, lo)]
    cells = cell_index(geo, *gidx) if geo.dim == 3 else cell_index(geo, gidx[0], gidx[1])
    order = np.argsort(cells)
    return Delta(cells[order], (vals/normalise)[order], vol)

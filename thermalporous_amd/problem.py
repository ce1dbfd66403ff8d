"""Problem description handed to the compute engine (plain data; no arithmetic of the hot path).

``build_spec`` turns the reference-shaped objects (geo, case, params) into the flat description
the C-ABI consumes through tp_create / tp_set_field / tp_set_params / tp_set_sources:

* internal axis order: internal axis 0 is fastest in memory, internal axis 2 is the slab axis
  of the 1-D multi-GPU decomposition.  3-D: a0 = z (the thin, strongly coupled SPE10 direction
  stays whole inside every ILU tile and is the first AMG semi-coarsening direction), a2 = the
  longer of x,y (smallest slab cross-section), a1 = the other.  2-D: (x, y, -).
* fields transposed to internal order, shape (n2, n1, n0);
* wells/heaters flattened to per-cell source entries with their Peaceman index
  (wellcase.py:182-191 of the reference: hard-coded Dx=Dy=h=5, rw=0.1; Kx,Ky at the cell).
"""
import numpy as np

from .wellcase import peaceman_WI

PROD, INJ, HEATER = 0, 1, 2


def choose_axes(geo):
    if geo.dim == 2:
        return (0, 1, 2)
    return (2, 0, 1) if geo.Ny >= geo.Nx else (2, 1, 0)


def to_internal(arr, geo, axes):
    """Physical array (Nx,Ny[,Nz]) or scalar -> internal array (n2,n1,n0)."""
    N = (geo.Nx, geo.Ny, geo.Nz)
    a = np.asarray(arr, dtype=float)
    if a.ndim == 0:
        return np.full((N[axes[2]], N[axes[1]], N[axes[0]]), float(a))
    a = a.reshape(N)
    return np.ascontiguousarray(a.transpose(axes[2], axes[1], axes[0]))


def from_internal(arr, geo, axes):
    """Internal (n2,n1,n0) -> physical (Nx,Ny,Nz)."""
    inv = [0, 0, 0]
    for num_ax, phys in enumerate((axes[2], axes[1], axes[0])):
        inv[phys] = num_ax
    return np.ascontiguousarray(np.asarray(arr).transpose(inv))


def phys_flat_to_internal(cells, geo, axes):
    cells = np.asarray(cells, dtype=np.int64)
    ix = cells % geo.Nx
    iy = (cells // geo.Nx) % geo.Ny
    iz = cells // (geo.Nx*geo.Ny)
    co = (ix, iy, iz)
    N = (geo.Nx, geo.Ny, geo.Nz)
    n0, n1 = N[axes[0]], N[axes[1]]
    return co[axes[0]] + n0*(co[axes[1]] + n1*co[axes[2]])


def field_major_to_internal(u, geo, axes, b):
    """User state (b, Nx*Ny*Nz) in x-fastest order -> (b, n2, n1, n0)."""
    u = np.asarray(u, dtype=float).reshape(b, geo.Nz, geo.Ny, geo.Nx)      # [f, iz, iy, ix]
    out = []
    for f in range(b):
        out.append(to_internal(u[f].transpose(2, 1, 0), geo, axes))
    return np.array(out)


def internal_to_field_major(u, geo, axes, b):
    u = np.asarray(u).reshape((b,) + tuple(np.array((geo.Nx, geo.Ny, geo.Nz))[[axes[2], axes[1], axes[0]]]))
    out = []
    for f in range(b):
        ph = from_internal(u[f], geo, axes)            # (Nx,Ny,Nz)
        out.append(ph.transpose(2, 1, 0).reshape(-1))  # x fastest
    return np.array(out)


def collect_entries(case):
    ent = []
    if hasattr(case, "source_entries"):
        ent += case.source_entries()
    if hasattr(case, "heater_entries"):
        ent += case.heater_entries()
    return ent


def build_spec(geo, case, params, nphase, axes=None):
    if getattr(geo, "gravity2D", False):
        raise NotImplementedError("gravity2D is orientation-dependent in the reference "
                                  "(singlephase.py:105-109) and used by no configuration")
    axes = tuple(axes) if axes is not None else choose_axes(geo)
    N = (geo.Nx, geo.Ny, geo.Nz)
    D = (geo.Dx, geo.Dy, geo.Dz if geo.dim == 3 else 1.0)
    Kphys = [geo.K_x, geo.K_y, getattr(geo, "K_z", 0.0)]
    spec = {
        "nphase": int(nphase),
        "dim": geo.dim,
        "axes": axes,
        "n": tuple(N[a] for a in axes),
        "h": tuple(D[a] for a in axes),
        "gaxis": axes.index(2) if geo.dim == 3 else -1,
        "phi": to_internal(geo.phi, geo, axes),
        "K": [to_internal(Kphys[a], geo, axes) for a in axes],
        "kT": to_internal(geo.kT, geo, axes),
        "prm": params.as_dict(),
    }
    ent = collect_entries(case) if case is not None else []
    ent.sort(key=lambda e: (phys_flat_to_internal([e[0]], geo, axes)[0], e[1]))
    cells = np.array([e[0] for e in ent], dtype=np.int64)
    kx = _flat(geo.K_x, geo)
    ky = _flat(geo.K_y, geo)
    spec["sources"] = {
        "cell": phys_flat_to_internal(cells, geo, axes) if len(ent) else np.zeros(0, np.int64),
        "kind": np.array([e[1] for e in ent], dtype=np.int32),
        "wt": np.array([e[2] for e in ent], dtype=float),
        "bhp": np.array([e[3] for e in ent], dtype=float),
        "max_rate": np.array([e[4] for e in ent], dtype=float),
        "WI": np.array([peaceman_WI(kx[c], ky[c]) if e[1] != HEATER else 0.0 for e, c in zip(ent, cells)], dtype=float),
        "const": np.array([1 if e[5] else 0 for e in ent], dtype=np.int32),
    }
    return spec


def _flat(field, geo):
    """Physical field as flat x-fastest array."""
    a = np.asarray(field, dtype=float)
    n = geo.Nx*geo.Ny*geo.Nz
    if a.ndim == 0:
        return np.full(n, float(a))
    return a.reshape(geo.Nx, geo.Ny, geo.Nz).transpose(2, 1, 0).reshape(-1)

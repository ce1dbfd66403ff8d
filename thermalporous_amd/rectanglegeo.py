"""Structured 2-D grid metadata (mirror of /root/reference/thermalporous/rectanglegeo.py:4-65).

The reference builds a Firedrake ``RectangleMesh(Nx, Ny, L, Ly, quadrilateral=True)`` and a DQ0
space (:33,:65).  Here the "mesh" is just the grid description the HIP kernels need; fields are
numpy arrays of shape (Nx, Ny) (or scalars for the homogeneous models) indexed [i, j] exactly
like the ``slice_*.npy`` files (SPE10model.py:36-40).  Mesh hierarchies (:36-61) are geometric
multigrid experiments outside the hot path and are rejected.
"""
import numpy as np

from .mesh import StructuredMesh


class RectangleGeo():
    def __init__(self, Nx, Ny, params, Length=365.76, Length_y=365.76, mg={}):
        self.Nx = int(Nx)
        self.Ny = int(Ny)
        self.Nz = 1
        self.dim = 2
        self.params = params
        self.Length = Length
        self.Length_y = Length_y
        if bool(mg):
            raise NotImplementedError("mesh hierarchies (geometric MG) are outside the hot path")
        self.mesh = self.generate_mesh(self.Nx, self.Ny)
        self.comm = self.mesh.comm
        self.init_function_space()
        self.generate_geo_fields()  # defined in subclass
        try:
            self.K_x = self.K_x
            self.K_y = self.K_y
        except AttributeError:
            # isotropic fallback (rectanglegeo.py:20-26)
            self.K_x = self.K
            self.K_y = self.K
        self.gravity2D = False

    def generate_mesh(self, Nx, Ny):
        self.Dx = self.Length/Nx
        self.Dy = self.Length_y/Ny
        self.Dz = 1.0
        return StructuredMesh((Nx, Ny, 1), (self.Dx, self.Dy, 1.0), dim=2)

    def init_function_space(self):
        self.V = self.mesh.dq0()

    def cell_centres(self):
        x = (np.arange(self.Nx) + 0.5)*self.Dx
        y = (np.arange(self.Ny) + 0.5)*self.Dy
        return np.meshgrid(x, y, indexing="ij")

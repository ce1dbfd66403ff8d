"""Structured 3-D grid metadata (mirror of /root/reference/thermalporous/boxgeo.py:3-86).

Reference: ``ExtrudedMesh(RectangleMesh(Nx,Ny,...), Nz, Dz)`` + DQ0 (:42-43,:86).  Fields are
numpy arrays of shape (Nx, Ny, Nz), index k increasing upwards (z up; gravity acts along -z).
"""
import numpy as np

from .mesh import StructuredMesh


class BoxGeo():
    def __init__(self, Nx, Ny, Nz, params, Length=365.76, Length_y=365.76, Length_z=1.8288, mg=False):
        self.Nx = int(Nx)
        self.Ny = int(Ny)
        self.Nz = int(Nz)
        self.dim = 3
        self.params = params
        self.Length = Length
        self.Length_y = Length_y
        self.Length_z = Length_z
        if bool(mg):
            raise NotImplementedError("mesh hierarchies (geometric MG) are outside the hot path")
        self.mesh = self.generate_mesh(self.Nx, self.Ny, self.Nz)
        self.comm = self.mesh.comm
        self.init_function_space()
        self.generate_geo_fields()  # defined in subclass
        try:
            self.K_x = self.K_x
            self.K_y = self.K_y
            self.K_z = self.K_z
        except AttributeError:
            # isotropic fallback (boxgeo.py:21-29)
            self.K_x = self.K
            self.K_y = self.K
            self.K_z = self.K
        self.gravity2D = False

    def generate_mesh(self, Nx, Ny, Nz):
        self.Dx = self.Length/Nx
        self.Dy = self.Length_y/Ny
        self.Dz = self.Length_z/Nz
        return StructuredMesh((Nx, Ny, Nz), (self.Dx, self.Dy, self.Dz), dim=3)

    def init_function_space(self):
        self.V = self.mesh.dq0()

    def cell_centres(self):
        x = (np.arange(self.Nx) + 0.5)*self.Dx
        y = (np.arange(self.Ny) + 0.5)*self.Dy
        z = (np.arange(self.Nz) + 0.5)*self.Dz
        return np.meshgrid(x, y, z, indexing="ij")

"""Minimal stand-ins for the Firedrake objects the reference's driver scripts touch
(``geo.mesh``, ``geo.mesh.comm``, ``geo.V``): plain descriptors, no arithmetic."""
import os


class Comm():
    """Rank/size of the one-process-per-GPU job (torchrun environment), like ``mesh.comm``."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.size = int(os.environ.get("WORLD_SIZE", "1"))


class DQ0():
    """Piecewise-constant space on the structured grid: one dof per cell, x fastest."""

    def __init__(self, mesh):
        self._mesh = mesh

    def mesh(self):
        return self._mesh

    def dim(self):
        n = self._mesh.N
        return n[0]*n[1]*n[2]


class StructuredMesh():
    def __init__(self, N, D, dim):
        self.N = tuple(int(v) for v in N)
        self.D = tuple(float(v) for v in D)
        self.dim = dim
        self.comm = Comm()

    def dq0(self):
        return DQ0(self)

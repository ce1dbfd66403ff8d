"""SPE10 3-D model (mirror of /root/reference/thermalporous/SPE10model3D.py:6-80).

Cell sizes 6.096 x 3.048 x 0.6096 m (:11-13).  Loads ``data/slice_{phi,perm_x,perm_y,perm_z}.npy``
of shape (Nx, Ny, Nz) when present, else the synthetic field of SURVEY.md 8d.
``phi += 1e-10`` (:28); ``kT = phi*ko + (1-phi)*kr`` (:72).
"""
import os

import numpy as np

from .boxgeo import BoxGeo
from .data.synthetic_spe10 import kill_cells, synthetic_spe10, upsample


class SPE10Model3D(BoxGeo):
    def __init__(self, Nx, Ny, Nz, params, save=False, data_dir=None, seed=10, refine=1):
        self.geotype = "SPE10 " + str(Nx) + 'X' + str(Ny) + 'X' + str(Nz)
        self.name = self.geotype
        self.save = save
        self.data_dir = data_dir
        self.seed = seed
        self.refine = int(refine)
        Dx, Dy, Dz = 6.096/refine, 3.048/refine, 0.6096/refine
        BoxGeo.__init__(self, Nx, Ny, Nz, params, Length=Nx*Dx, Length_y=Ny*Dy, Length_z=Nz*Dz)

    def generate_geo_fields(self):
        d = self.data_dir or os.path.join(os.path.dirname(__file__), "data")
        names = ("phi", "perm_x", "perm_y", "perm_z")
        r = self.refine
        if os.path.exists(os.path.join(d, "slice_perm_z.npy")):
            f = {k: np.load(os.path.join(d, "slice_%s.npy" % k)) for k in names}
            self.data_source = "slice_*.npy in " + d
        else:
            # refined models (config 5): K keeps the coarse field's piecewise-constant contrast, the zero-porosity cells
            # are punched AFTER the refinement (isolated dead cells, as in the unrefined model)
            f = synthetic_spe10(-(-self.Nx//r), -(-self.Ny//r), -(-self.Nz//r), seed=self.seed, dead=(r == 1))
            self.data_source = "synthetic SPE10-like field, default_rng(%d)" % self.seed
        if r > 1:
            f = upsample(f, r)
            if not os.path.exists(os.path.join(d, "slice_perm_z.npy")):
                kill_cells(f["phi"], self.seed + 1)
                self.data_source += ", refined x%d, dead cells drawn at the fine resolution (default_rng(%d))" % (r, self.seed + 1)
        sl = (slice(0, self.Nx), slice(0, self.Ny), slice(0, self.Nz))
        self.phi = f["phi"][sl] + 1e-10          # removing rock only cells (:28)
        self.K_x = f["perm_x"][sl].copy()
        self.K_y = f["perm_y"][sl].copy()
        self.K_z = f["perm_z"][sl].copy()
        self.kT = self.phi*self.params.ko + (1-self.phi)*self.params.kr

"""SPE10 2-D layer model (mirror of /root/reference/thermalporous/SPE10model.py:6-70).

Loads ``data/slice_{phi,perm_x,perm_y}.npy`` (shape (Nx, Ny), indexed by cell centre ->
[floor(x/Dx), floor(y/Dy)], :36-40) when present; otherwise the synthetic SPE10-like layer of
SURVEY.md 8d (the raw SPE10 data are not shipped with the reference).  ``phi += 1e-10`` (:34);
``kT = phi*ko + (1-phi)*kr`` (:64).
"""
import os

import numpy as np

from .rectanglegeo import RectangleGeo
from .data.synthetic_spe10 import synthetic_spe10


class SPE10Model(RectangleGeo):
    def __init__(self, Nx, Ny, params, save=False, plane='xy', data_dir=None, seed=10):
        self.geotype = "SPE10"
        self.name = self.geotype
        self.save = save
        self.data_dir = data_dir
        self.seed = seed
        if plane == 'xy':
            Dx, Dy = 6.096, 3.048
        elif plane == 'xz':
            Dx, Dy = 6.096, 0.6096
        elif plane == 'yz':
            Dx, Dy = 3.048, 0.6096
        RectangleGeo.__init__(self, Nx, Ny, params, Length=Nx*Dx, Length_y=Ny*Dy)

    def generate_geo_fields(self):
        d = self.data_dir or os.path.join(os.path.dirname(__file__), "data")
        if os.path.exists(os.path.join(d, "slice_phi.npy")):
            f = {k: np.load(os.path.join(d, "slice_%s.npy" % k)) for k in ("phi", "perm_x", "perm_y")}
            self.data_source = "slice_*.npy in " + d
        else:
            f = synthetic_spe10(self.Nx, self.Ny, seed=self.seed)
            self.data_source = "synthetic SPE10-like field, default_rng(%d)" % self.seed
        sl = (slice(0, self.Nx), slice(0, self.Ny))
        self.phi = f["phi"][sl] + 1e-10          # removing rock only cells (:34)
        self.K_x = f["perm_x"][sl].copy()
        self.K_y = f["perm_y"][sl].copy()
        self.kT = self.phi*self.params.ko + (1-self.phi)*self.params.kr

"""Homogeneous 3-D model (mirror of /root/reference/thermalporous/homogeneousboxgeo.py:4-20)."""
from .boxgeo import BoxGeo


class HomogeneousBoxGeo(BoxGeo):
    def __init__(self, Nx, Ny, Nz, params, Length=365.76, Length_y=365.76, Length_z=365.76, mg=None):
        self.geotype = "Homogeneous"
        self.name = self.geotype + " " + str(Nx) + "X" + str(Ny) + "X" + str(Nz) + " grid"
        BoxGeo.__init__(self, Nx, Ny, Nz, params, Length=Length, Length_y=Length_y, Length_z=Length_z, mg=mg)

    def generate_geo_fields(self):
        self.phi = 0.2                       # (:13)
        self.K = 3E-7                        # mm^2 (:16)
        self.kT = self.phi*self.params.ko + (1-self.phi)*self.params.kr   # (:19-20)

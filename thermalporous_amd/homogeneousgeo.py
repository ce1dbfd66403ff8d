"""Homogeneous 2-D model (mirror of /root/reference/thermalporous/homogeneousgeo.py:4-21)."""
from .rectanglegeo import RectangleGeo


class HomogeneousGeo(RectangleGeo):
    def __init__(self, Nx, Ny, params, Length, Length_y, mg={}):
        self.geotype = "Homogeneous"
        RectangleGeo.__init__(self, Nx, Ny, params, Length, Length_y, mg)
        self.name = self.geotype + " " + str(self.Nx) + "X" + str(self.Ny) + " grid"

    def generate_geo_fields(self):
        self.phi = 0.2                       # (:13)
        self.K = 3E-7                        # mm^2 (:16)
        self.kT = self.phi*self.params.ko + (1-self.phi)*self.params.kr   # (:19-20)

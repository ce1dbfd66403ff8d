"""Heaters: energy source U*(T_inj - T)*delta (mirror of /root/reference/thermalporous/heatercase.py:6-118)."""
from . import utils


class HeaterCase():

    def __init__(self, params, geo, well_case=None, heater_points=list()):
        self.name = 'Heaters'
        self.Length = geo.Length
        self.Length_y = geo.Length_y
        if geo.dim == 3:
            self.Length_z = geo.Length_z
        self.V = geo.V
        self.mesh = geo.mesh
        self.geo = geo
        self.params = params
        L, Ly = self.Length, self.Length_y
        pts = None
        if self.geo.dim == 2:
            if well_case == "default":
                pts = [[0.2*L, Ly/2]] + [[0.8*L, Ly/2]]
            elif well_case == "SPE10_60x120":
                pts = [[140.0, 210.0]] + [[265.0, 260.0]]
            elif well_case == "test0":
                pts = ([[2., Ly/4.], [2., Ly/2.], [2., 3.*Ly/4]]
                       + [[L-2., Ly/4.], [L-2., Ly/2.], [L-2., 3.*Ly/4]])
            elif well_case == "test":
                f = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]
                pts = [[10., a*Ly] for a in f] + [[L - 10., a*Ly] for a in f]
        elif self.geo.dim == 3:
            Lz = self.Length_z
            if well_case == "default":
                pts = [[L/2, Ly/2, Lz*0.2]] + [[L/2, Ly/2, Lz*0.8]]
            if well_case == "multiple":
                pts = ([[L/4, Ly/2, Lz*0.2], [L/2, Ly/2, Lz*0.2], [3*L/4, Ly/2, Lz*0.2]]
                       + [[L/4, Ly/2, Lz*0.8], [L/2, Ly/2, Lz*0.8], [3*L/4, Ly/2, Lz*0.8]])
        if pts is not None:
            heater_points = pts
        self.init_heaters(heater_points, 'circle')

    def init_heaters(self, heater_points, wellfunc):
        self.heaters = []
        self.heatercount = 0
        for point in heater_points:
            self.heaters.append(self.make_heater(point, wellfunc))

    def make_heater(self, w, wellfunc):
        if wellfunc == 'delta':
            delta = utils.well_delta(self.geo, w)
        elif self.geo.dim == 2:
            delta = utils.well_circle(self.geo, w, 0.1)                 # (:81)
        else:
            delta = utils.well_circle(self.geo, w, 0.1, height=1.0)     # (:97-98)
        current_count = str(self.heatercount)
        self.heatercount += 1
        return {'name': 'heater' + current_count, 'location': w, 'delta': delta}

    def heater_entries(self):
        from .problem import HEATER
        out = []
        for h in self.heaters:
            d = h['delta']
            for c, wt in zip(d.cells, d.weights):
                out.append((int(c), HEATER, float(wt), 0.0, 0.0, False))
        return out

"""Wells as normalised DG0 delta functions with Peaceman or constant rates.

Mirror of /root/reference/thermalporous/wellcase.py:6-266: same constructor, same named well
patterns (:26-64), same well dicts ``{'name','bhp','location','delta','max_rate','rate'}`` (:108).
The rate laws themselves (flow_rate_ :171-199, flow_rate_twophase_ :204-235 and the constant
variants) are evaluated on the device inside the assembly kernels; here they are only
described (``constant_rate`` flag, bhp, max_rate) and re-evaluated on the host for the
per-time-step rate diagnostics (thermalmodel.py:232-294).
"""
import numpy as np

from . import utils


def peaceman_WI(Kx, Ky):
    """2*pi*h*Ke/ln(ro/rw) with the hard-coded Dx=Dy=h=5, rw=0.1 of wellcase.py:182-191."""
    h, rw, Dx, Dy = 5.0, 0.1, 5.0, 5.0
    ro = 0.28*((Ky/Kx)**0.5*Dx**2 + (Kx/Ky)**0.5*Dy**2)**0.5/((Ky/Kx)**0.25 + (Kx/Ky)**0.25)
    Ke = (Kx*Ky)**0.5
    return 2*np.pi*h*Ke/np.log(ro/rw)


class WellCase():

    def __init__(self, params, geo, well_case=None, prod_points=None, inj_points=None, constant_rate=False):
        self.name = 'Wells'
        self.Length = geo.Length
        self.Length_y = geo.Length_y
        if geo.dim == 3:
            self.Length_z = geo.Length_z
        self.V = geo.V
        self.mesh = geo.mesh
        self.geo = geo
        self.params = params
        self.wellfunc = 'circle'
        self.constant_rate = bool(constant_rate)
        pts = self.named_points(well_case)
        if pts is not None:
            prod_points, inj_points = pts
        self.init_wells(prod_points or [], inj_points or [], self.wellfunc)

    def named_points(self, well_case):
        """Named patterns of wellcase.py:26-64 (including the duplicated point of 'large')."""
        L, Ly = self.Length, self.Length_y
        if self.geo.dim == 2:
            if well_case == "default":
                return [[0.2*L, Ly/2]], [[0.8*L, Ly/2]]
            elif well_case == "SPE10_60x120":
                return [[140.0, 210.0]], [[265.0, 260.0]]
            elif well_case == "test0":
                return ([[2., Ly/4.], [2., Ly/2.], [2., 3.*Ly/4]],
                        [[L-2., Ly/4.], [L-2., Ly/2.], [L-2., 3.*Ly/4]])
            elif well_case == "test":
                f = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]
                return [[10., a*Ly] for a in f], [[L - 10., a*Ly] for a in f]
            elif well_case == "SPE10_40x40":
                Dx = self.geo.Dx
                return ([[2*Dx, Ly/4.], [2*Dx, Ly/2.], [2*Dx, 3.*Ly/4]],
                        [[L-2*Dx, Ly/4.], [L-2*Dx, Ly/2.], [L-2*Dx, 3.*Ly/4]])
        elif self.geo.dim == 3:
            Lz = self.Length_z
            if well_case == "default":
                return [[L/2, Ly/2, Lz*0.2]], [[L/2, Ly/2, Lz*0.8]]
            if well_case == "large":
                xs = [L/8, L/4, 3*L/8, L/2, 5*L/8, 3*L/4, 7*L/8]

                def pattern(z):
                    third = [[x, 3*Ly/4, z] for x in xs[:-1]] + [[7*L/8, Ly/4, z]]   # sic, :63-64
                    return [[x, Ly/2, z] for x in xs] + [[x, Ly/4, z] for x in xs] + third
                return pattern(Lz*0.2), pattern(Lz*0.8)
        return None

    def init_wells(self, prod_points, inj_points, wellfunc):
        self.prod_wells = []
        self.inj_wells = []
        self.prodcount = 0
        self.injcount = 0
        for point in prod_points:
            self.prod_wells.append(self.make_well(point, wellfunc, 'prod'))
        for point in inj_points:
            self.inj_wells.append(self.make_well(point, wellfunc, 'inj'))

    def make_well(self, w, wellfunc, welltype):
        rate = self.params.rate
        if wellfunc == 'delta':
            delta = self.well_delta(w)
        else:
            delta = self.well_circle(w) if self.geo.dim == 2 else self.well_circle3D(w)
        if welltype == 'prod':
            current_count = str(self.prodcount)
            bhp = self.params.p_prod
            max_rate = -rate
            self.prodcount += 1
        elif welltype == 'inj':
            current_count = str(self.injcount)
            bhp = self.params.p_inj
            max_rate = rate
            self.injcount += 1
        return {'name': welltype + current_count, 'bhp': bhp, 'location': w, 'delta': delta,
                'max_rate': max_rate, 'rate': 0.0}

    def well_circle(self, w):
        return utils.well_circle(self.geo, w, self.params.well_radius)

    def well_circle3D(self, w):
        return utils.well_circle(self.geo, w, self.params.well_radius, height=1.0)   # (:145)

    def well_delta(self, w):
        return utils.well_delta(self.geo, w)

    # ---- description for the compute engine --------------------------------------------
    def source_entries(self):
        """Flatten wells into per-cell source entries (cell, kind, weight, bhp, max_rate, WI, const)."""
        from .problem import PROD, INJ
        out = []
        for kind, wells in ((PROD, self.prod_wells), (INJ, self.inj_wells)):
            for well in wells:
                d = well['delta']
                for c, wt in zip(d.cells, d.weights):
                    out.append((int(c), kind, float(wt), float(well['bhp']), float(well['max_rate']),
                                self.constant_rate))
        return out

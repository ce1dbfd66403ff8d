"""Source terms with ONE summed delta field per well type (mirror of
/root/reference/thermalporous/sourceterms.py:6-269): same arithmetic as WellCase/HeaterCase, cheaper
for many wells; selected by ``case.name.startswith("Sources")`` (singlephase.py:134, twophase.py:186).
Bhp/rates come from ``params.p_prod/p_inj`` and ``params.prod_rate/inj_rate`` (:19-26,:159-160,:186-187).
In 3-D the bump height is 0.1 here (:116), not 1.0."""
import numpy as np

from . import utils
from .wellcase import WellCase


class SourceTerms():

    def __init__(self, params, geo, well_case=None, prod_points=list(), inj_points=list(),
                 heater_points=list(), constant_rate=False):
        self.name = 'Sources'
        self.Length = geo.Length
        self.Length_y = geo.Length_y
        if geo.dim == 3:
            self.Length_z = geo.Length_z
        self.V = geo.V
        self.mesh = geo.mesh
        self.geo = geo
        self.params = params
        if not hasattr(self.params, "prod_rate"):
            self.params.prod_rate = self.params.rate
        if not hasattr(self.params, "inj_rate"):
            self.params.inj_rate = self.params.rate
        self.constant_rate = bool(constant_rate)
        pts = WellCase.named_points(self, well_case)
        if pts is not None:
            prod_points, inj_points = pts
        self.init_deltas(prod_points, inj_points, heater_points, 'circle')

    def init_deltas(self, prod_points, inj_points, heater_points, well_func):
        deltas = self.make_deltas if well_func == 'delta' else self.make_circles
        self.deltas_prod = deltas(prod_points)
        self.deltas_inj = deltas(inj_points)
        self.deltas_heaters = deltas(heater_points)

    def _merge(self, ds, accumulate):
        vol = utils.cell_volume(self.geo)
        acc = {}
        for d in ds:
            for c, v in zip(d.cells, d.values):
                acc[int(c)] = (acc.get(int(c), 0.0) + v) if accumulate else v
        cells = sorted(acc)
        return utils.Delta(cells, [acc[c] for c in cells], vol)

    def make_circles(self, ws):                      # (:88-96): deltas accumulate
        h = None if self.geo.dim == 2 else 0.1       # (:116)
        return self._merge([utils.well_circle(self.geo, w, self.params.well_radius, height=h) for w in ws], True)

    def make_deltas(self, ws):                       # (:124-137): vec[node] = 1.0 (no accumulation)
        return self._merge([utils.well_delta(self.geo, w) for w in ws], False)

    def source_entries(self):
        from .problem import PROD, INJ, HEATER
        p = self.params
        out = []
        for c, wt in zip(self.deltas_prod.cells, self.deltas_prod.weights):
            out.append((int(c), PROD, float(wt), float(p.p_prod), -float(p.prod_rate), self.constant_rate))
        for c, wt in zip(self.deltas_inj.cells, self.deltas_inj.weights):
            out.append((int(c), INJ, float(wt), float(p.p_inj), float(p.inj_rate), self.constant_rate))
        for c, wt in zip(self.deltas_heaters.cells, self.deltas_heaters.weights):
            out.append((int(c), HEATER, float(wt), 0.0, 0.0, False))
        return out

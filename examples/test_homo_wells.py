"""BASELINE config 1: single-phase 2-D homogeneous N x N grid with constant-rate wells -- the reference's
tests/test_homo_wells.py (:1-179) on the HIP engine.  Usage (same CLI):  python test_homo_wells.py cpr 1.0 100"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from thermalporous_amd.physicalparameters import PhysicalParameters as Params
from thermalporous_amd.homogeneousgeo import HomogeneousGeo as GeoModel
from thermalporous_amd.wellcase import WellCase as TestCase
from thermalporous_amd.singlephase import SinglePhase as ThermalModel

params = Params()
params.rate = 1e-6               # (:10)
params.T_prod = 320.0            # (:12)

pcname = sys.argv[1] if len(sys.argv) > 1 else "cpr"
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
end = 2*dt                       # (:31)
maxdt = dt
N = int(sys.argv[3]) if len(sys.argv) > 3 else 50
L = 20.

geo = GeoModel(N, N, params, L, L)
case = TestCase(params, geo, well_case="test0", constant_rate=True)   # (:51)

presets = {"cpr": "pc_cpr", "cpr_QI": "pc_cpr_QI", "cpr_TI": "pc_cpr_TI"}
if pcname not in presets:
    raise SystemExit("pcname must be one of %s (the other presets of the reference are outside the hot path)"
                     % sorted(presets))
suffix = os.path.splitext(__file__)[0]
model = ThermalModel(geo, case, params, end=end, maxdt=maxdt, save=False, small_dt_start=False,
                     solver_parameters=presets[pcname], filename=suffix + "_" + pcname + "_results.txt")
model.solve()

"""BASELINE config 3: two-phase 2-D SPE10 slice with Peaceman wells -- the reference's
tests_twophase/test_60x120_wells_default.py (:1-42) on the HIP engine, same CLI:
    python test_60x120_wells_default.py cptr 1.0 10.0        # pcname maxdt[days] end[days]
SPE10 data: `slice_*.npy` written by thermalporous_amd/data/create_SPE10_slice*.py when the SPE10 .dat files
are present; otherwise the seeded synthetic SPE10-like layer."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from thermalporous_amd.physicalparameters import PhysicalParameters as Params
from thermalporous_amd.SPE10model import SPE10Model as GeoModel
from thermalporous_amd.wellcase import WellCase as TestCase
from thermalporous_amd.twophase import TwoPhase as ThermalModel

params = Params()
params.rate = 2e-4               # (:8)
params.S_o = 0.9                 # (:9)

pcname = sys.argv[1] if len(sys.argv) > 1 else "cptr"
maxdt = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
end = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0

Nx, Ny = 60, 120                 # (:23-24)
geo = GeoModel(Nx, Ny, params)
case = TestCase(params, geo, well_case="SPE10_60x120")       # (:33)

suffix = os.path.splitext(__file__)[0]
model = ThermalModel(geo, case, params, end=end, maxdt=maxdt, save=False, n_save=1, small_dt_start=False,
                     solver_parameters="pc_" + pcname, filename=suffix + "_" + pcname + "_results.txt")
model.solve()

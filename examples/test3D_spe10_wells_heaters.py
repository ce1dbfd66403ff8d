"""BASELINE config 4: two-phase 3-D SPE10-like box with wells and heaters, pc_cptr -- the workload of bench.py as a
driver script in the reference's style (cf. tests_twophase/test3D_homo_wells.py).  Usage:
    python test3D_spe10_wells_heaters.py cptr 0.1 1.0 [Nx Ny Nz]      # pcname maxdt[days] end[days]
Multi-GPU: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 test3D_spe10_wells_heaters.py ..."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from thermalporous_amd.physicalparameters import PhysicalParameters as Params
from thermalporous_amd.SPE10model3D import SPE10Model3D as GeoModel
from thermalporous_amd.wellheatercase import WellHeaterCase as TestCase
from thermalporous_amd.twophase import TwoPhase as ThermalModel

params = Params()
params.rate = 2e-4
params.S_o = 0.9
params.T_inj = 373.15

pcname = sys.argv[1] if len(sys.argv) > 1 else "cptr"
maxdt = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
end = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
Nx, Ny, Nz = (int(v) for v in sys.argv[4:7]) if len(sys.argv) > 6 else (60, 220, 85)

geo = GeoModel(Nx, Ny, Nz, params)
L, Ly, Lz = geo.Length, geo.Length_y, geo.Length_z
case = TestCase(params, geo, prod_points=[[140.0/365.76*L, 210.0/670.56*Ly, 0.2*Lz]],
                inj_points=[[265.0/365.76*L, 260.0/670.56*Ly, 0.8*Lz]])

suffix = os.path.splitext(__file__)[0]
model = ThermalModel(geo, case, params, end=end, maxdt=maxdt, save=False, small_dt_start=True,
                     solver_parameters="pc_" + pcname, filename=suffix + "_" + pcname + "_results.txt")
model.solve()
print("total Newton its %d, total Krylov its %d, last dt %.4g days, failed solves %d"
      % (model.total_nits, model.total_lits, model.last_dt/86400.0, model.failed_solves))

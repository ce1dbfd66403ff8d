"""Host-side mirror of the reference API: geometry, wells, problem description, solver options,
the ThermalModel time loop (driven by the CPU oracle engine injected through the test hook) and
the device-state cache protocol.  No GPU needed."""
import numpy as np
import pytest

import cases
from oracle.engine import OracleEngine
from thermalporous_amd import exceptions
from thermalporous_amd.physicalparameters import PhysicalParameters
from thermalporous_amd.homogeneousgeo import HomogeneousGeo
from thermalporous_amd.homogeneousboxgeo import HomogeneousBoxGeo
from thermalporous_amd.SPE10model import SPE10Model
from thermalporous_amd.SPE10model3D import SPE10Model3D
from thermalporous_amd.wellcase import WellCase
from thermalporous_amd.heatercase import HeaterCase
from thermalporous_amd.wellheatercase import WellHeaterCase
from thermalporous_amd.sourceterms import SourceTerms
from thermalporous_amd.singlephase import SinglePhase
from thermalporous_amd.twophase import TwoPhase
from thermalporous_amd import problem, utils
from thermalporous_amd.solver_options import engine_options


def test_well_cell_selection_and_tie_break():
    p = PhysicalParameters()
    g = HomogeneousGeo(10, 10, p, 20., 20.)
    c = WellCase(p, g, well_case="test0", constant_rate=True)
    # wells at x=2 are equidistant from centres x=1 and x=3: first (lowest flat index) wins (utils.py:14-18)
    assert [int(w["delta"].cells[0]) for w in c.prod_wells] == [20, 40, 70]
    assert all(np.isclose(w["delta"].weights.sum(), 1.0) for w in c.prod_wells + c.inj_wells)
    assert c.prod_wells[0]["max_rate"] == -p.rate and c.inj_wells[0]["bhp"] == p.p_inj
    assert c.prod_wells[0]["name"] == "prod0" and c.inj_wells[2]["name"] == "inj2"


def test_well_circle_normalisation_and_3d_height():
    p = PhysicalParameters()
    g = HomogeneousBoxGeo(20, 20, 10, p, Length=1.0, Length_y=1.0, Length_z=5.0)   # 5 cm cells: bump covers several
    d = utils.well_circle(g, [0.5, 0.5, 2.5], 0.1, height=1.0)
    assert len(d.cells) > 4 and np.isclose(d.weights.sum(), 1.0)
    zc = (d.cells//(20*20) + 0.5)*0.5
    assert np.all(np.abs(zc - 2.5) < 1.0)
    d2 = utils.well_circle(g, [0.5, 0.5, 2.5], 0.1, height=0.1)               # sourceterms.py:116
    assert len(d2.cells) < len(d.cells)


def test_large_pattern_has_duplicated_well():
    p = PhysicalParameters()
    g = HomogeneousBoxGeo(16, 16, 10, p, Length=50., Length_y=50., Length_z=50.)
    c = WellCase(p, g, well_case="large", constant_rate=True)
    assert len(c.prod_wells) == 21 and len(c.inj_wells) == 21
    locs = [tuple(w["location"]) for w in c.prod_wells]
    assert len(set(locs)) == 20          # the duplicated [7Lx/8, Ly/4] of wellcase.py:63-64 is preserved


def test_sources_variant_flattens_to_same_entries():
    p = PhysicalParameters()
    g = HomogeneousGeo(12, 12, p, 20., 20.)
    a = problem.build_spec(g, WellCase(p, g, well_case="test0"), p, 1)["sources"]
    b = problem.build_spec(g, SourceTerms(p, g, well_case="test0"), p, 1)["sources"]
    for k in ("cell", "kind", "wt", "bhp", "max_rate", "WI"):
        assert np.allclose(a[k], b[k])
    h = problem.build_spec(g, HeaterCase(p, g, well_case="test0"), p, 1)["sources"]
    assert (h["kind"] == problem.HEATER).all() and len(h["cell"]) == 6
    wh = problem.build_spec(g, WellHeaterCase(p, g, well_case="test0"), p, 1)["sources"]
    assert len(wh["cell"]) == 12


def test_spec_axes_and_roundtrip():
    p = PhysicalParameters()
    g = SPE10Model3D(6, 11, 5, p)
    spec = problem.build_spec(g, None, p, 2)
    assert spec["axes"] == (2, 0, 1) and spec["n"] == (5, 6, 11) and spec["gaxis"] == 0
    assert np.isclose(spec["h"][0], 0.6096) and np.isclose(spec["h"][2], 3.048)
    u = np.arange(3*6*11*5, dtype=float).reshape(3, -1)
    ui = problem.field_major_to_internal(u, g, spec["axes"], 3)
    assert ui.shape == (3, 11, 6, 5)
    assert np.array_equal(problem.internal_to_field_major(ui, g, spec["axes"], 3), u)
    cells = np.arange(6*11*5)
    assert np.array_equal(ui[0].reshape(-1)[problem.phys_flat_to_internal(cells, g, spec["axes"])], u[0])
    # fields: phi gets +1e-10 (SPE10model3D.py:28) and kT = phi ko + (1-phi) kr
    assert spec["phi"].min() >= 1e-10
    assert np.allclose(spec["kT"], spec["phi"]*p.ko + (1 - spec["phi"])*p.kr)
    g2 = SPE10Model(7, 9, p)
    assert problem.build_spec(g2, None, p, 1)["axes"] == (0, 1, 2)


def test_synthetic_field_is_reproducible_and_spe10_like():
    from thermalporous_amd.data.synthetic_spe10 import synthetic_spe10, upsample, MD_TO_MM2
    a = synthetic_spe10(20, 30, 8)
    b = synthetic_spe10(20, 30, 8)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    k_md = a["perm_x"]/MD_TO_MM2
    assert k_md.min() >= 1e-3 and k_md.max() <= 2e4 and np.log10(k_md).std() > 1.0
    assert 0.01 < (a["phi"] == 0).mean() < 0.05 and a["phi"].max() <= 0.5
    up = upsample(a, 2)
    assert up["phi"].shape == (40, 60, 16) and up["phi"][1, 1, 1] == a["phi"][0, 0, 0]


def test_spe10_slice_maker_layout(tmp_path):
    """create_SPE10_slice: x fastest, then y, then z with the TOP layer first; z flipped; mD -> mm^2."""
    from thermalporous_amd.data import create_SPE10_slice as mk
    n = mk.NX*mk.NY*mk.NZ
    phi = np.arange(n, dtype=float)
    np.savetxt(tmp_path/"spe_phi.dat", phi.reshape(-1, 6))
    np.savetxt(tmp_path/"spe_perm.dat", np.concatenate([phi, 2*phi, 3*phi]).reshape(-1, 6))
    mk.create_SPE10_slice(3, 4, 5, x_shift=1, y_shift=2, z_shift=3, dirname=str(tmp_path))
    s = np.load(tmp_path/"slice_phi.npy")
    assert s.shape == (3, 4, 5)
    i, j, kk = 2, 1, 4
    assert s[i, j, 5 - 1 - kk] == phi[(i + 1) + (j + 2)*60 + (kk + 3)*220*60]
    ky = np.load(tmp_path/"slice_perm_y.npy")
    assert np.isclose(ky[i, j, 5 - 1 - kk], 2*phi[(i + 1) + (j + 2)*60 + (kk + 3)*220*60]*9.869233e-10)
    # vertical sections (create_SPE10_slicexz.py:9-99): no z flip there; perm_y is Kz
    mk.create_SPE10_slicexz(4, 6, x_shift=2, y_shift=5, z_shift=1, dirname=str(tmp_path))
    s = np.load(tmp_path/"slice_phi.npy")
    assert s.shape == (4, 6) and s[3, 2] == phi[(3 + 2) + 5*60 + (2 + 1)*220*60]
    assert np.isclose(np.load(tmp_path/"slice_perm_y.npy")[3, 2], 3*phi[(3 + 2) + 5*60 + (2 + 1)*220*60]*9.869233e-10)
    mk.create_SPE10_sliceyz(5, 3, x_shift=7, y_shift=2, z_shift=4, dirname=str(tmp_path))
    s = np.load(tmp_path/"slice_phi.npy")
    assert s.shape == (5, 3) and s[4, 1] == phi[7 + (4 + 2)*60 + (1 + 4)*220*60]
    assert np.isclose(np.load(tmp_path/"slice_perm_x.npy")[4, 1], 2*phi[7 + (4 + 2)*60 + (1 + 4)*220*60]*9.869233e-10)


def test_solver_option_mapping_and_rejections():
    p = PhysicalParameters()
    g = HomogeneousGeo(8, 8, p, 20., 20.)
    c = WellCase(p, g, well_case="test0", constant_rate=True)
    m = SinglePhase(g, c, p, solver_parameters="pc_cpr_QI", filename=None, verbosity=False,
                    _engine_factory=OracleEngine)
    assert m.engine_opts["pc"] == "cpr" and m.engine_opts["decoup"] == "QI" and m.decoup == "QI"
    assert m.engine_opts["snes_max_it"] == 15 and m.engine_opts["ksp_restart"] == 200
    assert m.engine_opts["ksp_rtol"] == 1e-7           # Firedrake default made explicit
    assert set(m.appctx) >= {"pressure_space", "temperature_space", "params", "geo", "dt", "u_", "case", "decoup"}
    p2 = PhysicalParameters()
    p2.S_o = 0.9
    m2 = TwoPhase(g, WellCase(p2, g, well_case="test0", constant_rate=True), p2, solver_parameters="pc_cptr",
                  filename=None, verbosity=False, _engine_factory=OracleEngine)
    assert m2.engine_opts["pc"] == "cptr" and m2.engine_opts["ksp_rtol"] == 1e-8 and m2.engine_opts["snes_max_it"] == 25
    assert m2.i_S_o == 2 and m2.appctx["saturation_space"] == 2
    m3 = SinglePhase(g, c, p, solver_parameters="pc_fieldsplit_cd", filename=None, verbosity=False,
                     _engine_factory=OracleEngine)
    assert m3.engine_opts["pc"] == "fieldsplit_cd" and m3.engine_opts["decoup"] == "No"
    with pytest.raises(NotImplementedError):     # other Schur preconditioners (self, full, user) are not on the path
        engine_options({"pc_type": "fieldsplit", "pc_fieldsplit_type": "schur", "pc_fieldsplit_schur_fact_type": "FULL",
                        "pc_fieldsplit_schur_precondition": "self"}, "Single phase")
    with pytest.raises(NotImplementedError):     # the single-phase block PC does not exist for two phases
        engine_options(m3.solver_parameters, "Two-phase")
    m3s = SinglePhase(g, c, p, solver_parameters="pc_fieldsplit_selfp", filename=None, verbosity=False,       # (:322-330)
                      _engine_factory=OracleEngine)
    assert m3s.engine_opts["pc"] == "fieldsplit_cd" and m3s.engine_opts["schur_selfp"] is True and m3s.engine_opts["schur_a11"] is False
    assert m3.engine_opts["schur_selfp"] is False
    m7 = SinglePhase(g, c, p, solver_parameters="pc_fieldsplit_a11", filename=None, verbosity=False, _engine_factory=OracleEngine)
    assert m7.engine_opts["pc"] == "fieldsplit_cd" and m7.engine_opts["schur_a11"] is True and m3.engine_opts["schur_a11"] is False
    m8 = TwoPhase(g, WellCase(p2, g, well_case="test0", constant_rate=True), p2, solver_parameters="pc_cptr_a11", filename=None,
                  verbosity=False, _engine_factory=OracleEngine)
    assert m8.engine_opts["pc"] == "cptr" and m8.engine_opts["schur_a11"] is True and m2.engine_opts["schur_a11"] is False
    m9 = TwoPhase(g, WellCase(p2, g, well_case="test0", constant_rate=True), p2, solver_parameters="pc_cptramg_QI", filename=None,
                  verbosity=False, _engine_factory=OracleEngine)      # (:552-566) forces vector=True (:935-943)
    assert m9.engine_opts["pc"] == "cptramg" and m9.engine_opts["decoup"] == "QI" and m9.vector and m9.i_S_o == 1
    with pytest.raises(NotImplementedError):
        TwoPhase(g, c, p2, solver_parameters="pc_cptrlu", filename=None, _engine_factory=OracleEngine)
    # the pure-PETSc "*_gmres" emulations (twophase.py:619-699, singlephase.py:355-368) are the same algebra
    c2 = WellCase(p2, g, well_case="test0", constant_rate=True)
    m4 = TwoPhase(g, c2, p2, filename=None, verbosity=False, _engine_factory=OracleEngine)      # default preset (:930)
    assert m4.engine_opts["pc"] == "cptr" and m4.engine_opts["decoup"] == "No"
    m5 = TwoPhase(g, c2, p2, solver_parameters="pc_cpr_gmres", filename=None, verbosity=False, _engine_factory=OracleEngine)
    assert m5.engine_opts["pc"] == "cpr" and m5.engine_opts["decoup"] == "No"
    m6 = SinglePhase(g, c, p, solver_parameters="pc_cpr_gmres", filename=None, verbosity=False, _engine_factory=OracleEngine)
    assert m6.engine_opts["pc"] == "cpr"
    # ILU(1) second stage: pc_cprilu1_gmres (twophase.py:653-668); deeper fill is not built
    assert engine_options({**m5.solver_parameters, "sub_1_sub_pc_factor_levels": 1}, "Two-phase")["ilu_levels"] == 1
    assert m5.engine_opts["ilu_levels"] == 0
    m7 = TwoPhase(g, c2, p2, solver_parameters="pc_cprilu1_gmres", filename=None, verbosity=False, _engine_factory=OracleEngine)
    assert m7.engine_opts["pc"] == "cpr" and m7.engine_opts["ilu_levels"] == 1
    with pytest.raises(NotImplementedError):
        engine_options({**m5.solver_parameters, "sub_1_sub_pc_factor_levels": 2}, "Two-phase")
    # pc_fieldsplit_diag (singlephase.py:371-375): additive fieldsplit = V(A_pp) + V(A_TT)
    m8d = SinglePhase(g, c, p, solver_parameters="pc_fieldsplit_diag", filename=None, verbosity=False, _engine_factory=OracleEngine)
    assert m8d.engine_opts["pc"] == "fieldsplit_cd" and m8d.engine_opts["fs_additive"] is True and m3.engine_opts["fs_additive"] is False
    # pc_cptramg_gmres (twophase.py:698-713): the pure-PETSc emulation of pc_cptramg; forces vector=True (:953-955)
    m8e = TwoPhase(g, c2, p2, solver_parameters="pc_cptramg_gmres", filename=None, verbosity=False, _engine_factory=OracleEngine)
    assert m8e.engine_opts["pc"] == "cptramg" and m8e.vector is True and m8e.engine_opts["decoup"] == "No"
    assert m5.engine_opts["pc"] == "cpr"            # (pc_cpr_gmres names its fields: pressure only)
    # pc_bilu (twophase.py:758-762, singlephase.py:402-406): bjacobi + ILU(1) alone
    m8b = TwoPhase(g, c2, p2, solver_parameters="pc_bilu", filename=None, verbosity=False, _engine_factory=OracleEngine)
    assert m8b.engine_opts["pc"] == "bilu" and m8b.engine_opts["ilu_levels"] == 1
    m8c = SinglePhase(g, c, p, solver_parameters="pc_bilu", filename=None, verbosity=False, _engine_factory=OracleEngine)
    assert m8c.engine_opts["pc"] == "bilu" and m8c.engine_opts["ilu_levels"] == 1
    with pytest.raises(NotImplementedError):
        engine_options({**m8c.solver_parameters, "sub_pc_type": "lu"}, "Single phase")
    with pytest.raises(NotImplementedError):
        engine_options({"pc_type": "lu", "ksp_type": "preonly"}, "Single phase")
    with pytest.raises(KeyError):
        engine_options({**m.solver_parameters, "bogus_key": 1}, "Single phase")
    # a dict in the style of tests/test_homo_wells.py of the reference
    d = {"snes_type": "newtonls", "snes_max_it": 15, "ksp_type": "fgmres", "ksp_max_it": 200,
         "pc_type": "composite", "pc_composite_type": "multiplicative", "pc_composite_pcs": "python,bjacobi",
         "sub_0_pc_python_type": "thermalporous.preconditioners.CPRStage1PC",
         "sub_0_cpr_stage1": {"ksp_type": "preonly", "pc_type": "hypre", "pc_hypre_type": "boomeramg"},
         "sub_1_pc_bjacobi_blocks": 1, "sub_1_sub_pc_type": "ilu", "sub_1_sub_pc_factor_levels": 0, "mat_type": "aij"}
    assert engine_options(d, "Single phase")["pc"] == "cpr"


def test_fieldsplit_cd_oracle_time_loop():
    """pc_fieldsplit_cd (singlephase.py:309-319) drives BASELINE config 1 to the same state as pc_cpr."""
    sols = []
    for preset in ("pc_cpr", "pc_fieldsplit_cd"):
        spec, u0, p, g, c = cases.c1_homogeneous(N=10)
        m = SinglePhase(g, c, p, end=2.0, maxdt=1.0, small_dt_start=False, solver_parameters=preset,
                        filename=None, verbosity=False, _engine_factory=OracleEngine)
        m.solve()
        assert m.failed_solves == 0
        sols.append([m.u.dat.data_ro[f].copy() for f in range(2)])
    for f in range(2):
        assert np.linalg.norm(sols[0][f] - sols[1][f]) <= 1e-6*np.linalg.norm(sols[0][f])


def test_field_output_pvd_vti(tmp_path):
    """save=True writes pressure/temperature[/saturation_o].pvd collections of .vti pieces at t = 0 and every n_save
    steps (thermalmodel.py:113-133, 303-322); the last piece holds the final state."""
    import xml.etree.ElementTree as ET
    from thermalporous_amd.output import read_vti
    spec, u0, p, g, c = cases.c1_homogeneous(N=8, nphase=2)
    m = TwoPhase(g, c, p, end=4.0, maxdt=1.0, small_dt_start=False, save=True, n_save=2, solver_parameters="pc_cptr",
                 filename=str(tmp_path/"res.txt"), verbosity=False, _engine_factory=OracleEngine)
    m.solve()
    assert len(m.dt_vec) == 4
    for f, name in enumerate(("pressure", "temperature", "saturation_o")):
        coll = ET.parse(tmp_path/(name + ".pvd")).getroot().findall("Collection/DataSet")
        assert [float(d.get("timestep")) for d in coll] == [0.0, 2.0, 4.0]          # initial + steps 2 and 4
        nm, first = read_vti(tmp_path/coll[0].get("file"))
        nm2, last = read_vti(tmp_path/coll[-1].get("file"))
        assert nm == nm2 == name and first.size == last.size == 64
        assert np.array_equal(last, m.u.dat.data_ro[f])
        assert np.allclose(first, m.initial_condition[f] if np.ndim(m.initial_condition) > 1 else first)
    root = ET.parse(tmp_path/"pressure_0.vti").getroot().find("ImageData")
    assert root.get("WholeExtent") == "0 8 0 8 0 1" and root.get("Spacing").split()[0] == repr(g.Dx)


def test_config1_time_loop_with_oracle_engine(tmp_path):
    """BASELINE config 1 (tests/test_homo_wells.py: 2 steps of dt, const-rate wells, pc cpr) end to end."""
    spec, u0, p, g, c = cases.c1_homogeneous(N=12)
    f = tmp_path/"res.txt"
    m = SinglePhase(g, c, p, end=2.0, maxdt=1.0, small_dt_start=False, solver_parameters="pc_cpr",
                    filename=str(f), verbosity=True, _engine_factory=OracleEngine)
    m.solve()
    assert len(m.dt_vec) == 2 and m.last_dt == 86400.0
    assert m.total_nits == sum(m.nits_vec) and m.total_lits == sum(m.lits_vec) and m.total_nits >= 4
    txt = f.read_text()
    for key in ("nits = ", "lits = ", "dts = ", "timings = ", "Total CPU time (s):", "Total Linear iterations: ",
                "Total Nonlinear iterations: ", "Number of time-steps: ", "Average Linear iteration per Nonlinear iteration: "):
        assert key in txt
    pf = m.u.dat.data_ro[0]
    assert pf.shape == (144,) and pf.min() < p.p_ref < pf.max()      # producers draw down, injectors build up
    # mass balance: const-rate wells inject/produce equal volumes -> mean pressure stays near p_ref
    assert abs(pf.mean() - p.p_ref) < 1.0


class FlakyEngine(OracleEngine):
    """Fails the first `nfail` Newton solves (reason DIVERGED_MAX_IT) to exercise the dt-halving policy."""
    nfail = 2

    def newton_solve(self):
        if FlakyEngine.nfail > 0:
            FlakyEngine.nfail -= 1
            self.last = dict(nits=15, lits=0, reason=-5, fnorm=1.0, fnorm0=1.0)
            return self.last
        return OracleEngine.newton_solve(self)


def test_dt_halving_on_convergence_error():
    spec, u0, p, g, c = cases.c1_homogeneous(N=8)
    FlakyEngine.nfail = 2
    m = SinglePhase(g, c, p, end=0.5, maxdt=1.0, small_dt_start=False, solver_parameters="pc_cpr",
                    filename=None, verbosity=False, _engine_factory=FlakyEngine)
    m.solve()
    assert m.failed_solves == 2
    assert np.isclose(m.dt_vec[0], 86400.0/4)         # halved twice (thermalmodel.py:170-180)
    assert np.isclose(sum(m.dt_vec), 0.5*86400.0)     # clipped to the end time (:346-348)


def test_spe10_adaptive_dt_and_saturation_guard():
    spec, u0, p, g, c = cases.c3_spe10_2d(10, 14, 2)
    m = TwoPhase(g, c, p, end=0.01, maxdt=0.004, solver_parameters="pc_cptr", filename=None, verbosity=False,
                 _engine_factory=OracleEngine)
    m.solve()
    dts = np.array(m.dt_vec)
    assert np.isclose(dts[0], 2**-10*0.004*86400)     # dt_init_fact ramp (:97-102)
    grow = dts[1:]/dts[:-1]
    for k, n in enumerate(m.nits_vec[:-2]):
        if n < 6 and dts[k + 1] < 0.004*86400 - 1e-9:
            assert np.isclose(grow[k], 1 + min(1.0, (6 - n)**2/9.0))     # (:337-341)
    S = m.u.dat.data_ro[2]
    assert S.min() >= 0.0 and S.max() <= 1.0


def test_state_cache_protocol_moves_nothing_when_host_does_not_touch_state():
    spec, u0, p, g, c = cases.c1_homogeneous(N=8)

    class Counting(OracleEngine):
        pushes = pulls = 0

        def set_state(self, u):
            Counting.pushes += 1
            OracleEngine.set_state(self, u)

        def get_state(self):
            Counting.pulls += 1
            return OracleEngine.get_state(self)
    m = SinglePhase(g, c, p, end=3.0, maxdt=1.0, small_dt_start=False, solver_parameters="pc_cpr", filename=None,
                    verbosity=False, _engine_factory=Counting)
    m.solve()
    assert Counting.pushes == 1 and Counting.pulls == 0         # only the initial condition crosses
    _ = m.u.dat.data_ro[0]
    assert Counting.pulls == 1
    m.u.dat.data[1][...] += 1.0                                  # host write -> next solve pushes
    m.solver.solve()
    assert Counting.pushes == 2


def test_pc_classes_mirror_pcbase_protocol():
    from thermalporous_amd import preconditioners as pcs

    class Eng:
        def __init__(self):
            self.opts = {"pc": "cpr", "decoup": "No"}
            self.calls = []

        def set_options(self, **kw):
            self.opts.update(kw)
            self.calls.append(("set_options", kw))

        def pc_setup(self):
            self.calls.append("pc_setup")

        def stage1_apply(self, x, y):
            self.calls.append(("stage1_apply", x, y))
    e = Eng()
    pc = pcs.PC(e, {"decoup": "QI", "vector": False}, prefix="sub_0_")
    s1 = pcs.CPTRStage1PC()
    s1.setUp(pc)          # initialize
    s1.setUp(pc)          # update
    s1.apply(pc, "x", "y")
    assert e.opts == {"pc": "cptr", "decoup": "QI"}
    assert e.calls.count("pc_setup") == 2 and e.calls[-1] == ("stage1_apply", "x", "y")
    assert pc.getOptionsPrefix() == "sub_0_"
    e.b = 2                                        # single-phase engine: the _temp decouplings need (T, S)
    with pytest.raises(NotImplementedError):
        pcs.CPRStage1PC().setUp(pcs.PC(e, {"decoup": "QI_temp"}))
    with pytest.raises(NotImplementedError):
        pcs.CPRStage1PC().setUp(pcs.PC(e, {"decoup": "bogus"}))
    e.b = 3
    pcs.CPRStage1PC().setUp(pcs.PC(e, {"decoup": "TI_temp"}))
    assert e.opts["decoup"] == "TI_temp"


def test_convergence_error_is_raised_like_firedrake():
    spec, u0, p, g, c = cases.c1_homogeneous(N=8)
    FlakyEngine.nfail = 1
    m = SinglePhase(g, c, p, end=1.0, maxdt=1.0, small_dt_start=False, solver_parameters="pc_cpr", filename=None,
                    verbosity=False, _engine_factory=FlakyEngine)
    m.start()
    with pytest.raises(exceptions.ConvergenceError):
        m.solver.solve()


def test_bjacobi_blocks_option_maps_to_tiles_or_raises():
    """``sub_1_pc_bjacobi_blocks`` (tests/test_homo_wells.py:112,125 of the reference) is consumed: N boxes over the grid.
    The GPU engine realises it when a tile fits one wavefront (<= 64 columns) and raises otherwise; the oracle has no
    limit (one block = whole-grid ILU(0))."""
    from thermalporous_amd.engine import tiles_for_blocks
    from oracle.engine import blocks_to_tile
    assert tiles_for_blocks((400, 400, 1), 16) == (400, 25, 1)          # pc_cpr_bilu of test_homo_wells.py on N = 400
    assert tiles_for_blocks((85, 60, 220), 224) == (85, 9, 7)
    with pytest.raises(NotImplementedError):
        tiles_for_blocks((400, 400, 1), 1)                               # 400 columns > one wavefront
    with pytest.raises(NotImplementedError):
        tiles_for_blocks((85, 60, 220), 1)
    assert tiles_for_blocks((60, 8, 8), 1) == (60, 8, 8)
    assert blocks_to_tile((400, 400, 1), 1) == (400, 400, 1)
    assert blocks_to_tile((85, 60, 220), 1) == (85, 60, 220)
    d = {"snes_type": "newtonls", "ksp_type": "fgmres", "pc_type": "composite", "pc_composite_type": "multiplicative",
         "pc_composite_pcs": "python,bjacobi", "sub_0_pc_python_type": "thermalporous.preconditioners.CPRStage1PC",
         "sub_0_cpr_stage1": {"ksp_type": "preonly", "pc_type": "hypre", "pc_hypre_type": "boomeramg",
                              "pc_hypre_boomeramg_max_iter": 1},
         "sub_1_pc_bjacobi_blocks": 16, "sub_1_sub_pc_type": "ilu", "sub_1_sub_pc_factor_levels": 0, "mat_type": "aij"}
    assert engine_options(d, "Single phase")["bjacobi_blocks"] == 16
    # keys that used to pass silently by prefix are now errors or checked values
    with pytest.raises(NotImplementedError):
        engine_options({**d, "snes_linesearch_type": "l2"}, "Single phase")
    with pytest.raises(KeyError):
        engine_options({**d, "sub_1_sub_pc_factor_fill": 2.0}, "Single phase")
    with pytest.raises(NotImplementedError):
        engine_options({**d, "sub_0_cpr_stage1_pc_hypre_boomeramg_strong_threshold": 0.5}, "Single phase")
    with pytest.raises(NotImplementedError):        # PETSc's default side for gmres is LEFT
        engine_options({**d, "ksp_type": "gmres"}, "Single phase")
    # one block on a grid the oracle engine can take: whole-grid ILU(0)
    spec, u0, p, g, c = cases.c1_homogeneous(N=10)
    m = SinglePhase(g, c, p, end=1.0, maxdt=1.0, small_dt_start=False, solver_parameters={**d, "sub_1_pc_bjacobi_blocks": 1},
                    filename=None, verbosity=False, _engine_factory=OracleEngine)
    assert m.engine.opts["ilu_tile"] == (10, 10, 1)
    m.solve()
    assert m.failed_solves == 0


def test_cptramg_oracle_time_loop_and_vector_view():
    """pc_cptramg_QI (twophase.py:552-566) drives a two-phase case to the same state as pc_cptr; with vector=True the
    host view follows the reference's VectorFunctionSpace x V layout: u.dat.data = [pT (ncell x 2), S_o]."""
    sols = []
    for preset in ("pc_cptr", "pc_cptramg_QI"):
        spec, u0, p, g, c = cases.c3_spe10_2d(12, 16, 2)
        m = TwoPhase(g, c, p, end=0.004, maxdt=0.002, solver_parameters=preset, filename=None, verbosity=False,
                     _engine_factory=OracleEngine)
        m.solve()
        assert m.failed_solves == 0
        sols.append(m)
    a, b = sols
    assert b.vector and not a.vector
    pT, S = b.u.dat.data_ro
    assert pT.shape == (12*16, 2) and S.shape == (12*16,) and not pT.flags.writeable
    ua = a.u.dat.data_ro
    assert np.linalg.norm(pT[:, 0] - ua[0]) <= 1e-6*np.linalg.norm(ua[0])
    assert np.linalg.norm(pT[:, 1] - ua[1]) <= 1e-6*np.linalg.norm(ua[1])
    assert np.abs(S - ua[2]).max() < 1e-6
    b.u.dat.data[b.i_S_o][...] = 0.5            # the entry the reference's time loop writes (thermalmodel.py:229)
    assert np.all(b.u._store[2] == 0.5) and b.u.dev_stale

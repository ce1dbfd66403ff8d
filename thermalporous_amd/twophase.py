"""Two-phase (water + oil) pressure-temperature-saturation model.

Mirror of /root/reference/thermalporous/twophase.py ``TwoPhase`` (:7-1007): same constructor, field
order (0 = p / weighted pressure equation, 1 = T / energy, 2 = S_o / oil; :99-101,1007), presets
pc_cptr (:531-550), pc_cpr, pc_cpr_QI, pc_cpr_TI (:582-595) on top of newton_krylov (:416-433).
The UFL residual of init_variational_form_2D/3D (:67-411) is evaluated by csrc/tp_assembly.hip.
"""
import numpy as np

from .thermalmodel import ThermalModel


class TwoPhase(ThermalModel):
    def __init__(self, geo, case, params, end=1.0, maxdt=0.005, save=False, n_save=2, small_dt_start=True,
                 checkpointing={}, solver_parameters=None, filename="results/results.txt", dt_init_fact=2**(-10),
                 vector=False, gravity2D=False, verbosity=True, _engine_factory=None):
        self.name = "Two-phase"
        self.geo = geo
        self.case = case
        self.params = params
        self.mesh = geo.mesh
        self.comm = self.mesh.comm
        self.V = geo.V
        self.vector = vector
        self.solver_parameters = solver_parameters
        self.init_solver_parameters()
        if self.vector:
            # VectorFunctionSpace(dim=2) x V (:19-21): in the reference this interleaves (p,T) in memory so that hypre sees
            # 2x2 blocks (pc_cptramg*).  Here it is a LAYOUT FLAG: the device keeps field planes (the system AMG reads the
            # (p,T) blocks from them); only the host view follows the reference: u.dat.data = [pT (n x 2), S_o]
            self.W = ("VectorDQ0(dim=2)", "DQ0")
            self.i_S_o = 1
        else:
            self.W = ("DQ0", "DQ0", "DQ0")
            self.i_S_o = 2
        self.save = save
        self.n_save = n_save
        self.small_dt_start = small_dt_start
        self.scaled_eqns = True       # (:29)
        self.pressure_eqn = True      # (:30)
        self.geo.gravity2D = gravity2D
        self._engine_factory = _engine_factory      # test hook: inject the CPU oracle engine
        for attr in ("prod_wells", "inj_wells", "heaters"):
            if not hasattr(self.case, attr):
                setattr(self.case, attr, list())
        self.bcs = []
        ThermalModel.__init__(self, end=end, maxdt=maxdt, save=save, n_save=n_save, small_dt_start=small_dt_start,
                              checkpointing=checkpointing, filename=filename, dt_init_fact=dt_init_fact,
                              verbosity=verbosity)

    def init_IC_uniform(self):
        ic = np.zeros((3, self.geo.Nx*self.geo.Ny*self.geo.Nz))
        ic[0] = self.params.p_ref     # (:62)
        ic[1] = self.params.T_prod    # (:63)
        ic[2] = self.params.S_o       # (:64)
        return ic

    def init_solver_parameters(self):
        newton_krylov = {             # (:416-433)
            "snes_type": "newtonls",
            "snes_monitor": None,
            "snes_converged_reason": None,
            "snes_max_it": 25,
            "ksp_type": "fgmres",
            "ksp_converged_reason": None,
            "ksp_max_it": 200,
            "ksp_gmres_restart": 200,
            "ksp_rtol": 1e-8,
        }
        v_cycle = {"ksp_type": "preonly", "pc_type": "hypre", "pc_hypre_type": "boomeramg",   # (:478-482)
                   "pc_hypre_boomeramg_max_iter": 1}
        pc_cptr = {"pc_type": "composite",    # (:531-550)
                   "pc_composite_type": "multiplicative",
                   "pc_composite_pcs": "python,bjacobi",
                   "sub_0_pc_python_type": "thermalporous.preconditioners.CPTRStage1PC",
                   "sub_0_cpr_stage1_pc_type": "fieldsplit",
                   "sub_0_cpr_stage1_pc_fieldsplit_type": "schur",
                   "sub_0_cpr_stage1_pc_fieldsplit_schur_fact_type": "FULL",
                   "sub_0_cpr_stage1_fieldsplit_1_ksp_type": "preonly",
                   "sub_0_cpr_stage1_fieldsplit_1_pc_type": "python",
                   "sub_0_cpr_stage1_fieldsplit_1_pc_python_type":
                       "thermalporous.preconditioners.ConvDiffSchurTwoPhasesPC",
                   "sub_0_cpr_stage1_fieldsplit_1_schur": v_cycle,
                   "sub_0_cpr_stage1_fieldsplit_0": v_cycle,
                   "sub_1_sub_pc_type": "ilu",
                   "sub_1_sub_pc_factor_levels": 0,
                   "mat_type": "aij"}
        pc_cpr = {"pc_type": "composite",     # (:582-592)
                  "pc_composite_type": "multiplicative",
                  "pc_composite_pcs": "python,bjacobi",
                  "sub_0_pc_python_type": "thermalporous.preconditioners.CPRStage1PC",
                  "sub_0_cpr_stage1": v_cycle,
                  "sub_1_sub_pc_type": "ilu",
                  "sub_1_sub_pc_factor_levels": 0,
                  "mat_type": "aij"}
        gmres0 = {"ksp_type": "gmres", "ksp_max_it": 0, "pc_type": "none"}     # "apply nothing": returns zero
        pc_cpr_gmres = {"pc_type": "composite",       # (:619-634) pure-PETSc emulation of pc_cpr
                        "pc_composite_type": "multiplicative",
                        "pc_composite_pcs": "fieldsplit,bjacobi",
                        "sub_0_pc_fieldsplit_0_fields": "0",
                        "sub_0_pc_fieldsplit_1_fields": "1,2",
                        "sub_0_pc_fieldsplit_type": "additive",
                        "sub_0_fieldsplit_0": v_cycle,
                        "sub_0_fieldsplit_1": gmres0,
                        "sub_1_sub_pc_type": "ilu",
                        "sub_1_sub_pc_factor_levels": 0,
                        "mat_type": "aij"}
        pc_cptr_gmres = {"pc_type": "composite",      # (:670-699) pure-PETSc emulation of pc_cptr; the default (:930)
                         "pc_composite_type": "multiplicative",
                         "pc_composite_pcs": "fieldsplit,bjacobi",
                         "sub_0_pc_fieldsplit_0_fields": "0,1",
                         "sub_0_pc_fieldsplit_1_fields": "2",
                         "sub_0_pc_fieldsplit_type": "additive",
                         "sub_0_fieldsplit_0_pc_type": "fieldsplit",
                         "sub_0_fieldsplit_0_pc_fieldsplit_type": "schur",
                         "sub_0_fieldsplit_0_pc_fieldsplit_schur_fact_type": "FULL",
                         "sub_0_fieldsplit_0_fieldsplit_1_ksp_type": "preonly",
                         "sub_0_fieldsplit_0_fieldsplit_1_pc_type": "python",
                         "sub_0_fieldsplit_0_fieldsplit_1_pc_python_type":
                             "thermalporous.preconditioners.ConvDiffSchurTwoPhasesPC",
                         "sub_0_fieldsplit_0_fieldsplit_1_schur": v_cycle,
                         "sub_0_fieldsplit_0_fieldsplit_0": v_cycle,
                         "sub_0_fieldsplit_1": gmres0,
                         "sub_1_sub_pc_type": "ilu",
                         "sub_1_sub_pc_factor_levels": 0,
                         "mat_type": "aij"}
        pc_cptramg = {"pc_type": "composite",     # (:552-563) system AMG on the interleaved (p,T) operator
                      "pc_composite_type": "multiplicative",
                      "pc_composite_pcs": "python,bjacobi",
                      "sub_0_pc_python_type": "thermalporous.preconditioners.CPTRStage1PC",
                      "sub_0_cpr_stage1_pc_type": "hypre",
                      "sub_0_cpr_stage1_pc_hypre_type": "boomeramg",
                      "sub_0_cpr_stage1_pc_hypre_boomeramg_max_iter": 1,
                      "sub_1_sub_pc_type": "ilu",
                      "sub_1_sub_pc_factor_levels": 0,
                      "mat_type": "aij"}
        pc_cptr_a11 = {k: v for k, v in pc_cptr.items() if not k.startswith("sub_0_cpr_stage1_fieldsplit_1")}   # (:598-616)
        pc_cptr_a11.update({"sub_0_cpr_stage1_pc_fieldsplit_schur_precondition": "a11",
                            "sub_0_cpr_stage1_fieldsplit_1": v_cycle, "sub_1_pc_bjacobi_blocks": 1})
        presets = {"pc_cptr": pc_cptr, "pc_cptr_a11": pc_cptr_a11, "pc_cpr": pc_cpr,
                   "pc_cpr_QI": {**pc_cpr, "sub_0_cpr_decoup": "QI"},      # (:594)
                   "pc_cpr_TI": {**pc_cpr, "sub_0_cpr_decoup": "TI"},      # (:595)
                   "pc_cpr_QI_temp": {**pc_cpr, "sub_0_cpr_decoup": "QI_temp"},      # (:596)
                   "pc_cpr_TI_temp": {**pc_cpr, "sub_0_cpr_decoup": "TI_temp"},      # (:597)
                   "pc_cpr_gmres": pc_cpr_gmres, "pc_cptr_gmres": pc_cptr_gmres,
                   "pc_cprilu1_gmres": {**pc_cpr_gmres, "sub_1_sub_pc_factor_levels": 1},     # (:653-668) block-ILU(1) second stage
                   "pc_bilu": {"pc_type": "bjacobi", "sub_pc_type": "ilu", "sub_pc_factor_levels": 1, "mat_type": "aij"},   # (758-762)
                   "pc_cptramg_gmres": {k: v for k, v in pc_cpr_gmres.items()                    # (:698-713) pure-PETSc emulation
                                        if k not in ("sub_0_pc_fieldsplit_0_fields", "sub_0_pc_fieldsplit_1_fields")},
                   "pc_cptramg": pc_cptramg,                                           # (:552-563)
                   "pc_cptramg_QI": {**pc_cptramg, "sub_0_cpr_decoup": "QI"},        # (:565)
                   "pc_cptramg_TI": {**pc_cptramg, "sub_0_cpr_decoup": "TI"}}        # (:566)
        parameters = newton_krylov
        if self.solver_parameters is None:
            self.solver_parameters = "pc_cptr_gmres"       # the reference's default (:930); same algebra as pc_cptr
        if isinstance(self.solver_parameters, str):
            if self.solver_parameters not in presets:
                raise NotImplementedError("two-phase preset %r is outside the hot path; available: %s"
                                          % (self.solver_parameters, sorted(presets)))
            if self.solver_parameters.startswith("pc_cptramg"):
                self.vector = True                           # forced by the reference (:935-943)
            parameters.update(presets[self.solver_parameters])
            self.solver_parameters = parameters
        if "sub_0_cpr_decoup" in self.solver_parameters:      # (:999-1002)
            self.decoup = self.solver_parameters["sub_0_cpr_decoup"]
        else:
            self.decoup = "No"

    @property
    def appctx(self):                                          # (:1005-1007)
        return {"pressure_space": 0, "temperature_space": 1, "saturation_space": 2, "params": self.params,
                "geo": self.geo, "dt": self.dt, "case": self.case, "u_": self.u_, "decoup": self.decoup,
                "vector": self.vector}

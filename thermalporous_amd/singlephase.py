"""Single-phase (oil) pressure-temperature model.

Mirror of /root/reference/thermalporous/singlephase.py ``SinglePhase`` (:6-450): same constructor,
same solver presets for the preconditioners that are on the hot path (pc_cpr, pc_cpr_QI, pc_cpr_TI;
:341-354) and the same appctx keys (:448-450).  The UFL residual of init_variational_form_2D/3D
(:60-273) is what csrc/tp_assembly.hip evaluates; here only the problem description is built.
"""
from .thermalmodel import ThermalModel
from .preconditioners import ConvDiffSchurPC, CPRStage1PC  # noqa: F401  (same import surface as the reference)


class SinglePhase(ThermalModel):
    def __init__(self, geo, case, params, end=1.0, maxdt=0.005, save=False, n_save=2, small_dt_start=True,
                 checkpointing={}, solver_parameters=None, filename="results/results.txt", dt_init_fact=2**(-10),
                 vector=False, gravity2D=False, verbosity=True, _engine_factory=None):
        self.name = "Single phase"
        self.geo = geo
        self.case = case
        self.params = params
        self.mesh = geo.mesh
        self.comm = self.mesh.comm
        self.V = geo.V
        self.W = ("DQ0", "DQ0")
        self.save = save
        self.n_save = n_save
        self.small_dt_start = small_dt_start
        self.vector = vector
        self.solver_parameters = solver_parameters
        self.init_solver_parameters()
        self.scaled_eqns = False
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
        import numpy as np
        ic = np.zeros((2, self.geo.Nx*self.geo.Ny*self.geo.Nz))
        ic[0] = self.params.p_ref            # (:56)
        ic[1] = self.params.T_prod           # (:57)
        return ic

    def init_solver_parameters(self):
        newton = {                            # (:289-301)
            "snes_type": "newtonls",
            "snes_monitor": None,
            "snes_converged_reason": None,
            "snes_max_it": 15,
            "ksp_type": "gmres",
            "ksp_pc_side": "right",
            "ksp_converged_reason": None,
            "ksp_max_it": 200,
            "ksp_gmres_restart": 200,
        }
        v_cycle = {"ksp_type": "preonly", "pc_type": "hypre", "pc_hypre_type": "boomeramg",   # (:303-307)
                   "pc_hypre_boomeramg_max_iter": 1}
        pc_cpr = {"pc_type": "composite",     # (:341-351)
                  "pc_composite_type": "multiplicative",
                  "pc_composite_pcs": "python,bjacobi",
                  "sub_0_pc_python_type": "thermalporous.preconditioners.CPRStage1PC",
                  "sub_0_cpr_stage1": v_cycle,
                  "sub_1_sub_pc_type": "ilu",
                  "sub_1_sub_pc_factor_levels": 0,
                  "mat_type": "aij"}
        pc_fieldsplit_cd = {"pc_type": "fieldsplit",     # (:309-319) the block preconditioner of Roy et al. 2019
                            "pc_fieldsplit_type": "schur",
                            "pc_fieldsplit_schur_fact_type": "FULL",
                            "fieldsplit_0": v_cycle,
                            "fieldsplit_1_ksp_type": "preonly",
                            "fieldsplit_1_pc_type": "python",
                            "fieldsplit_1_pc_python_type": "thermalporous.preconditioners.ConvDiffSchurPC",
                            "fieldsplit_1_schur": v_cycle}
        pc_fieldsplit_a11 = {"pc_type": "fieldsplit",    # (:331-338) A_TT stands in for the Schur complement
                             "pc_fieldsplit_type": "schur",
                             "pc_fieldsplit_schur_fact_type": "FULL",
                             "pc_fieldsplit_schur_precondition": "a11",
                             "fieldsplit_0": v_cycle,
                             "fieldsplit_1": v_cycle}
        pc_fieldsplit_selfp = {**pc_fieldsplit_a11,      # (:322-330) Sp = A_TT - A_Tp diag(A_pp)^-1 A_pT
                               "pc_fieldsplit_schur_precondition": "selfp"}
        pc_fieldsplit_diag = {"pc_type": "fieldsplit", "pc_fieldsplit_type": "additive",       # (:371-375)
                              "fieldsplit_0": v_cycle, "fieldsplit_1": v_cycle}
        presets = {"pc_fieldsplit_diag": pc_fieldsplit_diag, "pc_fieldsplit_a11": pc_fieldsplit_a11, "pc_fieldsplit_selfp": pc_fieldsplit_selfp, "pc_cpr": pc_cpr,
                   "pc_cpr_QI": {**pc_cpr, "sub_0_cpr_decoup": "QI"},      # (:353)
                   "pc_cpr_TI": {**pc_cpr, "sub_0_cpr_decoup": "TI"},      # (:354)
                   "pc_bilu": {"pc_type": "bjacobi", "sub_pc_type": "ilu", "sub_pc_factor_levels": 1, "mat_type": "aij"},   # (402-406)
                   "pc_fieldsplit_cd": pc_fieldsplit_cd,
                   "pc_cpr_gmres": {"pc_type": "composite",        # (:355-368) pure-PETSc emulation of pc_cpr
                                    "pc_composite_type": "multiplicative",
                                    "pc_composite_pcs": "fieldsplit,bjacobi",
                                    "sub_0_pc_fieldsplit_type": "additive",
                                    "sub_0_fieldsplit_0": v_cycle,
                                    "sub_0_fieldsplit_1": {"ksp_type": "gmres", "ksp_max_it": 0, "pc_type": "none"},
                                    "sub_1_sub_pc_type": "ilu",
                                    "sub_1_sub_pc_factor_levels": 0,
                                    "mat_type": "aij"}}
        parameters = newton
        if self.solver_parameters is None:
            # the reference's default name "pc_fieldsplit" matches no branch (:410-439) and silently runs
            # bare GMRES; the hot-path build defaults to its CPR preset instead.
            self.solver_parameters = "pc_cpr"
        if isinstance(self.solver_parameters, str):
            if self.solver_parameters not in presets:
                raise NotImplementedError("single-phase preset %r is outside the hot path; available: %s"
                                          % (self.solver_parameters, sorted(presets)))
            parameters.update(presets[self.solver_parameters])
            self.solver_parameters = parameters
        if "sub_0_cpr_decoup" in self.solver_parameters:      # (:441-444)
            self.decoup = self.solver_parameters["sub_0_cpr_decoup"]
        else:
            self.decoup = "No"

    @property
    def appctx(self):                                          # (:448-450)
        return {"pressure_space": 0, "temperature_space": 1, "params": self.params, "geo": self.geo,
                "dt": self.dt, "u_": self.u_, "case": self.case, "decoup": self.decoup}

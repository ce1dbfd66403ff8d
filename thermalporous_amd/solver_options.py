"""PETSc-style ``solver_parameters`` of the reference -> options of the compute engine.

The reference configures its whole linear/nonlinear stack through PETSc option dicts
(singlephase.py:289-354,410-439; twophase.py:416-433,531-597,929-997).  The hot path honours the
subset that selects the solvers it implements and rejects everything else loudly:

  snes_type newtonls  (line search `basic`: Firedrake's default, not settable in the reference)
  ksp_type fgmres|gmres (right preconditioning), ksp_rtol/atol/max_it, ksp_gmres_restart
  pc_type composite, pc_composite_type multiplicative, pc_composite_pcs "python,bjacobi"
     sub_0_pc_python_type  ...CPRStage1PC | ...CPTRStage1PC   sub_0_cpr_decoup  No|QI|TI
     sub_0_cpr_stage1*     boomeramg V-cycle / fieldsplit-schur-FULL with ConvDiffSchurTwoPhasesPC
     sub_1_sub_pc_type ilu, sub_1_sub_pc_factor_levels 0, sub_1_pc_bjacobi_blocks

Defaults the reference inherits silently from Firedrake/PETSc are fixed here explicitly
(SURVEY.md 8c): ksp_rtol 1e-7 (Firedrake), snes_rtol 1e-8, snes_atol 1e-50, snes_stol 1e-8,
ksp_atol 1e-50.  Build-specific tuning keys (not PETSc): amg_omega, amg_nu, amg_min_cells,
ilu_tile.  Two-phase string presets are layered on plain Newton-Krylov, not on the reference's
experimental FAS nonlinear preconditioner (twophase.py:927; needs mesh hierarchies + MUMPS).
"""

_IGNORED = {"snes_monitor", "snes_converged_reason", "ksp_converged_reason", "ksp_view", "snes_view", "ksp_monitor",
            "mat_type", "ksp_pc_side", "snes_linesearch_type", "ksp_monitor_residuals"}

_V_CYCLE = {"ksp_type": "preonly", "pc_type": "hypre", "pc_hypre_type": "boomeramg",
            "pc_hypre_boomeramg_max_iter": 1}


def _flatten(d, prefix=""):
    """PETSc semantics: a nested dict is a prefix (``"sub_0_cpr_stage1": v_cycle``)."""
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "_"))
        else:
            out[prefix + k] = v
    return out


def engine_options(solver_parameters, model_name, decoup="No"):
    from .engine import DEFAULT_OPTS
    sp = _flatten(dict(solver_parameters))
    o = dict(DEFAULT_OPTS)
    o["decoup"] = decoup
    o["schur_a11"] = False
    build_keys = ("amg_omega", "amg_nu", "amg_min_cells", "amg_full_levels", "amg_coarse_pre", "amg_coarse_post", "amg_mid_skip", "amg_tail_post", "amg_single",
                  "amg_gather_cells", "ilu_tile")
    for k in build_keys:
        if k in sp:
            o[k] = sp.pop(k)
    if sp.get("snes_type", "newtonls") != "newtonls":
        raise NotImplementedError("snes_type %r: only newtonls is on the hot path" % sp["snes_type"])
    if sp.get("ksp_type", "gmres") not in ("fgmres", "gmres"):
        raise NotImplementedError("ksp_type %r: only (f)gmres with right preconditioning" % sp["ksp_type"])
    if sp.get("ksp_type", "gmres") == "gmres" and sp.get("ksp_pc_side", "right") != "right":
        raise NotImplementedError("left-preconditioned GMRES is not implemented")
    for k_src, k_dst in (("ksp_rtol", "ksp_rtol"), ("ksp_atol", "ksp_atol"), ("ksp_max_it", "ksp_max_it"),
                         ("ksp_gmres_restart", "ksp_restart"), ("snes_max_it", "snes_max_it"),
                         ("snes_rtol", "snes_rtol"), ("snes_atol", "snes_atol"), ("snes_stol", "snes_stol")):
        if k_src in sp:
            o[k_dst] = sp[k_src]
    if sp.get("pc_type") == "fieldsplit":
        # pc_fieldsplit_cd (singlephase.py:309-319): Schur FULL on (p,T), V-cycle on A_pp, ConvDiffSchurPC on S
        a11 = sp.get("pc_fieldsplit_schur_precondition") == "a11" and sp.get("fieldsplit_1_pc_type") == "hypre"
        if sp.get("pc_fieldsplit_type") != "schur" or str(sp.get("pc_fieldsplit_schur_fact_type", "")).upper() != "FULL" \
                or not (a11 or (str(sp.get("fieldsplit_1_pc_python_type", "")).endswith("ConvDiffSchurPC")
                                and "pc_fieldsplit_schur_precondition" not in sp)):
            raise NotImplementedError("fieldsplit preconditioners on the hot path: pc_fieldsplit_cd (schur FULL with "
                                      "ConvDiffSchurPC) and pc_fieldsplit_a11; selfp / additive variants are not")
        o["schur_a11"] = bool(a11)
        if model_name == "Two-phase":
            raise NotImplementedError("pc_fieldsplit_cd is the single-phase block preconditioner")
        if o["decoup"] != "No":
            raise NotImplementedError("pc_fieldsplit_cd has no decoupling stage")
        o["pc"] = "fieldsplit_cd"
        for k in sp:
            if not (k in _IGNORED or k.startswith(("fieldsplit_", "pc_", "ksp_", "snes_"))):
                raise KeyError("unknown solver parameter %r" % k)
        return o
    if sp.get("pc_type") == "composite" and sp.get("pc_composite_pcs") == "fieldsplit,bjacobi":
        # the reference's pure-PETSc emulations of its python stage-1 classes (singlephase.py:355-368,
        # twophase.py:619-634,670-699): additive fieldsplit whose second split is "gmres, max_it 0, pc none" -- i.e.
        # returns zero, exactly the y_nonp = 0 of CPRStage1PC/CPTRStage1PC.apply -- with decoupling "No".
        # pc_cpr_gmres == pc_cpr and pc_cptr_gmres == pc_cptr (the two-phase default, twophase.py:930) as algebra.
        if sp.get("pc_composite_type", "multiplicative") != "multiplicative" or sp.get("sub_0_pc_fieldsplit_type") != "additive" \
                or sp.get("sub_0_fieldsplit_1_ksp_type") != "gmres" or int(sp.get("sub_0_fieldsplit_1_ksp_max_it", -1)) != 0 \
                or sp.get("sub_0_fieldsplit_1_pc_type") != "none":
            raise NotImplementedError("fieldsplit,bjacobi composite: only the *_gmres emulations of pc_cpr / pc_cptr")
        if sp.get("sub_1_sub_pc_type", "ilu") != "ilu" or int(sp.get("sub_1_sub_pc_factor_levels", 0)) != 0:
            raise NotImplementedError("stage 2 must be ILU(0) (pc_cprilu1_gmres is not on the hot path)")
        if o["decoup"] != "No":
            raise NotImplementedError("the fieldsplit emulations have no decoupling stage")
        if sp.get("sub_0_fieldsplit_0_pc_type") == "hypre":
            if sp.get("sub_0_pc_fieldsplit_0_fields", "0") != "0":
                raise NotImplementedError("system AMG on several fields (pc_cptramg_gmres) is not on the hot path")
            o["pc"] = "cpr"
        elif sp.get("sub_0_fieldsplit_0_pc_type") == "fieldsplit" \
                and sp.get("sub_0_fieldsplit_0_pc_fieldsplit_type") == "schur" \
                and str(sp.get("sub_0_fieldsplit_0_pc_fieldsplit_schur_fact_type", "")).upper() == "FULL" \
                and str(sp.get("sub_0_fieldsplit_0_fieldsplit_1_pc_python_type", "")).endswith("ConvDiffSchurTwoPhasesPC"):
            if model_name != "Two-phase":
                raise NotImplementedError("pc_cptr_gmres needs the two-phase model")
            o["pc"] = "cptr"
        else:
            raise NotImplementedError("unsupported first split of the fieldsplit,bjacobi composite "
                                      "(mg/LU/system-AMG variants are not on the hot path)")
        for k in sp:
            if not (k in _IGNORED or k.startswith(("sub_0_", "sub_1_", "pc_", "ksp_", "snes_"))):
                raise KeyError("unknown solver parameter %r" % k)
        return o
    if sp.get("pc_type") != "composite" or sp.get("pc_composite_type", "multiplicative") != "multiplicative" \
            or sp.get("pc_composite_pcs") != "python,bjacobi":
        raise NotImplementedError(
            "only the composite multiplicative 'python,bjacobi' preconditioners (pc_cpr*, pc_cptr) are on the hot "
            "path; got pc_type=%r pc_composite_pcs=%r" % (sp.get("pc_type"), sp.get("pc_composite_pcs")))
    pytype = str(sp.get("sub_0_pc_python_type", ""))
    if pytype.endswith("CPRStage1PC"):
        o["pc"] = "cpr"
    elif pytype.endswith("CPTRStage1PC"):
        if model_name != "Two-phase":
            raise NotImplementedError("CPTRStage1PC needs the two-phase model")
        o["pc"] = "cptr"
        if sp.get("sub_0_cpr_stage1_pc_type") != "fieldsplit":
            raise NotImplementedError("CPTRStage1PC is implemented with the fieldsplit-Schur stage-1 solver of "
                                      "pc_cptr; system-AMG/LU variants (pc_cptramg*, pc_cptrlu*) are not")
        pre = sp.get("sub_0_cpr_stage1_pc_fieldsplit_schur_precondition")
        if pre not in (None, "a11"):
            raise NotImplementedError("Schur preconditioning %r (only the ConvDiffSchurTwoPhasesPC operator and a11)" % pre)
        o["schur_a11"] = pre == "a11"                   # pc_cptr_a11 (twophase.py:598-616)
    else:
        raise NotImplementedError("sub_0_pc_python_type %r" % pytype)
    if o["decoup"] not in ("No", "QI", "TI", "QI_temp", "TI_temp"):
        raise NotImplementedError("unknown decoupling %r" % o["decoup"])
    if o["decoup"].endswith("_temp") and (o["pc"] != "cpr" or model_name != "Two-phase"):
        raise NotImplementedError("QI_temp/TI_temp decouple temperature AND saturation from the pressure: "
                                  "two-phase pc_cpr only (preconditioners.py:367-368)")
    if sp.get("sub_1_sub_pc_type", "ilu") != "ilu" or int(sp.get("sub_1_sub_pc_factor_levels", 0)) != 0:
        raise NotImplementedError("stage 2 must be ILU(0)")
    known_prefixes = ("sub_0_", "sub_1_", "pc_", "ksp_", "snes_")
    for k in sp:
        if k in _IGNORED or k.startswith(known_prefixes):
            continue
        raise KeyError("unknown solver parameter %r" % k)
    return o

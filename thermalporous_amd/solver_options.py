"""PETSc-style ``solver_parameters`` of the reference -> options of the compute engine.

The reference configures its whole linear/nonlinear stack through PETSc option dicts
(singlephase.py:289-354,410-439; twophase.py:416-433,531-597,929-997).  The hot path honours the
subset that selects the solvers it implements and rejects everything else loudly -- every key is either
CONSUMED (and its value checked against what is implemented), purely cosmetic (monitors/views), or an error:

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

# keys that only print / name the matrix type: no effect on the arithmetic
_IGNORED = {"snes_monitor", "snes_converged_reason", "ksp_converged_reason", "ksp_view", "snes_view", "ksp_monitor",
            "ksp_monitor_residuals"}      # (ksp_monitor_residuals is honoured by ThermalModel.init_solver, like the reference)

# every PETSc key the hot path CONSUMES (checked against the value it implements) -- anything else raises
_VCYCLE_SUFFIXES = {"ksp_type": "preonly", "pc_type": "hypre", "pc_hypre_type": "boomeramg",
                    "pc_hypre_boomeramg_max_iter": 1}


def _take_vcycle(sp, prefix, used):
    """``<prefix>`` must configure exactly the reference's one-V-cycle solver (v_cycle dicts, singlephase.py:303-307,
    twophase.py:478-482); hypre tuning keys are not honoured and therefore rejected."""
    for suf, want in _VCYCLE_SUFFIXES.items():
        k = prefix + suf
        if k not in sp:
            if suf == "pc_hypre_boomeramg_max_iter":      # PETSc's default is already 1
                continue
            raise NotImplementedError("%s missing: the stage-1 solver must be one BoomerAMG V-cycle (v_cycle)" % k)
        if sp[k] != want:
            raise NotImplementedError("%s = %r: only %r (one V-cycle per application)" % (k, sp[k], want))
        used.add(k)
    for k in sp:
        if k.startswith(prefix + "pc_hypre_") and k not in used:
            raise NotImplementedError("%s: hypre tuning options do not apply to this build's own AMG" % k)


def _take(sp, used, key, allowed=None, default=None):
    """Consume `key`; `allowed` = the values the hot path implements."""
    if key not in sp:
        return default
    v = sp[key]
    if allowed is not None and v not in allowed:
        raise NotImplementedError("%s = %r is not implemented on the hot path (supported: %s)"
                                  % (key, v, ", ".join(repr(a) for a in allowed)))
    used.add(key)
    return v


def _reject_unused(sp, used):
    for k in sp:
        if k in used or k in _IGNORED:
            continue
        raise KeyError("solver parameter %r is not consumed by the hot path (it would be silently ignored)" % k)

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


def engine_options(solver_parameters, model_name, decoup="No", vector=False):
    from .engine import DEFAULT_OPTS
    sp = _flatten(dict(solver_parameters))
    o = dict(DEFAULT_OPTS)
    o["decoup"] = decoup
    o["schur_a11"] = False
    o["schur_selfp"] = False
    o["fs_additive"] = False
    used = set()
    build_keys = ("amg_omega", "amg_nu", "amg_min_cells", "amg_full_levels", "amg_coarse_pre", "amg_coarse_post", "amg_mid_skip", "amg_tail_post", "amg_single",
                  "amg_gather_cells", "amg_dom_tau", "ilu_tile", "ilu_levels", "ilu_whole")
    for k in build_keys:
        if k in sp:
            o[k] = sp.pop(k)
    # ---- Newton / Krylov (singlephase.py:289-301, twophase.py:416-433) --------------------------------------
    _take(sp, used, "snes_type", ("newtonls",))
    # Firedrake's default line search is `basic`; the reference never sets another one on its Newton-Krylov path
    # (l2 only inside the FAS presets, twophase.py:437): anything but basic would silently change the algorithm
    _take(sp, used, "snes_linesearch_type", ("basic",))
    _take(sp, used, "mat_type", ("aij",))
    ksp = _take(sp, used, "ksp_type", ("fgmres", "gmres"), "gmres")
    side = _take(sp, used, "ksp_pc_side", ("right",))
    if ksp == "gmres" and side is None:
        raise NotImplementedError("ksp_type gmres without ksp_pc_side: PETSc would precondition from the LEFT; only "
                                  "right-preconditioned (F)GMRES is implemented (the reference sets ksp_pc_side right, "
                                  "singlephase.py:296)")
    for k_src, k_dst in (("ksp_rtol", "ksp_rtol"), ("ksp_atol", "ksp_atol"), ("ksp_max_it", "ksp_max_it"),
                         ("ksp_gmres_restart", "ksp_restart"), ("snes_max_it", "snes_max_it"),
                         ("snes_rtol", "snes_rtol"), ("snes_atol", "snes_atol"), ("snes_stol", "snes_stol")):
        if k_src in sp:
            o[k_dst] = sp[k_src]
            used.add(k_src)
    pc_type = _take(sp, used, "pc_type", ("fieldsplit", "composite", "bjacobi"))
    if pc_type == "bjacobi":
        # pc_bilu (twophase.py:758-762, singlephase.py:402-406): bjacobi + ILU(levels) is the whole preconditioner
        _take(sp, used, "sub_pc_type", ("ilu",))
        if "sub_pc_type" not in used:
            raise NotImplementedError("pc_type bjacobi: sub_pc_type ilu only (pc_bilu)")
        levels = int(_take(sp, used, "sub_pc_factor_levels", None, 0))
        if levels not in (0, 1):
            raise NotImplementedError("bjacobi sub-solver is block-ILU(0) or block-ILU(1) (sub_pc_factor_levels %d)" % levels)
        o["ilu_levels"] = levels
        nb = _take(sp, used, "pc_bjacobi_blocks")
        if nb is not None:
            o["bjacobi_blocks"] = int(nb)
        _take(sp, used, "mat_type", ("aij",))
        if o["decoup"] != "No":
            raise NotImplementedError("pc_bilu has no decoupling stage")
        o["pc"] = "bilu"
        _reject_unused(sp, used)
        return o
    if pc_type is None:
        raise NotImplementedError("pc_type missing: only the composite CPR/CPTR preconditioners and pc_fieldsplit_cd/_a11 "
                                  "are on the hot path")
    if pc_type == "fieldsplit":
        # pc_fieldsplit_cd (singlephase.py:309-319): Schur FULL on (p,T), V-cycle on A_pp, ConvDiffSchurPC on S;
        # pc_fieldsplit_a11 (:331-338): A_TT stands in for the Schur complement;
        # pc_fieldsplit_selfp (:322-330): Sp = A_TT - A_Tp diag(A_pp)^-1 A_pT
        if _take(sp, used, "pc_fieldsplit_type", ("schur", "additive")) == "additive":
            # pc_fieldsplit_diag (singlephase.py:371-375): block-diagonal, one V-cycle on A_pp and one on A_TT
            _take_vcycle(sp, "fieldsplit_0_", used)
            _take_vcycle(sp, "fieldsplit_1_", used)
            if model_name == "Two-phase" or o["decoup"] != "No":
                raise NotImplementedError("pc_fieldsplit_diag is a single-phase preconditioner without decoupling")
            o["pc"], o["schur_a11"], o["fs_additive"] = "fieldsplit_cd", True, True
            _reject_unused(sp, used)
            return o
        fact = str(_take(sp, used, "pc_fieldsplit_schur_fact_type", None, "")).upper()
        pre = _take(sp, used, "pc_fieldsplit_schur_precondition", ("a11", "selfp"))
        if "pc_fieldsplit_type" not in used or fact != "FULL":
            raise NotImplementedError("fieldsplit preconditioners on the hot path: schur FULL with ConvDiffSchurPC "
                                      "(pc_fieldsplit_cd), a11 (pc_fieldsplit_a11) or selfp (pc_fieldsplit_selfp)")
        _take_vcycle(sp, "fieldsplit_0_", used)
        if pre in ("a11", "selfp"):
            _take_vcycle(sp, "fieldsplit_1_", used)
        else:
            _take(sp, used, "fieldsplit_1_ksp_type", ("preonly",))
            _take(sp, used, "fieldsplit_1_pc_type", ("python",))
            if not str(_take(sp, used, "fieldsplit_1_pc_python_type", None, "")).endswith("ConvDiffSchurPC"):
                raise NotImplementedError("fieldsplit_1 must be ConvDiffSchurPC (pc_fieldsplit_cd) or a V-cycle "
                                          "(pc_fieldsplit_a11, pc_fieldsplit_selfp)")
            _take_vcycle(sp, "fieldsplit_1_schur_", used)
        o["schur_a11"] = pre == "a11"
        o["schur_selfp"] = pre == "selfp"                # (singlephase.py:322-330)
        if model_name == "Two-phase":
            raise NotImplementedError("pc_fieldsplit_cd is the single-phase block preconditioner")
        if o["decoup"] != "No":
            raise NotImplementedError("pc_fieldsplit_cd has no decoupling stage")
        o["pc"] = "fieldsplit_cd"
        _reject_unused(sp, used)
        return o
    # ---- composite multiplicative (stage 1, bjacobi/ILU(0)) -----------------------------------------------------
    _take(sp, used, "pc_composite_type", ("multiplicative",))
    pcs = _take(sp, used, "pc_composite_pcs", ("python,bjacobi", "fieldsplit,bjacobi"))
    if pcs is None:
        raise NotImplementedError("pc_composite_pcs missing")
    # stage 2: bjacobi + ILU(0) (singlephase.py:348-349); block count: see engine.tiles_for_blocks
    _take(sp, used, "sub_1_sub_pc_type", ("ilu",))
    levels = int(_take(sp, used, "sub_1_sub_pc_factor_levels", None, 0))
    if levels not in (0, 1):
        raise NotImplementedError("stage 2 is block-ILU(0) or block-ILU(1) (sub_1_sub_pc_factor_levels %d)" % levels)
    o["ilu_levels"] = levels
    nb = _take(sp, used, "sub_1_pc_bjacobi_blocks")
    if nb is not None:
        o["bjacobi_blocks"] = int(nb)
    _take(sp, used, "sub_0_cpr_decoup", ("No", "QI", "TI", "QI_temp", "TI_temp"))    # read by the model class (:441-444)
    if pcs == "fieldsplit,bjacobi":
        # the reference's pure-PETSc emulations of its python stage-1 classes (singlephase.py:355-368,
        # twophase.py:619-634,670-699): additive fieldsplit whose second split is "gmres, max_it 0, pc none" -- i.e.
        # returns zero, exactly the y_nonp = 0 of CPRStage1PC/CPTRStage1PC.apply -- with decoupling "No".
        # pc_cpr_gmres == pc_cpr and pc_cptr_gmres == pc_cptr (the two-phase default, twophase.py:930) as algebra.
        _take(sp, used, "sub_0_pc_fieldsplit_type", ("additive",))
        _take(sp, used, "sub_0_fieldsplit_1_ksp_type", ("gmres",))
        _take(sp, used, "sub_0_fieldsplit_1_pc_type", ("none",))
        if "sub_0_pc_fieldsplit_type" not in used or "sub_0_fieldsplit_1_ksp_type" not in used \
                or int(_take(sp, used, "sub_0_fieldsplit_1_ksp_max_it", None, -1)) != 0 \
                or "sub_0_fieldsplit_1_pc_type" not in used:
            raise NotImplementedError("fieldsplit,bjacobi composite: only the *_gmres emulations of pc_cpr / pc_cptr")
        if o["decoup"] != "No":
            raise NotImplementedError("the fieldsplit emulations have no decoupling stage")
        f0 = _take(sp, used, "sub_0_pc_fieldsplit_0_fields", None, "0")
        f1 = _take(sp, used, "sub_0_pc_fieldsplit_1_fields")
        first = _take(sp, used, "sub_0_fieldsplit_0_pc_type", ("hypre", "fieldsplit"))
        if first == "hypre":
            if f0 != "0":
                raise NotImplementedError("one AMG V-cycle on an explicit multi-field split is not on the hot path")
            _take_vcycle(sp, "sub_0_fieldsplit_0_", used)
            if model_name == "Two-phase" and vector and "sub_0_pc_fieldsplit_0_fields" not in used:
                # pc_cptramg_gmres (twophase.py:698-713): no explicit fields, so with vector=True (forced at :953-955) the
                # splits are the function space's own sub-spaces: (p,T) interleaved | S_o -- ONE V-cycle on the (p,T) system
                o["pc"] = "cptramg"
            else:
                o["pc"] = "cpr"
        elif first == "fieldsplit":
            _take(sp, used, "sub_0_fieldsplit_0_pc_fieldsplit_type", ("schur",))
            fact = str(_take(sp, used, "sub_0_fieldsplit_0_pc_fieldsplit_schur_fact_type", None, "")).upper()
            _take(sp, used, "sub_0_fieldsplit_0_fieldsplit_1_ksp_type", ("preonly",))
            _take(sp, used, "sub_0_fieldsplit_0_fieldsplit_1_pc_type", ("python",))
            py = str(_take(sp, used, "sub_0_fieldsplit_0_fieldsplit_1_pc_python_type", None, ""))
            if "sub_0_fieldsplit_0_pc_fieldsplit_type" not in used or fact != "FULL" or not py.endswith("ConvDiffSchurTwoPhasesPC"):
                raise NotImplementedError("unsupported first split of the fieldsplit,bjacobi composite")
            if f0 not in ("0,1", "0, 1") or f1 not in (None, "2"):
                raise NotImplementedError("pc_cptr_gmres splits fields (0,1 | 2)")
            _take_vcycle(sp, "sub_0_fieldsplit_0_fieldsplit_0_", used)
            _take_vcycle(sp, "sub_0_fieldsplit_0_fieldsplit_1_schur_", used)
            if model_name != "Two-phase":
                raise NotImplementedError("pc_cptr_gmres needs the two-phase model")
            o["pc"] = "cptr"
        else:
            raise NotImplementedError("unsupported first split of the fieldsplit,bjacobi composite "
                                      "(mg/LU/system-AMG variants are not on the hot path)")
        _reject_unused(sp, used)
        return o
    pytype = str(_take(sp, used, "sub_0_pc_python_type", None, ""))
    if pytype.endswith("CPRStage1PC"):
        o["pc"] = "cpr"
        _take_vcycle(sp, "sub_0_cpr_stage1_", used)
    elif pytype.endswith("CPTRStage1PC"):
        if model_name != "Two-phase":
            raise NotImplementedError("CPTRStage1PC needs the two-phase model")
        kind = _take(sp, used, "sub_0_cpr_stage1_pc_type", ("fieldsplit", "hypre"))
        if kind is None:
            raise NotImplementedError("CPTRStage1PC needs sub_0_cpr_stage1_pc_type fieldsplit (pc_cptr) or hypre "
                                      "(pc_cptramg*); the LU variants (pc_cptrlu*) are not on the hot path")
        if kind == "hypre":
            # pc_cptramg[_QI|_TI] (twophase.py:552-566): ONE BoomerAMG V-cycle on the interleaved (p,T) system
            o["pc"] = "cptramg"
            _take(sp, used, "sub_0_cpr_stage1_pc_hypre_type", ("boomeramg",))
            _take(sp, used, "sub_0_cpr_stage1_pc_hypre_boomeramg_max_iter", (1,))
            _take(sp, used, "sub_0_cpr_stage1_ksp_type", ("preonly",))
            if "sub_0_cpr_stage1_pc_hypre_type" not in used:
                raise NotImplementedError("sub_0_cpr_stage1_pc_hypre_type boomeramg expected")
            for k in sp:
                if k.startswith("sub_0_cpr_stage1_pc_hypre_") and k not in used:
                    raise NotImplementedError("%s: hypre tuning options do not apply to this build's own AMG" % k)
            if o["decoup"] not in ("No", "QI", "TI"):
                raise NotImplementedError("pc_cptramg: decoupling No, QI or TI (twophase.py:552-566)")
        else:
            o["pc"] = "cptr"
            _take(sp, used, "sub_0_cpr_stage1_pc_fieldsplit_type", ("schur",))
            fact = str(_take(sp, used, "sub_0_cpr_stage1_pc_fieldsplit_schur_fact_type", None, "")).upper()
            if "sub_0_cpr_stage1_pc_fieldsplit_type" not in used or fact != "FULL":
                raise NotImplementedError("CPTR stage 1: fieldsplit schur FULL only (twophase.py:536-538)")
            pre = _take(sp, used, "sub_0_cpr_stage1_pc_fieldsplit_schur_precondition", ("a11",))
            o["schur_a11"] = pre == "a11"                   # pc_cptr_a11 (twophase.py:598-616)
            _take_vcycle(sp, "sub_0_cpr_stage1_fieldsplit_0_", used)
            if pre == "a11":
                _take_vcycle(sp, "sub_0_cpr_stage1_fieldsplit_1_", used)
            else:
                _take(sp, used, "sub_0_cpr_stage1_fieldsplit_1_ksp_type", ("preonly",))
                _take(sp, used, "sub_0_cpr_stage1_fieldsplit_1_pc_type", ("python",))
                py = str(_take(sp, used, "sub_0_cpr_stage1_fieldsplit_1_pc_python_type", None, ""))
                if not py.endswith("ConvDiffSchurTwoPhasesPC"):
                    raise NotImplementedError("CPTR stage 1: the Schur split must be ConvDiffSchurTwoPhasesPC or a11")
                _take_vcycle(sp, "sub_0_cpr_stage1_fieldsplit_1_schur_", used)
    else:
        raise NotImplementedError("sub_0_pc_python_type %r" % pytype)
    if o["decoup"] not in ("No", "QI", "TI", "QI_temp", "TI_temp"):
        raise NotImplementedError("unknown decoupling %r" % o["decoup"])
    if o["decoup"].endswith("_temp") and (o["pc"] != "cpr" or model_name != "Two-phase"):
        raise NotImplementedError("QI_temp/TI_temp decouple temperature AND saturation from the pressure: "
                                  "two-phase pc_cpr only (preconditioners.py:367-368)")
    _reject_unused(sp, used)
    return o

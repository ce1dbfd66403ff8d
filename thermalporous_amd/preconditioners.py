"""Preconditioner plug-in classes: the PCBase surface of the reference on top of the C ABI.

Mirrors /root/reference/thermalporous/preconditioners.py:
    CPRStage1PC              :335-906    initialize :339, update :875-878, apply :881-903
    CPTRStage1PC             :1243-1571  initialize :1247, update :1545-1548, apply :1550-1567
    ConvDiffSchurPC          :11-163     (single-phase S~; apply :152-155)
    ConvDiffSchurTwoPhasesPC :165-333    (two-phase S~; update :318-319, apply :321-324)
In the reference these are instantiated by PETSc's PCPYTHON from the option
``-<prefix>pc_python_type`` and receive a petsc4py ``PC`` plus PETSc ``Vec``s.  Here ``pc`` is the light
``PC`` handle below (options prefix + appctx + compute engine) and ``x``/``y`` are names of device
vectors of the engine; the arithmetic of every method is a HIP kernel sequence behind one C-ABI call:

    initialize/update -> tp_stage1_update   (sub-block extraction is zero-copy on the stencil-of-blocks
                                             Jacobian, Quasi-/True-IMPES row operation, AMG set-up --
                                             no extra UFL assemblies, no SpGEMM)
    apply             -> tp_stage1_apply    r_p = x_p - D_ps D_ss^-1 x_s ; y_p = V-cycle(r_p) ; y_s = 0
    ConvDiffSchur*.apply -> tp_schur_apply  one V-cycle on S~

The production path does not go through Python per Krylov iteration (tp_newton_solve runs the whole
Newton/FGMRES loop natively); these classes exist so that code written against the reference's PC
API -- and the tests -- can drive the same stages one call at a time.  ``CompositePC`` reproduces
PCCOMPOSITE multiplicative("python,bjacobi") from the pieces.
"""


import numpy as np

class PC():
    """Minimal stand-in for the petsc4py PC handed to PCBase methods."""

    def __init__(self, engine, appctx, prefix=""):
        self.engine = engine
        self.appctx = appctx
        self._prefix = prefix

    def getOptionsPrefix(self):
        return self._prefix


class PCBase():
    """Firedrake's PCBase protocol: setUp -> initialize (first call) / update (later calls)."""

    def __init__(self):
        self.initialized = False

    def get_appctx(self, pc):
        return pc.appctx

    def setUp(self, pc):
        if self.initialized:
            self.update(pc)
        else:
            self.initialize(pc)
            self.initialized = True

    def view(self, pc, viewer=None):
        print("%s (decoupling %s) on the HIP engine" % (type(self).__name__, getattr(self, "decoup", "-")))


class CPRStage1PC(PCBase):
    '''
    1st stage solver for constrained pressure residual
    '''
    kind = "cpr"

    def initialize(self, pc):
        appctx = self.get_appctx(pc)
        self.decoup = appctx["decoup"]                 # (:370)
        if self.decoup not in ("No", "QI", "TI", "QI_temp", "TI_temp"):
            raise NotImplementedError("unknown decoupling %r" % self.decoup)
        if self.decoup.endswith("_temp") and (self.kind != "cpr" or pc.engine.b != 3):
            raise NotImplementedError("QI_temp/TI_temp: two-phase CPRStage1PC only (:367-368)")
        eng = pc.engine
        if eng.opts["pc"] != self.kind or eng.opts["decoup"] != self.decoup:
            eng.set_options(pc=self.kind, decoup=self.decoup)
        self.update(pc)

    def update(self, pc):
        # assemble_blocks + create_decoup + pc_schur.setOperators(Atildepp)   (:875-878)
        pc.engine.pc_setup()

    def apply(self, pc, x, y):
        # x_p - Aps*inv(Dss)*x_s -> V-cycle -> y_p ; y_nonp = 0                (:881-903)
        pc.engine.stage1_apply(x, y)

    # should not be used
    applyTranspose = apply


class CPTRStage1PC(CPRStage1PC):
    '''
    1st stage solver for constrained pressure-temperature residual
    '''
    kind = "cptr"

    def initialize(self, pc):
        appctx = self.get_appctx(pc)
        # the stage-1 solver is read from the options prefix <prefix>cpr_stage1_ (:1374): fieldsplit-Schur (pc_cptr) or
        # one system-AMG V-cycle on the interleaved (p,T) operator (pc_cptramg*, which force vector=True :935-955)
        if pc.engine.opts["pc"] == "cptramg" or (appctx.get("vector") and pc.engine.opts["pc"] != "cptr"):
            self.kind = "cptramg"
        CPRStage1PC.initialize(self, pc)


class ConvDiffSchurTwoPhasesPC(PCBase):
    """Schur-complement approximation S~ = temperature convection-diffusion operator frozen at the
    current Newton state (:165-333); assembled by the fused assembly kernel as a by-product."""

    needs = "cptr"

    def initialize(self, pc):
        if pc.engine.opts["pc"] != self.needs:
            raise NotImplementedError("this S~ is assembled for pc=%r only" % self.needs)
        self.update(pc)

    def update(self, pc):
        pc.engine.pc_setup()          # re-assembles nothing: S~ comes with the Jacobian (tp_jacobian)

    def apply(self, pc, X, Y):
        pc.engine.schur_apply(X, Y)   # one V-cycle on S~, temperature field

    applyTranspose = apply


class ConvDiffSchurPC(ConvDiffSchurTwoPhasesPC):
    """Single-phase variant (:11-163): S~ = accumulation + upwinded advection of c_v rho/mu T + conduction
    - producer / heater source derivatives, frozen at the Newton state.  The Schur block of the
    reference's pc_fieldsplit_cd preset (singlephase.py:309-319); one V-cycle per application."""
    needs = "fieldsplit_cd"


class FieldsplitSchurPC():
    """PCFIELDSPLIT schur FULL on (p,T) with K(A00) = V-cycle and K(S) = the given Schur PC
    (singlephase.py:309-319; the same three solves as pc_cptr's stage 1, twophase.py:536-545):
        y0 = K(A00) x0 ;  y1 = K(S)(x1 - A10 y0) ;  y0 = K(A00)(x0 - A01 y1).
    Assembled from the stage objects so tests can show that it reproduces tp_pc_apply."""

    def __init__(self, schur_pc):
        self.schur = schur_pc

    def setUp(self, pc):
        self.schur.setUp(pc)

    def apply(self, pc, x, y):
        eng = pc.engine
        xh = eng.vec_get(x)
        eng.amg_vcycle(0, x, 0, "_fs_w", 0)
        y0 = eng.vec_get("_fs_w")[0]
        # couplings through MatMult on a vector with one live field (decoupling "No": A10, A01 are blocks of J)
        z = np.zeros_like(xh)
        z[0] = y0
        eng.vec_set("_fs_w", z)
        eng.spmv("_fs_w", "_fs_t")
        t = xh.copy()
        t[1] = xh[1] - eng.vec_get("_fs_t")[1]                     # x1 - A10 y0
        eng.vec_set("_fs_t", t)
        self.schur.apply(pc, "_fs_t", "_fs_w")                     # y1 = K(S~)(...)
        y1 = eng.vec_get("_fs_w")[1]
        z[:] = 0.0
        z[1] = y1
        eng.vec_set("_fs_w", z)
        eng.spmv("_fs_w", "_fs_t")
        t = xh.copy()
        t[0] = xh[0] - eng.vec_get("_fs_t")[0]                     # x0 - A01 y1
        eng.vec_set("_fs_t", t)
        eng.amg_vcycle(0, "_fs_t", 0, "_fs_w", 0)
        out = np.zeros_like(xh)
        out[0] = eng.vec_get("_fs_w")[0]
        out[1] = y1
        eng.vec_set(y, out)


class CompositePC():
    """PCCOMPOSITE multiplicative("python,bjacobi") (singlephase.py:341-343) assembled from the
    stage objects: y = B1 x ; r = x - J y ; y += ILU0(r).  Used by tests to show that driving the
    stages through the PCBase API reproduces tp_pc_apply."""

    def __init__(self, stage1):
        self.stage1 = stage1

    def setUp(self, pc):
        self.stage1.setUp(pc)

    def apply(self, pc, x, y):
        eng = pc.engine
        self.stage1.apply(pc, x, y)
        eng.spmv(y, "_cmp_Jy")
        eng.vec_axpby("_cmp_r", 1.0, x, -1.0, "_cmp_Jy")
        eng.ilu_solve("_cmp_r", "_cmp_z")
        eng.vec_axpby(y, 1.0, y, 1.0, "_cmp_z")

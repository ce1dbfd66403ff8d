"""Child process of tests/test_gpu_parity.py::test_env_selected_sweep_kernels: the kernel variants that are selected by
environment variables read once per process (ILU(0) sweeps: TP_ILU_MW, TP_ILU_BLOCK, TP_ILU_YLDS; ILU(1): TP_ILU1_PACK, TP_ILU1_FACTOR_TILE, TP_ILU1_PF; system AMG: TP_BAMG_*;
LDS-tiled assembly: TP_ASM_LDS) against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases                                            # noqa: E402
from oracle.engine import OracleEngine                  # noqa: E402
from thermalporous_amd.engine import HipEngine          # noqa: E402


def rel2(a, b):
    return np.linalg.norm((a - b).ravel())/max(np.linalg.norm(b.ravel()), 1e-300)


CASES = [(cases.c4_spe10_3d, dict(Nx=11, Ny=13, Nz=17, nphase=2), dict(pc="cptr", ilu_tile=(5, 4, 7))),
         (cases.c4_spe10_3d, dict(Nx=9, Ny=14, Nz=40, nphase=2), dict(pc="cptr")),                  # whole lines of 40 cells
         (cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=1), dict(pc="cpr", decoup="TI")),
         (cases.c3_spe10_2d, dict(Nx=30, Ny=41, nphase=2), dict(pc="cptr")),
         # block-ILU(1): packed (default) or padded (TP_ILU1_PACK=0) stream of the sweeps, partial tiles and 2x2 blocks
         (cases.c4_spe10_3d, dict(Nx=11, Ny=13, Nz=17, nphase=2), dict(pc="cpr", ilu_levels=1, ilu_tile=(5, 4, 7))),
         (cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=1), dict(pc="cpr", decoup="TI", ilu_levels=1, ilu_tile=(1 << 30, 8, 8))),
         # system AMG: tail kernel / fused correction / LDS dense inverse (defaults) or the per-level kernels
         # (TP_BAMG_TAIL_CELLS=0, TP_BAMG_FUSE_BELOW=0, TP_BAMG_DENSE_LDS=0)
         (cases.c4_spe10_3d, dict(Nx=12, Ny=20, Nz=10, nphase=2), dict(pc="cptramg", decoup="QI"))]
for builder, kw, opts in CASES:
    spec, u0, *_ = builder(**kw)
    o, h = OracleEngine(spec, opts), HipEngine(spec, opts)
    u = cases.perturbed_state(spec, seed=5, amp=0.3)
    for e in (o, h):
        e.set_old(u0)
        e.set_dt(8640.0)
        e.set_state(u)
    schur = opts["pc"] == "cptr"
    out = o.jacobian(want_schur=schur)
    J, Sm = out if schur else (out, None)
    outh = h.jacobian(want_schur=schur)
    Jh, Smh = outh if schur else (outh, None)
    # assembly entries (the kernel variant TP_ASM_LDS selects must reproduce the default's entries)
    assert np.abs(h.residual() - o.residual()).max() <= 1e-11*np.abs(o.residual()).max()
    for r in range(o.b):
        for cc in range(o.b):
            sc = np.abs(J[:, r, cc]).max()
            assert np.abs(Jh[:, r, cc] - J[:, r, cc]).max() <= 1e-11*sc, (builder.__name__, kw, r, cc)
    if schur:
        assert np.abs(Smh - Sm).max() <= 1e-11*np.abs(Sm).max()
    o.pc.setup(J, Sm)
    h.pc_setup()
    x = np.random.default_rng(11).standard_normal(u.shape)
    h.vec_set("x", x)
    h.ilu_solve("x", "y")
    assert rel2(h.vec_get("y"), o.pc.ilu.solve(x)) < 1e-10, (builder.__name__, kw)
    h.pc_apply("x", "y")
    assert rel2(h.vec_get("y"), o.pc.apply(x)) < 1e-9, (builder.__name__, kw)
    h.close()
print("ok")

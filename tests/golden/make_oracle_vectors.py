"""Regenerates oracle_vectors.npz: inputs and expected outputs of the hot path on three small seeded cases,
computed by the CPU oracle (oracle/), NOT by the reference (which cannot be imported here: Firedrake is not
installed, and it ships no golden vectors -- parity with the reference itself stays unpinned, see oracle/__init__.py).

What the fixture is for: (a) the oracle is pinned against drift (tests/test_golden_vectors.py, CPU);
(b) the HIP path is compared with committed numbers, not only with an oracle computed in the same test run (GPU).

Per case: the perturbed state u, the old state u0, dt -> residual R, Jacobian J (7,b,b,...), [S~], and one
Newton solve from u0 (converged state, Newton / Krylov iteration counts).
usage: python tests/golden/make_oracle_vectors.py [r1|r2|r3|all]     (default r3)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

import cases                                    # noqa: E402
from oracle.engine import OracleEngine          # noqa: E402

CASES = {
    "c1_1ph_2d": (cases.c1_homogeneous, dict(N=8, nphase=1), dict(pc="cpr", ilu_tile=(1 << 30, 64, 1), amg_dom_tau=0.0, amg_omega=0.8), 86400.0),
    "c3_2ph_2d": (cases.c3_spe10_2d, dict(Nx=10, Ny=12, nphase=2),
                  dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, ilu_tile=(1 << 30, 64, 1), amg_dom_tau=0.0, amg_omega=0.8), 864.0),
    "c4_2ph_3d": (cases.c4_spe10_3d, dict(Nx=6, Ny=7, Nz=5, nphase=2), dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, ilu_tile=(1 << 30, 8, 8), amg_dom_tau=0.0, amg_omega=0.8), 86.4),   # (tile and amg_dom_tau pinned: the fixtures predate the balanced tile and the relaxation-only truncation)
    "c4_1ph_3d_fscd": (cases.c4_spe10_3d, dict(Nx=6, Ny=7, Nz=5, nphase=1), dict(pc="fieldsplit_cd", ksp_rtol=1e-8, amg_dom_tau=0.0, amg_omega=0.8), 864.0),
}

# round 2 additions, in a file of their own (oracle_vectors_r2.npz) so that the round-1 fixture stays byte-identical:
# block-ILU(1), selfp, the system AMG of pc_cptramg, and the engines' current defaults (balanced tiles, amg_dom_tau 0.25:
# the S~ hierarchy of the last case ends with relaxation only)
CASES_R2 = {
    "c4_2ph_3d_ilu1": (cases.c4_spe10_3d, dict(Nx=6, Ny=7, Nz=5, nphase=2), dict(pc="cpr", ilu_levels=1, ksp_rtol=1e-8, snes_max_it=25, amg_omega=0.8), 86.4),
    "c2_1ph_2d_selfp": (cases.c3_spe10_2d, dict(Nx=10, Ny=12, nphase=1), dict(pc="fieldsplit_cd", schur_selfp=True, ksp_rtol=1e-8, amg_omega=0.8), 8640.0),
    "c4_2ph_3d_cptramg_QI": (cases.c4_spe10_3d, dict(Nx=6, Ny=7, Nz=5, nphase=2), dict(pc="cptramg", decoup="QI", ksp_rtol=1e-8, snes_max_it=25, amg_omega=0.8), 86.4),
    "c4_2ph_3d_defaults": (cases.c4_spe10_3d, dict(Nx=6, Ny=9, Nz=7, nphase=2), dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, amg_omega=0.8), 86.4),
}
# round 3 additions (oracle_vectors_r3.npz): ONE bjacobi block = ILU(0) of the whole grid (sub_1_pc_bjacobi_blocks: 1: the GPU
# engine sweeps it tile-diagonal by tile-diagonal, the oracle has one tile), with pc_cptr and inside pc_cptr_a11; 13 x 9 = 117 > 64
# columns in the 3-D cases: more than one wavefront
CASES_R3 = {
    "c4_2ph_3d_whole": (cases.c4_spe10_3d, dict(Nx=9, Ny=13, Nz=6, nphase=2), dict(pc="cptr", bjacobi_blocks=1, ksp_rtol=1e-8, snes_max_it=25, amg_omega=0.8), 86.4),
    "c4_2ph_3d_a11_whole": (cases.c4_spe10_3d, dict(Nx=9, Ny=13, Nz=6, nphase=2),
                            dict(pc="cptr", schur_a11=True, bjacobi_blocks=1, ksp_rtol=1e-8, snes_max_it=25, amg_omega=0.8), 86.4),
    "c2_1ph_2d_whole": (cases.c3_spe10_2d, dict(Nx=20, Ny=70, nphase=1), dict(pc="cpr", decoup="QI", bjacobi_blocks=1, ksp_rtol=1e-8, amg_omega=0.8), 8640.0),
    # the engines' defaults as of round 3 (damped-Jacobi weight 0.9): nothing pinned
    "c4_2ph_3d_defaults_r3": (cases.c4_spe10_3d, dict(Nx=6, Ny=9, Nz=7, nphase=2), dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25), 86.4),
}
# (amg_omega = 0.8 is pinned in every case: the fixtures predate the round-3 default of 0.9)
ALL_CASES = {**CASES, **CASES_R2, **CASES_R3}


def compute(name):
    builder, kw, opts, dt = ALL_CASES[name]
    spec, u0, *_ = builder(**kw)
    o = OracleEngine(spec, opts)
    u = cases.perturbed_state(spec, seed=3)
    o.set_old(u0)
    o.set_dt(dt)
    o.set_state(u)
    out = {"u": u, "u0": u0, "dt": np.float64(dt), "R": o.residual()}
    schur = opts["pc"] in ("cptr", "fieldsplit_cd")
    j = o.jacobian(want_schur=schur)
    out["J"] = j[0] if schur else j
    if schur:
        out["Sm"] = j[1]
    o.set_state(u0)
    o.set_old(u0)
    info = o.newton_solve()
    out["newton_state"] = o.get_state()
    out["newton_info"] = np.array([info["nits"], info["lits"], info["reason"]])
    return out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "r3"          # "r1" / "r2": regenerate the earlier rounds' files
    for tag, table, fname in (("r1", CASES, "oracle_vectors.npz"), ("r2", CASES_R2, "oracle_vectors_r2.npz"),
                              ("r3", CASES_R3, "oracle_vectors_r3.npz")):
        if tag != which and which != "all":
            continue
        blob = {}
        for name in table:
            for k, v in compute(name).items():
                blob[name + "/" + k] = v
        np.savez_compressed(os.path.join(HERE, fname), **blob)
        print("wrote", fname, len(blob), "arrays,", os.path.getsize(os.path.join(HERE, fname)), "bytes")

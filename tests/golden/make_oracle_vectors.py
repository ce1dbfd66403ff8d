"""Regenerates oracle_vectors.npz: inputs and expected outputs of the hot path on three small seeded cases,
computed by the CPU oracle (oracle/), NOT by the reference (which cannot be imported here: Firedrake is not
installed, and it ships no golden vectors -- parity with the reference itself stays unpinned, see oracle/__init__.py).

What the fixture is for: (a) the oracle is pinned against drift (tests/test_golden_vectors.py, CPU);
(b) the HIP path is compared with committed numbers, not only with an oracle computed in the same test run (GPU).

Per case: the perturbed state u, the old state u0, dt -> residual R, Jacobian J (7,b,b,...), [S~], and one
Newton solve from u0 (converged state, Newton / Krylov iteration counts).
usage: python tests/golden/make_oracle_vectors.py
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
    "c1_1ph_2d": (cases.c1_homogeneous, dict(N=8, nphase=1), dict(pc="cpr", ilu_tile=(1 << 30, 64, 1), amg_dom_tau=0.0), 86400.0),
    "c3_2ph_2d": (cases.c3_spe10_2d, dict(Nx=10, Ny=12, nphase=2),
                  dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, ilu_tile=(1 << 30, 64, 1), amg_dom_tau=0.0), 864.0),
    "c4_2ph_3d": (cases.c4_spe10_3d, dict(Nx=6, Ny=7, Nz=5, nphase=2), dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, ilu_tile=(1 << 30, 8, 8), amg_dom_tau=0.0), 86.4),   # (tile and amg_dom_tau pinned: the fixtures predate the balanced tile and the relaxation-only truncation)
    "c4_1ph_3d_fscd": (cases.c4_spe10_3d, dict(Nx=6, Ny=7, Nz=5, nphase=1), dict(pc="fieldsplit_cd", ksp_rtol=1e-8, amg_dom_tau=0.0), 864.0),
}


def compute(name):
    builder, kw, opts, dt = CASES[name]
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
    blob = {}
    for name in CASES:
        for k, v in compute(name).items():
            blob[name + "/" + k] = v
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **blob)
    print("wrote", len(blob), "arrays,", os.path.getsize(os.path.join(HERE, "oracle_vectors.npz")), "bytes")

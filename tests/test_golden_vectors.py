"""Committed golden vectors (tests/golden/oracle_vectors.npz and oracle_vectors_r2.npz, made by tests/golden/make_oracle_vectors.py from the
CPU oracle -- the reference itself cannot run here, parity with it stays unpinned).  CPU: the oracle still
reproduces them.  GPU: the HIP path, through the C ABI, reproduces them."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_oracle_vectors", os.path.join(HERE, "golden", "make_oracle_vectors.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)

_files = [np.load(os.path.join(HERE, "golden", f), allow_pickle=False) for f in ("oracle_vectors.npz", "oracle_vectors_r2.npz", "oracle_vectors_r3.npz")]
GOLD = {k: f[k] for f in _files for k in f.files}          # (every round's cases live in a file of their own)


def relmax(a, b):
    return np.abs(a - b).max()/max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("name", sorted(gen.ALL_CASES))
def test_oracle_reproduces_golden_vectors(name):
    out = gen.compute(name)
    for k, v in out.items():
        g = GOLD[name + "/" + k]
        if k == "newton_info":
            assert list(v) == list(g)
        else:
            assert v.shape == g.shape and relmax(v, g) < 1e-9, k       # same code, same machine class: round-off only


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(gen.ALL_CASES))
def test_hip_path_reproduces_golden_vectors(name):
    from thermalporous_amd.engine import HipEngine
    builder, kw, opts, dt = gen.ALL_CASES[name]
    spec, u0, *_ = builder(**kw)
    g = {k.split("/", 1)[1]: v for k, v in GOLD.items() if k.startswith(name + "/")}
    assert np.array_equal(g["u0"], u0)                                   # the seeded inputs are the fixture's
    h = HipEngine(spec, opts)
    h.set_old(g["u0"])
    h.set_dt(float(g["dt"]))
    h.set_state(g["u"])
    R = h.residual()
    for f in range(R.shape[0]):
        assert relmax(R[f], g["R"][f]) < 1e-11
    schur = "Sm" in g
    j = h.jacobian(want_schur=schur)
    J = j[0] if schur else j
    b = J.shape[1]
    for r in range(b):
        for c in range(b):
            scale = np.abs(g["J"][:, r, c]).max()
            assert np.abs(J[:, r, c] - g["J"][:, r, c]).max() <= 1e-11*scale
    if schur:
        assert relmax(j[1], g["Sm"]) < 1e-11
    h.set_state(g["u0"])
    h.set_old(None)
    info = h.newton_solve()
    nits, lits, reason = (int(v) for v in g["newton_info"])
    assert info["reason"] == reason and info["nits"] == nits and abs(info["lits"] - lits) <= max(2, 0.1*lits)
    us = h.get_state()
    for f in range(2):
        assert np.linalg.norm(us[f] - g["newton_state"][f]) <= 1e-8*np.linalg.norm(g["newton_state"][f])
    if b == 3:
        assert np.abs(us[2] - g["newton_state"][2]).max() < 1e-8
    h.close()

"""Oracle closures vs the committed known-answer values (tests/golden/closure_kats.json, produced by
tests/golden/make_closure_kats.py from a scalar restatement of physicalparameters.py:37-90) and vs
complex-step derivatives.  The reference ships no golden vectors: parity with it is unpinned."""
import json
import os

import numpy as np

from oracle import closures as cl

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "closure_kats.json")))


def test_closure_known_answers():
    for k in KATS["points"]:
        p, T = k["p"], k["T"]
        assert np.isclose(cl.oil_rho(p, T, 10.0)[0], k["oil_rho"], rtol=1e-14)
        assert np.isclose(cl.oil_mu(T, 10.0)[0], k["oil_mu"], rtol=1e-13)
        assert np.isclose(cl.water_rho(p, T)[0], k["water_rho"], rtol=1e-14)
        assert np.isclose(cl.water_mu(T)[0], k["water_mu"], rtol=1e-14)


def test_host_closures_agree_with_oracle():
    """thermalporous_amd.physicalparameters (host diagnostics) is a third, independent copy."""
    from thermalporous_amd.physicalparameters import PhysicalParameters
    prm = PhysicalParameters()
    for k in KATS["points"]:
        p, T = k["p"], k["T"]
        assert np.isclose(prm.oil_rho(p, T), k["oil_rho"], rtol=1e-14)
        assert np.isclose(prm.oil_mu(T), k["oil_mu"], rtol=1e-13)
        assert np.isclose(prm.water_rho(p, T), k["water_rho"], rtol=1e-14)
        assert np.isclose(prm.water_mu(T), k["water_mu"], rtol=1e-14)
    assert np.isclose(0.2*prm.ko + 0.8*prm.kr, KATS["kT_homogeneous"], rtol=1e-15)


def test_closure_derivatives_complex_step():
    rng = np.random.default_rng(0)
    p = 20.0 + 60.0*rng.random(50)
    T = 280.0 + 150.0*rng.random(50)
    h = 1e-30
    r, rp, rT = cl.oil_rho(p, T, 10.0)
    assert np.allclose(rp, cl.oil_rho(p + 1j*h, T, 10.0)[0].imag/h, rtol=1e-13)
    assert np.allclose(rT, cl.oil_rho(p, T + 1j*h, 10.0)[0].imag/h, rtol=1e-13)
    m, mT = cl.oil_mu(T, 10.0)
    assert np.allclose(mT, cl.oil_mu(T + 1j*h, 10.0)[0].imag/h, rtol=1e-12)
    r, rp, rT = cl.water_rho(p, T)
    assert np.allclose(rp, cl.water_rho(p + 1j*h, T)[0].imag/h, rtol=1e-13)
    assert np.allclose(rT, cl.water_rho(p, T + 1j*h)[0].imag/h, rtol=1e-12)
    m, mT = cl.water_mu(T)
    assert np.allclose(mT, cl.water_mu(T + 1j*h)[0].imag/h, rtol=1e-13)


def test_peaceman_and_weights():
    from thermalporous_amd.wellcase import peaceman_WI
    K = KATS["peaceman"]["K"]
    assert np.isclose(peaceman_WI(K, K), KATS["peaceman"]["WI"], rtol=1e-14)
    import cases
    spec, *_ = cases.c4_spe10_3d(6, 8, 5)
    from oracle.tpfa import Problem
    P = Problem(spec)
    assert np.isclose(P.w0, KATS["weights"]["p_weight"])
    assert np.isclose(P.w2, KATS["weights"]["o_weight_So09"], rtol=1e-14)

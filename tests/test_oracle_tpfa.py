"""Identities the discrete equations must satisfy (SURVEY.md 8c iii-iv) + exact-Jacobian checks.
All on the CPU oracle; sizes chosen so the file runs in seconds."""
import numpy as np
import pytest

import cases
from oracle.tpfa import Problem, harmonic


def _strip_sources(spec):
    s = dict(spec)
    s["sources"] = None
    return s


@pytest.mark.parametrize("nphase", [1, 2])
def test_uniform_state_no_sources_is_equilibrium_2d(nphase):
    spec, u0, *_ = cases.c3_spe10_2d(9, 11, nphase)
    P = Problem(_strip_sources(spec))
    P.set_old(u0)
    P.set_dt(3600.0)
    R = P.residual(u0)
    assert np.abs(R).max() == 0.0


@pytest.mark.parametrize("nphase", [1, 2])
def test_flux_terms_conserve(nphase):
    """Sum over cells of every pure-flux term vanishes: sum(R) equals the sum of the accumulation."""
    spec, u0, *_ = cases.c4_spe10_3d(6, 7, 5, nphase)
    P = Problem(_strip_sources(spec))
    u = cases.perturbed_state(spec, seed=1)
    P.set_old(u0)
    P.set_dt(1000.0)
    R = P.residual(u)
    w = [P.w0, 1.0, P.w2]
    acc = (P.accum(*P.split(u)) - P.old)*(P.V/P.dt)
    for q in range(P.b):
        total, ref = R[q].sum(), (w[q]*acc[q]).sum()
        scale = np.abs(R[q]).sum()
        assert abs(total - ref) < 1e-12*scale


def test_hydrostatic_column_has_zero_vertical_flux():
    """p_lower = p_upper + rho_bar g Dz  =>  z_flow = 0 (singlephase.py:215; '+' = lower cell)."""
    spec, u0, p, g, c = cases.c4_spe10_3d(3, 4, 6, nphase=1, homogeneous=True)
    P = Problem(_strip_sources(spec))
    ga = spec["gaxis"]
    assert ga == 0                      # z is internal axis 0
    n0 = spec["n"][0]
    T = np.full(P.shape, 300.0)
    pz = np.zeros(n0)
    pz[0] = 45.0
    from oracle import closures as cl
    for k in range(1, n0):              # march upwards solving p_k + g Dz/2 rho(p_k) = p_{k-1} - g Dz/2 rho(p_{k-1})
        lo = pz[k-1]
        x = lo
        for _ in range(50):
            f = lo - x - p.g*spec["h"][0]*0.5*(cl.oil_rho(lo, 300.0, p.API)[0] + cl.oil_rho(x, 300.0, p.API)[0])
            df = -1.0 - p.g*spec["h"][0]*0.5*cl.oil_rho(x, 300.0, p.API)[1]
            x -= f/df
        pz[k] = x
    u = np.array([np.broadcast_to(pz, P.shape).copy(), T])
    P.set_old(u)
    P.set_dt(1.0)
    R = P.residual(u)
    assert np.abs(R[0]).max() < 1e-12*P.TK[0].max()*1e3
    assert pz[1] < pz[0]                # pressure decreases upwards


def test_xy_swap_symmetry():
    """Swapping x<->y together with Kx<->Ky, Dx<->Dy gives the transposed residual."""
    spec, u0, *_ = cases.c3_spe10_2d(7, 9, 2)
    spec = _strip_sources(spec)
    P = Problem(spec)
    u = cases.perturbed_state(spec, seed=2)
    P.set_old(u0)
    P.set_dt(500.0)
    R = P.residual(u)
    sw = dict(spec)
    sw["n"] = (spec["n"][1], spec["n"][0], 1)
    sw["h"] = (spec["h"][1], spec["h"][0], spec["h"][2])
    t = lambda a: np.ascontiguousarray(np.swapaxes(a, 1, 2))
    sw["phi"], sw["kT"] = t(spec["phi"]), t(spec["kT"])
    sw["K"] = [t(spec["K"][1]), t(spec["K"][0]), t(spec["K"][2])]
    P2 = Problem(sw)
    P2.set_old(np.array([t(f) for f in u0]))
    P2.set_dt(500.0)
    R2 = P2.residual(np.array([t(f) for f in u]))
    for q in range(3):
        assert np.allclose(t(R2[q]), R[q], rtol=1e-12, atol=1e-12*np.abs(R[q]).max())


def test_two_cell_flux_by_hand():
    """1-phase, two cells along x: mass flux = H(K) Dy/Dx * (rho/mu)_up * (p+ - p-)."""
    from oracle import closures as cl
    prm = cases.PhysicalParameters().as_dict()
    Kp, Km, Dx, Dy = 2e-7, 5e-7, 3.0, 2.0
    spec = dict(nphase=1, n=(2, 1, 1), h=(Dx, Dy, 1.0), gaxis=-1, phi=np.full((1, 1, 2), 0.2),
                K=[np.array([[[Kp, Km]]]), np.ones((1, 1, 2)), np.ones((1, 1, 2))], kT=np.full((1, 1, 2), 1.4),
                prm=prm, sources=None)
    P = Problem(spec)
    u = np.array([[[[42.0, 41.0]]], [[[300.0, 310.0]]]])
    P.set_old(u)
    P.set_dt(10.0)
    R = P.residual(u)
    H = 2*Kp*Km/(Kp + Km)
    rho, mu = cl.oil_rho(42.0, 300.0, 10.0)[0], cl.oil_mu(300.0, 10.0)[0]
    F = H*Dy/Dx*rho/mu*(42.0 - 41.0)
    assert np.isclose(R[0, 0, 0, 0], F, rtol=1e-13) and np.isclose(R[0, 0, 0, 1], -F, rtol=1e-13)
    cond = 1.4*Dy/Dx*(300.0 - 310.0)
    assert np.isclose(R[1, 0, 0, 0], prm["c_v_o"]*300.0*F + cond, rtol=1e-13)
    assert harmonic(np.array(0.0), np.array(0.0)) == 0.0


@pytest.mark.parametrize("builder,kw", [(cases.c1_homogeneous, dict(N=7, nphase=1, constant_rate=False)),
                                        (cases.c3_spe10_2d, dict(Nx=6, Ny=8, nphase=2)),
                                        (cases.c4_spe10_3d, dict(Nx=5, Ny=6, Nz=4, nphase=2)),
                                        (cases.c4_spe10_3d, dict(Nx=4, Ny=5, Nz=6, nphase=1))])
def test_analytic_jacobian_equals_complex_step(builder, kw):
    spec, u0, *_ = builder(**kw)
    P = Problem(spec)
    u = cases.perturbed_state(spec, seed=4)
    P.set_old(u0)
    P.set_dt(2000.0)
    J = P.jacobian(u)
    Jc = P.jacobian_complex_step(u)
    assert np.abs(J - Jc).max() <= 1e-13*np.abs(Jc).max()


def test_jacobian_vs_central_differences():
    spec, u0, *_ = cases.c4_spe10_3d(4, 5, 4, 2)
    P = Problem(spec)
    u = cases.perturbed_state(spec, seed=6)
    P.set_old(u0)
    P.set_dt(2000.0)
    import oracle.linalg as la
    J = P.jacobian(u)
    rng = np.random.default_rng(0)
    d = rng.standard_normal(u.shape)*np.array([1e-3, 1e-2, 1e-4]).reshape(3, 1, 1, 1)
    eps = 1e-3
    fd = (P.residual(u + eps*d) - P.residual(u - eps*d))/(2*eps)
    Jd = la.spmv_block(J, d)
    assert np.linalg.norm(fd - Jd)/np.linalg.norm(Jd) < 1e-6


def test_schur_operator_is_frozen_energy_block():
    """S~ (preconditioners.py:165-333) = d R_E/d T with densities/mobilities/kT frozen: for a uniform
    saturation/pressure state with zero flow only accumulation + conduction remain, and they are
    exactly the terms of J_TT that do not involve d rho/dT."""
    spec, u0, *_ = cases.c3_spe10_2d(6, 7, 2)
    spec = _strip_sources(spec)
    P = Problem(spec)
    u = u0.copy()
    u[1] += 5.0*np.random.default_rng(0).random(u[1].shape)
    P.set_old(u0)
    P.set_dt(1000.0)
    J, Sm = P.jacobian(u, want_schur=True)
    # off-diagonal entries: pure conduction (no flow): identical in S~ and J_TT
    for s in range(1, 5):
        assert np.allclose(Sm[s], J[s, 1, 1], rtol=1e-12, atol=1e-14*np.abs(Sm).max())

"""Oracle linear algebra: each piece against an independent formulation (scipy CSR algebra written
directly from the reference's PETSc calls, or the defining property of the method)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import cases
import oracle.linalg as la
from oracle.engine import OracleEngine


def setup_case(builder=cases.c4_spe10_3d, opts=None, **kw):
    spec, u0, *_ = builder(**kw)
    o = OracleEngine(spec, opts or dict(pc="cptr"))
    u = cases.perturbed_state(spec, seed=7, amp=0.3)
    o.set_old(u0)
    o.set_dt(4000.0)
    o.set_state(u)
    return spec, o


def flat(x):
    b = x.shape[0]
    return x.reshape(b, -1).T.reshape(-1)          # cell-interleaved, matching to_csr


def test_spmv_block_matches_csr():
    spec, o = setup_case(Nx=5, Ny=6, Nz=4)
    J = o.jacobian()
    x = np.random.default_rng(0).standard_normal(J.shape[1:2] + J.shape[3:])
    assert np.allclose(flat(la.spmv_block(J, x)), la.to_csr(J) @ flat(x), rtol=1e-13, atol=1e-9)


def _field_major_blocks(J):
    """Field sub-blocks A_qr as scipy matrices (what ExtractSubBlock + assemble give the reference)."""
    b = J.shape[1]
    return [[la.to_csr(J[:, q:q+1, r:r+1]) for r in range(b)] for q in range(b)]


@pytest.mark.parametrize("kind", ["QI", "TI"])
@pytest.mark.parametrize("nphase", [1, 2])
def test_cpr_decoupling_matches_petsc_algebra(kind, nphase):
    """Atildepp = App - Dps Dss^-1 Asp with D = diagonals (QI, preconditioners.py:785-808) or column sums
    (TI: transpose + getRowSum, :684-711), written here with scipy exactly as the reference writes it
    with petsc4py -- versus the oracle's per-cell row operation on the stencil."""
    spec, o = setup_case(Nx=5, Ny=6, Nz=4, nphase=nphase, opts=dict(pc="cpr", decoup=kind))
    J = o.jacobian()
    A = _field_major_blocks(J)
    s = J.shape[1] - 1
    App, Aps, Asp, Ass = A[0][0], A[0][s], A[s][0], A[s][s]
    if kind == "QI":
        invdiag = 1.0/Ass.diagonal()
        diag = Aps.diagonal()
    else:
        invdiag = 1.0/np.asarray(Ass.T.sum(axis=1)).ravel()
        diag = np.asarray(Aps.T.sum(axis=1)).ravel()
    apsinvdss = sp.diags(diag) @ sp.diags(invdiag)
    Atilde = App - apsinvdss @ Asp
    At, d = la.decouple(J, kind, [0])
    assert abs(la.to_csr(At) - Atilde).max() < 1e-12*abs(Atilde).max()
    x = np.random.default_rng(1).standard_normal(J.shape[1:2] + J.shape[3:])
    r_ref = x[0].reshape(-1) - apsinvdss @ x[s].reshape(-1)            # :894-895
    # TI column sums cancel to ~1e-9 of their terms: summation order shows up at ~1e-7
    assert np.allclose((x[0] - d[0]*x[s]).reshape(-1), r_ref, rtol=1e-12 if kind == "QI" else 1e-5)


@pytest.mark.parametrize("kind", ["QI_temp", "TI_temp"])
def test_cpr_temp_decoupling_matches_petsc_algebra(kind):
    """QI_temp / TI_temp (preconditioners.py:714-783, 810-873): per cell invDss = inverse of the 2x2 block
    [[TT, TS], [ST, SS]] of diagonals / column sums, Dps = [pT, pS]; Atildepp = App - Dps invDss Asp with
    s = (T,S).  Written with scipy matrices in the reference's order of operations."""
    spec, o = setup_case(Nx=5, Ny=6, Nz=4, nphase=2, opts=dict(pc="cpr", decoup=kind))
    J = o.jacobian()
    A = _field_major_blocks(J)
    n = A[0][0].shape[0]

    def vec(M):
        return M.diagonal() if kind == "QI_temp" else np.asarray(M.T.sum(axis=1)).ravel()
    blk = np.empty((n, 2, 2))
    blk[:, 0, 0], blk[:, 0, 1], blk[:, 1, 0], blk[:, 1, 1] = vec(A[1][1]), vec(A[1][2]), vec(A[2][1]), vec(A[2][2])
    inv = np.linalg.inv(blk)                                         # the per-cell np.linalg.inv(block) loop (:746-755)
    invDss = sp.bmat([[sp.diags(inv[:, 0, 0]), sp.diags(inv[:, 0, 1])], [sp.diags(inv[:, 1, 0]), sp.diags(inv[:, 1, 1])]])
    Dps = sp.hstack([sp.diags(vec(A[0][1])), sp.diags(vec(A[0][2]))])
    Asp = sp.vstack([A[1][0], A[2][0]])
    apsinvdss = Dps @ invDss
    Atilde = A[0][0] - apsinvdss @ Asp
    At, d = la.decouple(J, kind, [0])
    tol = 1e-12 if kind == "QI_temp" else 1e-6                       # TI column sums cancel (see above)
    assert abs(la.to_csr(At) - Atilde).max() < tol*abs(Atilde).max()
    x = np.random.default_rng(1).standard_normal(J.shape[1:2] + J.shape[3:])
    r_ref = x[0].reshape(-1) - apsinvdss @ np.concatenate([x[1].reshape(-1), x[2].reshape(-1)])     # :889-895
    assert np.allclose((x[0] - d[0][0]*x[1] - d[0][1]*x[2]).reshape(-1), r_ref, rtol=1e-12 if kind == "QI_temp" else 1e-5)


@pytest.mark.parametrize("kind", ["QI", "TI"])
def test_cptr_decoupling_matches_petsc_algebra(kind):
    """Atilde00 = A00 - D0s Dss^-1 As0, primary = (p,T) (preconditioners.py:1445-1543)."""
    spec, o = setup_case(Nx=5, Ny=6, Nz=4, opts=dict(pc="cptr", decoup=kind))
    J = o.jacobian()
    A = _field_major_blocks(J)
    inv = (1.0/A[2][2].diagonal()) if kind == "QI" else 1.0/np.asarray(A[2][2].T.sum(axis=1)).ravel()
    At, d = la.decouple(J, kind, [0, 1])
    for i in range(2):
        D = A[i][2].diagonal() if kind == "QI" else np.asarray(A[i][2].T.sum(axis=1)).ravel()
        for j in range(2):
            ref = A[i][j] - sp.diags(D*inv) @ A[2][j]
            # TI column sums suffer cancellation (~1e-9 of their terms): summation-order noise ~1e-7 relative
            tol = 1e-12 if kind == "QI" else 1e-6
            assert abs(la.to_csr(At[:, i:i+1, j:j+1]) - ref).max() < tol*abs(ref).max()


def test_ilu0_defining_property():
    """(L U)_ij = A_ij on the sparsity pattern of A: checked block-wise for the whole-domain tile."""
    spec, o = setup_case(Nx=4, Ny=5, Nz=3)
    J = o.jacobian()
    b = J.shape[1]
    ilu = la.TiledILU0(J.shape[3:], (1 << 30,)*3).factor(J)
    n = int(np.prod(J.shape[3:]))
    # apply M = (D~ + L_A) D~^-1 (D~ + U_A) to unit vectors via solve^-1: check M x = A x on pattern by
    # comparing M^-1 A e_j ~ e_j only where fill-in is absent is awkward; instead rebuild M explicitly.
    A = la.to_csr(J).toarray()
    Dt = np.zeros_like(A)
    Dinv = ilu.Dinv
    for c in range(n):
        Dt[c*b:(c+1)*b, c*b:(c+1)*b] = np.linalg.inv(Dinv[:, :, c])
    blk = sp.kron(sp.eye(n), np.ones((b, b))).toarray() > 0
    L = np.tril(A, -1)*(~blk) + 0.0
    U = np.triu(A, 1)*(~blk) + 0.0
    Dti = np.linalg.inv(Dt)
    M = (Dt + L) @ Dti @ (Dt + U)
    pattern = A != 0
    assert np.abs((M - A)[pattern]).max() < 1e-10*np.abs(A).max()
    r = np.random.default_rng(0).standard_normal((b,) + J.shape[3:])
    assert np.allclose(flat(ilu.solve(r)), np.linalg.solve(M, flat(r)), rtol=1e-9, atol=1e-12)


def test_tiled_ilu_equals_ilu_of_block_diagonal_restriction():
    spec, o = setup_case(Nx=6, Ny=7, Nz=5)
    J = o.jacobian()
    tile = (3, 2, 4)
    ilu = la.TiledILU0(J.shape[3:], tile).factor(J)
    # zero every coupling that leaves a tile, then whole-domain ILU(0) must give the same solve
    n2, n1, n0 = J.shape[3:]
    Jc = J.copy()
    i2, i1, i0 = np.meshgrid(np.arange(n2), np.arange(n1), np.arange(n0), indexing="ij")
    for a, (idx, t) in enumerate(zip((i0, i1, i2), tile)):
        Jc[1 + 2*a][:, :, idx % t == 0] = 0.0
        Jc[2 + 2*a][:, :, (idx % t == t - 1)] = 0.0
    ref = la.TiledILU0(J.shape[3:], (1 << 30,)*3).factor(Jc)
    r = np.random.default_rng(3).standard_normal(J.shape[1:2] + J.shape[3:])
    assert np.allclose(ilu.solve(r), ref.solve(r), rtol=1e-11, atol=1e-13)


def _dense_block_iluk(A, b, lev):
    """Textbook block ILU(lev): symbolic phase by the level-of-fill rule on the cell graph (lev(i,j) = min over k of
    lev(i,k) + lev(k,j) + 1), numeric phase = IKJ elimination restricted to that pattern.  Returns (L, U) dense."""
    n = A.shape[0]//b
    B = lambda M, i, j: M[i*b:(i + 1)*b, j*b:(j + 1)*b]
    lv = np.full((n, n), 10**6)
    for i in range(n):
        for j in range(n):
            if i == j or np.any(B(A, i, j) != 0):
                lv[i, j] = 0
    for i in range(n):
        for k in range(i):
            if lv[i, k] <= lev:
                for j in range(k + 1, n):
                    if lv[k, j] <= lev:
                        lv[i, j] = min(lv[i, j], lv[i, k] + lv[k, j] + 1)
    P = lv <= lev
    F = A.copy()
    for i in range(n):
        for k in range(i):
            if P[i, k]:
                Lik = B(F, i, k) @ np.linalg.inv(B(F, k, k))
                B(F, i, k)[:] = Lik
                for j in range(k + 1, n):
                    if P[i, j] and P[k, j]:
                        B(F, i, j)[:] -= Lik @ B(F, k, j)
    L, U = np.eye(n*b), np.zeros((n*b, n*b))
    for i in range(n):
        for j in range(n):
            if P[i, j]:
                (B(L, i, j) if j < i else B(U, i, j))[:] = B(F, i, j)
    return L, U, P


@pytest.mark.parametrize("shape,tile,b", [((3, 4, 5), None, 3), ((1, 5, 6), None, 3), ((2, 2, 7), None, 2),
                                          ((5, 7, 6), (4, 3, 2), 2)])
def test_ilu1_is_level_of_fill_one(shape, tile, b):
    """TiledILU1 (sub_1_sub_pc_factor_levels 1, twophase.py:665-666): its fixed 13-offset pattern and sweep order give the
    textbook level-of-fill ILU(1) of every tile's diagonal block in natural order (and TiledILU0 gives ILU(0))."""
    rng = np.random.default_rng(0)
    n2, n1, n0 = shape
    J = rng.standard_normal((7, b, b) + shape)*0.3
    J[0] += 4*np.eye(b)[:, :, None, None, None]

    def cut(M):
        M[1][..., 0] = 0; M[2][..., -1] = 0; M[3][..., 0, :] = 0; M[4][..., -1, :] = 0; M[5][..., 0, :, :] = 0; M[6][..., -1, :, :] = 0
        return M
    cut(J)
    big = (1 << 30,)*3
    t = tile or big
    r = rng.standard_normal((b,) + shape)
    for lev, cls in ((0, la.TiledILU0), (1, la.TiledILU1)):
        x = cls(shape, t).factor(J).solve(r)
        for k0 in range(0, n2, min(t[2], n2)):
            for j0 in range(0, n1, min(t[1], n1)):
                for i0 in range(0, n0, min(t[0], n0)):
                    sl = (slice(k0, min(k0 + t[2], n2)), slice(j0, min(j0 + t[1], n1)), slice(i0, min(i0 + t[0], n0)))
                    Jt = cut(J[(slice(None),)*3 + sl].copy())
                    L, U, P = _dense_block_iluk(la.to_csr(Jt).toarray(), b, lev)
                    rt = r[(slice(None),) + sl]
                    xd = np.linalg.solve(U, np.linalg.solve(L, rt.reshape(b, -1).T.reshape(-1))).reshape(-1, b).T.reshape(rt.shape)
                    assert np.abs(x[(slice(None),) + sl] - xd).max() < 1e-13
    if tile is None and n2 > 2 and n1 > 2 and n0 > 2:
        assert P.sum(axis=1).max() == 13           # interior rows: 7 stencil entries + 6 level-1 fill entries


@pytest.mark.parametrize("builder,kw", [(cases.c4_spe10_3d, dict(Nx=7, Ny=9, Nz=6, nphase=1)),
                                        (cases.c3_spe10_2d, dict(Nx=14, Ny=19, nphase=1))])
def test_selfp_operator_is_the_explicit_sparse_product(builder, kw):
    """SelfpSchur (pc_fieldsplit_schur_precondition selfp, singlephase.py:322-330) against PETSc's definition
    Sp = A11 - A10 diag(A00)^-1 A01 formed explicitly with scipy: the matrix-free product and the diagonal are exact,
    the 7-point collapse keeps the 7-point entries and the row sums of Sp, and the resulting preconditioner is as good
    as an exact Sp solve to within a few Krylov iterations."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    from oracle.engine import OracleEngine
    spec, u0, *_ = builder(**kw)
    o = OracleEngine(spec, dict(pc="fieldsplit_cd", ksp_rtol=1e-8, schur_selfp=True))
    o.set_old(u0)
    o.set_dt(8640.0)
    o.set_state(cases.perturbed_state(spec, seed=5, amp=0.3))
    J, Sm = o.jacobian(want_schur=True)
    o.pc.setup(J, Sm)
    shape = J.shape[3:]
    n = int(np.prod(shape))
    csr = lambda A: la.to_csr(A[:, None, None])
    A00, A01, A10, A11 = (csr(o.pc.At[:, i, j]) for i, j in ((0, 0), (0, 1), (1, 0), (1, 1)))
    Sp = (A11 - A10 @ sp.diags(1.0/A00.diagonal()) @ A01).tocsc()
    sel = o.pc.selfp
    x = np.random.default_rng(0).standard_normal(shape)
    d = np.abs(Sp.diagonal()).max()
    assert np.abs(sel.mult(x).ravel() - Sp @ x.ravel()).max() < 1e-13*np.abs(Sp @ x.ravel()).max()
    assert np.abs(sel.diag.ravel() - Sp.diagonal()).max() < 1e-13*d
    S7 = csr(sel.S7)
    assert np.abs(np.asarray(S7.sum(axis=1)).ravel() - np.asarray(Sp.sum(axis=1)).ravel()).max() < 1e-12*d
    offd = (abs(S7) > 0).astype(float) - sp.eye(n)
    assert abs(Sp.multiply(offd) - S7.multiply(offd)).max() < 1e-13*d
    assert Sp.nnz > S7.nnz                                      # (13- / 25-point vs 5- / 7-point)
    F = o.residual()
    mv = lambda v: la.spmv_block(J, v)
    _, its, reason, _ = la.fgmres(mv, o.pc.apply, F, rtol=1e-8)
    o.pc.selfp = type("Exact", (), {"vcycle": staticmethod(lambda b: spl.spsolve(Sp, b.ravel()).reshape(shape))})()
    _, its_x, reason_x, _ = la.fgmres(mv, o.pc.apply, F, rtol=1e-8)
    assert reason == reason_x == 2 and its <= its_x + 2, (its, its_x)


def test_amg_vcycle_is_a_convergent_preconditioner():
    spec, o = setup_case(builder=cases.c3_spe10_2d, Nx=24, Ny=31, nphase=2)
    J = o.jacobian()
    A = J[:, 0, 0]
    st = [float(np.mean(o.prob.TK[a][la._lo(a)])) if o.prob.n[a] > 1 else 0.0 for a in range(3)]
    amg = la.SemiAMG(o.prob.n, st, omega=0.8, nu=2).setup(A)
    assert all(l.shape[0] == 7 for l in amg.levels) and np.prod(amg.levels[-1].shape[1:]) <= 64
    b = np.random.default_rng(0).standard_normal(A.shape[1:])
    x, its, reason, _ = la.fgmres(lambda v: la.spmv_scalar(A, v), amg.vcycle, b, rtol=1e-8, maxit=60)
    assert reason == 2 and its < 40
    xd = spla.spsolve(la.to_csr(A[:, None, None]).tocsc(), b.reshape(-1))
    assert np.linalg.norm(x.reshape(-1) - xd)/np.linalg.norm(xd) < 1e-6
    # coarse operators keep M-matrix signs and a positive diagonal
    for lvl in amg.levels:
        assert (lvl[0] > 0).all() and (lvl[1:] <= 1e-300).all()


@pytest.mark.parametrize("pc,decoup,nphase", [("cpr", "No", 1), ("cpr", "QI", 2), ("cptr", "No", 2), ("cptr", "TI", 2)])
def test_fgmres_two_stage_vs_direct_solve(pc, decoup, nphase):
    spec, o = setup_case(Nx=6, Ny=8, Nz=5, nphase=nphase, opts=dict(pc=pc, decoup=decoup, ksp_rtol=1e-10))
    schur = pc == "cptr"
    out = o.jacobian(want_schur=schur)
    J, Sm = out if schur else (out, None)
    F = o.residual()
    x, its, reason, hist = o.linear_solve(J, Sm, F)
    xd = spla.spsolve(la.to_csr(J).tocsc(), flat(F))
    assert reason == 2 and its < 60
    # forward error = residual tolerance (1e-10) x condition number of the scaled system
    assert np.linalg.norm(flat(x) - xd)/np.linalg.norm(xd) < 1e-5
    assert np.linalg.norm(la.to_csr(J) @ flat(x) - flat(F)) <= 1.01e-10*np.linalg.norm(F)*10
    assert all(h1 <= h0*(1 + 1e-12) for h0, h1 in zip(hist, hist[1:]))      # GMRES residuals are monotone


def test_slab_emulation_keeps_iteration_counts():
    """N-GPU algorithm emulated in one process: ILU tiles restart at slab boundaries, stage 1 global."""
    spec, u0, *_ = cases.c4_spe10_3d(8, 20, 7)
    res = []
    for ns in (1, 2, 4):
        o = OracleEngine(spec, dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, nslabs=ns))
        o.set_state(u0)
        o.set_old(u0)
        o.set_dt(86.4)
        r = o.newton_solve()
        assert r["reason"] > 0
        res.append((r["nits"], r["lits"], o.get_state()))
    for nits, lits, u in res[1:]:
        assert nits == res[0][0] and abs(lits - res[0][1]) <= max(3, 0.15*res[0][1])
        assert np.linalg.norm(u[0] - res[0][2][0])/np.linalg.norm(res[0][2][0]) < 1e-8

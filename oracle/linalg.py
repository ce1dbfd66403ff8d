"""Linear algebra of the hot path on the structured 7-point stencil (oracle; test infrastructure).

What the reference delegates to PETSc/hypre, restated so that the HIP
kernels have a CPU checker computing the SAME algorithm:
  * MatMult on the 7-point block stencil                      (PETSc MatMult AIJ)
  * stage 2: block-Jacobi over tiles + block-ILU(0) per tile   (sub_1_* bjacobi/ilu levels 0,
    singlephase.py:348-349, twophase.py:547-548; levels 1: twophase.py:653-668; ``-sub_1_pc_bjacobi_blocks`` tests/test_homo_wells.py:112,125)
  * stage 1 operators: Quasi-/True-IMPES decoupling            (preconditioners.py:684-711,785-808,1445-1543)
  * pressure / temperature AMG V-cycle                         (v_cycle dicts singlephase.py:303-307)
  * fieldsplit Schur FULL apply for pc_cptr                    (twophase.py:536-545)
  * FGMRES, right preconditioned, classical Gram-Schmidt       (twophase.py:426-432)
hypre's BoomerAMG and PETSc's point-ILU on DMPlex numbering are not reproducible; the AMG here
is a structured semicoarsening AMG (SemiAMG below) and the ILU is the exact block ILU(0) of each tile in
natural order.  Iteration counts are therefore this build's own ("parity unpinned").

Stencil slots: 0 diag, 1 (-a0), 2 (+a0), 3 (-a1), 4 (+a1), 5 (-a2), 6 (+a2); arrays (.., n2, n1, n0).
"""
import numpy as np

from .tpfa import _lo, _hi


# ------------------------------------------------------------------ stencil mat-vec
def spmv_scalar(A, x):
    y = A[0] * x
    for a in range(3):
        if x.shape[2 - a] == 1:
            continue
        lo, hi = _lo(a), _hi(a)
        y[lo] += A[2 + 2 * a][lo] * x[hi]
        y[hi] += A[1 + 2 * a][hi] * x[lo]
    return y


def spmv_block(J, x):
    b = J.shape[1]
    y = np.zeros_like(x)
    for r in range(b):
        for c in range(b):
            y[r] += spmv_scalar(J[:, r, c], x[c])
    return y


def to_csr(J):
    """Assemble the block stencil into scipy CSR (cell-interleaved ordering) for direct-solve checks."""
    import scipy.sparse as sp
    b = J.shape[1]
    shape = J.shape[3:]
    n = int(np.prod(shape))
    n2, n1, n0 = shape
    idx = np.arange(n).reshape(shape)
    rows, cols, vals = [], [], []
    strides = (1, n0, n0 * n1)
    for s in range(7):
        if s == 0:
            sel, off = (slice(None),) * 3, 0
        else:
            a = (s - 1) // 2
            if shape[2 - a] == 1:
                continue
            sel = _hi(a) if s % 2 == 1 else _lo(a)
            off = -strides[a] if s % 2 == 1 else strides[a]
        for r in range(b):
            for c in range(b):
                rr = idx[sel].reshape(-1)
                rows.append(rr * b + r)
                cols.append((rr + off) * b + c)
                vals.append(J[s, r, c][sel].reshape(-1))
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n * b, n * b))


# ------------------------------------------------------------------ stage 2: tiled block ILU(0)
class TiledILU0:
    """Block-Jacobi over box tiles, exact block-ILU(0) in natural order inside each tile.

    On a 7-point stencil ILU(0) creates no off-diagonal updates (the lower neighbours of a cell
    are not adjacent to each other), so the factor is  M = (D~ + L_A) D~^-1 (D~ + U_A)  with
    D~_c = A_cc - sum_{m lower, same tile} A_cm D~_m^-1 A_mc.  Couplings that leave the tile are
    dropped (= PETSc bjacobi with one block per tile)."""

    def __init__(self, shape, tile, slabs=None):
        """slabs: list of (lo, hi) plane ranges along axis 2 (the multi-GPU decomposition); tiles
        restart at every slab boundary because each GPU factors only its own rows."""
        n2, n1, n0 = shape
        self.shape = shape
        t0, t1, t2 = (max(1, min(int(t), n)) for t, n in zip(tile, (n0, n1, n2)))
        self.tile = (t0, t1, t2)
        i2, i1, i0 = np.meshgrid(np.arange(n2), np.arange(n1), np.arange(n0), indexing="ij")
        slabs = slabs or [(0, n2)]
        lo2 = np.zeros(n2, dtype=int)
        hi2 = np.zeros(n2, dtype=int)
        for lo, hi in slabs:
            lo2[lo:hi], hi2[lo:hi] = lo, hi
        r2 = i2 - lo2[i2]                       # plane index relative to the owning slab
        self.l = [i0 % t0, i1 % t1, r2 % t2]
        self.tdim = [np.minimum(t0, n0 - (i0 // t0) * t0), np.minimum(t1, n1 - (i1 // t1) * t1),
                     np.minimum(t2, (hi2[i2] - lo2[i2]) - (r2 // t2) * t2)]
        self.level = (self.l[0] + self.l[1] + self.l[2]).reshape(-1)
        self.nlev = int(self.level.max()) + 1
        self.strides = (1, n0, n0 * n1)
        order = np.argsort(self.level, kind="stable")
        bounds = np.searchsorted(self.level[order], np.arange(self.nlev + 1))
        self.cells_at = [order[bounds[s]:bounds[s + 1]] for s in range(self.nlev)]
        lf = [x.reshape(-1) for x in self.l]
        td = [x.reshape(-1) for x in self.tdim]
        self.has_lo = [lf[a] > 0 for a in range(3)]
        self.has_hi = [lf[a] < td[a] - 1 for a in range(3)]
        self.Dinv = None

    def factor(self, J):
        b = J.shape[1]
        n = int(np.prod(self.shape))
        Jf = J.reshape(7, b, b, n)
        self.J = Jf
        Dinv = np.zeros((b, b, n))
        for s in range(self.nlev):
            c = self.cells_at[s]
            D = Jf[0][:, :, c].copy()
            for a in range(3):
                m = self.has_lo[a][c]
                cc = c[m]
                if len(cc) == 0:
                    continue
                nb = cc - self.strides[a]
                # A_cm D~_m^-1 A_mc : A_cm = slot(-a) at c ; A_mc = slot(+a) at m
                t = np.einsum("ijn,jkn,kln->iln", Jf[1 + 2 * a][:, :, cc], Dinv[:, :, nb], Jf[2 + 2 * a][:, :, nb])
                D[:, :, m] -= t
            Dinv[:, :, c] = np.linalg.inv(D.transpose(2, 0, 1)).transpose(1, 2, 0)
        self.Dinv = Dinv
        return self

    def solve(self, r):
        b = self.J.shape[1]
        n = int(np.prod(self.shape))
        Jf, Dinv = self.J, self.Dinv
        rf = r.reshape(b, n)
        y = np.zeros((b, n))
        for s in range(self.nlev):                     # (I + L_A D~^-1) y = r
            c = self.cells_at[s]
            t = rf[:, c].copy()
            for a in range(3):
                m = self.has_lo[a][c]
                cc = c[m]
                if len(cc) == 0:
                    continue
                nb = cc - self.strides[a]
                t[:, m] -= np.einsum("ijn,jkn,kn->in", Jf[1 + 2 * a][:, :, cc], Dinv[:, :, nb], y[:, nb])
            y[:, c] = t
        x = np.zeros((b, n))
        for s in range(self.nlev - 1, -1, -1):         # (D~ + U_A) x = y
            c = self.cells_at[s]
            t = y[:, c].copy()
            for a in range(3):
                m = self.has_hi[a][c]
                cc = c[m]
                if len(cc) == 0:
                    continue
                nb = cc + self.strides[a]
                t[:, m] -= np.einsum("ijn,jn->in", Jf[2 + 2 * a][:, :, cc], x[:, nb])
            x[:, c] = np.einsum("ijn,jn->in", Dinv[:, :, c], t)
        return x.reshape(r.shape)


class TiledILU1(TiledILU0):
    """Block-Jacobi over box tiles, block-ILU(1) in natural order inside each tile
    (sub_1_sub_pc_factor_levels 1: twophase.py:665-666 pc_cprilu1_gmres).

    Level-1 fill on the 7-point cell graph in natural order: a lower neighbour k of cell c and an upper neighbour
    j of k give a new entry (c, j) when j is not already coupled to c.  The factor pattern of a row is therefore
    the 13 offsets below (6 lower, diagonal, 6 upper; (d0, d1, d2) along axes 0, 1, 2), the numeric phase is the
    IKJ elimination restricted to that pattern (what PETSc's MatLUFactorNumeric does on the symbolic ILU(k)
    pattern), blocks are dense b x b, and entries that leave the tile do not exist.  M = L U with L unit lower.
    Cells with equal l0 + 2*l1 + 4*l2 (tile-local coordinates) are mutually independent, all rows a row
    depends on have a smaller value: that is the sweep order here and on the GPU."""
    LOWER = [(0, 0, -1), (1, 0, -1), (0, 1, -1), (0, -1, 0), (1, -1, 0), (-1, 0, 0)]     # increasing global index
    UPPER = [(1, 0, 0), (-1, 1, 0), (0, 1, 0), (0, -1, 1), (-1, 0, 1), (0, 0, 1)]
    SLOT = {(-1, 0, 0): 1, (1, 0, 0): 2, (0, -1, 0): 3, (0, 1, 0): 4, (0, 0, -1): 5, (0, 0, 1): 6, (0, 0, 0): 0}

    def __init__(self, shape, tile, slabs=None):
        super().__init__(shape, tile, slabs)
        self.level = (self.l[0] + 2 * self.l[1] + 4 * self.l[2]).reshape(-1)
        self.nlev = int(self.level.max()) + 1
        order = np.argsort(self.level, kind="stable")
        bounds = np.searchsorted(self.level[order], np.arange(self.nlev + 1))
        self.cells_at = [order[bounds[s]:bounds[s + 1]] for s in range(self.nlev)]
        lf = [x.reshape(-1) for x in self.l]
        td = [x.reshape(-1) for x in self.tdim]
        self.inside = {}
        for o in self.LOWER + self.UPPER + [(0, 0, 0)]:
            m = np.ones(lf[0].shape, dtype=bool)
            for a in range(3):
                m &= (lf[a] + o[a] >= 0) & (lf[a] + o[a] < td[a])
            self.inside[o] = m
        self.off = {o: o[0] * self.strides[0] + o[1] * self.strides[1] + o[2] * self.strides[2] for o in self.inside}

    def factor(self, J):
        b = J.shape[1]
        n = int(np.prod(self.shape))
        Jf = J.reshape(7, b, b, n)
        F = {}
        for o in self.inside:
            F[o] = np.zeros((b, b, n))
            if o in self.SLOT:
                F[o][:, :, self.inside[o]] = Jf[self.SLOT[o]][:, :, self.inside[o]]
        Dinv = np.zeros((b, b, n))
        for s in range(self.nlev):
            c = self.cells_at[s]
            for ok in self.LOWER:
                cc = c[self.inside[ok][c]]
                if len(cc) == 0:
                    continue
                k = cc + self.off[ok]
                Lck = np.einsum("ijn,jkn->ikn", F[ok][:, :, cc], Dinv[:, :, k])
                F[ok][:, :, cc] = Lck
                for oj in self.UPPER:
                    o = (ok[0] + oj[0], ok[1] + oj[1], ok[2] + oj[2])
                    if o in F:            # (k, j) entries that leave the tile are stored as zeros
                        F[o][:, :, cc] -= np.einsum("ijn,jkn->ikn", Lck, F[oj][:, :, k])
            Dinv[:, :, c] = np.linalg.inv(F[(0, 0, 0)][:, :, c].transpose(2, 0, 1)).transpose(1, 2, 0)
        self.F, self.Dinv = F, Dinv
        return self

    def solve(self, r):
        b = self.Dinv.shape[0]
        n = int(np.prod(self.shape))
        F, Dinv = self.F, self.Dinv
        rf = r.reshape(b, n)
        y = np.zeros((b, n))
        for s in range(self.nlev):                     # L y = r
            c = self.cells_at[s]
            t = rf[:, c].copy()
            for o in self.LOWER:
                m = self.inside[o][c]
                cc = c[m]
                if len(cc):
                    t[:, m] -= np.einsum("ijn,jn->in", F[o][:, :, cc], y[:, cc + self.off[o]])
            y[:, c] = t
        x = np.zeros((b, n))
        for s in range(self.nlev - 1, -1, -1):         # U x = y
            c = self.cells_at[s]
            t = y[:, c].copy()
            for o in self.UPPER:
                m = self.inside[o][c]
                cc = c[m]
                if len(cc):
                    t[:, m] -= np.einsum("ijn,jn->in", F[o][:, :, cc], x[:, cc + self.off[o]])
            x[:, c] = np.einsum("ijn,jn->in", Dinv[:, :, c], t)
        return x.reshape(r.shape)


# ------------------------------------------------------------------ semicoarsening AMG (7-point on every level)
def _axsl(ax, sl, ndim=3):
    s = [slice(None)] * ndim
    s[ax] = sl
    return tuple(s)


class SemiAMG:
    """Structured semicoarsening AMG in the spirit of hypre's PFMG with non-Galerkin 7-point
    coarse operators (the reference's v_cycle is hypre BoomerAMG, singlephase.py:303-307, which
    cannot be reproduced; this is the build's own pressure AMG).

    Level l -> l+1 along internal axis a: C points = even indices along a.  An F point g between
    C points interpolates with operator weights  w-(g) = -a_-(g)/c(g), w+(g) = -a_+(g)/c(g),
    c(g) = a_0(g) + sum of its cross-axis off-diagonals (stencil collapsed onto the a-line).
    R = P^T.  Coarse operator at C point f (neighbours g- = f-a, g+ = f+a):
        A_c[-a] = a_-(f) w-(g-)          A_c[+a] = a_+(f) w+(g+)
        A_c[d]  = a_d(f) + w+(g-) a_d(g-) + w-(g+) a_d(g+)          (cross slots d)
        A_c[0]  = -sum(off-diagonals of the coarse row) + rho(f) + w+(g-) rho(g-) + w-(g+) rho(g+)
    with rho = fine row sums (the accumulation part), so M-matrix structure and the zero-order
    term are kept and every level stays a 7-point stencil.  V(nu,nu) cycle, damped Jacobi.
    """

    def __init__(self, n, strength, omega=0.8, min_cells=64, nu=1, max_levels=40, full_levels=99, coarse_pre=None,
                 coarse_post=None, single=False, tail_post=None, mid_skip=False, dom_tau=0.0):
        """V(nu,nu) on the first `full_levels` levels, V(coarse_pre, coarse_post) below (the coarse levels of
        the GPU cycle are launch-latency bound: dropping their pre-smoothing costs no Krylov iterations)."""
        self.n = tuple(n)
        self.omega, self.nu = omega, nu
        self.single = bool(single)      # operators / weights / inverse diagonals stored in fp32 (GPU amg_single)
        self.full_levels = full_levels
        self.coarse_pre = nu if coarse_pre is None else coarse_pre
        self.coarse_post = nu if coarse_post is None else coarse_post
        # levels of <= 1024 cells (the GPU's single-workgroup tail) may smooth more: V(coarse_pre, tail_post)
        self.tail_post = self.coarse_post if tail_post is None else tail_post
        # every second level between the full ones and the <= 1024-cell ones is a pure transfer level
        self.mid_skip = bool(mid_skip)
        # Relaxation-only truncation: the first of the V(nu,nu) levels whose operator is strongly diagonally dominant in
        # every row (max_i sum_{j != i} |a_ij| / |a_ii| <= dom_tau) ends the cycle with two damped-Jacobi sweeps from a
        # zero guess -- damped Jacobi contracts the error there by <= 1 - omega (1 - dom_tau) per sweep, so a coarse-grid
        # correction has nothing left to do.  (BoomerAMG's own rule for such rows, -pc_hypre_boomeramg_max_row_sum 0.9:
        # rows dominated by their diagonal have no strong connections and are not coarsened.)  This is the temperature
        # operator S~ of pc_cptr at every time step of the BASELINE configurations (ratio 0.03-0.14); the pressure
        # operator has ratio 1 and is never truncated.  0 disables.
        self.dom_tau = float(dom_tau)
        self.trunc = None
        self.sched = self._schedule(n, strength, min_cells, max_levels)

    @staticmethod
    def _schedule(n, strength, min_cells, max_levels):
        n = list(n)
        s = [float(v) if n[a] > 1 else -1.0 for a, v in enumerate(strength)]
        sched = []
        while n[0] * n[1] * n[2] > min_cells and len(sched) < max_levels:
            cand = [a for a in range(3) if n[a] > 1]
            if not cand:
                break
            a = max(cand, key=lambda q: (s[q], -q))
            sched.append(a)
            n[a] = (n[a] + 1) // 2
            for q in range(3):
                s[q] = s[q] * 0.5 if q == a else s[q] * 2.0
        return sched

    @staticmethod
    def weights(A, a):
        """Interpolation weights (w-, w+) at every cell (only odd cells along a are used)."""
        cross = [s for s in range(1, 7) if (s - 1) // 2 != a]
        c = A[0] + sum(A[s] for s in cross)
        return -A[1 + 2 * a] / c, -A[2 + 2 * a] / c

    @staticmethod
    def coarsen(A, a, w=None):
        ax = 2 - a
        n = A.shape[1 + ax]
        nc = (n + 1) // 2
        wm, wp = w if w is not None else SemiAMG.weights(A, a)
        rho = A.sum(axis=0)
        ev = _axsl(ax, slice(0, n, 2))

        def nb(x, side):
            """value of x at the F neighbour g- (side=-1) / g+ (side=+1) of each C point, 0 if none."""
            out = np.zeros(x[ev].shape)
            if side < 0:
                out[_axsl(ax, slice(1, None))] = x[_axsl(ax, slice(1, n, 2))][_axsl(ax, slice(0, nc - 1))]
            else:
                src = x[_axsl(ax, slice(1, n, 2))]
                out[_axsl(ax, slice(0, src.shape[ax]))] = src
            return out
        Pm = nb(wp, -1)      # P[g-, I] = w+(g-)
        Pp = nb(wm, +1)      # P[g+, I] = w-(g+)
        Ac = np.zeros((7,) + A[0][ev].shape)
        lo_s, hi_s = 1 + 2 * a, 2 + 2 * a
        Ac[lo_s] = A[lo_s][ev] * nb(wm, -1)
        Ac[hi_s] = A[hi_s][ev] * nb(wp, +1)
        for s in range(1, 7):
            if s in (lo_s, hi_s):
                continue
            Ac[s] = A[s][ev] + Pm * nb(A[s], -1) + Pp * nb(A[s], +1)
        Ac[0] = -Ac[1:].sum(axis=0) + rho[ev] + Pm * nb(rho, -1) + Pp * nb(rho, +1)
        return Ac, (wm, wp)

    def _store(self, x):
        return x.astype(np.float32).astype(np.float64) if self.single else x

    def setup(self, A):
        self.levels = [self._store(A)]
        self.W = []
        for a in self.sched:
            A_l = self.levels[-1]
            w = tuple(self._store(v) for v in self.weights(A_l, a))
            Ac, _ = self.coarsen(A_l, a, w)
            self.levels.append(self._store(Ac))
            self.W.append(w)
        self.invd = [self._store(self.omega / l[0]) for l in self.levels]
        self.trunc = None
        if self.dom_tau > 0.0:
            for l in range(min(self.full_levels, len(self.levels) - 1)):
                A_l = self.levels[l]
                if float((np.abs(A_l[1:]).sum(axis=0) / np.abs(A_l[0])).max()) <= self.dom_tau:
                    self.trunc = l
                    break
        import scipy.sparse.linalg as spla
        M = to_csr(self.levels[-1][:, None, None])
        self.coarse = spla.splu(M.tocsc()) if M.shape[0] > 1 else None
        self.coarse_scalar = M[0, 0] if M.shape[0] == 1 else None
        return self

    def opcomplexity(self):
        return sum(np.prod(l.shape[1:]) for l in self.levels) / np.prod(self.levels[0].shape[1:])

    def restrict(self, r, lvl):
        a = self.sched[lvl]
        ax = 2 - a
        n = r.shape[ax]
        wm, wp = self.W[lvl]
        rc = r[_axsl(ax, slice(0, n, 2))].copy()
        odd = _axsl(ax, slice(1, n, 2))
        t = wm[odd] * r[odd]                      # F point g contributes w-(g) r(g) to its left C
        rc[_axsl(ax, slice(0, t.shape[ax]))] += t
        t = wp[odd] * r[odd]                      # and w+(g) r(g) to its right C (if any)
        k = min(t.shape[ax], rc.shape[ax] - 1)
        rc[_axsl(ax, slice(1, 1 + k))] += t[_axsl(ax, slice(0, k))]
        return rc

    def prolong(self, ec, lvl, shape):
        a = self.sched[lvl]
        ax = 2 - a
        n = shape[ax]
        wm, wp = self.W[lvl]
        e = np.zeros(shape)
        e[_axsl(ax, slice(0, n, 2))] = ec
        odd = _axsl(ax, slice(1, n, 2))
        no = e[odd].shape[ax]
        left = ec[_axsl(ax, slice(0, no))]
        right = np.zeros(left.shape)
        k = min(no, ec.shape[ax] - 1)
        right[_axsl(ax, slice(0, k))] = ec[_axsl(ax, slice(1, 1 + k))]
        e[odd] = wm[odd] * left + wp[odd] * right
        return e

    def _smooth(self, lvl, b, x):
        return x + self.invd[lvl] * (b - spmv_scalar(self.levels[lvl], x))

    def vcycle(self, b, lvl=0):
        A = self.levels[lvl]
        if self.trunc is not None and lvl == self.trunc:      # relaxation-only level (see dom_tau)
            return self._smooth(lvl, b, self.invd[lvl] * b)
        if lvl == len(self.levels) - 1:
            if self.coarse is None:
                return b / self.coarse_scalar
            return self.coarse.solve(b.reshape(-1)).reshape(b.shape)
        if lvl < self.full_levels:
            pre, post = self.nu, self.nu
        else:
            pre, post = self.coarse_pre, (self.tail_post if b.size <= 1024 else self.coarse_post)
            if self.mid_skip and b.size > 1024 and (lvl - self.full_levels) % 2 == 1:
                pre, post = 0, 0
        if pre == 0:
            x, r = np.zeros_like(b), b
        else:
            x = self.invd[lvl] * b
            for _ in range(pre - 1):
                x = self._smooth(lvl, b, x)
            r = b - spmv_scalar(A, x)
        ec = self.vcycle(self.restrict(r, lvl), lvl + 1)
        x = x + self.prolong(ec, lvl, b.shape)
        for _ in range(post):
            x = self._smooth(lvl, b, x)
        return x


class BlockSemiAMG(SemiAMG):
    """System version of SemiAMG for the interleaved (p,T) operator of pc_cptramg[_QI|_TI] (twophase.py:552-566:
    hypre BoomerAMG on Atilde_00 with the VectorFunctionSpace layout, ``vector=True`` :935-955).  hypre cannot be
    reproduced; this is the build's own "unknown-based" system AMG on the same semicoarsening grids:
      * every stencil entry is an nb x nb block, arrays (7, nb, nb, n2, n1, n0); vectors (nb, n2, n1, n0);
      * interpolation per unknown q from its own diagonal block A^{qq} (weights w-_q, w+_q as in SemiAMG);  R = P^T;
      * coarse block (q, r): rows combined with the restriction weights of q, columns along the coarsening axis
        interpolated with the weights of r, the same lumped non-Galerkin 7-point formula block by block;
      * smoother: damped block-Jacobi with the nb x nb diagonal blocks; dense solve on the coarsest grid."""

    def __init__(self, n, strength, nb=2, **kw):
        kw = dict(kw, dom_tau=0.0)            # the (p,T) system contains the pressure rows: never relaxation-only
        super().__init__(n, strength, **kw)
        self.nb = nb

    def bweights(self, A, a):
        cross = [s for s in range(1, 7) if (s - 1) // 2 != a]
        wm, wp = [], []
        for q in range(self.nb):
            c = A[0, q, q] + sum(A[s, q, q] for s in cross)
            wm.append(-A[1 + 2 * a, q, q] / c)
            wp.append(-A[2 + 2 * a, q, q] / c)
        return np.array(wm), np.array(wp)

    def bcoarsen(self, A, a, w):
        ax = 2 - a
        n = A.shape[3 + ax]
        nc = (n + 1) // 2
        wm, wp = w
        ev = _axsl(ax, slice(0, n, 2))

        def nb_(x, side):
            out = np.zeros(x[ev].shape)
            if side < 0:
                out[_axsl(ax, slice(1, None))] = x[_axsl(ax, slice(1, n, 2))][_axsl(ax, slice(0, nc - 1))]
            else:
                src = x[_axsl(ax, slice(1, n, 2))]
                out[_axsl(ax, slice(0, src.shape[ax]))] = src
            return out
        lo_s, hi_s = 1 + 2 * a, 2 + 2 * a
        Ac = np.zeros((7, self.nb, self.nb) + A[0, 0, 0][ev].shape)
        for q in range(self.nb):
            Pm, Pp = nb_(wp[q], -1), nb_(wm[q], +1)          # row restriction weights of unknown q
            for r in range(self.nb):
                B = A[:, q, r]
                rho = B.sum(axis=0)
                Ac[lo_s, q, r] = B[lo_s][ev] * nb_(wm[r], -1)    # column interpolation weights of unknown r
                Ac[hi_s, q, r] = B[hi_s][ev] * nb_(wp[r], +1)
                for s in range(1, 7):
                    if s in (lo_s, hi_s):
                        continue
                    Ac[s, q, r] = B[s][ev] + Pm * nb_(B[s], -1) + Pp * nb_(B[s], +1)
                Ac[0, q, r] = -Ac[1:, q, r].sum(axis=0) + rho[ev] + Pm * nb_(rho, -1) + Pp * nb_(rho, +1)
        return Ac

    def setup(self, A):
        self.levels = [self._store(A)]
        self.W = []
        for a in self.sched:
            A_l = self.levels[-1]
            w = tuple(self._store(v) for v in self.bweights(A_l, a))
            self.levels.append(self._store(self.bcoarsen(A_l, a, w)))
            self.W.append(w)
        self.invD = []
        for l in self.levels:
            D = l[0].transpose(2, 3, 4, 0, 1)                       # (..., nb, nb)
            self.invD.append(self._store(self.omega * np.linalg.inv(D).transpose(3, 4, 0, 1, 2)))
        import scipy.sparse.linalg as spla
        self.coarse = spla.splu(to_csr(self.levels[-1]).tocsc())
        return self

    def _bsmooth(self, lvl, b, x):
        r = b - spmv_block(self.levels[lvl], x)
        return x + np.einsum("qr...,r...->q...", self.invD[lvl], r)

    def _each(self, fn, v, lvl, *a):
        saveW = self.W[lvl]
        out = []
        for q in range(self.nb):
            self.W[lvl] = (saveW[0][q], saveW[1][q])
            out.append(fn(v[q], lvl, *a))
        self.W[lvl] = saveW
        return np.array(out)

    def vcycle(self, b, lvl=0):
        if lvl == len(self.levels) - 1:
            nbk = b.shape[0]
            xf = self.coarse.solve(b.reshape(nbk, -1).T.reshape(-1))       # cell-interleaved like to_csr
            return xf.reshape(-1, nbk).T.reshape(b.shape)
        ncell = b[0].size
        if lvl < self.full_levels:
            pre, post = self.nu, self.nu
        else:
            pre, post = self.coarse_pre, (self.tail_post if ncell <= 1024 else self.coarse_post)
            if self.mid_skip and ncell > 1024 and (lvl - self.full_levels) % 2 == 1:
                pre, post = 0, 0
        if pre == 0:
            x, r = np.zeros_like(b), b
        else:
            x = np.einsum("qr...,r...->q...", self.invD[lvl], b)
            for _ in range(pre - 1):
                x = self._bsmooth(lvl, b, x)
            r = b - spmv_block(self.levels[lvl], x)
        ec = self.vcycle(self._each(self.restrict, r, lvl), lvl + 1)
        x = x + self._each(self.prolong, ec, lvl, b.shape[1:])
        for _ in range(post):
            x = self._bsmooth(lvl, b, x)
        return x


# ------------------------------------------------------------------ stage 1 (CPR / CPTR)
def decouple(J, kind, primary):
    """Atilde = A_00 - D_0s D_ss^-1 A_s0 on the cell-interleaved stencil.

    s = LAST field (preconditioners.py:358,1408).  QI: D = diagonal entries (:785-808,:1505-1543);
    TI: column sums (:684-711,:1445-1503).  Returns (Atilde [7,len(primary),len(primary),...], d)
    with d[q] = D_qs/D_ss per cell, used by apply as r_q = x_q - d[q] x_s (:894-895,:1559-1560)."""
    b = J.shape[1]
    s = b - 1
    shape = J.shape[3:]
    if kind == "No":
        return J[:, primary][:, :, primary].copy(), None
    if kind in ("QI_temp", "TI_temp"):
        return _decouple_temp(J, kind, primary)
    if kind == "QI":
        Dss = J[0, s, s]
        D0s = [J[0, q, s] for q in primary]
    elif kind == "TI":
        ones = np.ones(shape)

        def colsum(q):      # column sums of block (q, s): sum over rows i of A_qs[i, j]  == (A^T 1)_j
            AT = J[:, q, s]
            out = AT[0].copy()
            for a in range(3):
                if shape[2 - a] == 1:
                    continue
                lo, hi = _lo(a), _hi(a)
                out[hi] += AT[2 + 2 * a][lo]      # row lo has entry in column hi
                out[lo] += AT[1 + 2 * a][hi]
            return out
        Dss = colsum(s)
        D0s = [colsum(q) for q in primary]
    else:
        raise ValueError("unknown decoupling " + str(kind))
    d = [D / Dss for D in D0s]
    At = np.zeros((7, len(primary), len(primary)) + shape)
    for i, q in enumerate(primary):
        for j, c in enumerate(primary):
            At[:, i, j] = J[:, q, c] - d[i][None] * J[:, s, c]
    return At, d


def _decouple_temp(J, kind, primary):
    """QI_temp / TI_temp (preconditioners.py:714-783, 810-873): two-phase CPR where BOTH non-pressure fields
    (T, S) are decoupled: per cell D_ss is the 2x2 block [[D_TT, D_TS], [D_ST, D_SS]] and D_ps = [D_pT, D_pS]
    (diagonal entries for QI_temp, column sums for TI_temp);  Atilde_pp = A_pp - D_ps D_ss^-1 A_sp  and
    r_p = x_p - D_ps D_ss^-1 x_s.  Returns (Atilde [7,1,1,...], d) with d[0] = (d_T, d_S) per cell."""
    assert J.shape[1] == 3 and list(primary) == [0], "the _temp decouplings are two-phase pressure-only variants"
    shape = J.shape[3:]

    def entry(q, c):
        if kind == "QI_temp":
            return J[0, q, c]
        AT = J[:, q, c]
        out = AT[0].copy()
        for a in range(3):
            if shape[2 - a] == 1:
                continue
            lo, hi = _lo(a), _hi(a)
            out[hi] += AT[2 + 2 * a][lo]
            out[lo] += AT[1 + 2 * a][hi]
        return out
    DTT, DTS, DST, DSS = entry(1, 1), entry(1, 2), entry(2, 1), entry(2, 2)
    DpT, DpS = entry(0, 1), entry(0, 2)
    det = DTT * DSS - DTS * DST
    dT = (DpT * DSS - DpS * DST) / det          # [DpT DpS] . inv([[DTT DTS],[DST DSS]])
    dS = (DpS * DTT - DpT * DTS) / det
    At = np.zeros((7, 1, 1) + shape)
    At[:, 0, 0] = J[:, 0, 0] - dT[None] * J[:, 1, 0] - dS[None] * J[:, 2, 0]
    return At, [(dT, dS)]


class SelfpSchur:
    """K(S) for pc_fieldsplit_schur_precondition selfp (pc_fieldsplit_selfp, singlephase.py:322-330): one V-cycle on
    Sp = A11 - A10 diag(A00)^-1 A01, which PETSc forms explicitly (a 13-point operator in 2-D, 25-point in 3-D).

    Here the hierarchy is the 7-point semicoarsening AMG of Sp's collapse S7 -- the 7-point entries of Sp exactly, the far
    entries (c -> m -> j with j neither c nor a stencil neighbour of c) lumped onto the diagonal (row sums preserved) --
    followed by one damped-Jacobi sweep on the EXACT Sp, applied matrix-free (three 7-point products and a diagonal
    scaling) with its exact diagonal D:
        x = V7(b) ;  x += w D^-1 (b - Sp x).
    (Sp is indefinite on hard states -- diag(A00)^-1 is a poor stand-in for A00^-1 near wells at large dt -- and a
    pre-smoothing sweep from the zero guess, x = w D^-1 b, then makes FGMRES stall; measured on C2-like cases, where
    this form stays within 10 % of an exact Sp solve.)"""

    def __init__(self, amg, omega):
        self.amg, self.omega = amg, omega

    def setup(self, A00, A01, A10, A11):
        shape = A00.shape[1:]
        dinv = 1.0/A00[0]
        S7 = A11.copy()
        t0 = A10[0]*dinv
        S7[0] -= t0*A01[0]
        lump = np.zeros(shape)
        for s in range(1, 7):
            a = (s - 1)//2
            if shape[2 - a] == 1:
                continue
            odd = s % 2 == 1                              # odd slots look down the axis: cells _hi have that neighbour
            sc, sm = (_hi(a), _lo(a)) if odd else (_lo(a), _hi(a))
            opp = s + 1 if odd else s - 1
            S7[s][sc] -= t0[sc]*A01[s][sc]                 # c -> c -> m
            w = A10[s][sc]*dinv[sm]                        # c -> m
            S7[s][sc] -= w*A01[0][sm]                      # c -> m -> m
            S7[0][sc] -= w*A01[opp][sm]                    # c -> m -> c
            for t in range(1, 7):
                at = (t - 1)//2
                if t == opp or shape[2 - at] == 1:
                    continue
                # m's own boundary entries are zero by assembly; mask them anyway through the validity of m + off_t
                ok = np.zeros(shape, dtype=bool)
                ok[_hi(at) if t % 2 == 1 else _lo(at)] = True
                lump[sc] -= np.where(ok[sm], w*A01[t][sm], 0.0)
        self.diag = S7[0].copy()                           # diag(Sp), exact
        S7[0] += lump
        self.S7 = S7
        self.ops = (A01, A10, A11, dinv)
        self.invd = self.omega/self.diag
        self.amg.setup(S7)
        return self

    def mult(self, x):
        A01, A10, A11, dinv = self.ops
        return spmv_scalar(A11, x) - spmv_scalar(A10, dinv*spmv_scalar(A01, x))

    def vcycle(self, b):
        x = self.amg.vcycle(b)
        return x + self.invd*(b - self.mult(x))


def slab_ranges(n2, nslabs):
    """Planes [lo, hi) of internal axis 2 per slab (same rule as thermalporous_amd.engine.slab_range)."""
    base, rem = divmod(n2, nslabs)
    out = []
    for r in range(nslabs):
        lo = r * base + min(r, rem)
        out.append((lo, lo + base + (1 if r < rem else 0)))
    return out


class TwoStagePC:
    """Composite multiplicative PC: y = B1 x; r = x - J y; y += B2 r (PCCOMPOSITE multiplicative,
    singlephase.py:341-343).  B1 = CPR or CPTR stage 1, B2 = tiled block-ILU(0).
    opts["nslabs"] > 1 emulates the N-GPU algorithm in one process: the ILU tiles restart at every slab
    boundary (bjacobi across ranks, twophase.py:533: each GPU factors only its own rows); stage 1 is
    the same global operator as on one GPU (every rank runs the V-cycles on the gathered pressure system)."""

    def __init__(self, prob, opts):
        self.prob = prob
        self.o = opts
        n = prob.n
        shape = prob.shape
        self.slabs = slab_ranges(n[2], int(opts.get("nslabs", 1)))
        kw = dict(omega=opts["amg_omega"], min_cells=opts["amg_min_cells"], nu=opts["amg_nu"],
                  full_levels=opts.get("amg_full_levels", 99), coarse_pre=opts.get("amg_coarse_pre"),
                  coarse_post=opts.get("amg_coarse_post"), single=opts.get("amg_single", False),
                  tail_post=opts.get("amg_tail_post"), mid_skip=opts.get("amg_mid_skip", False),
                  dom_tau=opts.get("amg_dom_tau", 0.0))

        # coarsening schedule from the mean interior-face transmissibility per axis
        st = [float(np.mean(prob.TK[a][_lo(a)])) if n[a] > 1 else 0.0 for a in range(3)]
        self.amg_p = SemiAMG(n, st, **kw)
        self.amg_T = SemiAMG(n, [prob.G[a] if n[a] > 1 else 0.0 for a in range(3)], **kw) \
            if opts["pc"] in ("cptr", "fieldsplit_cd") else None
        # pc_cptramg: ONE system V-cycle on the (p,T) 2x2-block operator (coarsening schedule of the pressure)
        self.amg_pT = BlockSemiAMG(n, st, nb=2, **kw) if opts["pc"] == "cptramg" else None
        levels = int(opts.get("ilu_levels", 0))
        if levels not in (0, 1):
            raise ValueError("ilu_levels must be 0 or 1")
        self.ilu = (TiledILU1 if levels else TiledILU0)(shape, opts["ilu_tile"], self.slabs)
        self.vcycles = 0

    def setup(self, J, Sm=None):
        o = self.o
        self.J = J
        self.ilu.factor(J)
        if o["pc"] == "bilu":
            return                                   # pc_bilu (twophase.py:758-762, singlephase.py:402-406): stage 2 alone
        if o["pc"] == "cpr":
            At, self.d = decouple(J, o["decoup"], [0])
            self.amg_p.setup(At[:, 0, 0])
        elif o["pc"] == "cptr":
            # pc_cptr: decoup "No", fieldsplit Schur FULL on (p,T) with V(App), V(S~) (twophase.py:531-550)
            At, self.d = decouple(J, o["decoup"], [0, 1])
            self.At = At
            self.amg_p.setup(At[:, 0, 0])
            # K(S): the convection-diffusion operator S~, or (schur_precondition a11, twophase.py:598-616) A_11
            self.amg_T.setup(At[:, 1, 1] if o.get("schur_a11") else Sm)
        elif o["pc"] == "cptramg":
            # pc_cptramg[_QI|_TI] (twophase.py:552-566): CPTRStage1PC whose stage-1 solver is ONE AMG V-cycle on the
            # interleaved (p,T) system Atilde_00 (preconditioners.py:1505-1543)
            At, self.d = decouple(J, o["decoup"], [0, 1])
            self.At = At
            self.amg_pT.setup(At)
        elif o["pc"] == "fieldsplit_cd":
            # single-phase block preconditioner (singlephase.py:309-319): the same Schur FULL stage on the
            # undecoupled (p,T) system with the ConvDiffSchurPC operator (preconditioners.py:11-163); no stage 2
            assert J.shape[1] == 2 and o["decoup"] == "No"
            self.At, self.d = decouple(J, "No", [0, 1])
            self.amg_p.setup(self.At[:, 0, 0])
            if o.get("schur_selfp"):                       # (singlephase.py:322-330: selfp)
                At = self.At
                self.selfp = SelfpSchur(self.amg_T, o["amg_omega"]).setup(At[:, 0, 0], At[:, 0, 1], At[:, 1, 0], At[:, 1, 1])
            else:
                self.selfp = None
                self.amg_T.setup(self.At[:, 1, 1] if (o.get("schur_a11") or o.get("fs_additive")) else Sm)     # (singlephase.py:331-338: a11)
        else:
            raise ValueError(o["pc"])

    def stage1(self, x):
        y = np.zeros_like(x)
        o = self.o
        s = x.shape[0] - 1
        if o["pc"] == "cpr":
            if self.d is None:
                r = x[0]
            elif o["decoup"] in ("QI_temp", "TI_temp"):
                r = x[0] - self.d[0][0] * x[1] - self.d[0][1] * x[2]
            else:
                r = x[0] - self.d[0] * x[s]
            y[0] = self.amg_p.vcycle(r)
            self.vcycles += 1
        elif o["pc"] == "cptramg":
            r0 = x[0] if self.d is None else x[0] - self.d[0] * x[s]
            r1 = x[1] if self.d is None else x[1] - self.d[1] * x[s]
            y[:2] = self.amg_pT.vcycle(np.array([r0, r1]))
            self.vcycles += 1
        else:
            r0 = x[0] if self.d is None else x[0] - self.d[0] * x[s]
            r1 = x[1] if self.d is None else x[1] - self.d[1] * x[s]
            At = self.At
            if o.get("fs_additive"):
                # PCFIELDSPLIT additive (pc_fieldsplit_diag, singlephase.py:371-375): block-diagonal, one V-cycle per field
                y[0], y[1] = self.amg_p.vcycle(r0), self.amg_T.vcycle(r1)
                self.vcycles += 2
                return y
            # PCFIELDSPLIT schur FULL: y0 = K(A00) r0; y1 = K(S)(r1 - A10 y0); y0 = K(A00)(r0 - A01 y1)
            y0 = self.amg_p.vcycle(r0)
            KS = self.selfp.vcycle if getattr(self, "selfp", None) is not None else self.amg_T.vcycle
            y1 = KS(r1 - spmv_scalar(At[:, 1, 0], y0))
            y0 = self.amg_p.vcycle(r0 - spmv_scalar(At[:, 0, 1], y1))
            y[0], y[1] = y0, y1
            self.vcycles += 3
        return y

    def apply(self, x):
        if self.o["pc"] == "bilu":
            return self.ilu.solve(x)
        y = self.stage1(x)
        if self.o["pc"] == "fieldsplit_cd":
            return y
        r = x - spmv_block(self.J, y)
        return y + self.ilu.solve(r)


# ------------------------------------------------------------------ FGMRES
def fgmres(matvec, pc, b, rtol=1e-8, atol=1e-50, restart=200, maxit=200, dot=None):
    """Right-preconditioned flexible GMRES from x0 = 0 with classical Gram-Schmidt (PETSc default,
    no refinement) and Givens-rotated Hessenberg; convergence on the recurrence residual norm,
    ||r|| <= max(rtol*||b||, atol) (KSP default test).  Returns (x, its, reason, resnorms)."""
    dot = dot or (lambda u, v: float(np.vdot(u, v).real))
    x = np.zeros_like(b)
    bnorm = np.sqrt(dot(b, b))
    if bnorm == 0.0:
        return x, 0, 2, [0.0]
    tol = max(rtol * bnorm, atol)
    its = 0
    hist = [bnorm]
    r = b.copy()
    beta = bnorm
    while True:
        m = min(restart, maxit - its)
        V = [r / beta]
        Z = []
        H = np.zeros((m + 1, m))
        cs, sn = np.zeros(m), np.zeros(m)
        g = np.zeros(m + 1)
        g[0] = beta
        k = 0
        reason = 0
        for j in range(m):
            z = pc(V[j])
            w = matvec(z)
            Z.append(z)
            h = np.array([dot(V[i], w) for i in range(j + 1)])
            for i in range(j + 1):
                w = w - h[i] * V[i]
            hn = np.sqrt(dot(w, w))
            H[:j + 1, j] = h
            H[j + 1, j] = hn
            for i in range(j):
                t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]
                H[i + 1, j] = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]
                H[i, j] = t
            d = np.hypot(H[j, j], H[j + 1, j])
            cs[j], sn[j] = H[j, j] / d, H[j + 1, j] / d
            H[j, j] = d
            H[j + 1, j] = 0.0
            g[j + 1] = -sn[j] * g[j]
            g[j] = cs[j] * g[j]
            its += 1
            k = j + 1
            res = abs(g[j + 1])
            hist.append(res)
            if res <= tol:
                reason = 2          # KSP_CONVERGED_RTOL
                break
            if hn == 0.0:
                reason = 2
                break
            V.append(w / hn)
        yk = np.linalg.solve(np.triu(H[:k, :k]), g[:k]) if k else np.zeros(0)
        for i in range(k):
            x = x + yk[i] * Z[i]
        if reason:
            return x, its, reason, hist
        if its >= maxit:
            return x, its, -3, hist   # KSP_DIVERGED_ITS
        r = b - matvec(x)
        beta = np.sqrt(dot(r, r))
        if beta <= tol:
            return x, its, 2, hist

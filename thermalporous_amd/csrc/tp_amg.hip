// Structured semicoarsening AMG for the stage-1 pressure (and temperature) operators.
//
// Reference: the `v_cycle` dicts (singlephase.py:303-307, twophase.py:478-482) = one hypre BoomerAMG
// V-cycle per application, set up again at every Newton step (preconditioners.py:878).  hypre is not
// reproducible; this is the build's own AMG, designed for the machine: every level is a 7-point
// stencil on a box stored as 7 coalesced planes (no index arrays), set-up is one kernel per level
// (cheap enough to redo every Newton step), coarsening direction per level is decided on the host
// from mean face couplings (no device sync).  Algorithm (mirrored by oracle/linalg.py:SemiAMG):
//   C points = even indices along the level's axis a;  F point g:  w-(g) = -a_-(g)/c(g),
//   w+(g) = -a_+(g)/c(g),  c(g) = a_0(g) + sum of cross-axis off-diagonals;  R = P^T;
//   coarse row at C point f (F neighbours g-, g+):
//     A_c[-a] = a_-(f) w-(g-),  A_c[+a] = a_+(f) w+(g+),
//     A_c[d]  = a_d(f) + w+(g-) a_d(g-) + w-(g+) a_d(g+)            (cross slots),
//     A_c[0]  = -sum(off-diagonals) + rho(f) + w+(g-) rho(g-) + w-(g+) rho(g+),  rho = row sums.
//   V(nu,nu), damped Jacobi, dense inverse on the coarsest grid.
#include "tp_common.hpp"
#include <algorithm>

namespace tp {

static inline dim3 grid_for(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

__device__ __forceinline__ void cell_ijk(const GridDev &g, long tid, int &i0, int &i1, int &i2) {
    i2 = (int)(tid / g.np);
    const int rem = (int)(tid - (long)i2 * g.np);
    i1 = rem / g.n0;
    i0 = rem - i1 * g.n0;
}

// interpolation weights of every cell w.r.t. axis a (only odd cells are used) + invd = omega/diag
__global__ void k_amg_weights(GridDev g, Stencil A, int axis, double omega, double *wm, double *wp, double *invd) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid;
    const double a0 = A.slot(0)[c];
    invd[c] = omega / a0;
    if (axis < 0) return;
    double cc = a0;
#pragma unroll
    for (int s = 1; s < 7; ++s)
        if ((s - 1) / 2 != axis) cc += A.slot(s)[c];
    wm[c] = -A.slot(1 + 2 * axis)[c] / cc;
    wp[c] = -A.slot(2 + 2 * axis)[c] / cc;
}

// coarse operator: one thread per coarse cell
__global__ void k_amg_coarsen(GridDev gf, GridDev gc, Stencil A, int axis, const double *wm, const double *wp,
                              double *Ac) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= gc.nown) return;
    int I[3];
    cell_ijk(gc, tid, I[0], I[1], I[2]);
    const long cc = gc.np + tid;
    int F[3] = {I[0], I[1], I[2]};
    F[axis] = 2 * I[axis];
    const int nfa = axis == 0 ? gf.n0 : (axis == 1 ? gf.n1 : gf.n2);
    const long stride = axis == 0 ? 1 : (axis == 1 ? gf.n0 : gf.np);
    const long f = gf.np + (long)F[0] + (long)gf.n0 * F[1] + gf.np * F[2];
    const bool hm = F[axis] - 1 >= 0, hp = F[axis] + 1 < nfa;
    const long gm = f - stride, gp = f + stride;
    const double Pm = hm ? wp[gm] : 0.0;     // P[g-, I] = w+(g-)
    const double Pp = hp ? wm[gp] : 0.0;     // P[g+, I] = w-(g+)
    double rho_f = 0.0, rho_m = 0.0, rho_p = 0.0;
    double out[7];
    double offsum = 0.0;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const double af = A.slot(s)[f];
        const double am = hm ? A.slot(s)[gm] : 0.0;
        const double ap = hp ? A.slot(s)[gp] : 0.0;
        rho_f += af; rho_m += am; rho_p += ap;
        if (s == 0) continue;
        double v;
        if (s == 1 + 2 * axis)      v = af * (hm ? wm[gm] : 0.0);
        else if (s == 2 + 2 * axis) v = af * (hp ? wp[gp] : 0.0);
        else                        v = af + Pm * am + Pp * ap;
        out[s] = v;
        offsum += v;
    }
    out[0] = -offsum + rho_f + Pm * rho_m + Pp * rho_p;
#pragma unroll
    for (int s = 0; s < 7; ++s) Ac[(long)s * gc.ntot + cc] = out[s];
}

// x = invd * b  (first Jacobi sweep from a zero guess)
__global__ void k_amg_jacobi0(GridDev g, const double *invd, const double *b, double *x) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid;
    x[c] = invd[c] * b[c];
}

// xout = x + invd * (b - A x)
__global__ __launch_bounds__(256) void k_amg_jacobi(GridDev g, Stencil A, const double *__restrict__ invd,
                                                    const double *__restrict__ b, const double *__restrict__ x,
                                                    double *__restrict__ xout) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid;
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += A.slot(k)[c] * x[c + off[k]];
    xout[c] = x[c] + invd[c] * (b[c] - s);
}

// r = b - A x on the fine level, written to r (plain residual)
__global__ __launch_bounds__(256) void k_amg_resid(GridDev g, Stencil A, const double *__restrict__ b,
                                                   const double *__restrict__ x, double *__restrict__ r) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid;
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += A.slot(k)[c] * x[c + off[k]];
    r[c] = b[c] - s;
}

// rc = P^T r : one thread per coarse cell
__global__ void k_amg_restrict(GridDev gf, GridDev gc, int axis, const double *wm, const double *wp,
                               const double *r, double *rc) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= gc.nown) return;
    int I[3];
    cell_ijk(gc, tid, I[0], I[1], I[2]);
    int F[3] = {I[0], I[1], I[2]};
    F[axis] = 2 * I[axis];
    const int nfa = axis == 0 ? gf.n0 : (axis == 1 ? gf.n1 : gf.n2);
    const long stride = axis == 0 ? 1 : (axis == 1 ? gf.n0 : gf.np);
    const long f = gf.np + (long)F[0] + (long)gf.n0 * F[1] + gf.np * F[2];
    double v = r[f];
    if (F[axis] - 1 >= 0) v += wp[f - stride] * r[f - stride];
    if (F[axis] + 1 < nfa) v += wm[f + stride] * r[f + stride];
    rc[gc.np + tid] = v;
}

// x += P ec : one thread per fine cell
__global__ void k_amg_prolong_add(GridDev gf, GridDev gc, int axis, const double *wm, const double *wp,
                                  const double *ec, double *x) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= gf.nown) return;
    int F[3];
    cell_ijk(gf, tid, F[0], F[1], F[2]);
    const long c = gf.np + tid;
    int I[3] = {F[0], F[1], F[2]};
    I[axis] = F[axis] >> 1;
    const long ci = gc.np + (long)I[0] + (long)gc.n0 * I[1] + gc.np * I[2];
    const long cstride = axis == 0 ? 1 : (axis == 1 ? gc.n0 : gc.np);
    const int nca = axis == 0 ? gc.n0 : (axis == 1 ? gc.n1 : gc.n2);
    double e;
    if ((F[axis] & 1) == 0) e = ec[ci];
    else {
        e = wm[c] * ec[ci];
        if (I[axis] + 1 < nca) e += wp[c] * ec[ci + cstride];
    }
    x[c] += e;
}

// coarsest grid: dense inverse by Gauss-Jordan in one workgroup (n <= 64 ... a few hundred)
__global__ void k_amg_dense_inverse(GridDev g, Stencil A, int n, double *M, double *Minv) {
    // build dense M from the stencil
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) { M[e] = 0.0; Minv[e] = (e / n == e % n) ? 1.0 : 0.0; }
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        const long c = g.np + r;
        int i0, i1, i2;
        cell_ijk(g, r, i0, i1, i2);
        const bool has[7] = {true, i0 > 0, i0 < g.n0 - 1, i1 > 0, i1 < g.n1 - 1, i2 > 0, i2 < g.n2 - 1};
        for (int s = 0; s < 7; ++s)
            if (has[s]) M[(long)r * n + (r + off[s])] += A.slot(s)[c];
    }
    __syncthreads();
    for (int p = 0; p < n; ++p) {
        const double piv = M[(long)p * n + p];
        __syncthreads();
        for (int e = threadIdx.x; e < n; e += blockDim.x) {
            M[(long)p * n + e] /= piv;
            Minv[(long)p * n + e] /= piv;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
            const int r = e / n, q = e % n;
            if (r == p) continue;
            const double fct = M[(long)r * n + p];
            // column p of M must stay readable for every q: update Minv fully, M for q != p
            Minv[(long)r * n + q] -= fct * Minv[(long)p * n + q];
            if (q != p) M[(long)r * n + q] -= fct * M[(long)p * n + q];
        }
        __syncthreads();
        for (int r = threadIdx.x; r < n; r += blockDim.x)
            if (r != p) M[(long)r * n + p] = 0.0;
        __syncthreads();
    }
}

__global__ void k_amg_dense_apply(GridDev g, int n, const double *Minv, const double *b, double *x) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double s = 0.0;
    for (int q = 0; q < n; ++q) s += Minv[(long)r * n + q] * b[g.np + q];
    x[g.np + r] = s;
}

// ---- host side ------------------------------------------------------------------------------------
static std::vector<int> schedule(const int n_[3], const double strength[3], int min_cells) {
    int n[3] = {n_[0], n_[1], n_[2]};
    double s[3];
    for (int a = 0; a < 3; ++a) s[a] = n[a] > 1 ? strength[a] : -1.0;
    std::vector<int> sched;
    while ((long)n[0] * n[1] * n[2] > min_cells && sched.size() < 40) {
        int best = -1;
        for (int a = 0; a < 3; ++a)
            if (n[a] > 1 && (best < 0 || s[a] > s[best])) best = a;
        if (best < 0) break;
        sched.push_back(best);
        n[best] = (n[best] + 1) / 2;
        for (int q = 0; q < 3; ++q) s[q] = (q == best) ? s[q] * 0.5 : s[q] * 2.0;
    }
    return sched;
}

void amg_build(tp_ctx *c, Amg *&amg, const double strength[3]) {
    delete amg;
    amg = new Amg();
    const int n[3] = {c->g.n0, c->g.n1, c->g.n2};
    amg->sched = schedule(n, strength, std::max(1, c->opt.amg_min_cells));
    int m[3] = {n[0], n[1], n[2]};
    for (size_t l = 0; l <= amg->sched.size(); ++l) {
        AmgLevel *L = new AmgLevel();
        // coarse levels are local boxes: no live halos (multi-GPU AMG is block-Jacobi per slab)
        L->g = make_grid(m[0], m[1], m[2], m[2], 0);
        const size_t nt = (size_t)L->g.ntot;
        if (l > 0) {
            L->A.alloc(7 * nt);
            L->op.base = L->A.p;
            L->op.slot_stride = (long)nt;
        }
        L->invd.alloc(nt);
        L->b.alloc(nt); L->x.alloc(nt); L->x2.alloc(nt); L->r.alloc(nt); L->e.alloc(nt);
        if (l < amg->sched.size()) {
            L->axis = amg->sched[l];
            L->wm.alloc(nt); L->wp.alloc(nt);
            m[L->axis] = (m[L->axis] + 1) / 2;
        }
        amg->lv.push_back(L);
    }
    amg->ncoarse = (int)amg->lv.back()->g.nown;
    TP_REQUIRE(amg->ncoarse <= 1024, "coarsest AMG grid too large for the dense solve");
    amg->coarse_inv.alloc((size_t)2 * amg->ncoarse * amg->ncoarse);
}

void amg_setup(tp_ctx *c, Amg *amg, const Stencil &A0) {
    // level 0 works on the slab's own grid descriptor but with dead halos (couplings into other
    // slabs are ignored inside the AMG)
    AmgLevel *L0 = amg->lv[0];
    L0->op = A0;
    for (size_t l = 0; l < amg->lv.size(); ++l) {
        AmgLevel *L = amg->lv[l];
        hipLaunchKernelGGL(k_amg_weights, grid_for(L->g.nown), dim3(256), 0, c->stream, L->g, L->op, L->axis,
                           c->opt.amg_omega, L->wm.p, L->wp.p, L->invd.p);
        if (L->axis >= 0) {
            AmgLevel *Lc = amg->lv[l + 1];
            hipLaunchKernelGGL(k_amg_coarsen, grid_for(Lc->g.nown), dim3(256), 0, c->stream, L->g, Lc->g, L->op,
                               L->axis, L->wm.p, L->wp.p, Lc->A.p);
        }
    }
    AmgLevel *Lc = amg->lv.back();
    const int n = amg->ncoarse;
    hipLaunchKernelGGL(k_amg_dense_inverse, dim3(1), dim3(256), 0, c->stream, Lc->g, Lc->op, n, amg->coarse_inv.p,
                       amg->coarse_inv.p + (size_t)n * n);
    TP_HIP(hipGetLastError());
}

static void vcycle_rec(tp_ctx *c, Amg *amg, size_t l, const double *b, double *x) {
    AmgLevel *L = amg->lv[l];
    const GridDev &g = L->g;
    const dim3 gr = grid_for(g.nown), bl(256);
    if (L->axis < 0) {
        const int n = amg->ncoarse;
        hipLaunchKernelGGL(k_amg_dense_apply, grid_for(n, 64), dim3(64), 0, c->stream, g, n,
                           amg->coarse_inv.p + (size_t)n * n, b, x);
        return;
    }
    const int nu = std::max(1, c->opt.amg_nu);
    // pre-smoothing from a zero guess: ping-pong so that the result lands in L->x
    double *cur = (nu % 2 == 1) ? L->x.p : L->x2.p, *oth = (nu % 2 == 1) ? L->x2.p : L->x.p;
    hipLaunchKernelGGL(k_amg_jacobi0, gr, bl, 0, c->stream, g, L->invd.p, b, cur);
    for (int k = 1; k < nu; ++k) {
        hipLaunchKernelGGL(k_amg_jacobi, gr, bl, 0, c->stream, g, L->op, L->invd.p, b, cur, oth);
        std::swap(cur, oth);
    }
    // cur == L->x
    hipLaunchKernelGGL(k_amg_resid, gr, bl, 0, c->stream, g, L->op, b, cur, L->r.p);
    AmgLevel *Lc = amg->lv[l + 1];
    hipLaunchKernelGGL(k_amg_restrict, grid_for(Lc->g.nown), bl, 0, c->stream, g, Lc->g, L->axis, L->wm.p, L->wp.p,
                       L->r.p, Lc->b.p);
    vcycle_rec(c, amg, l + 1, Lc->b.p, Lc->e.p);
    hipLaunchKernelGGL(k_amg_prolong_add, gr, bl, 0, c->stream, g, Lc->g, L->axis, L->wm.p, L->wp.p, Lc->e.p, cur);
    // post-smoothing: nu sweeps; x (caller's buffer) is distinct from L->x / L->x2, last sweep writes it
    double *src = cur;
    for (int k = 0; k < nu; ++k) {
        double *dst = (k == nu - 1) ? x : (src == L->x.p ? L->x2.p : L->x.p);
        hipLaunchKernelGGL(k_amg_jacobi, gr, bl, 0, c->stream, g, L->op, L->invd.p, b, src, dst);
        src = dst;
    }
}

void amg_vcycle(tp_ctx *c, Amg *amg, const double *b, double *x) {
    TP_REQUIRE(amg && !amg->lv.empty(), "AMG not set up");
    TP_REQUIRE(x != amg->lv[0]->x.p && x != amg->lv[0]->x2.p && b != x, "aliasing in amg_vcycle");
    vcycle_rec(c, amg, 0, b, x);
    TP_HIP(hipGetLastError());
    c->vcycles++;
}

}  // namespace tp

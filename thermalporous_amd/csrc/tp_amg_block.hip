// System (2x2-block) semicoarsening AMG: the stage-1 solver of pc_cptramg[_QI|_TI].
//
// Reference: twophase.py:552-566 -- CPTRStage1PC (preconditioners.py:1243-1571) with ONE hypre BoomerAMG V-cycle on
// the interleaved (p,T) operator Atilde_00 (VectorFunctionSpace layout, `vector=True` forced at twophase.py:935-955;
// 2*i, 2*i+1 indexing preconditioners.py:1534-1536).  hypre is not reproducible; like the scalar hierarchy of
// tp_amg.hip this is the build's own AMG, mirrored by oracle/linalg.py:BlockSemiAMG -- an "unknown-based" system AMG
// on the same semicoarsening grids:
//   * every stencil entry is an NB x NB block (7*NB*NB planes per level), vectors are NB planes;
//   * interpolation per unknown q from its own diagonal block A^{qq} (weights w-_q, w+_q), R = P^T;
//   * coarse block (q,r): rows combined with the restriction weights of q, columns along the coarsening axis
//     interpolated with the weights of r -- the same lumped non-Galerkin 7-point formula block by block;
//   * smoother: damped block-Jacobi with the NB x NB diagonal blocks; dense inverse (partial pivoting) on the coarsest
//     grid; cycle shape and coarsening schedule are those of the pressure hierarchy (same options).
// This is a "next" row of SURVEY.md 8f-3: thread-per-cell kernels (each HBM-bound: 28 operator planes per sweep) on the
// big levels; like tp_amg.hip the cycle fuses the two zero-guess pre-sweeps, the coarse-grid correction with the first
// post-sweep (mid levels), and runs every level of <= 1024 cells in ONE single-workgroup tail kernel (k_bamg_tail).
// Multi-GPU (round 3): like the scalar
// hierarchy -- levels with more than amg_gather_cells cells stay distributed over the slabs (C points = even GLOBAL planes,
// a halo exchange in front of every kernel that reads across the slab boundary, weights / inverse diagonal blocks / -- for
// slab-axis levels -- operator rows exchanged once per set-up), the first smaller level is gathered in place and the rest
// of the cycle runs replicated; a grid below the threshold is replicated from the top (dist_levels = 0).
#include "tp_common.hpp"
#include <algorithm>

namespace tp {

struct BAmgLevel {
    GridDev g;
    DBuf<double> A;            // 7*NB*NB planes (levels >= 1; level 0 views the Jacobian / decoupled operator)
    BStencil op;
    DBuf<double> invD;         // NB*NB planes: omega * inverse of the diagonal block
    DBuf<double> wm, wp;       // NB planes each
    DBuf<double> b, x, x2, r, e;
    int axis = -1;
    int pre = 0, post = 0;
};

struct BAmg {
    int nb = 2;
    std::vector<BAmgLevel *> lv;
    std::vector<int> sched;
    DBuf<double> dense;        // [M | Minv] of the coarsest grid, (nb*ncoarse)^2 each
    int ncoarse = 0;
    int dist_levels = 0;       // levels [0, dist_levels) are this rank's slab of the level; the rest global + replicated
    int tail_level = 0;        // first level of the single-workgroup tail kernel (k_bamg_tail)
    long fuse_below = 0;       // replicated levels with fewer cells fuse prolongation + first post-sweep
    DBuf<char> lvdev;          // device array of BTailLevel descriptors
    std::vector<char> lvhost;
    std::vector<std::vector<std::pair<int, int>>> ranges;   // [level][rank] -> owned global planes along axis 2
    ~BAmg() { for (auto *l : lv) delete l; }
};

// slab parity / open ends along the slab axis (as in tp_amg.hip)
__device__ __forceinline__ int b_par(const GridDev &gf, int a) { return a == 2 ? (gf.off2 & 1) : 0; }
__device__ __forceinline__ bool b_open_lo(const GridDev &g, int a) { return a == 2 && g.nb_lo; }
__device__ __forceinline__ bool b_open_hi(const GridDev &g, int a) { return a == 2 && g.nb_hi; }

static inline dim3 grid_for(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

template <int NB>
struct BLevelDev {
    GridDev g;
    BStencil op;
    const double *invD, *wm, *wp;
    int axis;
};

__device__ __forceinline__ void b_ijk(const GridDev &g, long tid, int &i0, int &i1, int &i2) {
    // 32-bit unsigned divisions (every slab has fewer than 2^31 cells, tp_create): a 64-bit division is ~150 instructions on
    // this ISA, and the few-thousand-cell levels of a V-cycle are bound by exactly that kind of per-cell index arithmetic
    const unsigned t = (unsigned)tid, np = (unsigned)g.np, n0 = (unsigned)g.n0;
    const unsigned q2 = t / np, rem = t - q2 * np, q1 = rem / n0;
    i2 = (int)q2;
    i1 = (int)q1;
    i0 = (int)(rem - q1 * n0);
}

// ---- set-up -------------------------------------------------------------------------------------------
template <int NB>
__global__ void k_bamg_level(BLevelDev<NB> L, double omega, double *wm, double *wp, double *invD) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= L.g.nown) return;
    const long c = L.g.np + tid, nt = L.g.ntot;
    double D[NB][NB];
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int r = 0; r < NB; ++r) D[q][r] = L.op.at(0, q, r)[c];
    if (NB == 1) {
        invD[c] = omega / D[0][0];
    } else {
        const double det = D[0][0] * D[NB - 1][NB - 1] - D[0][NB - 1] * D[NB - 1][0];
        const double f = omega / det;
        invD[(0 * NB + 0) * nt + c] = D[NB - 1][NB - 1] * f;
        invD[(0 * NB + (NB - 1)) * nt + c] = -D[0][NB - 1] * f;
        invD[((NB - 1) * NB + 0) * nt + c] = -D[NB - 1][0] * f;
        invD[((NB - 1) * NB + (NB - 1)) * nt + c] = D[0][0] * f;
    }
    if (L.axis < 0) return;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        double cs = 0.0;
#pragma unroll
        for (int s = 1; s < 7; ++s)
            if ((s - 1) / 2 != L.axis) cs += L.op.at(s, q, q)[c];
        const double cc = D[q][q] + cs;
        wm[(long)q * nt + c] = -L.op.at(1 + 2 * L.axis, q, q)[c] / cc;
        wp[(long)q * nt + c] = -L.op.at(2 + 2 * L.axis, q, q)[c] / cc;
    }
}

template <int NB>
__global__ void k_bamg_coarsen(BLevelDev<NB> L, GridDev gc, BStencil Ac) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= gc.nown) return;
    const GridDev &gf = L.g;
    const int a = L.axis;
    int I[3];
    b_ijk(gc, tid, I[0], I[1], I[2]);
    const long cc = gc.np + tid;
    int F[3] = {I[0], I[1], I[2]};
    F[a] = 2 * I[a] + b_par(gf, a);
    const int nfa = a == 0 ? gf.n0 : (a == 1 ? gf.n1 : gf.n2);
    const long stride = a == 0 ? 1 : (a == 1 ? gf.n0 : gf.np);
    const long f = gf.np + (long)F[0] + (long)gf.n0 * F[1] + gf.np * F[2];
    const bool hm = F[a] - 1 >= 0 || b_open_lo(gf, a), hp = F[a] + 1 < nfa || b_open_hi(gf, a);
    const long gm = hm ? f - stride : f, gp = hp ? f + stride : f;
    const long ntf = gf.ntot;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const double Pm = hm ? L.wp[(long)q * ntf + gm] : 0.0;       // row restriction weights of unknown q
        const double Pp = hp ? L.wm[(long)q * ntf + gp] : 0.0;
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const double Wm = hm ? L.wm[(long)r * ntf + gm] : 0.0;   // column interpolation weights of unknown r
            const double Wp = hp ? L.wp[(long)r * ntf + gp] : 0.0;
            double rho_f = 0.0, rho_m = 0.0, rho_p = 0.0, out[7], offsum = 0.0;
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                const double af = L.op.at(s, q, r)[f];
                const double am = hm ? L.op.at(s, q, r)[gm] : 0.0, ap = hp ? L.op.at(s, q, r)[gp] : 0.0;
                rho_f += af; rho_m += am; rho_p += ap;
                if (s == 0) continue;
                double v;
                if (s == 1 + 2 * a) v = af * Wm;
                else if (s == 2 + 2 * a) v = af * Wp;
                else v = af + Pm * am + Pp * ap;
                out[s] = v;
                offsum += v;
            }
            out[0] = -offsum + rho_f + Pm * rho_m + Pp * rho_p;
#pragma unroll
            for (int s = 0; s < 7; ++s) Ac.at(s, q, r)[cc] = out[s];
        }
    }
}

// dense inverse of the coarsest block system (cell-interleaved unknowns i = cell*NB + q), Gauss-Jordan with partial
// pivoting in one workgroup (the (p,T) diagonal blocks are not ordered by dominance)
template <int NB>
__global__ __launch_bounds__(256) void k_bamg_dense_inverse(BLevelDev<NB> L, int ncell, double *M, double *Minv) {
    const int n = ncell * NB;
    const GridDev &g = L.g;
    __shared__ int s_piv;
    __shared__ double s_val[256];
    __shared__ int s_idx[256];
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) { M[e] = 0.0; Minv[e] = (e / n == e % n) ? 1.0 : 0.0; }
    __syncthreads();
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    for (int rc = threadIdx.x; rc < ncell; rc += blockDim.x) {
        int i0, i1, i2;
        b_ijk(g, rc, i0, i1, i2);
        const long c = g.np + rc;
        const bool has[7] = {true, i0 > 0, i0 < g.n0 - 1, i1 > 0, i1 < g.n1 - 1, i2 > 0, i2 < g.n2 - 1};
        for (int s = 0; s < 7; ++s)
            if (has[s])
                for (int q = 0; q < NB; ++q)
                    for (int r = 0; r < NB; ++r)
                        M[(long)(rc * NB + q) * n + (long)(rc + off[s]) * NB + r] += L.op.at(s, q, r)[c];
    }
    __syncthreads();
    for (int p = 0; p < n; ++p) {
        // pivot search over rows >= p
        double best = -1.0;
        int bi = p;
        for (int r = p + threadIdx.x; r < n; r += blockDim.x) {
            const double v = fabs(M[(long)r * n + p]);
            if (v > best) { best = v; bi = r; }
        }
        s_val[threadIdx.x] = best;
        s_idx[threadIdx.x] = bi;
        __syncthreads();
        if (threadIdx.x == 0) {
            double bv = -1.0;
            int bb = p;
            for (int t = 0; t < (int)blockDim.x; ++t)
                if (s_val[t] > bv || (s_val[t] == bv && s_idx[t] < bb)) { bv = s_val[t]; bb = s_idx[t]; }
            s_piv = bb;
        }
        __syncthreads();
        const int pr = s_piv;
        if (pr != p)
            for (int e = threadIdx.x; e < n; e += blockDim.x) {
                double t = M[(long)p * n + e]; M[(long)p * n + e] = M[(long)pr * n + e]; M[(long)pr * n + e] = t;
                t = Minv[(long)p * n + e]; Minv[(long)p * n + e] = Minv[(long)pr * n + e]; Minv[(long)pr * n + e] = t;
            }
        __syncthreads();
        const double piv = M[(long)p * n + p];
        __syncthreads();
        for (int e = threadIdx.x; e < n; e += blockDim.x) { M[(long)p * n + e] /= piv; Minv[(long)p * n + e] /= piv; }
        __syncthreads();
        for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
            const int r = e / n, q = e % n;
            if (r == p) continue;
            const double fct = M[(long)r * n + p];
            Minv[(long)r * n + q] -= fct * Minv[(long)p * n + q];
            if (q != p) M[(long)r * n + q] -= fct * M[(long)p * n + q];
        }
        __syncthreads();
        for (int r = threadIdx.x; r < n; r += blockDim.x)
            if (r != p) M[(long)r * n + p] = 0.0;
        __syncthreads();
    }
}

// The same inverse for n <= 128 unknowns, IN PLACE in LDS (n*n doubles <= 128 KB): Gauss-Jordan with partial pivoting on
// one matrix -- the pivot column of the identity half is never stored, row swaps are undone as column swaps at the end.
// The global-memory kernel above spends ~24 us per pivot on C4's 128 x 128 system (3.1 ms per set-up, 70 % of it).
template <int NB>
__global__ __launch_bounds__(1024) void k_bamg_dense_inverse_lds(BLevelDev<NB> L, int ncell, double *Minv) {
    extern __shared__ double A[];                   // n x n, row-major
    __shared__ int s_piv[128];
    const int n = ncell * NB, T = blockDim.x, t = threadIdx.x;
    const GridDev &g = L.g;
    for (int e = t; e < n * n; e += T) A[e] = 0.0;
    __syncthreads();
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    for (int rc = t; rc < ncell; rc += T) {
        int i0, i1, i2;
        b_ijk(g, rc, i0, i1, i2);
        const long c = g.np + rc;
        const bool has[7] = {true, i0 > 0, i0 < g.n0 - 1, i1 > 0, i1 < g.n1 - 1, i2 > 0, i2 < g.n2 - 1};
        for (int s = 0; s < 7; ++s)
            if (has[s])
                for (int q = 0; q < NB; ++q)
                    for (int r = 0; r < NB; ++r)
                        A[(rc * NB + q) * n + (int)(rc + off[s]) * NB + r] += L.op.at(s, q, r)[c];
    }
    __syncthreads();
    for (int p = 0; p < n; ++p) {
        // pivot search over rows >= p of column p: first wave, ties to the lowest row (as the global kernel)
        if (t < 64) {
            double best = -1.0;
            int bi = p;
            for (int r = p + t; r < n; r += 64) {
                const double v = fabs(A[r * n + p]);
                if (v > best) { best = v; bi = r; }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const double ov = __shfl_xor(best, d, 64);
                const int oi = __shfl_xor(bi, d, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (t == 0) s_piv[p] = bi;
        }
        __syncthreads();
        const int pr = s_piv[p];
        if (pr != p)
            for (int e = t; e < n; e += T) { const double v = A[p * n + e]; A[p * n + e] = A[pr * n + e]; A[pr * n + e] = v; }
        __syncthreads();
        const double ipiv = 1.0 / A[p * n + p];
        __syncthreads();
        // row p: the pivot column takes the identity's entry (1), then the row is scaled
        for (int e = t; e < n; e += T) A[p * n + e] = (e == p ? 1.0 : A[p * n + e]) * ipiv;
        __syncthreads();
        // every other row i: f = A[i][p]; A[i][p] <- 0; A[i][:] -= f * A[p][:]   (one row per 16-lane group at a time)
        const int grp = t >> 4, gl = t & 15, ng = T >> 4;
        for (int i = grp; i < n; i += ng) {
            if (i == p) continue;
            const double f = A[i * n + p];
            // all 16 lanes of the group have read f before lane (p & 15) overwrites A[i][p]: same wave, lock step
            for (int e = gl; e < n; e += 16) {
                const double base = (e == p) ? 0.0 : A[i * n + e];
                A[i * n + e] = base - f * A[p * n + e];
            }
        }
        __syncthreads();
    }
    // undo the row swaps: columns, in reverse order
    for (int p = n - 1; p >= 0; --p) {
        const int pr = s_piv[p];
        if (pr != p) {
            for (int r = t; r < n; r += T) { const double v = A[r * n + p]; A[r * n + p] = A[r * n + pr]; A[r * n + pr] = v; }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int e = t; e < n * n; e += T) Minv[e] = A[e];
}

// ---- cycle kernels -------------------------------------------------------------------------------------
// (b - A x)_q at cell c
template <int NB>
__device__ __forceinline__ void b_resid(const BLevelDev<NB> &L, const double *b, const double *x, long c, double (&r)[NB]) {
    const GridDev &g = L.g;
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    const long nt = g.ntot;
#pragma unroll
    for (int q = 0; q < NB; ++q) r[q] = b[(long)q * nt + c];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        double xv[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) xv[k] = x[(long)k * nt + c + off[s]];
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int k = 0; k < NB; ++k) r[q] -= L.op.at(s, q, k)[c] * xv[k];
    }
}

// out = invD b (first sweep from a zero guess) | out = x + invD (b - A x)
template <int NB, bool FIRST>
__global__ __launch_bounds__(256) void k_bamg_smooth(BLevelDev<NB> L, const double *b, const double *x, double *out) {
    const long tid = xcd_tid();
    if (tid >= L.g.nown) return;
    const long c = L.g.np + tid, nt = L.g.ntot;
    double r[NB];
    if (FIRST) {
#pragma unroll
        for (int q = 0; q < NB; ++q) r[q] = b[(long)q * nt + c];
    } else {
        b_resid<NB>(L, b, x, c, r);
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        double s = FIRST ? 0.0 : x[(long)q * nt + c];
#pragma unroll
        for (int k = 0; k < NB; ++k) s += L.invD[(long)(q * NB + k) * nt + c] * r[k];
        out[(long)q * nt + c] = s;
    }
}

template <int NB>
__global__ __launch_bounds__(256) void k_bamg_resid(BLevelDev<NB> L, const double *b, const double *x, double *r) {
    const long tid = xcd_tid();
    if (tid >= L.g.nown) return;
    const long c = L.g.np + tid;
    double v[NB];
    b_resid<NB>(L, b, x, c, v);
#pragma unroll
    for (int q = 0; q < NB; ++q) r[(long)q * L.g.ntot + c] = v[q];
}

template <int NB>
__global__ __launch_bounds__(256) void k_bamg_restrict(BLevelDev<NB> Lf, GridDev gc, const double *r, double *rc, long cstride) {
    const long tid = xcd_tid();
    if (tid >= gc.nown) return;
    const GridDev &gf = Lf.g;
    const int a = Lf.axis;
    int I[3];
    b_ijk(gc, tid, I[0], I[1], I[2]);
    int F[3] = {I[0], I[1], I[2]};
    F[a] = 2 * I[a] + b_par(gf, a);
    const int nfa = a == 0 ? gf.n0 : (a == 1 ? gf.n1 : gf.n2);
    const long stride = a == 0 ? 1 : (a == 1 ? gf.n0 : gf.np);
    const long f = gf.np + (long)F[0] + (long)gf.n0 * F[1] + gf.np * F[2];
    const bool hm = F[a] - 1 >= 0 || b_open_lo(gf, a), hp = F[a] + 1 < nfa || b_open_hi(gf, a);
    const long fm = hm ? f - stride : f, fp = hp ? f + stride : f;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const long o = (long)q * gf.ntot;
        double v = r[o + f];
        v += hp ? Lf.wm[o + fp] * r[o + fp] : 0.0;       // (order of SemiAMG.restrict: right F point first)
        v += hm ? Lf.wp[o + fm] * r[o + fm] : 0.0;
        rc[(long)q * cstride + gc.np + tid] = v;
    }
}

// x (+)= P ec
template <int NB>
__global__ __launch_bounds__(256) void k_bamg_prolong(BLevelDev<NB> Lf, GridDev gc, const double *ec, const double *xin,
                                                      double *xout, long cstride) {
    const long tid = xcd_tid();
    if (tid >= Lf.g.nown) return;
    const GridDev &g = Lf.g;
    const int a = Lf.axis;
    int i[3];
    b_ijk(g, tid, i[0], i[1], i[2]);
    const int p = b_par(g, a);
    const int Fa = i[a], Ia = (Fa - p) >> 1;          // (Fa - p may be -1: the C parent below the slab, in the halo plane)
    int I[3] = {i[0], i[1], i[2]};
    I[a] = Ia;
    const long ci = gc.np + (long)I[0] + (long)gc.n0 * I[1] + gc.np * I[2];
    const long cs = a == 0 ? 1 : (a == 1 ? gc.n0 : gc.np);
    const int nca = a == 0 ? gc.n0 : (a == 1 ? gc.n1 : gc.n2);
    const long c = g.np + tid;
    const bool isF = (Fa + p) & 1, hasR = isF && (Ia + 1 < nca || b_open_hi(gc, a));
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const double e0 = ec[(long)q * cstride + ci], e1 = ec[(long)q * cstride + (hasR ? ci + cs : ci)];
        const double v = isF ? Lf.wm[(long)q * g.ntot + c] * e0 + (hasR ? Lf.wp[(long)q * g.ntot + c] * e1 : 0.0) : e0;
        xout[(long)q * g.ntot + c] = (xin ? xin[(long)q * g.ntot + c] : 0.0) + v;
    }
}

// ---- fused building blocks -----------------------------------------------------------------------------
// two damped block-Jacobi sweeps from a zero guess: x1 = invD b ; out = x1 + invD (b - A x1)   (x1 of the neighbours is
// recomputed from their b and invD: the operator is read once, no intermediate vector)
template <int NB>
__device__ __forceinline__ void b_pre2_cell(const BLevelDev<NB> &L, const double *__restrict__ b, long c, double (&out)[NB]) {
    const GridDev &g = L.g;
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    const long nt = g.ntot;
    double r[NB], x1c[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) r[q] = b[(long)q * nt + c];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const long n = c + off[s];
        double bn[NB], xv[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) bn[k] = b[(long)k * nt + n];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < NB; ++k) v += L.invD[(long)(q * NB + k) * nt + n] * bn[k];
            xv[q] = v;
        }
        if (s == 0) {
#pragma unroll
            for (int q = 0; q < NB; ++q) x1c[q] = xv[q];
        }
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int k = 0; k < NB; ++k) r[q] -= L.op.at(s, q, k)[c] * xv[k];
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        double v = x1c[q];
#pragma unroll
        for (int k = 0; k < NB; ++k) v += L.invD[(long)(q * NB + k) * nt + c] * r[k];
        out[q] = v;
    }
}

template <int NB>
__global__ __launch_bounds__(256) void k_bamg_pre2(BLevelDev<NB> L, const double *b, double *out) {
    const long tid = xcd_tid();
    if (tid >= L.g.nown) return;
    const long c = L.g.np + tid;
    double v[NB];
    b_pre2_cell<NB>(L, b, c, v);
#pragma unroll
    for (int q = 0; q < NB; ++q) out[(long)q * L.g.ntot + c] = v[q];
}

// (P ec)_q at the fine cell (F0,F1,F2) (memory-safe coordinates; replicated levels: no slab parity)
template <int NB>
__device__ __forceinline__ void b_prolong_val(const BLevelDev<NB> &Lf, const GridDev &gc, const double *__restrict__ ec,
                                              long cstride, int F0, int F1, int F2, double (&v)[NB]) {
    const GridDev &g = Lf.g;
    const int a = Lf.axis;
    const int p = b_par(g, a);
    const int Fa = a == 0 ? F0 : (a == 1 ? F1 : F2);
    const int Ia = (Fa - p) >> 1;
    const bool isF = (Fa + p) & 1;
    const int I0 = a == 0 ? Ia : F0, I1 = a == 1 ? Ia : F1, I2 = a == 2 ? Ia : F2;
    const long ci = gc.np + (long)I0 + (long)gc.n0 * I1 + gc.np * I2;
    const long cf = g.np + (long)F0 + (long)g.n0 * F1 + g.np * F2;
    const long cs = a == 0 ? 1 : (a == 1 ? gc.n0 : gc.np);
    const int nca = a == 0 ? gc.n0 : (a == 1 ? gc.n1 : gc.n2);
    const bool hasR = isF && (Ia + 1 < nca || b_open_hi(gc, a));
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const double wm = Lf.wm[(long)q * g.ntot + cf], wp = Lf.wp[(long)q * g.ntot + cf];
        const double e0 = ec[(long)q * cstride + ci], e1 = ec[(long)q * cstride + (hasR ? ci + cs : ci)];
        v[q] = isF ? wm * e0 + (hasR ? wp * e1 : 0.0) : e0;
    }
}

// coarse-grid correction fused with the first post-sweep: x' = x + P ec ; out = x' + invD (b - A x')
// (branch-free as tp_amg.hip:prolong_jacobi_cell: clamped neighbour coordinates, absent neighbours have zero blocks)
template <int NB>
__device__ __forceinline__ void b_prolong_smooth_cell(const BLevelDev<NB> &Lf, const GridDev &gc, const double *__restrict__ b,
                                                      const double *x, const double *__restrict__ ec, long cstride, long tid,
                                                      double (&out)[NB]) {
    const GridDev &g = Lf.g;
    int i0, i1, i2;
    b_ijk(g, tid, i0, i1, i2);
    const long c = g.np + tid, nt = g.ntot;
    const int m0 = max(i0 - 1, 0), p0 = min(i0 + 1, g.n0 - 1);
    const int m1 = max(i1 - 1, 0), p1 = min(i1 + 1, g.n1 - 1);
    const int m2 = max(i2 - 1, g.nb_lo ? -1 : 0), p2 = min(i2 + 1, g.nb_hi ? g.n2 : g.n2 - 1);
    const int N0[7] = {i0, m0, p0, i0, i0, i0, i0}, N1[7] = {i1, i1, i1, m1, p1, i1, i1}, N2[7] = {i2, i2, i2, i2, i2, m2, p2};
    double r[NB], v0[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) r[q] = b[(long)q * nt + c];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        double v[NB];
        b_prolong_val<NB>(Lf, gc, ec, cstride, N0[s], N1[s], N2[s], v);
        if (x) {
            const long cn = g.np + (long)N0[s] + (long)g.n0 * N1[s] + g.np * N2[s];
#pragma unroll
            for (int k = 0; k < NB; ++k) v[k] += x[(long)k * nt + cn];
        }
        if (s == 0) {
#pragma unroll
            for (int q = 0; q < NB; ++q) v0[q] = v[q];
        }
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int k = 0; k < NB; ++k) r[q] -= Lf.op.at(s, q, k)[c] * v[k];
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        double sacc = v0[q];
#pragma unroll
        for (int k = 0; k < NB; ++k) sacc += Lf.invD[(long)(q * NB + k) * nt + c] * r[k];
        out[q] = sacc;
    }
}

template <int NB>
__global__ __launch_bounds__(256) void k_bamg_prolong_smooth(BLevelDev<NB> Lf, GridDev gc, const double *b, const double *x,
                                                             const double *ec, double *out, long cstride) {
    const long tid = xcd_tid();
    if (tid >= Lf.g.nown) return;
    double v[NB];
    b_prolong_smooth_cell<NB>(Lf, gc, b, x, ec, cstride, tid, v);
#pragma unroll
    for (int q = 0; q < NB; ++q) out[(long)q * Lf.g.ntot + Lf.g.np + tid] = v[q];
}

// (P^T r)_q at coarse cell tidc
template <int NB>
__device__ __forceinline__ void b_restrict_cell(const BLevelDev<NB> &Lf, const GridDev &gc, const double *__restrict__ r,
                                                long tidc, double (&out)[NB]) {
    const GridDev &gf = Lf.g;
    const int a = Lf.axis;
    int I[3];
    b_ijk(gc, tidc, I[0], I[1], I[2]);
    int F[3] = {I[0], I[1], I[2]};
    F[a] = 2 * I[a] + b_par(gf, a);
    const int nfa = a == 0 ? gf.n0 : (a == 1 ? gf.n1 : gf.n2);
    const long stride = a == 0 ? 1 : (a == 1 ? gf.n0 : gf.np);
    const long f = gf.np + (long)F[0] + (long)gf.n0 * F[1] + gf.np * F[2];
    const bool hm = F[a] - 1 >= 0 || b_open_lo(gf, a), hp = F[a] + 1 < nfa || b_open_hi(gf, a);
    const long fm = hm ? f - stride : f, fp = hp ? f + stride : f;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const long o = (long)q * gf.ntot;
        double v = r[o + f];
        v += hp ? Lf.wm[o + fp] * r[o + fp] : 0.0;
        v += hm ? Lf.wp[o + fm] * r[o + fm] : 0.0;
        out[q] = v;
    }
}

// ---- the tail: every level of <= 1024 cells, down to the dense solve and back, in one workgroup -----------------
// (vectors stay in global memory: a few KB per level, L2 resident; a phase boundary is a __syncthreads)
template <int NB>
struct BTailLevel {
    BLevelDev<NB> d;
    double *b, *x, *x2, *r, *e;
    int pre, post;
};

template <int NB>
__global__ __launch_bounds__(512) void k_bamg_tail(const BTailLevel<NB> *__restrict__ lvg, int l0, int nlev, int ncell, const double *Minv,
                                                    const double *b_top, double *e_top) {
    constexpr int T = 512;             // (the launch's block size, as a constant: tp_common.hpp:xcd_tid)
    const int t = threadIdx.x;
    // level descriptors are read with scalar loads (uniform addresses): they live in SGPRs, not in the lanes' registers
    const BTailLevel<NB> *__restrict__ lv = lvg;
    // pull the read-only arrays of the tail into the L2 at once (tp_amg.hip:k_amg_tail)
    {
        double sink = 0.0;
        for (int l = l0; l < nlev; ++l) {
            const BLevelDev<NB> &L = lv[l].d;
            for (long i = L.g.np + t; i < L.g.np + L.g.nown; i += T) {
#pragma unroll
                for (int s = 0; s < 7; ++s)
#pragma unroll
                    for (int q = 0; q < NB * NB; ++q) sink += L.op.at(s, q / NB, q % NB)[i];
#pragma unroll
                for (int q = 0; q < NB * NB; ++q) sink += L.invD[(long)q * L.g.ntot + i];
                if (L.axis >= 0) {
#pragma unroll
                    for (int q = 0; q < NB; ++q) sink += L.wm[(long)q * L.g.ntot + i] + L.wp[(long)q * L.g.ntot + i];
                }
            }
        }
        const int n = ncell * NB;
        for (int i = t; i < n * n; i += T) sink += Minv[i];
        if (sink == 1.2345678e-300) e_top[0] = sink;       // never true: keeps the loads alive
    }
    // down-sweep (b and e of the tail's top level are the caller's; their plane stride is the level's ntot like everybody's)
    for (int l = l0; l < nlev - 1; ++l) {
        const BLevelDev<NB> L = lv[l].d;
        const GridDev gc = lv[l + 1].d.g;
        const double *b = (l == l0) ? b_top : lv[l].b;
        const long nt = L.g.ntot;
        double *rc = lv[l + 1].b;
        const double *res = b;
        if (lv[l].pre > 0) {
            double *cur = lv[l].x, *oth = lv[l].x2;
            for (long i = t; i < L.g.nown; i += T) {
                const long c = L.g.np + i;
                double v[NB];
                if (lv[l].pre >= 2) b_pre2_cell<NB>(L, b, c, v);
                else {
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        double sacc = 0.0;
#pragma unroll
                        for (int k = 0; k < NB; ++k) sacc += L.invD[(long)(q * NB + k) * nt + c] * b[(long)k * nt + c];
                        v[q] = sacc;
                    }
                }
#pragma unroll
                for (int q = 0; q < NB; ++q) cur[(long)q * nt + c] = v[q];
            }
            __syncthreads();
            for (int k = 2; k < lv[l].pre; ++k) {
                for (long i = t; i < L.g.nown; i += T) {
                    const long c = L.g.np + i;
                    double r[NB];
                    b_resid<NB>(L, b, cur, c, r);
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        double sacc = cur[(long)q * nt + c];
#pragma unroll
                        for (int kk = 0; kk < NB; ++kk) sacc += L.invD[(long)(q * NB + kk) * nt + c] * r[kk];
                        oth[(long)q * nt + c] = sacc;
                    }
                }
                __syncthreads();
                double *tmp = cur; cur = oth; oth = tmp;
            }
            double *rr = lv[l].r;
            for (long i = t; i < L.g.nown; i += T) {
                const long c = L.g.np + i;
                double v[NB];
                b_resid<NB>(L, b, cur, c, v);
#pragma unroll
                for (int q = 0; q < NB; ++q) rr[(long)q * nt + c] = v[q];
            }
            __syncthreads();
            res = rr;
        }
        for (long i = t; i < gc.nown; i += T) {
            double v[NB];
            b_restrict_cell<NB>(L, gc, res, i, v);
#pragma unroll
            for (int q = 0; q < NB; ++q) rc[(long)q * gc.ntot + gc.np + i] = v[q];
        }
        __syncthreads();
    }
    // coarsest grid: e = Minv b, 16 lanes per row
    {
        const BLevelDev<NB> Lc = lv[nlev - 1].d;
        const double *b = (nlev - 1 == l0) ? b_top : lv[nlev - 1].b;
        double *e = (nlev - 1 == l0) ? e_top : lv[nlev - 1].e;
        const long bs = Lc.g.ntot, es = Lc.g.ntot;
        const int n = ncell * NB;
        const int grp = t >> 4, gl = t & 15;
        for (int r = grp; r < n; r += T >> 4) {
            double sacc = 0.0;
            for (int j = gl; j < n; j += 16) sacc += Minv[(long)r * n + j] * b[(long)(j % NB) * bs + Lc.g.np + j / NB];
            sacc += __shfl_xor(sacc, 8, 64);
            sacc += __shfl_xor(sacc, 4, 64);
            sacc += __shfl_xor(sacc, 2, 64);
            sacc += __shfl_xor(sacc, 1, 64);
            if (gl == 0) e[(long)(r % NB) * es + Lc.g.np + r / NB] = sacc;
        }
        __syncthreads();
    }
    // up-sweep
    for (int l = nlev - 2; l >= l0; --l) {
        const BLevelDev<NB> L = lv[l].d;
        const GridDev gc = lv[l + 1].d.g;
        const long nt = L.g.ntot;
        const double *b = (l == l0) ? b_top : lv[l].b;
        double *out = (l == l0) ? e_top : lv[l].e;
        const long os = nt;
        const double *ec = lv[l + 1].e;
        const int extra = lv[l].pre >= 2 ? lv[l].pre - 2 : 0;
        const double *src = lv[l].pre == 0 ? nullptr : ((extra % 2 == 0) ? lv[l].x : lv[l].x2);
        if (lv[l].post == 0) {
            for (long i = t; i < L.g.nown; i += T) {
                int i0, i1, i2;
                b_ijk(L.g, i, i0, i1, i2);
                double v[NB];
                b_prolong_val<NB>(L, gc, ec, gc.ntot, i0, i1, i2, v);
#pragma unroll
                for (int q = 0; q < NB; ++q) out[(long)q * os + L.g.np + i] = (src ? src[(long)q * nt + L.g.np + i] : 0.0) + v[q];
            }
            __syncthreads();
            continue;
        }
        double *dst = (src == lv[l].x) ? lv[l].x2 : lv[l].x;
        for (int k = 0; k < lv[l].post; ++k) {
            const bool last = k == lv[l].post - 1;
            double *d = last ? out : dst;
            const long ds = last ? os : nt;
            for (long i = t; i < L.g.nown; i += T) {
                const long c = L.g.np + i;
                double v[NB];
                if (k == 0) b_prolong_smooth_cell<NB>(L, gc, b, src, ec, gc.ntot, i, v);
                else {
                    double r[NB];
                    b_resid<NB>(L, b, src, c, r);
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        double sacc = src[(long)q * nt + c];
#pragma unroll
                        for (int kk = 0; kk < NB; ++kk) sacc += L.invD[(long)(q * NB + kk) * nt + c] * r[kk];
                        v[q] = sacc;
                    }
                }
#pragma unroll
                for (int q = 0; q < NB; ++q) d[(long)q * ds + c] = v[q];
            }
            __syncthreads();
            src = d;
            dst = (src == lv[l].x) ? lv[l].x2 : lv[l].x;
        }
    }
}

// ---- host ----------------------------------------------------------------------------------------------
template <int NB>
static BLevelDev<NB> bdev(const BAmgLevel *L) {
    BLevelDev<NB> d;
    d.g = L->g; d.op = L->op; d.invD = L->invD.p; d.wm = L->wm.p; d.wp = L->wp.p; d.axis = L->axis;
    return d;
}

void bamg_build(tp_ctx *c, BAmg *&amg, const GridDev &g0, const double strength[3]) {
    delete amg;
    amg = new BAmg();
    c->graph_epoch++;
    const int NB = amg->nb;
    int n[3] = {g0.n0, g0.n1, g0.n2};
    double s[3];
    for (int a = 0; a < 3; ++a) s[a] = n[a] > 1 ? strength[a] : -1.0;
    while ((long)n[0] * n[1] * n[2] > std::max(1, c->opt.amg_min_cells) && amg->sched.size() < 40) {
        int best = -1;
        for (int a = 0; a < 3; ++a)
            if (n[a] > 1 && (best < 0 || s[a] > s[best])) best = a;
        if (best < 0) break;
        amg->sched.push_back(best);
        n[best] = (n[best] + 1) / 2;
        for (int q = 0; q < 3; ++q) s[q] = (q == best) ? s[q] * 0.5 : s[q] * 2.0;
    }
    int m[3] = {g0.n0, g0.n1, g0.n2};
    const int nu = std::max(1, c->opt.amg_nu);
    // multi-GPU: the rule of tp_amg.hip:amg_build -- a level stays on the slabs while it has more than amg_gather_cells
    // cells and every rank owns at least two of its planes
    const int nranks = c->dist ? c->grid.nranks : 1, me = c->dist ? c->grid.rank : 0;
    std::vector<std::pair<int, int>> cur(nranks);
    for (int r = 0; r < nranks; ++r) {
        if (c->dist) slab_of(c, r, cur[r].first, cur[r].second);
        else cur[r] = {0, m[2]};
    }
    const long gather_cells = c->opt.amg_gather_cells;
    bool still = c->dist && gather_cells >= 0;
    for (size_t l = 0; l <= amg->sched.size(); ++l) {
        BAmgLevel *L = new BAmgLevel();
        if (still) {
            int minp = 1 << 30;
            for (auto &q : cur) minp = std::min(minp, q.second - q.first);
            still = (long)m[0] * m[1] * m[2] > gather_cells && minp >= 2 && l < amg->sched.size();
            if (still) amg->dist_levels = (int)l + 1;
        }
        amg->ranges.push_back(cur);
        L->g = still ? make_grid(m[0], m[1], cur[me].second - cur[me].first, m[2], cur[me].first)
                     : make_grid(m[0], m[1], m[2], m[2], 0);
        const size_t nt = (size_t)L->g.ntot;
        if (l > 0) {
            L->A.alloc(7 * NB * NB * nt);
            L->op.base = L->A.p; L->op.ss = (long)NB * NB * nt; L->op.rs = (long)NB * nt; L->op.cs = (long)nt;
        }
        L->invD.alloc(NB * NB * nt);
        L->b.alloc(NB * nt); L->x.alloc(NB * nt); L->x2.alloc(NB * nt); L->r.alloc(NB * nt); L->e.alloc(NB * nt);
        // cycle shape: the rules of tp_amg.hip:dev_of (mirrored by oracle/linalg.py)
        const bool full = (int)l < c->opt.amg_full_levels;
        const bool small = L->g.np * (long)L->g.gn2 <= 1024;
        L->pre = full ? nu : std::max(0, c->opt.amg_coarse_pre);
        L->post = full ? nu : std::max(1, small ? c->opt.amg_tail_post : c->opt.amg_coarse_post);
        if (c->opt.amg_mid_skip && !full && !small && (((int)l - c->opt.amg_full_levels) & 1)) { L->pre = 0; L->post = 0; }
        if (l < amg->sched.size()) {
            L->axis = amg->sched[l];
            L->wm.alloc(NB * nt); L->wp.alloc(NB * nt);
            m[L->axis] = (m[L->axis] + 1) / 2;
            if (L->axis == 2)
                for (auto &q : cur) q = {(q.first + 1) / 2, (q.second + 1) / 2};     // even global planes survive
        }
        amg->lv.push_back(L);
    }
    // first level of the single-workgroup tail; replicated levels below fuse_below cells fuse prolongation + first post-sweep
    const long tail_cells = getenv("TP_BAMG_TAIL_CELLS") ? atol(getenv("TP_BAMG_TAIL_CELLS")) : 1024;
    amg->fuse_below = getenv("TP_BAMG_FUSE_BELOW") ? atol(getenv("TP_BAMG_FUSE_BELOW")) : 200000;
    amg->tail_level = (int)amg->lv.size() - 1;
    for (size_t l = (size_t)amg->dist_levels; l < amg->lv.size(); ++l)
        if (amg->lv[l]->g.nown <= tail_cells) { amg->tail_level = (int)l; break; }
    TP_REQUIRE((int)amg->lv.size() - amg->tail_level <= 24, "system-AMG tail has too many levels");
    amg->lvdev.alloc(amg->lv.size() * sizeof(BTailLevel<2>));
    amg->ncoarse = (int)amg->lv.back()->g.nown;
    TP_REQUIRE(amg->ncoarse * NB <= 2048, "coarsest system-AMG grid too large for the dense solve");
    const size_t nd = (size_t)amg->ncoarse * NB;
    amg->dense.alloc(2 * nd * nd);
}

// level l+1 as its parent level l sees it: below the last distributed level that is this rank's planes of the global
// (replicated) arrays -- plane 0 of the view is the lower halo (tp_amg.hip:coarse_view)
struct BCoarseView { GridDev g; long off; };
static BCoarseView bcoarse_view(const tp_ctx *c, const BAmg *amg, int l) {
    const BAmgLevel *Lc = amg->lv[l + 1];
    BCoarseView v;
    if (l + 1 < amg->dist_levels || l >= amg->dist_levels) { v.g = Lc->g; v.off = 0; return v; }
    const auto &q = amg->ranges[l + 1][c->grid.rank];
    v.g = make_grid(Lc->g.n0, Lc->g.n1, q.second - q.first, Lc->g.n2, q.first);
    v.off = Lc->g.np * q.first;
    return v;
}

void bamg_setup(tp_ctx *c, BAmg *amg, const BStencil &A0) {
    constexpr int NB = 2;
    TP_REQUIRE(amg->nb == NB, "system AMG is built for 2x2 blocks");
    amg->lv[0]->op = A0;
    const int lg = amg->dist_levels;
    for (size_t l = 0; l < amg->lv.size(); ++l) {
        BAmgLevel *L = amg->lv[l];
        const BLevelDev<NB> Ld = bdev<NB>(L);
        hipLaunchKernelGGL(k_bamg_level<NB>, grid_for(L->g.nown), dim3(256), 0, c->stream, Ld, c->opt.amg_omega, L->wm.p,
                           L->wp.p, L->invD.p);
        if ((int)l < lg) {
            // distributed level: the cycle reads inverse diagonal blocks and weights of the neighbours' boundary planes,
            // coarsening along the slab axis also their operator rows
            const long nt = L->g.ntot;
            halo_exchange(c, L->g, L->invD.p, NB * NB, nt);
            if (L->axis >= 0) { halo_exchange(c, L->g, L->wm.p, NB, nt); halo_exchange(c, L->g, L->wp.p, NB, nt); }
            if (L->axis == 2)
                for (int sl = 0; sl < 7; ++sl)
                    for (int q = 0; q < NB; ++q)
                        for (int r = 0; r < NB; ++r) halo_exchange(c, L->g, L->op.at(sl, q, r), 1, 0);
        }
        if (L->axis >= 0) {
            BAmgLevel *Lc = amg->lv[l + 1];
            const BCoarseView cv = bcoarse_view(c, amg, (int)l);
            BStencil Ac = Lc->op;
            Ac.base += cv.off;
            hipLaunchKernelGGL(k_bamg_coarsen<NB>, grid_for(cv.g.nown), dim3(256), 0, c->stream, Ld, cv.g, Ac);
            if ((int)l + 1 == lg)        // first replicated level: everybody gets everybody's rows
                gather_ranges(c, Lc->A.p, Lc->g.np, amg->ranges[lg], 7 * NB * NB, (size_t)Lc->g.ntot * sizeof(double), sizeof(double));
        }
    }
    {
        amg->lvhost.resize(amg->lv.size() * sizeof(BTailLevel<NB>));
        BTailLevel<NB> *h = reinterpret_cast<BTailLevel<NB> *>(amg->lvhost.data());
        for (size_t l = 0; l < amg->lv.size(); ++l) {
            BAmgLevel *L = amg->lv[l];
            memset((void *)&h[l], 0, sizeof(h[l]));
            h[l].d = bdev<NB>(L);
            h[l].b = L->b.p; h[l].x = L->x.p; h[l].x2 = L->x2.p; h[l].r = L->r.p; h[l].e = L->e.p;
            h[l].pre = L->pre; h[l].post = L->post;
        }
        TP_HIP(hipMemcpyAsync(amg->lvdev.p, amg->lvhost.data(), amg->lvhost.size(), hipMemcpyHostToDevice, c->stream));
    }
    BAmgLevel *Lc = amg->lv.back();
    const size_t nd = (size_t)amg->ncoarse * NB;
    static const bool lds_inv = !(getenv("TP_BAMG_DENSE_LDS") && atoi(getenv("TP_BAMG_DENSE_LDS")) == 0);
    if (lds_inv && nd <= 128) {
        const size_t bytes = nd * nd * sizeof(double);
        TP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bamg_dense_inverse_lds<NB>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 128 * (int)sizeof(double)));
        hipLaunchKernelGGL(k_bamg_dense_inverse_lds<NB>, dim3(1), dim3(1024), bytes, c->stream, bdev<NB>(Lc), amg->ncoarse,
                           amg->dense.p + nd * nd);
    } else {
        hipLaunchKernelGGL(k_bamg_dense_inverse<NB>, dim3(1), dim3(256), 0, c->stream, bdev<NB>(Lc), amg->ncoarse, amg->dense.p,
                           amg->dense.p + nd * nd);
    }
    TP_HIP(hipGetLastError());
}

// x = V-cycle(b): b, x are NB planes with the stride of level 0 (= ntot of the grid the hierarchy was built on)
void bamg_vcycle(tp_ctx *c, BAmg *amg, const double *b, double *x) {
    constexpr int NB = 2;
    const int nlev = (int)amg->lv.size(), lt = amg->tail_level;
    const dim3 bl(256);
    std::vector<double *> xs(nlev, nullptr);
    const size_t nd = (size_t)amg->ncoarse * NB;
    const int lg = amg->dist_levels;
    auto hx = [&](int l, const double *v) {        // halo exchange of the NB planes of a level-l vector (no-op below lg)
        if (l < lg) halo_exchange(c, amg->lv[l]->g, const_cast<double *>(v), NB, amg->lv[l]->g.ntot);
    };
    // down-sweep over the big levels
    for (int l = 0; l < lt; ++l) {
        BAmgLevel *L = amg->lv[l], *Lc = amg->lv[l + 1];
        const BLevelDev<NB> Ld = bdev<NB>(L);
        const double *bl_ = (l == 0) ? b : L->b.p;
        const dim3 gr = xcd_grid(L->g.nown);
        const double *res = bl_;
        const bool slab_axis = l < lg && L->axis == 2;      // the transfer itself crosses the slab boundary
        if (L->pre > 0) {
            double *cur = L->x.p, *oth = L->x2.p;
            if (L->pre >= 2) {
                hx(l, bl_);                         // the fused double sweep reads invD b of the neighbours
                hipLaunchKernelGGL(k_bamg_pre2<NB>, gr, bl, 0, c->stream, Ld, bl_, cur);
            } else {
                hipLaunchKernelGGL((k_bamg_smooth<NB, true>), gr, bl, 0, c->stream, Ld, bl_, (const double *)nullptr, cur);
            }
            for (int k = 2; k < L->pre; ++k) {
                hx(l, cur);
                hipLaunchKernelGGL((k_bamg_smooth<NB, false>), gr, bl, 0, c->stream, Ld, bl_, (const double *)cur, oth);
                std::swap(cur, oth);
            }
            xs[l] = cur;
            hx(l, cur);
            hipLaunchKernelGGL(k_bamg_resid<NB>, gr, bl, 0, c->stream, Ld, bl_, (const double *)cur, L->r.p);
            res = L->r.p;
        }
        if (slab_axis) hx(l, res);
        const BCoarseView cv = bcoarse_view(c, amg, l);
        hipLaunchKernelGGL(k_bamg_restrict<NB>, xcd_grid(cv.g.nown), bl, 0, c->stream, Ld, cv.g, res, Lc->b.p + cv.off,
                           (long)Lc->g.ntot);
        if (l + 1 == lg)        // restricted residual of every slab -> the replicated levels' right-hand side
            gather_ranges(c, Lc->b.p, Lc->g.np, amg->ranges[lg], NB, (size_t)Lc->g.ntot * sizeof(double), sizeof(double));
    }
    {   // the tail: every level from lt down to the dense solve and back, one launch
        BAmgLevel *Lt = amg->lv[lt];
        const double *bt = (lt == 0) ? b : Lt->b.p;
        double *et = (lt == 0) ? x : Lt->e.p;
        hipLaunchKernelGGL(k_bamg_tail<NB>, dim3(1), dim3(512), 0, c->stream, (const BTailLevel<NB> *)amg->lvdev.p, lt, nlev,
                           amg->ncoarse, (const double *)(amg->dense.p + nd * nd), bt, et);
    }
    // up-sweep over the big levels
    for (int l = lt - 1; l >= 0; --l) {
        BAmgLevel *L = amg->lv[l], *Lc = amg->lv[l + 1];
        const BLevelDev<NB> Ld = bdev<NB>(L);
        const double *bl_ = (l == 0) ? b : L->b.p;
        double *out = (l == 0) ? x : L->e.p;
        const dim3 gr = xcd_grid(L->g.nown);
        hx(l + 1, Lc->e.p);                         // distributed coarse level: parents across the boundary
        const BCoarseView cv = bcoarse_view(c, amg, l);
        const double *ec = Lc->e.p + cv.off;
        double *src;
        int k0 = 0;
        if (L->post > 0 && L->g.nown < amg->fuse_below) {
            // coarse-grid correction + first post-sweep in one launch (xs[l]'s halo is the one exchanged before the residual)
            double *dst = (L->post == 1) ? out : (xs[l] == L->x.p ? L->x2.p : L->x.p);
            hipLaunchKernelGGL(k_bamg_prolong_smooth<NB>, gr, bl, 0, c->stream, Ld, cv.g, bl_, (const double *)xs[l], ec, dst,
                               (long)Lc->g.ntot);
            src = dst;
            k0 = 1;
        } else {
            // x <- x + P ec into a buffer that is not the final output unless no post-smoothing follows
            double *dst = (L->post == 0) ? out : (xs[l] == L->x.p ? L->x2.p : L->x.p);
            hipLaunchKernelGGL(k_bamg_prolong<NB>, gr, bl, 0, c->stream, Ld, cv.g, ec, (const double *)xs[l], dst,
                               (long)Lc->g.ntot);
            src = dst;
        }
        for (int k = k0; k < L->post; ++k) {
            double *dst = (k == L->post - 1) ? out : (src == L->x.p ? L->x2.p : L->x.p);
            hx(l, src);
            hipLaunchKernelGGL((k_bamg_smooth<NB, false>), gr, bl, 0, c->stream, Ld, bl_, (const double *)src, dst);
            src = dst;
        }
    }
    TP_HIP(hipGetLastError());
}

void bamg_destroy(BAmg *amg) { delete amg; }

int bamg_levels(const BAmg *amg) { return (int)amg->lv.size(); }
int bamg_dist_levels(const BAmg *amg) { return amg ? amg->dist_levels : 0; }
const std::vector<int> &bamg_sched(const BAmg *amg) { return amg->sched; }

}  // namespace tp

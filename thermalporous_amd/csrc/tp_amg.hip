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
//   Cycle shape: V(nu,nu) on the first amg_full_levels levels, V(coarse_pre, coarse_post) = V(0,1) below,
//   V(0, tail_post) on the levels of <= 1024 cells; every second level in between is a pure transfer level
//   (amg_mid_skip) folded into its parent's launches; damped Jacobi; dense inverse on the coarsest grid.
//
// Below the first three levels the V-cycle is bound by the ~5 us floor of a dependent kernel, not by bytes, so it
// is organised to minimise launches AND dependent memory round trips inside them:
//   * levels >= 200 000 cells: separate streaming kernels (pre-smoothing pair, residual, restriction,
//     prolongation, sweeps), XCD-aware block order, 75-80 % of HBM peak each;
//   * smaller levels: fused kernels -- [residual + restriction] and [prolongation + first post-sweep]; the cell
//     functions are branch-free (clamped addresses, unconditional loads, arithmetic selects) so that every load
//     of a kernel is issued in one batch;
//   * all levels at or below `tail_cells` cells run inside ONE single-workgroup kernel (down-sweep, dense coarse
//     solve, up-sweep) with workgroup barriers between phases, vectors in LDS, read-only arrays touched up
//     front -- for the small 2-D configurations the whole V-cycle is a single launch.
// Multi-GPU: levels [0, dist_levels) are this rank's slab of the global level (halo exchange in front of every
// kernel that reads across the slab boundary); the rest of the hierarchy is gathered and replicated.
#include "tp_common.hpp"
#include <algorithm>
#include <cstdlib>

namespace tp {

// Operator storage type R: double, or float (amg_single: the AMG is only a preconditioner inside the flexible
// Krylov method, so its operators, weights and inverse diagonals can be stored in fp32 -- 7 of the ~11 planes
// every smoothing sweep streams -- while all vectors and all arithmetic stay fp64).
template <class R>
struct StencilT {
    R *base;
    long slot_stride;
    __host__ __device__ const R *slot(int s) const { return base + (long)s * slot_stride; }
};

static inline dim3 grid_for(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

__device__ __forceinline__ void cell_ijk(const GridDev &g, long tid, int &i0, int &i1, int &i2) {
    // 32-bit unsigned divisions (every slab has fewer than 2^31 cells, tp_create): a 64-bit division is ~150 instructions on
    // this ISA, and the few-thousand-cell levels of a V-cycle are bound by exactly that kind of per-cell index arithmetic
    const unsigned t = (unsigned)tid, np = (unsigned)g.np, n0 = (unsigned)g.n0;
    const unsigned q2 = t / np, rem = t - q2 * np, q1 = rem / n0;
    i2 = (int)q2;
    i1 = (int)q1;
    i0 = (int)(rem - q1 * n0);
}

// Multi-GPU: along the slab axis (2) the C points are the even GLOBAL planes, so a slab that starts on an odd
// plane is shifted by one; its F neighbours / C parents across the slab boundary live in the halo planes.
__device__ __forceinline__ int par_of(const GridDev &gf, int a) { return a == 2 ? (gf.off2 & 1) : 0; }
__device__ __forceinline__ bool open_lo(const GridDev &g, int a) { return a == 2 && g.nb_lo; }
__device__ __forceinline__ bool open_hi(const GridDev &g, int a) { return a == 2 && g.nb_hi; }

// ---- set-up kernels --------------------------------------------------------------------------------
// interpolation weights of every cell w.r.t. axis a (only odd cells are used) + invd = omega/diag
template <class R>
__global__ void k_amg_weights(GridDev g, StencilT<R> A, int axis, double omega, R *wm, R *wp, R *invd,
                              unsigned long long *ratio_slots) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = tid < g.nown;
    const long c = g.np + (in ? tid : 0);
    const double a0 = A.slot(0)[c];
    if (in) invd[c] = (R)(omega / a0);
    if (ratio_slots) {
        // dominance ratio sum_{s>=1}|a_s| / |a_0| of the row, max over the level (tp_options.amg_dom_tau): wave max by
        // shuffles, workgroup max through LDS, then ONE atomic per WORKGROUP on one of 64 slots (non-negative doubles order
        // like their bit patterns).  One atomic per wave -- 17 500 on 64 addresses for C4's level 0 -- serialised in the L2
        // and made this kernel 54 us instead of the 15 us its 90 MB take.
        __shared__ double wmax[16];
        double so = 0.0;
#pragma unroll
        for (int s = 1; s < 7; ++s) so += fabs((double)A.slot(s)[c]);
        double r = in ? so / fabs(a0) : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) r = fmax(r, __shfl_down(r, o, 64));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = r;
        __syncthreads();
        if (threadIdx.x == 0) {
            double m = wmax[0];
            for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmax(m, wmax[w]);
            atomicMax(&ratio_slots[blockIdx.x & 63], (unsigned long long)__double_as_longlong(m));
        }
    }
    if (!in || axis < 0) return;
    double cc = a0;
#pragma unroll
    for (int s = 1; s < 7; ++s)
        if ((s - 1) / 2 != axis) cc += A.slot(s)[c];
    wm[c] = (R)(-A.slot(1 + 2 * axis)[c] / cc);
    wp[c] = (R)(-A.slot(2 + 2 * axis)[c] / cc);
}

// coarse operator: one thread per coarse cell
template <class R>
__global__ void k_amg_coarsen(GridDev gf, GridDev gc, StencilT<R> A, int axis, const R *wm, const R *wp, R *Ac,
                              long cstride) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= gc.nown) return;
    int I[3];
    cell_ijk(gc, tid, I[0], I[1], I[2]);
    const long cc = gc.np + tid;
    int F[3] = {I[0], I[1], I[2]};
    F[axis] = 2 * I[axis] + par_of(gf, axis);
    const int nfa = axis == 0 ? gf.n0 : (axis == 1 ? gf.n1 : gf.n2);
    const long stride = axis == 0 ? 1 : (axis == 1 ? gf.n0 : gf.np);
    const long f = gf.np + (long)F[0] + (long)gf.n0 * F[1] + gf.np * F[2];
    const bool hm = F[axis] - 1 >= 0 || open_lo(gf, axis), hp = F[axis] + 1 < nfa || open_hi(gf, axis);
    const long gm = hm ? f - stride : f, gp = hp ? f + stride : f;      // clamped: every load below is unconditional
    const double Pm = hm ? (double)wp[gm] : 0.0;     // P[g-, I] = w+(g-)
    const double Pp = hp ? (double)wm[gp] : 0.0;     // P[g+, I] = w-(g+)
    const double Wm = hm ? (double)wm[gm] : 0.0, Wp = hp ? (double)wp[gp] : 0.0;
    double rho_f = 0.0, rho_m = 0.0, rho_p = 0.0;
    double out[7];
    double offsum = 0.0;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const double af = A.slot(s)[f];
        const double lm = A.slot(s)[gm], lp = A.slot(s)[gp];
        const double am = hm ? lm : 0.0, ap = hp ? lp : 0.0;
        rho_f += af; rho_m += am; rho_p += ap;
        if (s == 0) continue;
        double v;
        if (s == 1 + 2 * axis)      v = af * Wm;
        else if (s == 2 + 2 * axis) v = af * Wp;
        else                        v = af + Pm * am + Pp * ap;
        out[s] = v;
        offsum += v;
    }
    out[0] = -offsum + rho_f + Pm * rho_m + Pp * rho_p;
#pragma unroll
    for (int s = 0; s < 7; ++s) Ac[(long)s * cstride + cc] = (R)out[s];
}

// coarsest grid: dense inverse by Gauss-Jordan in one workgroup (diagonally dominant: no pivoting)
template <class R>
__global__ void k_amg_dense_inverse(GridDev g, StencilT<R> A, int n, double *M, double *Minv) {
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
            Minv[(long)r * n + q] -= fct * Minv[(long)p * n + q];
            if (q != p) M[(long)r * n + q] -= fct * M[(long)p * n + q];
        }
        __syncthreads();
        for (int r = threadIdx.x; r < n; r += blockDim.x)
            if (r != p) M[(long)r * n + p] = 0.0;
        __syncthreads();
    }
}

// same, with the augmented matrix [M | I] held in LDS (n <= 64: 64 KiB, rows padded to 64 columns so that an element index
// splits into (row, column) by shift and mask) -- no global round trips, 1024 threads, three barriers per pivot: the pivot
// column is copied aside first, so the rank-1 update of a pivot is ONE phase for both halves.  Same operations in the same
// order as the global-memory kernel (bit-identical inverse); C4: 136 -> ~30 us per hierarchy and set-up.
template <class R>
__global__ __launch_bounds__(1024) void k_amg_dense_inverse_lds(GridDev g, StencilT<R> A, int n, double *Minv_out) {
    __shared__ double M[64 * 64];
    __shared__ double I[64 * 64];
    __shared__ double col[64];
    const int t = threadIdx.x, T = blockDim.x;
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    for (int e = t; e < 64 * 64; e += T) { M[e] = 0.0; I[e] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0; }
    __syncthreads();
    for (int r = t; r < n; r += T) {
        const long c = g.np + r;
        int i0, i1, i2;
        cell_ijk(g, r, i0, i1, i2);
        const bool has[7] = {true, i0 > 0, i0 < g.n0 - 1, i1 > 0, i1 < g.n1 - 1, i2 > 0, i2 < g.n2 - 1};
        for (int s = 0; s < 7; ++s)
            if (has[s]) M[r * 64 + (int)(r + off[s])] += (double)A.slot(s)[c];
    }
    __syncthreads();
    for (int p = 0; p < n; ++p) {
        if (t < n) col[t] = M[t * 64 + p];
        __syncthreads();
        const double piv = col[p];
        if (t < n) M[p * 64 + t] /= piv;
        else if (t >= 64 && t < 64 + n) I[p * 64 + (t - 64)] /= piv;
        __syncthreads();
        // eliminate column p from every other row: thread (r, q) updates both halves (M[r][p] becomes fct - fct * 1 = 0)
        for (int e = t; e < n * 64; e += T) {
            const int r = e >> 6, q = e & 63;
            if (r == p || q >= n) continue;
            const double fct = col[r];
            I[e] -= fct * I[p * 64 + q];
            M[e] = (q == p) ? 0.0 : M[e] - fct * M[p * 64 + q];
        }
        __syncthreads();
    }
    for (int e = t; e < n * n; e += T) Minv_out[e] = I[(e / n) * 64 + e % n];
}

// The MFMA experiment of north_star ("MFMA only in the batched small dense block factor/solve"): the same inverse as a BLOCKED
// Gauss-Jordan, four pivots at a time, whose trailing update [M | I] -= C R (C: the 64 x 4 multipliers, R: the 4 x 128 scaled
// pivot rows) is a rank-4 GEMM on v_mfma_f64_16x16x4_f64 -- 4 x 8 output tiles of 16 x 16, two per wave of a 1024-thread
// workgroup.  Operand maps (cdna_hip_programming.md): A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15],
// C/D[row = (lane >> 4) + 4 reg][col = lane & 15].  16 block steps of three barriers instead of 64 pivots of three; not
// bit-identical with the scalar kernel (different association), same inverse to round-off.  The default since it won (31 against 58 us on C4);
// TP_AMG_DENSE_MFMA=0 selects the scalar kernel; the measured comparison is in DESIGN.md 4.5.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
template <class R>
__global__ __launch_bounds__(1024) void k_amg_dense_inverse_mfma(GridDev g, StencilT<R> A, int n, double *Minv_out) {
    extern __shared__ double lds[];
    double *X = lds;                       // [64][128]: [M | I], row stride 128
    double *Rb = X + 64 * 128;             // [4][128] scaled pivot rows
    double *Cb = Rb + 4 * 128;             // [64][4] multipliers (pivot rows: 0)
    double *Pi = Cb + 64 * 4;              // [4][4] inverse of the pivot block
    const int t = threadIdx.x, T = blockDim.x, lane = t & 63, wave = t >> 6;
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    for (int e = t; e < 64 * 128; e += T) {
        const int r = e >> 7, q = e & 127;
        X[e] = (q == r + 64 || (q == r && r >= n)) ? 1.0 : 0.0;          // identity half; unit diagonal in the padding rows
    }
    __syncthreads();
    for (int r = t; r < n; r += T) {
        const long c = g.np + r;
        int i0, i1, i2;
        cell_ijk(g, r, i0, i1, i2);
        const bool has[7] = {true, i0 > 0, i0 < g.n0 - 1, i1 > 0, i1 < g.n1 - 1, i2 > 0, i2 < g.n2 - 1};
        for (int s = 0; s < 7; ++s)
            if (has[s]) X[r * 128 + (int)(r + off[s])] += (double)A.slot(s)[c];
    }
    __syncthreads();
    for (int p0 = 0; p0 < 64; p0 += 4) {
        // (1) inverse of the 4 x 4 pivot block: Gauss-Jordan in the registers of one thread
        if (t == 0) {
            double P[4][4], Q[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { P[i][j] = X[(p0 + i) * 128 + p0 + j]; Q[i][j] = i == j ? 1.0 : 0.0; }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const double ip = 1.0 / P[p][p];
#pragma unroll
                for (int j = 0; j < 4; ++j) { P[p][j] *= ip; Q[p][j] *= ip; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i == p) continue;
                    const double f = P[i][p];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { P[i][j] -= f * P[p][j]; Q[i][j] -= f * Q[p][j]; }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Pi[i * 4 + j] = Q[i][j];
        }
        __syncthreads();
        // (2) R = P^-1 [M | I](pivot rows, :) and the multipliers C = M(:, pivot columns), zero in the pivot rows
        if (t < 512) {
            const int i = t >> 7, q = t & 127;
            double v = 0.0;
#pragma unroll
            for (int m = 0; m < 4; ++m) v += Pi[i * 4 + m] * X[(p0 + m) * 128 + q];
            Rb[i * 128 + q] = v;
        } else if (t < 768) {
            const int r = (t - 512) >> 2, j = t & 3;
            Cb[r * 4 + j] = (r >= p0 && r < p0 + 4) ? 0.0 : X[r * 128 + p0 + j];
        }
        __syncthreads();
        // (3) [M | I] -= C R on the matrix cores: wave w owns the tiles (row tile w >> 2, column tiles 2 (w & 3), + 1);
        //     the pivot rows (C = 0 there) take R itself
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rt = wave >> 2, ct = 2 * (wave & 3) + h;
            const double a = -Cb[(rt * 16 + (lane & 15)) * 4 + (lane >> 4)];
            const double b = Rb[(lane >> 4) * 128 + ct * 16 + (lane & 15)];
            mfma_d4 acc;
            double *xp = X + (rt * 16 + (lane >> 4)) * 128 + ct * 16 + (lane & 15);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = xp[q * 4 * 128];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = rt * 16 + (lane >> 4) + 4 * q;
                xp[q * 4 * 128] = (row >= p0 && row < p0 + 4) ? Rb[(row - p0) * 128 + ct * 16 + (lane & 15)] : acc[q];
            }
        }
        __syncthreads();
    }
    for (int e = t; e < n * n; e += T) Minv_out[e] = X[(e / n) * 128 + 64 + e % n];
}

// ---- per-cell building blocks of the cycle (shared by the per-level kernels and the tail kernel) ------
template <class R>
struct LevelDevT {
    GridDev g;
    StencilT<R> op;
    const R *invd, *wm, *wp;
    double *b, *x, *x2, *e;
    int axis;
    int pre, post;           // smoothing sweeps of this level: V(pre, post); post == 0: pure transfer level
    int pad_;
};

__device__ __forceinline__ void nb_offsets(const GridDev &g, long (&off)[7]) {
    off[0] = 0; off[1] = -1; off[2] = 1; off[3] = -(long)g.n0; off[4] = g.n0; off[5] = -g.np; off[6] = g.np;
}

// two damped-Jacobi sweeps from a zero initial guess:  x1 = invd b ;  x2 = x1 + invd (b - A x1)
template <class R>
__device__ __forceinline__ double pre2_cell(const LevelDevT<R> &L, const double *__restrict__ b, long tid) {
    const long c = L.g.np + tid;
    long off[7];
    nb_offsets(L.g, off);
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const long n = c + off[k];
        s += L.op.slot(k)[c] * (L.invd[n] * b[n]);
    }
    const double x1 = L.invd[c] * b[c];
    return x1 + L.invd[c] * (b[c] - s);
}

template <class R>
__device__ __forceinline__ double jacobi_cell(const LevelDevT<R> &L, const double *__restrict__ b,
                                              const double *__restrict__ x, long tid) {
    const long c = L.g.np + tid;
    long off[7];
    nb_offsets(L.g, off);
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += L.op.slot(k)[c] * x[c + off[k]];
    return x[c] + L.invd[c] * (b[c] - s);
}

template <class R>
__device__ __forceinline__ double resid_at(const LevelDevT<R> &L, const double *__restrict__ b,
                                           const double *__restrict__ x, long c) {
    long off[7];
    nb_offsets(L.g, off);
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += L.op.slot(k)[c] * x[c + off[k]];
    return b[c] - s;
}

// (P^T (b - A x)) at coarse cell tidc
template <class R>
__device__ __forceinline__ double resid_restrict_cell(const LevelDevT<R> &Lf, const GridDev &gc,
                                                      const double *__restrict__ b, const double *__restrict__ x,
                                                      long tidc) {
    int I[3];
    cell_ijk(gc, tidc, I[0], I[1], I[2]);
    const int a = Lf.axis;
    int F[3] = {I[0], I[1], I[2]};
    const GridDev &gf = Lf.g;
    F[a] = 2 * I[a] + par_of(gf, a);       // (slab-axis levels of a multi-GPU hierarchy use the unfused path)
    const int nfa = a == 0 ? gf.n0 : (a == 1 ? gf.n1 : gf.n2);
    const long stride = a == 0 ? 1 : (a == 1 ? gf.n0 : gf.np);
    const long f = gf.np + (long)F[0] + (long)gf.n0 * F[1] + gf.np * F[2];
    const bool hm = F[a] - 1 >= 0, hp = F[a] + 1 < nfa;
    const long fm = hm ? f - stride : f, fp = hp ? f + stride : f;
    const double vm = (double)Lf.wp[fm] * resid_at(Lf, b, x, fm), vp = (double)Lf.wm[fp] * resid_at(Lf, b, x, fp);
    return resid_at(Lf, b, x, f) + (hm ? vm : 0.0) + (hp ? vp : 0.0);
}

// (P^T r) at coarse cell tidc, r given as a vector
template <class R>
__device__ __forceinline__ double restrict_cell(const LevelDevT<R> &Lf, const GridDev &gc, const double *__restrict__ r,
                                                long tidc) {
    int I[3];
    cell_ijk(gc, tidc, I[0], I[1], I[2]);
    const int a = Lf.axis;
    int F[3] = {I[0], I[1], I[2]};
    const GridDev &gf = Lf.g;
    F[a] = 2 * I[a] + par_of(gf, a);
    const int nfa = a == 0 ? gf.n0 : (a == 1 ? gf.n1 : gf.n2);
    const long stride = a == 0 ? 1 : (a == 1 ? gf.n0 : gf.np);
    const long f = gf.np + (long)F[0] + (long)gf.n0 * F[1] + gf.np * F[2];
    // branch-free (see prolong_jacobi_cell): clamped addresses, unconditional loads, selected results
    const bool hm = F[a] - 1 >= 0 || open_lo(gf, a), hp = F[a] + 1 < nfa || open_hi(gf, a);
    const long fm = hm ? f - stride : f, fp = hp ? f + stride : f;
    const double vm = (double)Lf.wp[fm] * r[fm], vp = (double)Lf.wm[fp] * r[fp];
    return r[f] + (hm ? vm : 0.0) + (hp ? vp : 0.0);
}

// coarse-grid correction fused with the first post-smoothing sweep:
//   x' = x + P ec ;  out = x' + invd (b - A x')
// BRANCH-FREE: (P ec) at each of the 7 stencil points is alpha*ec[left] + beta*ec[right] with (alpha, beta) =
// (1, 0) at a C point and (w-, w+) at an F point, every address clamped into its array and every load issued
// unconditionally.  Written with `if (C point) return ...` per neighbour the loads sit behind divergent branches
// and execute as seven dependent round trips (~5 us of a 9.6 us kernel on the 10^5-cell levels); like this they are
// one batch.  Neighbours that do not exist are multiplied by their zero stencil coefficient (every level keeps
// exact zeros towards physical boundaries), so no existence tests are needed -- only memory-safe indices.
template <class R, bool MASK = true>
__device__ __forceinline__ double prolong_val(const LevelDevT<R> &Lf, const GridDev &gc, const double *__restrict__ ec,
                                              int F0, int F1, int F2) {
    const GridDev &g = Lf.g;
    const int a = Lf.axis;
    const int p = par_of(g, a);
    const int Fa = a == 0 ? F0 : (a == 1 ? F1 : F2);
    const int Ia = (Fa - p) >> 1;
    const bool isF = (Fa + p) & 1;
    const int I0 = a == 0 ? Ia : F0, I1 = a == 1 ? Ia : F1, I2 = a == 2 ? Ia : F2;
    const long ci = gc.np + (long)I0 + (long)gc.n0 * I1 + gc.np * I2;
    const long cf = g.np + (long)F0 + (long)g.n0 * F1 + g.np * F2;
    const long cs = a == 0 ? 1 : (a == 1 ? gc.n0 : gc.np);
    const int nca = a == 0 ? gc.n0 : (a == 1 ? gc.n1 : gc.n2);
    const bool hasR = isF && (Ia + 1 < nca || open_hi(gc, a));
    const double wm = (double)Lf.wm[cf], wp = (double)Lf.wp[cf];
    const double e0 = ec[ci], e1 = ec[hasR ? ci + cs : ci];
    // 0/1 FACTORS, not selects: `isF ? expr : e0` lets the compiler sink the weight loads into a branch on isF, and the seven
    // calls of a stencil then execute as seven dependent load batches (29 branches and 42 vmcnt waits in k_amg_prolong2_jacobi's
    // ISA: 8.3 us for 4096 cells).  A product with a loaded value cannot be skipped; 1.0 * x and x + 0.0 * y are exact.
    // (MASK = false: the select form, for the bandwidth-bound top levels, where the loads a C point can skip are traffic:
    // k_amg_prolong_add 7.6 -> 9.5 us on C4's level 0 with factors, k_amg_prolong2_jacobi 8.3 -> 5.9 us on the small levels)
    if constexpr (!MASK) return isF ? wm * e0 + (hasR ? wp * e1 : 0.0) : e0;
    const double fF = isF ? 1.0 : 0.0, fR = hasR ? 1.0 : 0.0;
    return (1.0 - fF) * e0 + fF * (wm * e0 + fR * (wp * e1));
}

template <class R>
__device__ __forceinline__ double prolong_jacobi_cell(const LevelDevT<R> &Lf, const GridDev &gc,
                                                      const double *__restrict__ b, const double *x,
                                                      const double *__restrict__ ec, long tid) {
    const GridDev &g = Lf.g;
    int i0, i1, i2;
    cell_ijk(g, tid, i0, i1, i2);
    const long c = g.np + tid;
    // memory-safe neighbour coordinates (halo planes along axis 2 are part of the arrays)
    const int m0 = max(i0 - 1, 0), p0 = min(i0 + 1, g.n0 - 1);
    const int m1 = max(i1 - 1, 0), p1 = min(i1 + 1, g.n1 - 1);
    const int m2 = max(i2 - 1, g.nb_lo ? -1 : 0), p2 = min(i2 + 1, g.nb_hi ? g.n2 : g.n2 - 1);
    double v[7];
    v[0] = prolong_val(Lf, gc, ec, i0, i1, i2);
    v[1] = prolong_val(Lf, gc, ec, m0, i1, i2);
    v[2] = prolong_val(Lf, gc, ec, p0, i1, i2);
    v[3] = prolong_val(Lf, gc, ec, i0, m1, i2);
    v[4] = prolong_val(Lf, gc, ec, i0, p1, i2);
    v[5] = prolong_val(Lf, gc, ec, i0, i1, m2);
    v[6] = prolong_val(Lf, gc, ec, i0, i1, p2);
    if (x) {            // (uniform over the launch) x == nullptr: no pre-smoothing, the level's iterate is zero
        const long cn[7] = {c, c + (m0 - i0), c + (p0 - i0), c + (long)g.n0 * (m1 - i1), c + (long)g.n0 * (p1 - i1),
                            c + g.np * (m2 - i2), c + g.np * (p2 - i2)};
#pragma unroll
        for (int k = 0; k < 7; ++k) v[k] += x[cn[k]];
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += (double)Lf.op.slot(k)[c] * v[k];
    return v[0] + (double)Lf.invd[c] * (b[c] - s);
}

// ---- two levels at once: a smoothed level l followed by a pure transfer level l+1 (amg_mid_skip) ------------
// Replicated grids only (no slab parity).  Branch-free like the single-level versions.
// (P_l^T r) at the level-(l+1) cell (I0,I1,I2), by coordinates
template <class R>
__device__ __forceinline__ double restrict_at(const LevelDevT<R> &Lf, const double *__restrict__ r, int I0, int I1, int I2) {
    const GridDev &gf = Lf.g;
    const int a = Lf.axis;
    const int F0 = a == 0 ? 2 * I0 : I0, F1 = a == 1 ? 2 * I1 : I1, F2 = a == 2 ? 2 * I2 : I2;
    const int Fa = a == 0 ? F0 : (a == 1 ? F1 : F2);
    const int nfa = a == 0 ? gf.n0 : (a == 1 ? gf.n1 : gf.n2);
    const long stride = a == 0 ? 1 : (a == 1 ? gf.n0 : gf.np);
    const long f = gf.np + (long)F0 + (long)gf.n0 * F1 + gf.np * F2;
    const bool hm = Fa - 1 >= 0, hp = Fa + 1 < nfa;
    const long fm = hm ? f - stride : f, fp = hp ? f + stride : f;
    const double vm = (double)Lf.wp[fm] * r[fm], vp = (double)Lf.wm[fp] * r[fp];
    return r[f] + (hm ? vm : 0.0) + (hp ? vp : 0.0);
}
// (P_{l+1}^T P_l^T r) at the level-(l+2) cell tid2
template <class R>
__device__ __forceinline__ double restrict2_cell(const LevelDevT<R> &L0, const LevelDevT<R> &L1, const GridDev &g2,
                                                 const double *__restrict__ r, long tid2) {
    int I[3];
    cell_ijk(g2, tid2, I[0], I[1], I[2]);
    const GridDev &g1 = L1.g;
    const int a = L1.axis;
    int M[3] = {I[0], I[1], I[2]};
    M[a] = 2 * I[a];
    const int n1a = a == 0 ? g1.n0 : (a == 1 ? g1.n1 : g1.n2);
    const long stride = a == 0 ? 1 : (a == 1 ? g1.n0 : g1.np);
    const long m = g1.np + (long)M[0] + (long)g1.n0 * M[1] + g1.np * M[2];
    const bool hm = M[a] - 1 >= 0, hp = M[a] + 1 < n1a;
    int Mm[3] = {M[0], M[1], M[2]}, Mp[3] = {M[0], M[1], M[2]};
    Mm[a] -= hm ? 1 : 0;
    Mp[a] += hp ? 1 : 0;
    const double v0 = restrict_at(L0, r, M[0], M[1], M[2]);
    const double vm = (double)L1.wp[hm ? m - stride : m] * restrict_at(L0, r, Mm[0], Mm[1], Mm[2]);
    const double vp = (double)L1.wm[hp ? m + stride : m] * restrict_at(L0, r, Mp[0], Mp[1], Mp[2]);
    return v0 + (hm ? vm : 0.0) + (hp ? vp : 0.0);
}
// (P_l P_{l+1} e2) at the level-l cell (F0,F1,F2) (clamped coordinates)
template <class R>
__device__ __forceinline__ double prolong2_val(const LevelDevT<R> &L0, const LevelDevT<R> &L1, const GridDev &g2,
                                               const double *__restrict__ e2, int F0, int F1, int F2) {
    const GridDev &g = L0.g, &g1 = L1.g;
    const int a = L0.axis;
    const int Fa = a == 0 ? F0 : (a == 1 ? F1 : F2);
    const int Ia = Fa >> 1;
    const bool isF = Fa & 1;
    const int n1a = a == 0 ? g1.n0 : (a == 1 ? g1.n1 : g1.n2);
    const bool hasR = isF && (Ia + 1 < n1a);
    const int I0 = a == 0 ? Ia : F0, I1 = a == 1 ? Ia : F1, I2 = a == 2 ? Ia : F2;
    const int J0 = I0 + (a == 0 && hasR), J1 = I1 + (a == 1 && hasR), J2 = I2 + (a == 2 && hasR);
    const long cf = g.np + (long)F0 + (long)g.n0 * F1 + g.np * F2;
    const double wm = (double)L0.wm[cf], wp = (double)L0.wp[cf];
    const double v0 = prolong_val(L1, g2, e2, I0, I1, I2), v1 = prolong_val(L1, g2, e2, J0, J1, J2);
    const double fF = isF ? 1.0 : 0.0, fR = hasR ? 1.0 : 0.0;          // (factors, not selects: see prolong_val)
    return (1.0 - fF) * v0 + fF * (wm * v0 + fR * (wp * v1));
}
// x' = P_l P_{l+1} e2 ;  out = x' + invd (b - A x')     (level l has no pre-smoothing)
template <class R>
__device__ __forceinline__ double prolong2_jacobi_cell(const LevelDevT<R> &L0, const LevelDevT<R> &L1, const GridDev &g2,
                                                       const double *__restrict__ b, const double *__restrict__ e2,
                                                       long tid) {
    const GridDev &g = L0.g;
    int i0, i1, i2;
    cell_ijk(g, tid, i0, i1, i2);
    const long c = g.np + tid;
    const int m0 = max(i0 - 1, 0), p0 = min(i0 + 1, g.n0 - 1);
    const int m1 = max(i1 - 1, 0), p1 = min(i1 + 1, g.n1 - 1);
    const int m2 = max(i2 - 1, 0), p2 = min(i2 + 1, g.n2 - 1);
    double v[7];
    v[0] = prolong2_val(L0, L1, g2, e2, i0, i1, i2);
    v[1] = prolong2_val(L0, L1, g2, e2, m0, i1, i2);
    v[2] = prolong2_val(L0, L1, g2, e2, p0, i1, i2);
    v[3] = prolong2_val(L0, L1, g2, e2, i0, m1, i2);
    v[4] = prolong2_val(L0, L1, g2, e2, i0, p1, i2);
    v[5] = prolong2_val(L0, L1, g2, e2, i0, i1, m2);
    v[6] = prolong2_val(L0, L1, g2, e2, i0, i1, p2);
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += (double)L0.op.slot(k)[c] * v[k];
    return v[0] + (double)L0.invd[c] * (b[c] - s);
}

// ---- per-level kernels (big levels) -----------------------------------------------------------------
template <class R>
__global__ __launch_bounds__(256) void k_amg_pre(LevelDevT<R> L, const double *b, int two, double *out) {
    const long tid = xcd_tid();
    if (tid >= L.g.nown) return;
    const long c = L.g.np + tid;
    out[c] = two ? pre2_cell(L, b, tid) : L.invd[c] * b[c];
}
template <class R>
__global__ __launch_bounds__(256) void k_amg_jacobi(LevelDevT<R> L, const double *b, const double *x, double *out) {
    const long tid = xcd_tid();
    if (tid >= L.g.nown) return;
    out[L.g.np + tid] = jacobi_cell(L, b, x, tid);
}
template <class R>
__global__ __launch_bounds__(256) void k_amg_resid_restrict(LevelDevT<R> Lf, GridDev gc, const double *b,
                                                            const double *x, double *rc) {
    const long tid = xcd_tid();
    if (tid >= gc.nown) return;
    rc[gc.np + tid] = resid_restrict_cell(Lf, gc, b, x, tid);
}
template <class R>
__global__ __launch_bounds__(256) void k_amg_prolong_jacobi(LevelDevT<R> Lf, GridDev gc, const double *b,
                                                            const double *x, const double *ec, double *out) {
    const long tid = xcd_tid();
    if (tid >= Lf.g.nown) return;
    out[Lf.g.np + tid] = prolong_jacobi_cell(Lf, gc, b, x, ec, tid);
}

template <class R>
__global__ __launch_bounds__(256) void k_amg_restrict2(LevelDevT<R> L0, LevelDevT<R> L1, GridDev g2, const double *r,
                                                       double *rc) {
    const long tid = xcd_tid();
    if (tid >= g2.nown) return;
    rc[g2.np + tid] = restrict2_cell(L0, L1, g2, r, tid);
}
template <class R>
__global__ __launch_bounds__(256) void k_amg_prolong2_jacobi(LevelDevT<R> L0, LevelDevT<R> L1, GridDev g2, const double *b,
                                                             const double *e2, double *out) {
    const long tid = xcd_tid();
    if (tid >= L0.g.nown) return;
    out[L0.g.np + tid] = prolong2_jacobi_cell(L0, L1, g2, b, e2, tid);
}

// unfused variants for the top levels, where the fused kernels are issue-bound rather than HBM-bound
template <class R>
__global__ __launch_bounds__(256) void k_amg_resid(LevelDevT<R> L, const double *__restrict__ b,
                                                   const double *__restrict__ x, double *__restrict__ r) {
    const long tid = xcd_tid();
    if (tid >= L.g.nown) return;
    r[L.g.np + tid] = resid_at(L, b, x, L.g.np + tid);
}
template <class R>
__global__ __launch_bounds__(256) void k_amg_restrict(LevelDevT<R> Lf, GridDev gc, const double *__restrict__ r,
                                                      double *__restrict__ rc) {
    const long tid = xcd_tid();
    if (tid >= gc.nown) return;
    rc[gc.np + tid] = restrict_cell(Lf, gc, r, tid);
}
// x = P ec: the up-sweep of a pure transfer level
template <class R>
__global__ __launch_bounds__(256) void k_amg_prolong_set(LevelDevT<R> Lf, GridDev gc, const double *__restrict__ ec,
                                                         double *__restrict__ out) {
    const long tid = xcd_tid();
    if (tid >= Lf.g.nown) return;
    int i0, i1, i2;
    cell_ijk(Lf.g, tid, i0, i1, i2);
    out[Lf.g.np + tid] = prolong_val<R, false>(Lf, gc, ec, i0, i1, i2);
}
template <class R>
__global__ __launch_bounds__(256) void k_amg_prolong_add(LevelDevT<R> Lf, GridDev gc, const double *__restrict__ ec,
                                                         double *x) {
    const long tid = xcd_tid();
    if (tid >= Lf.g.nown) return;
    int i0, i1, i2;
    cell_ijk(Lf.g, tid, i0, i1, i2);
    x[Lf.g.np + tid] += prolong_val<R, false>(Lf, gc, ec, i0, i1, i2);
}

// ---- the tail: all small levels in one workgroup --------------------------------------------------------
template <class R>
__global__ __launch_bounds__(1024) void k_amg_tail(const LevelDevT<R> *lv, int l0, int nlev_all, int trunc, int ncoarse,
                                                   const double *Minv, const double *b_top, double *e_top,
                                                   int lds_doubles) {
    // trunc >= l0: the hierarchy ends at level `trunc` with two damped-Jacobi sweeps instead of the dense solve
    const int nlev = trunc >= 0 ? trunc + 1 : nlev_all;
    extern __shared__ double dyn[];          // 4 x lds_doubles: the b, e, x, x2 vectors of every tail level
    constexpr int T = 1024;            // (the launch's block size, as a constant: no blockDim fetch -- tp_common.hpp:xcd_tid)
    const int t = threadIdx.x;
    // level descriptors live in LDS: every phase below starts with LDS reads, not a global round trip
    static_assert(sizeof(LevelDevT<R>) % sizeof(long) == 0, "LevelDev must be a whole number of words");
    __shared__ long slv_raw[40 * sizeof(LevelDevT<R>) / sizeof(long)];
    {
        const int wpl = (int)(sizeof(LevelDevT<R>) / sizeof(long));
        const long *src = reinterpret_cast<const long *>(lv);
        for (int i = l0 * wpl + t; i < nlev * wpl; i += T) slv_raw[i] = src[i];
        if (lds_doubles > 0)
            for (int i = t; i < 4 * lds_doubles; i += T) dyn[i] = 0.0;      // (halo planes must read as zero)
        __syncthreads();
    }
    if (lds_doubles > 0) {
        // the tail's vectors live in LDS: a phase boundary is then an LDS store + barrier + LDS load instead of a
        // global store that must drain (s_waitcnt vmcnt(0)) before the barrier and an L2 round trip after it.
        // Done by re-pointing the LDS copy of the level descriptors; the sweeps below do not change.
        if (t == 0) {
            LevelDevT<R> *w = reinterpret_cast<LevelDevT<R> *>(slv_raw);
            long off = 0;
            for (int l = l0; l < nlev; ++l) {
                w[l].b = dyn + off;
                w[l].e = dyn + lds_doubles + off;
                w[l].x = dyn + 2 * lds_doubles + off;
                w[l].x2 = dyn + 3 * lds_doubles + off;
                off += w[l].g.ntot;
            }
        }
        __syncthreads();
    }
    lv = reinterpret_cast<const LevelDevT<R> *>(slv_raw);
    // Touch every read-only array of the tail (operators, inverse diagonals, weights, the dense inverse) NOW, all
    // at once: they were written by the set-up long ago and have left the L2; otherwise each of the ~10 phases below
    // starts with its own HBM + TLB round trip (~3 us of a ~4 us phase).
    {
        double sink = 0.0;
        for (int l = l0; l < nlev; ++l) {
            const LevelDevT<R> &L = lv[l];
            for (long i = L.g.np + t; i < L.g.np + L.g.nown; i += T) {
#pragma unroll
                for (int k = 0; k < 7; ++k) sink += (double)L.op.slot(k)[i];
                sink += (double)L.invd[i];
                if (L.axis >= 0) sink += (double)L.wm[i] + (double)L.wp[i];
            }
        }
        if (trunc < 0)
            for (int i = t; i < ncoarse * ncoarse; i += T) sink += Minv[i];
        if (sink == 1.2345678e-300) e_top[0] = sink;       // never true: keeps the loads alive
    }
    // down-sweep
    for (int l = l0; l < nlev - 1; ++l) {
        const LevelDevT<R> L = lv[l];
        const GridDev gc = lv[l + 1].g;
        const double *b = (l == l0) ? b_top : L.b;
        double *rc = lv[l + 1].b;
        if (L.pre == 0) {                      // V(0,post): x = 0, the residual is b itself
            for (long i = t; i < gc.nown; i += T) rc[gc.np + i] = restrict_cell(L, gc, b, i);
            __syncthreads();
            continue;
        }
        double *cur = L.x, *oth = L.x2;
        for (long i = t; i < L.g.nown; i += T)
            cur[L.g.np + i] = L.pre >= 2 ? pre2_cell(L, b, i) : L.invd[L.g.np + i] * b[L.g.np + i];
        __syncthreads();
        for (int k = 2; k < L.pre; ++k) {
            for (long i = t; i < L.g.nown; i += T) oth[L.g.np + i] = jacobi_cell(L, b, cur, i);
            __syncthreads();
            double *tmp = cur; cur = oth; oth = tmp;
        }
        for (long i = t; i < gc.nown; i += T) rc[gc.np + i] = resid_restrict_cell(L, gc, b, cur, i);
        __syncthreads();
    }
    // coarsest grid: dense solve
    {
        const LevelDevT<R> Lc = lv[nlev - 1];
        const double *b = (nlev - 1 == l0) ? b_top : Lc.b;
        double *e = (nlev - 1 == l0) ? e_top : Lc.e;
        if (trunc >= 0) {                 // relaxation-only level: x = x1 + invd (b - A x1), x1 = invd b
            for (long i = t; i < Lc.g.nown; i += T) e[Lc.g.np + i] = pre2_cell(Lc, b, i);
            __syncthreads();
        } else {
        // 16 lanes per row: coalesced reads of the row, 4 independent products per lane, shuffle reduction
        // (one lane per row walked its 64 entries one dependent L2 round trip at a time: ~30 us of a 44 us kernel)
        const int grp = t >> 4, gl = t & 15;
        for (int r = grp; r < ncoarse; r += T >> 4) {
            double s = 0.0;
            for (int q = gl; q < ncoarse; q += 16) s += Minv[(long)r * ncoarse + q] * b[Lc.g.np + q];
            s += __shfl_xor(s, 8, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 1, 64);
            if (gl == 0) e[Lc.g.np + r] = s;
        }
        __syncthreads();
        }
    }
    // up-sweep
    for (int l = nlev - 2; l >= l0; --l) {
        const LevelDevT<R> L = lv[l];
        const GridDev gc = lv[l + 1].g;
        const double *b = (l == l0) ? b_top : L.b;
        double *out = (l == l0) ? e_top : L.e;
        const double *ec = lv[l + 1].e;
        // where the pre-smoothed iterate lives: x after an even number of extra sweeps, else x2; none if pre == 0
        const int extra = L.pre >= 2 ? L.pre - 2 : 0;
        const double *src = L.pre == 0 ? nullptr : ((extra % 2 == 0) ? L.x : L.x2);
        if (L.post == 0) {                          // pure transfer level (amg_mid_skip): x = P ec
            for (long i = t; i < L.g.nown; i += T) {
                int i0, i1, i2;
                cell_ijk(L.g, i, i0, i1, i2);
                out[L.g.np + i] = prolong_val(L, gc, ec, i0, i1, i2);
            }
            __syncthreads();
            continue;
        }
        double *dst = (L.post == 1) ? out : (src == L.x ? L.x2 : L.x);
        for (long i = t; i < L.g.nown; i += T) dst[L.g.np + i] = prolong_jacobi_cell(L, gc, b, src, ec, i);
        __syncthreads();
        for (int k = 1; k < L.post; ++k) {
            src = dst;
            dst = (k == L.post - 1) ? out : (src == L.x ? L.x2 : L.x);
            for (long i = t; i < L.g.nown; i += T) dst[L.g.np + i] = jacobi_cell(L, b, src, i);
            __syncthreads();
        }
    }
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

template <class R>
static LevelDevT<R> dev_of(const AmgLevel *L, int level, const tp_options &o) {
    LevelDevT<R> d;
    const int nu = std::max(1, o.amg_nu);
    const bool full = level < o.amg_full_levels;
    // cycle shape: V(nu,nu) on the first levels, V(coarse_pre, coarse_post) below, and V(coarse_pre, tail_post) on
    // the levels of <= 1024 cells (a property of the level size, so that the oracle can mirror it)
    const bool small = L->g.np * (long)L->g.gn2 <= 1024;
    d.pre = full ? nu : std::max(0, o.amg_coarse_pre);
    d.post = full ? nu : std::max(1, small ? o.amg_tail_post : o.amg_coarse_post);
    // mid levels (neither full nor small): every second one is a pure transfer level -- the hierarchy then coarsens
    // two directions per smoothing level there, which costs no Krylov iterations (454 -> 456 on C4) and lets the
    // cycle fuse the two transfers
    if (o.amg_mid_skip && !full && !small && ((level - o.amg_full_levels) & 1)) { d.pre = 0; d.post = 0; }
    d.pad_ = 0;
    d.g = L->g;
    d.op.base = (R *)L->op.base;
    d.op.slot_stride = L->op.slot_stride;
    d.invd = (const R *)L->invd.p; d.wm = (const R *)L->wm.p; d.wp = (const R *)L->wp.p;
    d.b = L->b.p; d.x = L->x.p; d.x2 = L->x2.p; d.e = L->e.p;
    d.axis = L->axis;
    return d;
}

// g0: the grid the hierarchy coarsens -- the slab itself on one GPU, the GLOBAL grid on several.  There the top
// levels (more than amg_gather_cells cells, at least two planes on every rank) stay distributed over the slabs
// and the rest of the hierarchy is built on the gathered global grid, replicated on every rank; a problem
// smaller than amg_gather_cells is replicated from the top (dist_levels = 0: its V-cycle is launch-latency
// bound and would only get slower with a halo exchange between sweeps).
void amg_build(tp_ctx *c, Amg *&amg, const GridDev &g0, const double strength[3]) {
    delete amg;
    amg = new Amg();
    c->graph_epoch++;            // a new hierarchy may reuse the old one's addresses: never replay graphs across a rebuild
    amg->single = c->opt.amg_single != 0;
    const int n[3] = {g0.n0, g0.n1, g0.n2};
    amg->sched = schedule(n, strength, std::max(1, c->opt.amg_min_cells));
    const int nranks = c->dist ? c->grid.nranks : 1, me = c->dist ? c->grid.rank : 0;
    std::vector<std::pair<int, int>> cur(nranks);          // owned global planes of every rank at the current level
    for (int r = 0; r < nranks; ++r) {
        if (c->dist) slab_of(c, r, cur[r].first, cur[r].second);
        else cur[r] = {0, n[2]};
    }
    const long gather_cells = c->gather_override != -2 ? c->gather_override : (long)c->opt.amg_gather_cells;
    bool still = c->dist && gather_cells >= 0;
    int m[3] = {n[0], n[1], n[2]};
    for (size_t l = 0; l <= amg->sched.size(); ++l) {
        AmgLevel *L = new AmgLevel();
        if (still) {
            int minp = 1 << 30;
            for (auto &q : cur) minp = std::min(minp, q.second - q.first);
            still = (long)m[0] * m[1] * m[2] > gather_cells && minp >= 2 && l < amg->sched.size();
            if (still) amg->dist_levels = (int)l + 1;
        }
        amg->ranges.push_back(cur);
        // a distributed level is this rank's slab of the level (live halo planes towards the neighbours);
        // every other level is the whole box with dead halo planes
        L->g = still ? make_grid(m[0], m[1], cur[me].second - cur[me].first, m[2], cur[me].first)
                     : make_grid(m[0], m[1], m[2], m[2], 0);
        if (l < amg->sched.size()) {
            L->axis = amg->sched[l];
            m[L->axis] = (m[L->axis] + 1) / 2;
            if (L->axis == 2)
                for (auto &q : cur) q = {(q.first + 1) / 2, (q.second + 1) / 2};     // even global planes survive
        }
        amg->lv.push_back(L);
    }
    // carve every level's buffers out of one arena (32-double = 256-byte aligned slices)
    {
        auto pad = [](size_t v) { return (v + 31) & ~(size_t)31; };
        size_t total = 0;
        for (size_t l = 0; l < amg->lv.size(); ++l) {
            const size_t nt = pad((size_t)amg->lv[l]->g.ntot);
            total += ((l > 0 || amg->single) ? 7 : 0) * nt + 5 * nt + (amg->lv[l]->axis >= 0 ? 2 * nt : 0);
        }
        amg->arena.alloc(total);
        double *q = amg->arena.p;
        auto take = [&](DBuf<double> &d, size_t nd, size_t step) { d.view(q, nd); q += step; };
        for (size_t l = 0; l < amg->lv.size(); ++l) {
            AmgLevel *L = amg->lv[l];
            const size_t nt = (size_t)L->g.ntot, ntp = pad(nt);
            // operator/weight buffers are sized for doubles and hold floats when amg_single (slot stride in elements)
            if (l > 0 || amg->single) {
                take(L->A, 7 * nt, 7 * ntp);
                L->op.base = L->A.p;
                L->op.slot_stride = (long)nt;
            }
            take(L->invd, nt, ntp);
            take(L->b, nt, ntp); take(L->x, nt, ntp); take(L->x2, nt, ntp); take(L->e, nt, ntp);
            if (L->axis >= 0) { take(L->wm, nt, ntp); take(L->wp, nt, ntp); }
        }
    }
    amg->ncoarse = (int)amg->lv.back()->g.nown;
    TP_REQUIRE(amg->ncoarse <= 1024, "coarsest AMG grid too large for the dense solve");
    amg->ratio_dev.alloc(64 * 8);                       // 8 levels x 64 slots is more than amg_full_levels ever needs
    TP_HIP(hipHostMalloc((void **)&amg->ratio_host, 64 * 8 * sizeof(double)));
    TP_HIP(hipEventCreateWithFlags(&amg->ev_ratio, hipEventDisableTiming));
    amg->coarse_inv.alloc((size_t)2 * amg->ncoarse * amg->ncoarse);
    // first level handled by the single-workgroup tail kernel
    const long tail_cells = getenv("TP_AMG_TAIL_CELLS") ? atol(getenv("TP_AMG_TAIL_CELLS")) : 1024;
    amg->fuse_below = getenv("TP_AMG_FUSE_BELOW") ? atol(getenv("TP_AMG_FUSE_BELOW")) : 200000;
    amg->tail_level = (int)amg->lv.size() - 1;
    for (size_t l = (size_t)amg->dist_levels; l < amg->lv.size(); ++l)
        if (amg->lv[l]->g.nown <= tail_cells) { amg->tail_level = (int)l; break; }
    amg->lvdev.alloc(amg->lv.size() * sizeof(LevelDevT<double>));
    // the tail keeps its vectors (b, e, x, x2 of every level) in LDS when they fit next to the level descriptors
    long tot = 0;
    for (size_t l = (size_t)amg->tail_level; l < amg->lv.size(); ++l) tot += amg->lv[l]->g.ntot;
    const bool lds_on = !(getenv("TP_AMG_TAIL_LDS") && atoi(getenv("TP_AMG_TAIL_LDS")) == 0);
    amg->tail_lds = (lds_on && 4 * tot * (long)sizeof(double) <= 120 * 1024) ? (int)tot : 0;
    if (getenv("TP_DEBUG")) fprintf(stderr, "[tp] amg tail: level %d of %zu, %ld doubles per vector set, lds %d\n", amg->tail_level, amg->lv.size(), tot, amg->tail_lds);
    if (amg->tail_lds > 0) {
        const int bytes = 120 * 1024;      // per-function limit shared by every hierarchy: always the maximum
        TP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_amg_tail<double>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        TP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_amg_tail<float>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    }
}

// level l+1 as its parent level l sees it.  Below the last distributed level that is this rank's planes of the
// global (replicated) arrays -- pointer arithmetic, no copy: plane 0 of the view is the lower halo.
struct CoarseView {
    GridDev g;
    long off;          // element offset of the view inside the level's arrays
    long slot_stride;  // operator slot stride of the level's arrays
};
static CoarseView coarse_view(const tp_ctx *c, const Amg *amg, int l) {
    const AmgLevel *Lc = amg->lv[l + 1];
    CoarseView v;
    v.slot_stride = Lc->g.ntot;
    if (l + 1 < amg->dist_levels || l >= amg->dist_levels) { v.g = Lc->g; v.off = 0; return v; }
    const auto &q = amg->ranges[l + 1][c->grid.rank];
    v.g = make_grid(Lc->g.n0, Lc->g.n1, q.second - q.first, Lc->g.n2, q.first);
    v.off = Lc->g.np * q.first;
    return v;
}

// level-0 operator: double stencil view of the Jacobian -> storage type R
template <class R>
__global__ void k_amg_import(GridDev g, Stencil A0, R *out) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid;
#pragma unroll
    for (int s = 0; s < 7; ++s) out[(long)s * g.ntot + c] = (R)A0.slot(s)[c];
}

template <class R>
static void setup_impl(tp_ctx *c, Amg *amg, const Stencil &A0) {
    AmgLevel *L0 = amg->lv[0];
    const int lg = amg->dist_levels;
    if (sizeof(R) == sizeof(double)) {
        L0->op = A0;                     // zero-copy view of the Jacobian planes
    } else {
        hipLaunchKernelGGL(k_amg_import<R>, grid_for(L0->g.nown), dim3(256), 0, c->stream, L0->g, A0, (R *)L0->A.p);
        L0->op.base = L0->A.p;
        L0->op.slot_stride = L0->g.ntot;
    }
    // relaxation-only truncation: dominance ratios of the V(nu,nu) levels (slab-distributed levels: maximum over the ranks)
    const int nratio = c->opt.amg_dom_tau > 0.0 ? std::min({c->opt.amg_full_levels, (int)amg->lv.size() - 1, 8}) : 0;
    if (nratio > 0) TP_HIP(hipMemsetAsync(amg->ratio_dev.p, 0, sizeof(double) * 64 * nratio, c->stream));
    for (size_t l = 0; l < amg->lv.size(); ++l) {
        AmgLevel *L = amg->lv[l];
        const StencilT<R> op{(R *)L->op.base, L->op.slot_stride};
        const bool want_ratio = (int)l < nratio;
        hipLaunchKernelGGL(k_amg_weights<R>, grid_for(L->g.nown, want_ratio ? 1024 : 256), dim3(want_ratio ? 1024 : 256), 0, c->stream, L->g, op, L->axis,
                           c->opt.amg_omega, (R *)L->wm.p, (R *)L->wp.p, (R *)L->invd.p,
                           (int)l < nratio ? (unsigned long long *)amg->ratio_dev.p + 64 * l : (unsigned long long *)nullptr);
        if ((int)l < lg) {
            // distributed level: the cycle reads inverse diagonals and weights of the neighbours' boundary planes,
            // coarsening along the slab axis also their operator rows
            halo_exchange_raw(c, L->g, L->invd.p, 1, 0, sizeof(R));
            halo_exchange_raw(c, L->g, L->wm.p, 1, 0, sizeof(R));
            halo_exchange_raw(c, L->g, L->wp.p, 1, 0, sizeof(R));
            if (L->axis == 2) halo_exchange_raw(c, L->g, L->op.base, 7, (size_t)L->op.slot_stride * sizeof(R), sizeof(R));
        }
        if (L->axis >= 0) {
            AmgLevel *Lc = amg->lv[l + 1];
            const CoarseView cv = coarse_view(c, amg, (int)l);
            hipLaunchKernelGGL(k_amg_coarsen<R>, grid_for(cv.g.nown), dim3(256), 0, c->stream, L->g, cv.g, op, L->axis,
                               (const R *)L->wm.p, (const R *)L->wp.p, (R *)Lc->A.p + cv.off, cv.slot_stride);
            if ((int)l + 1 == lg)        // first replicated level: everybody gets everybody's rows
                gather_ranges(c, Lc->A.p, Lc->g.np, amg->ranges[lg], 7, (size_t)Lc->g.ntot * sizeof(R), sizeof(R));
        }
    }
    AmgLevel *Lc = amg->lv.back();
    const int n = amg->ncoarse;
    const StencilT<R> opc{(R *)Lc->op.base, Lc->op.slot_stride};
    // (TP_EXP_SKIP_DENSE=1: timing experiment only -- leaves the previous inverse in place; bounds what ANY faster coarse
    // inverse, e.g. a blocked Gauss-Jordan on v_mfma_f64_16x16x4, could gain on pc_setup: DESIGN.md 4.5)
    static const bool skip_dense = getenv("TP_EXP_SKIP_DENSE") && atoi(getenv("TP_EXP_SKIP_DENSE")) == 1;
    if (skip_dense && amg->dense_done) {
    } else if (n <= 64) {
        const bool mfma = !(getenv("TP_AMG_DENSE_MFMA") && atoi(getenv("TP_AMG_DENSE_MFMA")) == 0);  // (read per set-up: A/B in one process)
        if (mfma) {
            const int bytes = (64 * 128 + 4 * 128 + 64 * 4 + 16) * (int)sizeof(double);
            TP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_amg_dense_inverse_mfma<R>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            hipLaunchKernelGGL(k_amg_dense_inverse_mfma<R>, dim3(1), dim3(1024), bytes, c->stream, Lc->g, opc, n,
                               amg->coarse_inv.p + (size_t)n * n);
        } else {
            hipLaunchKernelGGL(k_amg_dense_inverse_lds<R>, dim3(1), dim3(1024), 0, c->stream, Lc->g, opc, n,
                               amg->coarse_inv.p + (size_t)n * n);
        }
    }
    else
        hipLaunchKernelGGL(k_amg_dense_inverse<R>, dim3(1), dim3(256), 0, c->stream, Lc->g, opc, n, amg->coarse_inv.p,
                           amg->coarse_inv.p + (size_t)n * n);
    amg->dense_done = true;
    std::vector<LevelDevT<R>> h;
    for (size_t l = 0; l < amg->lv.size(); ++l) h.push_back(dev_of<R>(amg->lv[l], (int)l, c->opt));
    amg->lvhost.assign((const char *)h.data(), (const char *)h.data() + h.size() * sizeof(LevelDevT<R>));
    TP_HIP(hipMemcpyAsync(amg->lvdev.p, amg->lvhost.data(), amg->lvhost.size(), hipMemcpyHostToDevice, c->stream));
    if (nratio > 0) {
        if (lg > 0) allreduce_max(c, amg->ratio_dev.p, 64 * nratio);      // every rank takes the same decision
        TP_HIP(hipMemcpyAsync(amg->ratio_host, amg->ratio_dev.p, sizeof(double) * 64 * nratio, hipMemcpyDeviceToHost, c->stream));
        TP_HIP(hipEventRecord(amg->ev_ratio, c->stream));
        amg->ratio_pending = true;
    } else if (amg->trunc != -1) {
        amg->trunc = -1;
        c->graph_epoch++;
    }
    TP_HIP(hipGetLastError());
}

// Called before the first cycle after a set-up (never inside a stream capture): waits for the ratios and fixes the
// truncation level.  Returns true when the cycle shape changed (captured pc_apply graphs are then stale).
bool amg_resolve_trunc(tp_ctx *c, Amg *amg) {
    if (!amg || !amg->ratio_pending) return false;
    TP_HIP(hipEventSynchronize(amg->ev_ratio));
    amg->ratio_pending = false;
    const int nratio = std::min({c->opt.amg_full_levels, (int)amg->lv.size() - 1, 8});
    int t = -1;
    for (int l = 0; l < nratio; ++l) {
        double r = 0.0;
        for (int q = 0; q < 64; ++q) r = std::max(r, amg->ratio_host[64 * l + q]);
        if (l == 0) amg->ratio0 = r;
        if (r <= c->opt.amg_dom_tau) { t = l; break; }
    }
    const bool changed = t != amg->trunc;
    amg->trunc = t;
    return changed;
}

void amg_setup(tp_ctx *c, Amg *amg, const Stencil &A0) {
    static_assert(sizeof(LevelDevT<float>) == sizeof(LevelDevT<double>), "descriptor size");
    if (amg->single) setup_impl<float>(c, amg, A0);
    else setup_impl<double>(c, amg, A0);
}

// Multi-GPU, levels [0, lg): the same kernels on this rank's slab of the level, with a halo exchange in front of
// every kernel that reads a neighbour's value across the slab boundary.  Levels >= lg: the gathered global
// grid, every rank computing the same thing.  (b's halo planes are overwritten by the exchange.)
template <class R>
static void vcycle_impl(tp_ctx *c, Amg *amg, const double *b, double *x) {
    const int nlev = (int)amg->lv.size(), lt = amg->tail_level, lg = amg->dist_levels;
    const int trunc = amg->trunc;               // >= 0: that level ends the cycle with two Jacobi sweeps (amg_dom_tau)
    const int ltop = (trunc >= 0 && trunc < lt) ? trunc : lt;      // big levels [0, ltop) run their normal down/up sweeps
    const dim3 bl(256);
    std::vector<double *> xs(nlev, nullptr);       // pre-smoothed iterate of each big level (null: none)
    // level l (smoothed, no pre-sweeps) + level l+1 (pure transfer): both transfers in one launch each way
    static const bool pair_on = !(getenv("TP_AMG_PAIR") && atoi(getenv("TP_AMG_PAIR")) == 0);
    auto paired = [&](int l) {
        if (!pair_on || l < lg || l + 2 > lt || amg->lv[l]->g.nown >= amg->fuse_below) return false;
        if (trunc >= 0 && l + 2 > trunc) return false;          // (the truncation level is among the V(nu,nu) levels: never paired)
        const LevelDevT<R> A = dev_of<R>(amg->lv[l], l, c->opt), B = dev_of<R>(amg->lv[l + 1], l + 1, c->opt);
        return A.pre == 0 && A.post >= 1 && B.pre == 0 && B.post == 0;
    };
    auto hx = [&](int l, const double *v) {        // halo exchange of a level-l vector (no-op below lg)
        if (l < lg) halo_exchange(c, amg->lv[l]->g, const_cast<double *>(v), 1, 0);
    };
    // down-sweep over the big levels
    for (int l = 0; l < ltop; ++l) {
        AmgLevel *L = amg->lv[l];
        AmgLevel *Lc = amg->lv[l + 1];
        const LevelDevT<R> Ld = dev_of<R>(L, l, c->opt);
        const CoarseView cv = coarse_view(c, amg, l);
        const double *bl_ = (l == 0) ? b : L->b.p;
        double *bc = Lc->b.p + cv.off;
        const dim3 gr = xcd_grid(L->g.nown);
        const bool slab_axis = l < lg && L->axis == 2;      // the transfer itself crosses the slab boundary
        if (paired(l)) {
            AmgLevel *L2 = amg->lv[l + 2];
            hipLaunchKernelGGL(k_amg_restrict2<R>, xcd_grid(L2->g.nown), bl, 0, c->stream, Ld,
                               dev_of<R>(amg->lv[l + 1], l + 1, c->opt), L2->g, bl_, L2->b.p);
            ++l;                                    // level l+1 has nothing else to do on the way down
            continue;
        }
        if (Ld.pre == 0) {                          // V(0,post): residual = b
            if (slab_axis) hx(l, bl_);
            hipLaunchKernelGGL(k_amg_restrict<R>, xcd_grid(cv.g.nown), bl, 0, c->stream, Ld, cv.g, bl_, bc);
        } else {
            double *cur = L->x.p, *oth = L->x2.p;
            if (Ld.pre >= 2) hx(l, bl_);            // the fused double sweep reads invd*b of the neighbours
            hipLaunchKernelGGL(k_amg_pre<R>, gr, bl, 0, c->stream, Ld, bl_, Ld.pre >= 2 ? 1 : 0, cur);
            for (int k = 2; k < Ld.pre; ++k) {
                hx(l, cur);
                hipLaunchKernelGGL(k_amg_jacobi<R>, gr, bl, 0, c->stream, Ld, bl_, (const double *)cur, oth);
                std::swap(cur, oth);
            }
            xs[l] = cur;
            hx(l, cur);
            if (slab_axis || L->g.nown >= amg->fuse_below) {
                hipLaunchKernelGGL(k_amg_resid<R>, gr, bl, 0, c->stream, Ld, bl_, (const double *)cur, oth);
                if (slab_axis) hx(l, oth);
                hipLaunchKernelGGL(k_amg_restrict<R>, xcd_grid(cv.g.nown), bl, 0, c->stream, Ld, cv.g, (const double *)oth, bc);
            } else {
                hipLaunchKernelGGL(k_amg_resid_restrict<R>, xcd_grid(cv.g.nown), bl, 0, c->stream, Ld, cv.g, bl_,
                                   (const double *)cur, bc);
            }
        }
        if (l + 1 == lg)        // restricted residual of every slab -> the replicated levels' right-hand side
            gather_ranges(c, Lc->b.p, Lc->g.np, amg->ranges[lg], 1, 0, sizeof(double));
    }
    if (trunc >= 0 && trunc < lt) {
        // relaxation-only big level: x = x1 + invd (b - A x1), x1 = invd b (the fused double sweep), nothing below it
        AmgLevel *L = amg->lv[trunc];
        const double *bt = (trunc == 0) ? b : L->b.p;
        double *et = (trunc == 0) ? x : L->e.p;
        hx(trunc, bt);                              // the fused double sweep reads invd*b of the neighbours
        hipLaunchKernelGGL(k_amg_pre<R>, xcd_grid(L->g.nown), bl, 0, c->stream, dev_of<R>(L, trunc, c->opt), bt, 1, et);
    } else {
        // the tail: every level from lt down to the coarsest (or the truncation level) and back, one launch
        AmgLevel *Lt = amg->lv[lt];
        const double *bt = (lt == 0) ? b : Lt->b.p;
        double *et = (lt == 0) ? x : Lt->e.p;
        const int n = amg->ncoarse;
        hipLaunchKernelGGL(k_amg_tail<R>, dim3(1), dim3(1024), (size_t)4 * amg->tail_lds * sizeof(double), c->stream,
                           (const LevelDevT<R> *)amg->lvdev.p, lt, nlev, trunc, n,
                           (const double *)(amg->coarse_inv.p + (size_t)n * n), bt, et, amg->tail_lds);
    }
    // up-sweep over the big levels
    for (int l = ltop - 1; l >= 0; --l) {
        AmgLevel *L = amg->lv[l];
        AmgLevel *Lc = amg->lv[l + 1];
        const LevelDevT<R> Ld = dev_of<R>(L, l, c->opt);
        const CoarseView cv = coarse_view(c, amg, l);
        const double *bl_ = (l == 0) ? b : L->b.p;
        double *out = (l == 0) ? x : L->e.p;
        const double *ec = Lc->e.p + cv.off;
        const dim3 gr = xcd_grid(L->g.nown);
        double *src = xs[l];                        // its halo is still the one exchanged before the residual
        double *dst = (Ld.post == 1) ? out : (src == L->x.p ? L->x2.p : L->x.p);
        if (l >= 1 && paired(l - 1)) continue;      // transfer-only level folded into its parent's launch
        if (paired(l)) {
            AmgLevel *L2 = amg->lv[l + 2];
            hipLaunchKernelGGL(k_amg_prolong2_jacobi<R>, gr, bl, 0, c->stream, Ld, dev_of<R>(Lc, l + 1, c->opt), L2->g, bl_,
                               (const double *)L2->e.p, dst);
            for (int k = 1; k < Ld.post; ++k) {
                src = dst;
                dst = (k == Ld.post - 1) ? out : (src == L->x.p ? L->x2.p : L->x.p);
                hipLaunchKernelGGL(k_amg_jacobi<R>, gr, bl, 0, c->stream, Ld, bl_, (const double *)src, dst);
            }
            continue;
        }
        hx(l + 1, Lc->e.p);                         // distributed coarse level: parents across the boundary
        if (Ld.post == 0) {                         // pure transfer level
            hipLaunchKernelGGL(k_amg_prolong_set<R>, gr, bl, 0, c->stream, Ld, cv.g, ec, out);
            continue;
        }
        if (src && L->g.nown >= amg->fuse_below) {
            hipLaunchKernelGGL(k_amg_prolong_add<R>, gr, bl, 0, c->stream, Ld, cv.g, ec, src);
            hx(l, src);
            hipLaunchKernelGGL(k_amg_jacobi<R>, gr, bl, 0, c->stream, Ld, bl_, (const double *)src, dst);
        } else {
            hipLaunchKernelGGL(k_amg_prolong_jacobi<R>, gr, bl, 0, c->stream, Ld, cv.g, bl_, (const double *)src, ec, dst);
        }
        for (int k = 1; k < Ld.post; ++k) {
            src = dst;
            dst = (k == Ld.post - 1) ? out : (src == L->x.p ? L->x2.p : L->x.p);
            hx(l, src);
            hipLaunchKernelGGL(k_amg_jacobi<R>, gr, bl, 0, c->stream, Ld, bl_, (const double *)src, dst);
        }
    }
    TP_HIP(hipGetLastError());
}

void amg_vcycle(tp_ctx *c, Amg *amg, const double *b, double *x) {
    TP_REQUIRE(amg && !amg->lv.empty(), "AMG not set up");
    TP_REQUIRE(!amg->ratio_pending, "amg_resolve_trunc must run after a set-up and before the first cycle");
    TP_REQUIRE(x != amg->lv[0]->x.p && x != amg->lv[0]->x2.p && b != x, "aliasing in amg_vcycle");
    if (amg->single) vcycle_impl<float>(c, amg, b, x);
    else vcycle_impl<double>(c, amg, b, x);
}

}  // namespace tp

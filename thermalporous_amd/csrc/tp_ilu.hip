// Stage 2 of the composite preconditioner: block-Jacobi over tiles + block-ILU(0) inside each tile
// (reference: PETSc PCBJACOBI + PCILU levels 0, options sub_1_sub_pc_type ilu /
// sub_1_sub_pc_factor_levels 0, singlephase.py:348-349, twophase.py:547-548; explicit block counts
// via sub_1_pc_bjacobi_blocks, tests/test_homo_wells.py:112,125).
//
// MI355X design.  On a 7-point stencil ILU(0) in natural order never updates an off-diagonal block
// (the lower neighbours of a cell are not adjacent to one another), so
//     M = (D~ + L_A) D~^-1 (D~ + U_A),   D~_c = A_cc - sum_{m lower} A_cm D~_m^-1 A_mc .
// A tile is the full axis-0 line times t1 x t2 cells (t1*t2 <= 64) and is swept by ONE 64-lane
// wavefront: lane <-> (i1,i2) inside the tile, step s <-> i0 = s - i1 - i2.  The three lower
// neighbours of a cell were all produced in the previous step -- by the same lane (axis 0), by lane-1
// (axis 1) and by lane-t1 (axis 2) -- so the recurrence runs entirely in registers with two DPP
// shuffles per step: no LDS, no barriers, no inter-workgroup flags.  Couplings that leave the tile
// are dropped (= one bjacobi block per tile).
//
// The factor is stored in CONSUMPTION ORDER, [entry][tile][step][lane], with the products the
// sweeps need premultiplied (B_cm = A_cm D~_m^-1 for the forward sweep, C_cm = D~_c^-1 A_cm and
// D~_c^-1 for the backward sweep): every load of the sweeps is a 512-byte fully coalesced
// wave access and each factor byte is read exactly once per application.  HBM-bound:
// (3+4) b^2 + ... doubles per cell, the same traffic as one block SpMV.
#include "tp_common.hpp"

namespace tp {

template <int B>
__device__ __forceinline__ void inv_block(const double (&A)[B][B], double (&I)[B][B]);

template <>
__device__ __forceinline__ void inv_block<2>(const double (&A)[2][2], double (&I)[2][2]) {
    const double det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
    const double r = 1.0 / det;
    I[0][0] = A[1][1] * r; I[0][1] = -A[0][1] * r;
    I[1][0] = -A[1][0] * r; I[1][1] = A[0][0] * r;
}
template <>
__device__ __forceinline__ void inv_block<3>(const double (&A)[3][3], double (&I)[3][3]) {
    const double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1];
    const double c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2];
    const double c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    const double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
    const double r = 1.0 / det;
    I[0][0] = c00 * r;
    I[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * r;
    I[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * r;
    I[1][0] = c01 * r;
    I[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * r;
    I[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * r;
    I[2][0] = c02 * r;
    I[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * r;
    I[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * r;
}

struct IluGeom {
    GridDev g;
    int t1, t2, nt1, nt2, nsteps;
    long slots;          // ntiles*nsteps*64
};

// tile/lane/step -> cell; returns false if the lane has no cell at this step
__device__ __forceinline__ bool tile_cell(const IluGeom &G, int tile, int lane, int s, int &i0, int &j, int &k,
                                          int &tj, int &tk, long &c) {
    const int T1 = tile % G.nt1, T2 = tile / G.nt1;
    j = lane % G.t1;
    k = lane / G.t1;
    tj = min(G.t1, G.g.n1 - T1 * G.t1);
    tk = min(G.t2, G.g.n2 - T2 * G.t2);
    i0 = s - j - k;
    const bool ok = (k < G.t2) && (j < tj) && (k < tk) && (i0 >= 0) && (i0 < G.g.n0);
    const int i1 = T1 * G.t1 + j, i2 = T2 * G.t2 + k;
    c = G.g.np + (long)i0 + (long)G.g.n0 * i1 + G.g.np * i2;
    return ok;
}

template <int B>
__global__ __launch_bounds__(64) void k_ilu_factor(IluGeom G, const double *__restrict__ J, double *fwd,
                                                   double *bwd) {
    const int tile = blockIdx.x, lane = threadIdx.x;
    const long nt = G.g.ntot;
    double Dp[B][B];                       // D~^-1 of this lane's previous cell (axis-0 lower neighbour)
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
        for (int q = 0; q < B; ++q) Dp[r][q] = 0.0;
    const long stride[3] = {1, (long)G.g.n0, G.g.np};
    for (int s = 0; s < G.nsteps; ++s) {
        int i0, j, k, tj, tk;
        long c;
        const bool ok = tile_cell(G, tile, lane, s, i0, j, k, tj, tk, c);
        // D~^-1 of the three lower neighbours (previous step): self, lane-1, lane-t1
        double Dn[3][B][B];
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int q = 0; q < B; ++q) {
                Dn[0][r][q] = Dp[r][q];
                Dn[1][r][q] = __shfl_up(Dp[r][q], 1, 64);
                Dn[2][r][q] = __shfl_up(Dp[r][q], G.t1, 64);
            }
        const bool has[3] = {ok && i0 > 0, ok && j > 0, ok && k > 0};
        double D[B][B], Di[B][B];
        const long slot_base = ((long)tile * G.nsteps + s) * 64 + lane;
        if (ok) {
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) D[r][q] = J[((long)(0 * B + r) * B + q) * nt + c];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double Bm[B][B];                   // B_cm = A_cm D~_m^-1
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) Bm[r][q] = 0.0;
            if (has[a]) {
                const long m = c - stride[a];
                double Acm[B][B], Amc[B][B];
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int q = 0; q < B; ++q) {
                        Acm[r][q] = J[((long)((1 + 2 * a) * B + r) * B + q) * nt + c];
                        Amc[r][q] = J[((long)((2 + 2 * a) * B + r) * B + q) * nt + m];
                    }
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int q = 0; q < B; ++q) {
                        double v = 0.0;
#pragma unroll
                        for (int t = 0; t < B; ++t) v += Acm[r][t] * Dn[a][t][q];
                        Bm[r][q] = v;
                    }
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int q = 0; q < B; ++q) {
                        double v = 0.0;
#pragma unroll
                        for (int t = 0; t < B; ++t) v += Bm[r][t] * Amc[t][q];
                        D[r][q] -= v;
                    }
            }
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) fwd[(long)((a * B + r) * B + q) * G.slots + slot_base] = Bm[r][q];
        }
        if (ok) {
            inv_block<B>(D, Di);
        } else {
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) Di[r][q] = 0.0;
        }
        // backward-sweep data: D~_c^-1 and C_cm = D~_c^-1 A_cm for the three upper neighbours in the tile
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int q = 0; q < B; ++q) bwd[(long)((3 * B + r) * B + q) * G.slots + slot_base] = Di[r][q];
        const bool hasu[3] = {ok && i0 < G.g.n0 - 1, ok && j < tj - 1, ok && k < tk - 1};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double Cm[B][B];
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) Cm[r][q] = 0.0;
            if (hasu[a]) {
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int q = 0; q < B; ++q) {
                        double v = 0.0;
#pragma unroll
                        for (int t = 0; t < B; ++t) v += Di[r][t] * J[((long)((2 + 2 * a) * B + t) * B + q) * nt + c];
                        Cm[r][q] = v;
                    }
            }
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) bwd[(long)((a * B + r) * B + q) * G.slots + slot_base] = Cm[r][q];
        }
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int q = 0; q < B; ++q) Dp[r][q] = Di[r][q];
    }
}

// x = addto + M^-1 r  (forward then backward sweep of one tile by one wavefront)
template <int B>
__global__ __launch_bounds__(64) void k_ilu_solve(IluGeom G, const double *__restrict__ fwd,
                                                  const double *__restrict__ bwd, const double *__restrict__ rhs,
                                                  double *__restrict__ ytmp, double *x,
                                                  const double *addto) {
    const int tile = blockIdx.x, lane = threadIdx.x;
    const long nt = G.g.ntot;
    double yp[B];
#pragma unroll
    for (int r = 0; r < B; ++r) yp[r] = 0.0;
    // ---- forward: y_c = r_c - sum_lower B_cm y_m -------------------------------------------------
    for (int s = 0; s < G.nsteps; ++s) {
        int i0, j, k, tj, tk;
        long c;
        const bool ok = tile_cell(G, tile, lane, s, i0, j, k, tj, tk, c);
        const long sb = ((long)tile * G.nsteps + s) * 64 + lane;
        double yn[3][B];
#pragma unroll
        for (int r = 0; r < B; ++r) {
            yn[0][r] = yp[r];
            yn[1][r] = __shfl_up(yp[r], 1, 64);
            yn[2][r] = __shfl_up(yp[r], G.t1, 64);
        }
        double y[B];
#pragma unroll
        for (int r = 0; r < B; ++r) y[r] = ok ? rhs[(long)r * nt + c] : 0.0;
        // B_cm is stored as zero where the neighbour is outside the tile, so no branches here
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) y[r] -= fwd[(long)((a * B + r) * B + q) * G.slots + sb] * yn[a][q];
#pragma unroll
        for (int r = 0; r < B; ++r) {
            y[r] = ok ? y[r] : 0.0;
            ytmp[(long)r * G.slots + sb] = y[r];
            yp[r] = y[r];
        }
    }
    // ---- backward: x_c = D~_c^-1 y_c - sum_upper C_cm x_m -----------------------------------------
    double xp[B];
#pragma unroll
    for (int r = 0; r < B; ++r) xp[r] = 0.0;
    for (int s = G.nsteps - 1; s >= 0; --s) {
        int i0, j, k, tj, tk;
        long c;
        const bool ok = tile_cell(G, tile, lane, s, i0, j, k, tj, tk, c);
        const long sb = ((long)tile * G.nsteps + s) * 64 + lane;
        double xn[3][B];
#pragma unroll
        for (int r = 0; r < B; ++r) {
            xn[0][r] = xp[r];
            xn[1][r] = __shfl_down(xp[r], 1, 64);
            xn[2][r] = __shfl_down(xp[r], G.t1, 64);
        }
        double y[B], xv[B];
#pragma unroll
        for (int r = 0; r < B; ++r) y[r] = ytmp[(long)r * G.slots + sb];
#pragma unroll
        for (int r = 0; r < B; ++r) {
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < B; ++q) v += bwd[(long)((3 * B + r) * B + q) * G.slots + sb] * y[q];
            xv[r] = v;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) xv[r] -= bwd[(long)((a * B + r) * B + q) * G.slots + sb] * xn[a][q];
#pragma unroll
        for (int r = 0; r < B; ++r) {
            xv[r] = ok ? xv[r] : 0.0;
            xp[r] = xv[r];
            if (ok) x[(long)r * nt + c] = (addto ? addto[(long)r * nt + c] : 0.0) + xv[r];
        }
    }
}

static IluGeom geom_of(const tp_ctx *c) {
    IluGeom G;
    G.g = c->g;
    G.t1 = c->ilu.t1; G.t2 = c->ilu.t2; G.nt1 = c->ilu.nt1; G.nt2 = c->ilu.nt2;
    G.nsteps = c->ilu.nsteps; G.slots = c->ilu.slots;
    return G;
}

void ilu_setup(tp_ctx *c) {
    IluData &d = c->ilu;
    const GridDev &g = c->g;
    int t1 = c->opt.ilu_t1, t2 = c->opt.ilu_t2;
    if (t1 <= 0) t1 = (g.n2 == 1) ? 64 : 8;
    if (t2 <= 0) t2 = (g.n2 == 1) ? 1 : 8;
    t1 = std::min(t1, g.n1);
    t2 = std::min(t2, g.n2);
    TP_REQUIRE(t1 >= 1 && t2 >= 1 && t1 * t2 <= 64, "ILU tile must satisfy t1*t2 <= 64 (one wavefront per tile)");
    d.t1 = t1; d.t2 = t2;
    d.nt1 = (g.n1 + t1 - 1) / t1;
    d.nt2 = (g.n2 + t2 - 1) / t2;
    d.ntiles = d.nt1 * d.nt2;
    d.nsteps = g.n0 + t1 + t2 - 2;
    d.slots = (long)d.ntiles * d.nsteps * 64;
    const int B = c->b;
    d.fwd.alloc((size_t)3 * B * B * d.slots);
    d.bwd.alloc((size_t)4 * B * B * d.slots);
    d.ytmp.alloc((size_t)B * d.slots);
}

void ilu_factor(tp_ctx *c) {
    TP_REQUIRE(c->jac_ready, "Jacobian not assembled");
    if (c->ilu.slots == 0) ilu_setup(c);
    const IluGeom G = geom_of(c);
    if (c->b == 3)
        hipLaunchKernelGGL(k_ilu_factor<3>, dim3(c->ilu.ntiles), dim3(64), 0, c->stream, G, c->J.p, c->ilu.fwd.p,
                           c->ilu.bwd.p);
    else
        hipLaunchKernelGGL(k_ilu_factor<2>, dim3(c->ilu.ntiles), dim3(64), 0, c->stream, G, c->J.p, c->ilu.fwd.p,
                           c->ilu.bwd.p);
    TP_HIP(hipGetLastError());
}

void ilu_solve(tp_ctx *c, const double *r, double *x, const double *addto) {
    TP_REQUIRE(c->ilu.slots > 0, "ILU not factored");
    const IluGeom G = geom_of(c);
    if (c->b == 3)
        hipLaunchKernelGGL(k_ilu_solve<3>, dim3(c->ilu.ntiles), dim3(64), 0, c->stream, G, c->ilu.fwd.p,
                           c->ilu.bwd.p, r, c->ilu.ytmp.p, x, addto);
    else
        hipLaunchKernelGGL(k_ilu_solve<2>, dim3(c->ilu.ntiles), dim3(64), 0, c->stream, G, c->ilu.fwd.p,
                           c->ilu.bwd.p, r, c->ilu.ytmp.p, x, addto);
    TP_HIP(hipGetLastError());
}

}  // namespace tp

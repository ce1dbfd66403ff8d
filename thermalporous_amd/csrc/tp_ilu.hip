// Stage 2 of the composite preconditioner: block-Jacobi over tiles + block-ILU(0) inside each tile
// (reference: PETSc PCBJACOBI + PCILU levels 0, options sub_1_sub_pc_type ilu /
// sub_1_sub_pc_factor_levels 0, singlephase.py:348-349, twophase.py:547-548; explicit block counts
// via sub_1_pc_bjacobi_blocks, tests/test_homo_wells.py:112,125).
//
// MI355X design.  On a 7-point stencil ILU(0) in natural order never updates an off-diagonal block
// (the lower neighbours of a cell are not adjacent to one another), so
//     M = (D~ + L_A) D~^-1 (D~ + U_A),   D~_c = A_cc - sum_{m lower} A_cm D~_m^-1 A_mc .
// A tile is t0 x t1 x t2 cells (t1*t2 <= 64; t0 = whole axis-0 line by default) and is swept by ONE
// 64-lane wavefront: lane <-> (i1,i2) inside the tile, step s <-> i0 = s - i1 - i2.  The three lower
// neighbours of a cell were all produced in the previous step -- by the same lane (axis 0), by lane-1
// (axis 1) and by lane-t1 (axis 2) -- so the recurrence runs entirely in registers with two
// cross-lane shuffles per step: no LDS, no barriers, no inter-workgroup flags.  Couplings that leave
// the tile are dropped (= one bjacobi block per tile).
//
// The factor is stored in CONSUMPTION ORDER: one contiguous chunk per (tile, step) holding the
// premultiplied blocks the sweep needs at that step (B_cm = A_cm D~_m^-1 forward; C_cm = D~_c^-1 A_cm
// and D~_c^-1 backward) as [entry pair][lane][2] doubles, so that every load of the sweeps is a
// 16-byte-per-lane, 1-KiB-per-wave fully coalesced access, each wave streams its own contiguous
// region of HBM front to back (then back to front), and each factor byte is read exactly once per
// application.  The chunks of the next steps are prefetched through a ring of register buffers.
// Traffic: (3+4) b^2 doubles per cell plus vectors, that of one block SpMV.  Measured: the solve
// moves 4.5 TB/s but is bound by the serial recurrence, not by HBM.
// Default sweep kernel since round 2: k_ilu_solve_mw (one wavefront per block ROW of a tile, same chunk stream, same
// schedule; DESIGN.md 4.4 vi); k_ilu_solve below is the one-wave kernel described here (TP_ILU_MW=0).
// The factorisation reads the Jacobian through k_ilu_gather, which re-orders it into the same chunk order
// with thousands of waves (the plane layout puts neighbouring lanes n0 doubles apart).
#include "tp_common.hpp"
#include <type_traits>
#include <cstdlib>

// compact chunk rows are padded to this many lanes: 8 lanes x 16 B = one 128-byte line, so that a row never straddles
// a line shared with its neighbour row (a 54-lane row of 864 B touches 8 lines where 7 would do)
#ifndef ILU_ROW_ALIGN
#define ILU_ROW_ALIGN 1
#endif
#ifndef TP_ILU_UNROLL
#define TP_ILU_UNROLL 4
#endif

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
    int t0, t1, t2, nt0, nt1, nt2, nsteps;
    int ntiles;    // nt0*nt1*nt2
    int nl;        // lanes of a wave that carry a column: t1*t2 <= 64
    int rs;        // doubles per chunk row: 2*nl when the rows are as wide as the tile (CP kernels), else 128
    int ws;        // whole-slab ILU(0) (tp_options.ilu_whole): couplings between tiles are kept
    const int *pref;   // ILU(1) packed factor copy: slots before each step of a tile (null: padded 64-lane rows)
    int ptot;          // slots per tile = pref[nsteps]
};

// number of double2 pairs per chunk
template <int B> struct IluLayout {
    static constexpr int NEF = 3 * B * B;                 // forward entries per cell
    static constexpr int NEB = 4 * B * B;                 // backward entries per cell
    static constexpr int PF = (NEF + 1) / 2;              // double2 pairs per forward chunk
    static constexpr int PB = (NEB + 1) / 2;
    static constexpr int PY = (B + 1) / 2;                // pairs of the intermediate vector y
    static constexpr int NEJ = 7 * B * B;                 // Jacobian entries per cell (7 blocks)
    static constexpr int PJ = (NEJ + 1) / 2;              // pairs per chunk of the re-ordered Jacobian
};

__device__ __forceinline__ long chunk_idx(const IluGeom &G, int tile, int s) {
    return (long)tile * G.nsteps + s;      // (step-major chunks, [step][tile], were measured: no gain -- DESIGN.md 4.4 vi)
}

// tile/lane/step -> cell; returns false if the lane has no cell at this step
struct TileInfo {
    int base0, base1, base2;   // first cell of the tile
    int tt0, tj, tk;           // actual tile extents
    int j, k;                  // lane coordinates
};

__device__ __forceinline__ TileInfo tile_info(const IluGeom &G, int tile, int lane) {
    TileInfo t;
    const int T0 = tile % G.nt0, T1 = (tile / G.nt0) % G.nt1, T2 = tile / (G.nt0 * G.nt1);
    t.base0 = T0 * G.t0; t.base1 = T1 * G.t1; t.base2 = T2 * G.t2;
    t.tt0 = min(G.t0, G.g.n0 - t.base0);
    t.tj = min(G.t1, G.g.n1 - t.base1);
    t.tk = min(G.t2, G.g.n2 - t.base2);
    t.j = lane % G.t1;
    t.k = lane / G.t1;
    return t;
}

__device__ __forceinline__ bool tile_cell(const IluGeom &G, const TileInfo &t, int s, int &l0, long &c) {
    l0 = s - t.j - t.k;
    const bool ok = (t.k < G.t2) && (t.j < t.tj) && (t.k < t.tk) && (l0 >= 0) && (l0 < t.tt0);
    c = G.g.np + (long)(t.base0 + l0) + (long)G.g.n0 * (t.base1 + t.j) + G.g.np * (t.base2 + t.k);
    return ok;
}

// Jacobian blocks -> the factorisation's consumption order.  In the plane layout neighbouring lanes of a tile
// (lane = column (j,k), step = i0 + j + k) sit n0 doubles apart, so a wave touches 64 cache lines per value and
// uses 8 bytes of each; the ONE wave that factors a tile cannot hide that.  This pass does the same gather with
// thousands of waves: a workgroup covers ILU_SEG consecutive steps of one (tile, group of entry pairs), so the 16 doubles
// of every line it touches are consumed by the workgroup itself (L1/L2 hits), and writes chunks
// [tile][step][entry pair][lane][2] (1 KiB contiguous per wave).  Couplings that leave the tile are written as
// zeros, so the factorisation needs no masks.
constexpr int ILU_SEG = 16;      // steps per workgroup
constexpr int ILU_PPT = 1;       // entry pairs per thread
template <int B, bool CP>
__global__ __launch_bounds__(64 * ILU_SEG) void k_ilu_gather(IluGeom G, const double *__restrict__ J,
                                                             double *__restrict__ Jt) {
    using L = IluLayout<B>;
    const int NL = CP ? G.nl : 64, RS = CP ? G.rs : 128;
    const int tile = blockIdx.x, P0 = blockIdx.y * ILU_PPT, s = blockIdx.z * ILU_SEG + (threadIdx.x >> 6),
              lane = threadIdx.x & 63;
    const int ns = G.nsteps;
    if (s >= ns) return;
    const long nt = G.g.ntot;
    const TileInfo ti = tile_info(G, tile, lane);
    int l0;
    long c;
    const bool ok = tile_cell(G, ti, s, l0, c);
    // which of the 7 blocks survive: couplings to cells outside the tile are dropped (bjacobi) -- outside the SLAB when the
    // whole slab is one block (G.ws)
    const int g0 = ti.base0 + l0, g1 = ti.base1 + ti.j, g2 = ti.base2 + ti.k;
    const bool ws = G.ws != 0;
    const bool keep[7] = {true,
                          ws ? g0 > 0 : l0 > 0, ws ? g0 < G.g.n0 - 1 : l0 < ti.tt0 - 1,
                          ws ? g1 > 0 : ti.j > 0, ws ? g1 < G.g.n1 - 1 : ti.j < ti.tj - 1,
                          ws ? g2 > 0 : ti.k > 0, ws ? g2 < G.g.n2 - 1 : ti.k < ti.tk - 1};
    double v[2 * ILU_PPT];
#pragma unroll
    for (int u = 0; u < 2 * ILU_PPT; ++u) {          // all loads of the thread in flight together
        const int e = 2 * P0 + u;
        v[u] = (ok && e < L::NEJ && keep[e / (B * B)]) ? J[(long)e * nt + c] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < ILU_PPT; ++u) {
        if (P0 + u >= L::PJ) break;
        double2 o;
        o.x = v[2 * u];
        o.y = v[2 * u + 1];
        if (!CP || lane < NL) reinterpret_cast<double2 *>(Jt + (chunk_idx(G, tile, s) * L::PJ + P0 + u) * RS)[lane] = o;
    }
}

// MW: the factor is written in the layout of the multi-wave sweep (k_ilu_solve_mw): one run of doubles per block ROW,
// [tile][step][row r][entry (a, q)][lane] -- forward 3B entries per row (B_a[r][q]), backward 4B (C_a[r][q], then D~^-1[r][q])
// (vmcnt counts loads and stores and retires in order: every condition around a load or a store -- a run-time `if (live)`, the
// whole-slab branch, `if (s + 1 < ns) load` -- makes the compiler wait with vmcnt(0) at the next use and drains the prefetched
// chunk.  Hence: WS is a template parameter, lanes without a column store into a dump chunk behind the arrays, and the loop body
// is two unconditional steps with clamped prefetch indices.)
template <int B, bool CP, bool MW, bool WS = false>
__global__ __launch_bounds__(64) void k_ilu_factor(IluGeom G, const double *__restrict__ Jt, double *fwd,
                                                   double *bwd, const int *__restrict__ tiles) {
    using L = IluLayout<B>;
    const int tile = tiles ? tiles[blockIdx.x] : blockIdx.x, lane = threadIdx.x;      // (tiles: one tile-diagonal, G.ws)
    const int NL = CP ? G.nl : 64, RS = CP ? G.rs : 128;
    // idle lanes (>= t1*t2) load a live lane's data and store nothing.  With wave-wide rows every lane is live and both
    // are compile-time facts: a (never false) run-time `if (live)` around the stores is a divergent branch to the
    // compiler, which then drains ALL outstanding loads (s_waitcnt vmcnt(0)) at every step -- measured +40 % on C1
    const int la = CP ? (lane < NL ? lane : NL - 1) : lane;
    const bool live = CP ? lane < NL : true;
    const TileInfo ti = tile_info(G, tile, lane);
    const int ns = G.nsteps;
    double Dp[B][B];                       // D~^-1 of this lane's previous cell (axis-0 lower neighbour)
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
        for (int q = 0; q < B; ++q) Dp[r][q] = 0.0;
    // the Jacobian blocks of a step are gathered one step ahead (their addresses do not depend on the recurrence)
    // The blocks of a step come from the re-ordered Jacobian (k_ilu_gather), one step ahead of their use.
    // A_mc (the +a block of the lower neighbour m = c - e_a) is NOT loaded: it is the A_up[a] block that the lane
    // owning m (this lane, lane-1, lane-t1) loaded for the previous step -> taken from its registers by shuffle.
    struct Blk {
        double D[B][B], Acm[3][B][B], Aup[3][B][B];
        bool ok;
    };
    Blk buf[2];
    double Aprev[3][B][B];                 // A_up of this lane's previous step
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int q = 0; q < B; ++q) Aprev[a][r][q] = 0.0;
    auto load = [&](Blk &k, int s) {
        int l0;
        long c;
        k.ok = tile_cell(G, ti, s, l0, c);
        const double2 *ch = reinterpret_cast<const double2 *>(Jt + chunk_idx(G, tile, s) * ((long)L::PJ * RS)) + la;
        double v[2 * L::PJ];
#pragma unroll
        for (int p = 0; p < L::PJ; ++p) {
            const double2 t = ch[p * (RS >> 1)];
            v[2 * p] = t.x;
            v[2 * p + 1] = t.y;
        }
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int q = 0; q < B; ++q) {
                k.D[r][q] = v[r * B + q];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    k.Acm[a][r][q] = v[((1 + 2 * a) * B + r) * B + q];
                    k.Aup[a][r][q] = v[((2 + 2 * a) * B + r) * B + q];
                }
            }
    };
    auto step = [&](const Blk &k, int s) {
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
        double Amc[3][B][B];                   // A_mc of the three lower neighbours, from their owners' registers
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int q = 0; q < B; ++q) {
                Amc[0][r][q] = Aprev[0][r][q];
                Amc[1][r][q] = __shfl_up(Aprev[1][r][q], 1, 64);
                Amc[2][r][q] = __shfl_up(Aprev[2][r][q], G.t1, 64);
            }
        double D[B][B], Di[B][B];
        constexpr int PFR = (3 * B + 1) / 2, PBR = (4 * B + 1) / 2;       // (IluMwLayout)
        if constexpr (MW && WS) {
            // whole-slab ILU(0): a lower neighbour in ANOTHER tile (finished in an earlier launch: smaller T0+T1+T2) is not
            // in this wave's registers.  Its D~^-1 comes from that tile's backward chunk, its A_mc from the re-ordered
            // Jacobian.  Lower neighbour tiles are never partial along the axis they are crossed in.
            int l0c;
            long ccell;
            const bool okc = tile_cell(G, ti, s, l0c, ccell);
            const int T0 = tile % G.nt0, T1 = (tile / G.nt0) % G.nt1, T2 = tile / (G.nt0 * G.nt1);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const bool cross = okc && (a == 0 ? (l0c == 0 && T0 > 0) : a == 1 ? (ti.j == 0 && T1 > 0) : (ti.k == 0 && T2 > 0));
                if (cross) {
                    const int tA = tile - (a == 0 ? 1 : a == 1 ? G.nt0 : G.nt0 * G.nt1);
                    const int lnA = a == 0 ? lane : a == 1 ? (G.t1 - 1) + G.t1 * ti.k : ti.j + G.t1 * (G.t2 - 1);
                    const int sA = a == 0 ? (G.t0 - 1) + ti.j + ti.k : a == 1 ? l0c + (G.t1 - 1) + ti.k : l0c + ti.j + (G.t2 - 1);
                    const double *bA = bwd + chunk_idx(G, tA, sA) * ((long)2 * B * PBR * G.nl);
                    const double *jA = Jt + chunk_idx(G, tA, sA) * ((long)L::PJ * RS);
#pragma unroll
                    for (int r = 0; r < B; ++r)
#pragma unroll
                        for (int q = 0; q < B; ++q) {
                            Dn[a][r][q] = bA[((long)(r * PBR + ((3 * B + q) >> 1)) * G.nl + lnA) * 2 + ((3 * B + q) & 1)];
                            const int e = ((2 + 2 * a) * B + r) * B + q;
                            Amc[a][r][q] = jA[(long)(e >> 1) * RS + lnA * 2 + (e & 1)];
                        }
                }
            }
        }
        // (multi-wave layout: lanes beyond the tile's columns store into the dump chunk behind the last real one -- ilu_setup
        // allocates it with 128 doubles of slack -- instead of not storing)
        const bool mlive = lane < G.nl;
        const long chs = MW ? (mlive ? chunk_idx(G, tile, s) : (long)G.ntiles * G.nsteps) : chunk_idx(G, tile, s);
        double *fch = fwd + chs * (MW ? (long)2 * B * PFR * G.nl : (long)L::PF * RS);
        double *bch = bwd + chs * (MW ? (long)2 * B * PBR * G.nl : (long)L::PB * RS);
        auto fidx = [&](int a, int r, int q) { return MW ? ((long)(r * PFR + ((a * B + q) >> 1)) * G.nl + lane) * 2 + ((a * B + q) & 1)
                                                         : (long)((((a * B + r) * B + q) >> 1) * RS + lane * 2 + (((a * B + r) * B + q) & 1)); };
        auto bidx = [&](int a, int r, int q) { return MW ? ((long)(r * PBR + ((a * B + q) >> 1)) * G.nl + lane) * 2 + ((a * B + q) & 1)
                                                         : (long)((((a * B + r) * B + q) >> 1) * RS + lane * 2 + (((a * B + r) * B + q) & 1)); };
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int q = 0; q < B; ++q) D[r][q] = k.D[r][q];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double Bm[B][B];                   // B_cm = A_cm D~_m^-1 (zero when the neighbour is outside the tile)
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) {
                    double v = 0.0;
#pragma unroll
                    for (int t = 0; t < B; ++t) v += k.Acm[a][r][t] * Dn[a][t][q];
                    Bm[r][q] = v;                  // zero when the neighbour is outside the tile (A_cm = 0)
                }
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) {
                    double v = 0.0;
#pragma unroll
                    for (int t = 0; t < B; ++t) v += Bm[r][t] * Amc[a][t][q];
                    D[r][q] -= v;
                    if (MW || live) fch[fidx(a, r, q)] = Bm[r][q];
                }
        }
        if (!MW && (L::NEF & 1) && live) fch[(L::NEF >> 1) * RS + lane * 2 + 1] = 0.0;    // padding half of the last pair
        if (k.ok) {
            inv_block<B>(D, Di);
        } else {
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) Di[r][q] = 0.0;
        }
        // backward-sweep data: C_cm = D~_c^-1 A_cm for the three upper neighbours in the tile, then D~_c^-1
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) {
                    double v = 0.0;
#pragma unroll
                    for (int t = 0; t < B; ++t) v += Di[r][t] * k.Aup[a][t][q];
                    if (MW || live) bch[bidx(a, r, q)] = v;
                }
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int q = 0; q < B; ++q) {
                if (MW || live) bch[bidx(3, r, q)] = Di[r][q];
                Dp[r][q] = Di[r][q];
            }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) Aprev[a][r][q] = k.Aup[a][r][q];
    };
    load(buf[0], 0);
    int s = 0;
    for (; s + 2 <= ns; s += 2) {
        load(buf[1], min(s + 1, ns - 1));
        step(buf[0], s);
        load(buf[0], min(s + 2, ns - 1));
        step(buf[1], s + 1);
    }
    if (s < ns) step(buf[0], s);
}

// one chunk = NP double2 per live lane, coalesced (rows of nl double2)
template <int NP>
__device__ __forceinline__ void load_chunk(const double *__restrict__ ch, int lane, int nl, double2 (&v)[NP]) {
    const double2 *p = reinterpret_cast<const double2 *>(ch) + lane;
#pragma unroll
    for (int i = 0; i < NP; ++i) v[i] = p[i * nl];
}
template <int NP>
__device__ __forceinline__ double chunk_get(const double2 (&v)[NP], int e) {
    return (e & 1) ? v[e >> 1].y : v[e >> 1].x;
}

// x = addto + M^-1 r  (forward then backward sweep of one tile by one wavefront).
// Three register buffers form a prefetch ring: while step s computes from one buffer the chunks of
// steps s+1 and s+2 are in flight into the other two, and the buffer just consumed is refilled with
// step s+3 -- a single wave per CU keeps ~40 KB of HBM reads outstanding.
// YLDS: the intermediate vector y of the tile (nsteps x B x 64 doubles, 147 KB on C4) stays in the CU's LDS between
// the two sweeps instead of going through HBM (a write, a read and their row padding: 10 % of the kernel's traffic).
template <int B, bool DEPTH2, bool YLDS, bool CP>
__global__ __launch_bounds__(64) void k_ilu_solve(IluGeom G, const double *__restrict__ fwd,
                                                  const double *__restrict__ bwd, const double *__restrict__ rhs,
                                                  double *__restrict__ ytmp, double *x, const double *addto,
                                                  int nadd) {
    using L = IluLayout<B>;
    extern __shared__ double ylds[];       // [step][field][lane] when YLDS
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int NL = CP ? G.nl : 64, RS = CP ? G.rs : 128;
    const int la = CP ? (lane < NL ? lane : NL - 1) : lane;      // idle lanes (>= t1*t2) shadow a live lane's loads
    const bool live = CP ? lane < NL : true;                     // (compile-time true with wave-wide rows: see k_ilu_factor)
    const long nt = G.g.ntot;
    const TileInfo ti = tile_info(G, tile, lane);
    const int ns = G.nsteps;
    const long rowF = (long)L::PF * RS, rowB = (long)L::PB * RS, rowY = (long)L::PY * RS;
    const long park = min((long)lane, G.g.np - 1);     // an entry of the lower halo plane: where cell-less lanes read / write
    constexpr int RING = DEPTH2 ? 3 : 2;
    constexpr int UN = TP_ILU_UNROLL;          // rings per steady-state loop iteration
    int l0;
    long c;
    // ---- forward: y_c = r_c - sum_lower B_cm y_m -------------------------------------------------
    {
        double yp[B];
#pragma unroll
        for (int r = 0; r < B; ++r) yp[r] = 0.0;
        double2 buf[RING][L::PF];
        double rr[RING][B];
        bool okk[RING];
        auto load = [&](int k, int step) {
            okk[k] = tile_cell(G, ti, step, l0, c);
            load_chunk<L::PF>(fwd + chunk_idx(G, tile, step) * rowF, la, RS >> 1, buf[k]);
            // BRANCH-FREE: a lane without a cell at this step reads entry `lane` of the lower halo plane (valid memory,
            // finite) and the value is dropped by a select.  A load or store behind a divergent branch makes the compiler
            // drain every outstanding load (s_waitcnt vmcnt(0)) at each step, prefetch ring included.
            const long cs = okk[k] ? c : park;
#pragma unroll
            for (int r = 0; r < B; ++r) rr[k][r] = rhs[(long)r * nt + cs];
        };
        auto step = [&](int k, int s) {
            double yn[3][B], y[B];
#pragma unroll
            for (int r = 0; r < B; ++r) {
                yn[0][r] = yp[r];
                yn[1][r] = __shfl_up(yp[r], 1, 64);
                yn[2][r] = __shfl_up(yp[r], G.t1, 64);
                y[r] = okk[k] ? rr[k][r] : 0.0;
            }
            // B_cm is stored as zero where the neighbour is outside the tile, so no branches here
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int q = 0; q < B; ++q) y[r] -= chunk_get<L::PF>(buf[k], (a * B + r) * B + q) * yn[a][q];
            double *ych = ytmp + chunk_idx(G, tile, s) * rowY;
            const long ydump = ((long)G.nsteps * gridDim.x - chunk_idx(G, tile, s)) * rowY + ((long)tile * 64 + lane) * B;
#pragma unroll
            for (int r = 0; r < B; ++r) {
                y[r] = okk[k] ? y[r] : 0.0;
                if (YLDS) ylds[((long)s * B + r) * 64 + lane] = y[r];
                else ych[live ? (long)(r >> 1) * RS + lane * 2 + (r & 1) : ydump + r] = y[r];      // (idle lanes: parking slot)
                yp[r] = y[r];
            }
        };
#pragma unroll
        for (int k = 0; k < RING; ++k)
            if (k < ns) load(k, k);
        int s = 0;
        // The compiler drains every outstanding load (s_waitcnt vmcnt(0)) at a loop header whose back edge carries
        // loads in flight: one exposed memory latency per loop iteration.  UN rings per iteration make that one per
        // UN*RING steps instead of one per RING.
        for (; s + (UN + 1) * RING <= ns; s += UN * RING) {      // steady state: no condition inside the body
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int k = 0; k < RING; ++k) {
                    step(k, s + u * RING + k);
                    load(k, s + u * RING + k + RING);
                }
        }
        for (; s + 2 * RING <= ns; s += RING) {
#pragma unroll
            for (int k = 0; k < RING; ++k) {
                step(k, s + k);
                load(k, s + k + RING);
            }
        }
        for (; s < ns; s += RING) {
#pragma unroll
            for (int k = 0; k < RING; ++k) {
                if (s + k < ns) {
                    step(k, s + k);
                    if (s + k + RING < ns) load(k, s + k + RING);
                }
            }
        }
    }
    // ---- backward: x_c = D~_c^-1 y_c - sum_upper C_cm x_m -----------------------------------------
    {
        double xp[B];
#pragma unroll
        for (int r = 0; r < B; ++r) xp[r] = 0.0;
        double2 buf[RING][L::PB], ybuf[RING][YLDS ? 1 : L::PY];
        double aa[RING][B];
        bool okk[RING];
        long cc[RING];
        // fields >= nadd of addto count as zero: read a valid array instead and multiply by a 0/1 mask (no branch)
        const double *asrc[B];
        double amask[B];
#pragma unroll
        for (int r = 0; r < B; ++r) {
            const bool use = addto && r < nadd;
            asrc[r] = (use ? addto : rhs) + (long)r * nt;
            amask[r] = use ? 1.0 : 0.0;
        }
        auto load = [&](int k, int step) {
            okk[k] = tile_cell(G, ti, step, l0, c);
            cc[k] = c;
            load_chunk<L::PB>(bwd + chunk_idx(G, tile, step) * rowB, la, RS >> 1, buf[k]);
            if (!YLDS) load_chunk<(YLDS ? 1 : L::PY)>(ytmp + chunk_idx(G, tile, step) * rowY, la, RS >> 1, ybuf[k]);
            const long cs = okk[k] ? c : park;                // branch-free, as in the forward sweep
#pragma unroll
            for (int r = 0; r < B; ++r) aa[k][r] = asrc[r][cs];
        };
        auto step = [&](int k, int s) {
            double xn[3][B], xv[B], yv[B];
#pragma unroll
            for (int r = 0; r < B; ++r) {
                xn[0][r] = xp[r];
                xn[1][r] = __shfl_down(xp[r], 1, 64);
                xn[2][r] = __shfl_down(xp[r], G.t1, 64);
                yv[r] = YLDS ? ylds[((long)s * B + r) * 64 + lane] : chunk_get<(YLDS ? 1 : L::PY)>(ybuf[k], r);
            }
#pragma unroll
            for (int r = 0; r < B; ++r) {
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < B; ++q) v += chunk_get<L::PB>(buf[k], (3 * B + r) * B + q) * yv[q];
                xv[r] = v;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int q = 0; q < B; ++q) xv[r] -= chunk_get<L::PB>(buf[k], (a * B + r) * B + q) * xn[a][q];
#pragma unroll
            for (int r = 0; r < B; ++r) {
                xv[r] = okk[k] ? xv[r] : 0.0;
                xp[r] = xv[r];
                // unconditional store: lanes without a cell write 0.0 to entry `lane` of x's lower halo plane (never read
                // with a non-zero coefficient on one GPU, overwritten by the next halo exchange on several)
                x[(long)r * nt + (okk[k] ? cc[k] : park)] = okk[k] ? amask[r] * aa[k][r] + xv[r] : 0.0;
            }
        };
        // the forward sweep's y of the last steps may still be in flight as stores: same-lane same-address
        // loads are ordered after them by the memory pipeline (YLDS: same lane, same LDS address, program order)
#pragma unroll
        for (int k = 0; k < RING; ++k)
            if (ns - 1 - k >= 0) load(k, ns - 1 - k);
        int s = ns - 1;
        for (; s - (UN + 1) * RING + 1 >= 0; s -= UN * RING) {   // steady state: no condition inside the body
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int k = 0; k < RING; ++k) {
                    step(k, s - u * RING - k);
                    load(k, s - u * RING - k - RING);
                }
        }
        for (; s - 2 * RING + 1 >= 0; s -= RING) {
#pragma unroll
            for (int k = 0; k < RING; ++k) {
                step(k, s - k);
                load(k, s - k - RING);
            }
        }
        for (; s >= 0; s -= RING) {
#pragma unroll
            for (int k = 0; k < RING; ++k) {
                if (s - k >= 0) {
                    step(k, s - k);
                    if (s - k - RING >= 0) load(k, s - k - RING);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-wave sweep: B wavefronts per tile, wave r computes block ROW r of every step (field r of y and x).  The serial
// recurrence of a step -- what bounds the one-wave kernel above: ~130 dependent instructions per step issued by a single
// wave -- is cut to 3B loads, 3B FMAs and one LDS exchange per wave: the fields of a step's result go through the LDS
// (where the intermediate vector y lives anyway), one s_barrier per step; neighbours' values are LDS reads at
// lane-1 / lane-t1 (clamped: the coefficient of a neighbour that does not exist is zero) instead of shuffles.
// The barrier orders LDS accesses only (workgroup-scope fences restricted to the local address space + s_barrier):
// __syncthreads() would also drain the prefetched global loads.
// LDS: y of every step when it fits (YLDS), else a two-step ring (y then goes through HBM for the backward sweep).
#define TP_LDS_BARRIER()                                                        \
    do {                                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");         \
        __builtin_amdgcn_s_barrier();                                           \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");         \
    } while (0)
// two doubles at 8-byte alignment: one 16-byte access per lane where the hardware allows unaligned vector accesses
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));

template <int B> struct IluMwLayout {
    static constexpr int PFR = (3 * B + 1) / 2;      // double2 pairs per forward row: B_a[r][q], a = 0..2 (+ padding half)
    static constexpr int PBR = (4 * B + 1) / 2;      // pairs per backward row: C_a[r][q], a = 0..2, then D~^-1[r][q]
};

// (The lambdas are force-inlined: left to its heuristics the compiler outlines them in the larger instantiations, the
// register arrays they capture by reference then live in scratch memory, and the sweep is ten times slower.)
// BLK: grid-layout vectors moved in blocks of RF / RB steps (below); off for the short 2-D tiles, where the delayed block stores
// only lengthen the tail
// WS (tp_options.ilu_whole): the whole slab is ONE block.  A launch sweeps the tiles of one tile-diagonal (`tiles`), forward
// (phase 1) or backward (phase 2); a neighbour value across a tile face was produced by an EARLIER launch and is read from
// global memory -- y from ytmp, the raw backward result from xtmp -- instead of the LDS.  WS implies !YLDS.
template <int B, bool YLDS, bool BLK, bool WS = false>
__global__ __launch_bounds__(64 * B) void k_ilu_solve_mw(IluGeom G, const double *__restrict__ fwd,
                                                         const double *__restrict__ bwd, const double *__restrict__ rhs,
                                                         double *ytmp, double *x, const double *addto, int nadd,
                                                         const int *__restrict__ tiles = nullptr, int phase = 3,
                                                         double *xtmp = nullptr) {
    static_assert(!(WS && YLDS), "whole-slab sweeps keep y in global memory");
    extern __shared__ double lds[];
    using M = IluMwLayout<B>;
    // ring depths: a wave may have 63 loads in flight; what the sweep's throughput follows is BYTES in flight per CU
    // (B waves x ring x 16-byte loads: 3 x 8 x 6 x 864 B = 124 KB on C4, against 44 KB for the one-wave kernel)
    constexpr int RF = 8, RB = YLDS ? 8 : 6;
    const int tile = WS ? tiles[blockIdx.x] : blockIdx.x, lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int NL = G.nl, ns = G.nsteps;
    const int la = min(lane, NL - 1);                 // idle lanes shadow the last live lane's loads ...
    const bool live = lane < NL;                      // ... and store to dump locations: no divergent branch in the loops
    const long nt = G.g.ntot;
    const TileInfo ti = tile_info(G, tile, la);
    const long park = min((long)lane, G.g.np - 1);
    const int lm1 = max(la - 1, 0), lmt = max(la - G.t1, 0), lp1 = min(la + 1, NL - 1), lpt = min(la + G.t1, NL - 1);
    const int slotsz = B * NL;
    double *yl = lds;                                  // slot s+1 holds y of step s (slot 0 = zeros); ring of 2 without YLDS
    double *xl = lds + (size_t)(YLDS ? ns + 1 : 2) * slotsz;      // two slots, by step parity
    double *ldump = xl + 2 * slotsz + threadIdx.x;     // where idle lanes store
    const long dump_off = (long)G.ntiles * ns * slotsz + (long)tile * 64 * B + threadIdx.x;
    double *gdump = ytmp + dump_off;
    // exact select between two already-loaded doubles by a lane mask (no branch: see the 0/1-factor note in `step`)
    auto pick = [](double a, double b, unsigned long long takeb) __attribute__((always_inline)) {
        return __longlong_as_double((long long)((((unsigned long long)__double_as_longlong(b)) & takeb) |
                                                (((unsigned long long)__double_as_longlong(a)) & ~takeb)));
    };
    const int T0 = tile % G.nt0, T1 = (tile / G.nt0) % G.nt1, T2 = tile / (G.nt0 * G.nt1);
    for (int i = threadIdx.x; i < slotsz; i += 64 * B) { yl[i] = 0.0; xl[i] = 0.0; xl[slotsz + i] = 0.0; }
    TP_LDS_BARRIER();
    int l0;
    long c;
    // ---- forward: y_c[r] = rhs_c[r] - sum_a B_a[r][:] y_(m_a) ------------------------------------------------------
    if (phase & 1) {
        double2 v[RF][M::PFR];
        double rr[RF], rrA[RF], rrB[RF];
        bool okk[RF];
        // WS: lower neighbours across a tile face.  Cell (l0, j, k) of this tile at step s; its -a1 neighbour when j == 0 is
        // lane (t1-1, k) of tile T1-1 at ITS step l0 + (t1-1) + k = s + t1 - 1; -a2 when k == 0: lane (j, t2-1) of tile
        // T2-1 at step s + t2 - 1; -a0 when l0 == 0 (step j + k): the same lane of tile T0-1 at its last cell, step
        // (t0-1) + j + k.  (Lower neighbour tiles are never partial along the crossed axis.)  Lanes without such a neighbour
        // load a valid dummy address and keep the LDS value (mask 0).
        double gy1[WS ? RF : 1][B], gy2[WS ? RF : 1][B], gy0[B];
        unsigned long long mk0[WS ? RF : 1];
        const unsigned long long mk1 = (WS && ti.j == 0 && T1 > 0) ? ~0ull : 0ull, mk2 = (WS && ti.k == 0 && T2 > 0) ? ~0ull : 0ull;
        const double *y1b = ytmp + ((long)(mk1 ? tile - G.nt0 : tile) * ns * B) * NL + (mk1 ? (G.t1 - 1) + G.t1 * ti.k : la);
        const double *y2b = ytmp + ((long)(mk2 ? tile - G.nt0 * G.nt1 : tile) * ns * B) * NL + (mk2 ? ti.j + G.t1 * (G.t2 - 1) : la);
        if (WS) {
            const bool c0 = T0 > 0;
            const double *y0b = ytmp + (((long)(c0 ? tile - 1 : tile) * ns + min(G.t0 - 1 + ti.j + ti.k, ns - 1)) * B) * NL + la;
#pragma unroll
            for (int q = 0; q < B; ++q) gy0[q] = c0 ? y0b[(long)q * NL] : 0.0;
        }
        // Right-hand side in the grid layout: a lane's cells of consecutive steps are consecutive doubles of ITS column, but
        // the lanes of a wave sit n0 doubles apart -- one cache line per lane.  Loaded one value per step, every line is
        // fetched from L2 sixteen times (the CU's L1 does not hold the ~270 lines the three waves touch per step until the
        // next step).  Loaded RF values at a time, back to back, it is fetched twice: blocks of RF steps, double buffered.
        const long cb = G.g.np + (long)ti.base0 + (long)G.g.n0 * (ti.base1 + ti.j) + G.g.np * (ti.base2 + ti.k) - (ti.j + ti.k);
        auto blockload = [&](double (&dst)[RF], int s) __attribute__((always_inline)) {      // values of steps s .. s+RF-1 (block clamped into the vector: unused values are masked)
            const double2_u *col = reinterpret_cast<const double2_u *>(rhs + (long)r * nt + min(max(cb + s, 0L), nt - RF));
#pragma unroll
            for (int q = 0; q < RF / 2; ++q) {
                const double2_u t = col[q];
                dst[2 * q] = t.x;
                dst[2 * q + 1] = t.y;
            }
        };
        auto load = [&](int k, int s) __attribute__((always_inline)) {
            okk[k] = tile_cell(G, ti, s, l0, c) && live;
            const double2 *ch = reinterpret_cast<const double2 *>(fwd + (chunk_idx(G, tile, s) * B + r) * (long)(2 * M::PFR * NL)) + la;
#pragma unroll
            for (int p = 0; p < M::PFR; ++p) v[k][p] = ch[(long)p * NL];
            if (!BLK) rr[k] = rhs[(long)r * nt + (okk[k] ? c : park)];
            if (WS) {
                const int s1 = min(s + G.t1 - 1, ns - 1), s2 = min(s + G.t2 - 1, ns - 1);
#pragma unroll
                for (int q = 0; q < B; ++q) {
                    gy1[WS ? k : 0][q] = y1b[((long)s1 * B + q) * NL];
                    gy2[WS ? k : 0][q] = y2b[((long)s2 * B + q) * NL];
                }
                mk0[WS ? k : 0] = (T0 > 0 && s == ti.j + ti.k) ? ~0ull : 0ull;
            }
        };
        auto step = [&](int k, int s, double rhs_k) __attribute__((always_inline)) {
            const double *yp = yl + (size_t)(YLDS ? s : (s & 1)) * slotsz;      // y of step s-1
            double acc[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int src = a == 0 ? la : a == 1 ? lm1 : lmt;
                double t = 0.0;
#pragma unroll
                for (int q = 0; q < B; ++q) {
                    double yv = yp[q * NL + src];
                    if (WS) yv = a == 0 ? pick(yv, gy0[q], mk0[WS ? k : 0]) : a == 1 ? pick(yv, gy1[WS ? k : 0][q], mk1) : pick(yv, gy2[WS ? k : 0][q], mk2);
                    t += chunk_get<M::PFR>(v[k], a * B + q) * yv;
                }
                acc[a] = t;
            }
            // (a 0/1 factor, not a select: the compiler turns `ok ? expr : 0` into a branch around the LDS reads, and a
            // divergent branch makes it drain the prefetched loads at every step)
            const double y = (okk[k] ? 1.0 : 0.0) * ((BLK ? rhs_k : rr[k]) - (acc[0] + acc[1] + acc[2]));
            double *dst = yl + (size_t)(YLDS ? s + 1 : ((s + 1) & 1)) * slotsz + r * NL + lane;
            *(live ? dst : ldump) = y;
            if (!YLDS) *(live ? ytmp + (chunk_idx(G, tile, s) * B + r) * (long)NL + lane : gdump) = y;
            TP_LDS_BARRIER();
        };
        // RF steps on the values of `cur`, with the block of the following RF steps loaded into `nxt` first
        auto half = [&](double (&cur)[RF], double (&nxt)[RF], int s, auto guarded) __attribute__((always_inline)) {
            if (BLK) blockload(nxt, s + RF);
#pragma unroll
            for (int k = 0; k < RF; ++k) {
                if (!decltype(guarded)::value || s + k < ns) {
                    step(k, s + k, cur[k]);
                    if (!decltype(guarded)::value || s + k + RF < ns) load(k, s + k + RF);
                }
            }
        };
        if (BLK) blockload(rrA, 0);
#pragma unroll
        for (int k = 0; k < RF; ++k)
            if (k < ns) load(k, k);
        int s = 0;
        for (; s + 3 * RF <= ns; s += 2 * RF) {          // steady state: no condition inside the body
            half(rrA, rrB, s, std::false_type{});
            half(rrB, rrA, s + RF, std::false_type{});
        }
        for (; s < ns; s += 2 * RF) {
            half(rrA, rrB, s, std::true_type{});
            if (s + RF < ns) half(rrB, rrA, s + RF, std::true_type{});
        }
    }
// ---- backward: x_c[r] = D~^-1[r][:] y_c - sum_a C_a[r][:] x_(m_a) -------------------------------------------------
    if (phase & 2) {
        double2 v[RB][M::PBR];
        double yb[RB][YLDS ? 1 : B], aA[RB], aB[RB], xq[RB], aa[RB];
        bool okk[RB];
        long cc[RB];
        // WS: upper neighbours across a tile face, from xtmp (the raw result of the tiles of earlier backward launches):
        // +a1 when j == tj-1: lane (0, k) of tile T1+1 at its step l0 + k = s - j; +a2 when k == tk-1: lane (j, 0) of tile
        // T2+1 at step s - k; +a0 when l0 == tt0-1 (step tt0-1 + j + k): the same lane of tile T0+1 at step j + k.
        double gx1[WS ? RB : 1][B], gx2[WS ? RB : 1][B], gx0[B];
        unsigned long long nk0[WS ? RB : 1];
        const unsigned long long nk1 = (WS && ti.j == ti.tj - 1 && T1 < G.nt1 - 1) ? ~0ull : 0ull,
                                 nk2 = (WS && ti.k == ti.tk - 1 && T2 < G.nt2 - 1) ? ~0ull : 0ull;
        const double *xsrc = WS ? xtmp : ytmp;
        const double *x1b = xsrc + ((long)(nk1 ? tile + G.nt0 : tile) * ns * B) * NL + (nk1 ? G.t1 * ti.k : la);
        const double *x2b = xsrc + ((long)(nk2 ? tile + G.nt0 * G.nt1 : tile) * ns * B) * NL + (nk2 ? ti.j : la);
        if (WS) {
            const bool c0 = T0 < G.nt0 - 1;
            const double *x0b = xsrc + (((long)(c0 ? tile + 1 : tile) * ns + min(ti.j + ti.k, ns - 1)) * B) * NL + la;
#pragma unroll
            for (int q = 0; q < B; ++q) gx0[q] = c0 ? x0b[(long)q * NL] : 0.0;
        }
        const bool use = addto && r < nadd;
        const double *asrc = (use ? addto : rhs) + (long)r * nt;
        const double amask = use ? 1.0 : 0.0;
        const long cb = G.g.np + (long)ti.base0 + (long)G.g.n0 * (ti.base1 + ti.j) + G.g.np * (ti.base2 + ti.k) - (ti.j + ti.k);
        // `addto` in blocks of RB steps, like the right-hand side of the forward sweep; the RB results of a block are stored
        // back to back at its end (x may alias addto: a block's loads are all issued before the stores of the block before)
        auto blockload = [&](double (&dst)[RB], int s) __attribute__((always_inline)) {      // values of steps s, s-1, .., s-RB+1
            const double2_u *col = reinterpret_cast<const double2_u *>(asrc + min(max(cb + s - (RB - 1), 0L), nt - RB));
#pragma unroll
            for (int q = 0; q < RB / 2; ++q) {
                const double2_u t = col[q];                // cells of steps s-(RB-1)+2q and s-(RB-1)+2q+1
                dst[RB - 1 - 2 * q] = t.x;
                dst[RB - 2 - 2 * q] = t.y;
            }
        };
        auto load = [&](int k, int s) __attribute__((always_inline)) {
            okk[k] = tile_cell(G, ti, s, l0, c) && live;
            const double2 *ch = reinterpret_cast<const double2 *>(bwd + (chunk_idx(G, tile, s) * B + r) * (long)(2 * M::PBR * NL)) + la;
#pragma unroll
            for (int p = 0; p < M::PBR; ++p) v[k][p] = ch[(long)p * NL];
            if (!YLDS) {
#pragma unroll
                for (int q = 0; q < B; ++q) yb[k][q] = ytmp[(chunk_idx(G, tile, s) * B + q) * (long)NL + la];
            }
            if (!BLK) { cc[k] = okk[k] ? c : park; aa[k] = asrc[cc[k]]; }
            if (WS) {
                const int s1 = max(s - ti.j, 0), s2 = max(s - ti.k, 0);
#pragma unroll
                for (int q = 0; q < B; ++q) {
                    gx1[WS ? k : 0][q] = x1b[((long)s1 * B + q) * NL];
                    gx2[WS ? k : 0][q] = x2b[((long)s2 * B + q) * NL];
                }
                nk0[WS ? k : 0] = (T0 < G.nt0 - 1 && s == ti.tt0 - 1 + ti.j + ti.k) ? ~0ull : 0ull;
            }
        };
        auto step = [&](int k, int s) __attribute__((always_inline)) {
            const double *xp = xl + (size_t)((s + 1) & 1) * slotsz;             // x of step s+1
            const double *yv = yl + (size_t)(s + 1) * slotsz;                   // (YLDS only)
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < B; ++q) t += chunk_get<M::PBR>(v[k], 3 * B + q) * (YLDS ? yv[q * NL + la] : yb[k][q]);
            double acc[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int src = a == 0 ? la : a == 1 ? lp1 : lpt;
                double u = 0.0;
#pragma unroll
                for (int q = 0; q < B; ++q) {
                    double xn = xp[q * NL + src];
                    if (WS) xn = a == 0 ? pick(xn, gx0[q], nk0[WS ? k : 0]) : a == 1 ? pick(xn, gx1[WS ? k : 0][q], nk1) : pick(xn, gx2[WS ? k : 0][q], nk2);
                    u += chunk_get<M::PBR>(v[k], a * B + q) * xn;
                }
                acc[a] = u;
            }
            const double xv = (okk[k] ? 1.0 : 0.0) * (t - (acc[0] + acc[1] + acc[2]));
            double *dst = xl + (size_t)(s & 1) * slotsz + r * NL + lane;
            *(live ? dst : ldump) = xv;
            if (WS) *(live ? xtmp + (chunk_idx(G, tile, s) * B + r) * (long)NL + lane : xtmp + dump_off) = xv;
            if (!BLK) x[(long)r * nt + cc[k]] = (okk[k] ? 1.0 : 0.0) * (amask * aa[k] + xv);
            xq[k] = xv;
            TP_LDS_BARRIER();
        };
        // RB steps with the addto values of `cur`, the block of the following RB steps loaded into `nxt` first
        auto half = [&](double (&cur)[RB], double (&nxt)[RB], int s, auto guarded) __attribute__((always_inline)) {
            bool okq[RB];
            if (BLK) blockload(nxt, s - RB);
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                okq[k] = false;
                if (!decltype(guarded)::value || s - k >= 0) {
                    okq[k] = okk[k];
                    step(k, s - k);
                    if (!decltype(guarded)::value || s - k - RB >= 0) load(k, s - k - RB);
                }
            }
            if (BLK) {
                // lanes without a cell write 0.0 to an entry of x's lower halo plane (as the one-wave sweep does); (16-byte stores
                // of neighbouring results, with single stores at the column ends, were measured: no gain)
#pragma unroll
                for (int k = 0; k < RB; ++k)
                    if (!decltype(guarded)::value || s - k >= 0)
                        x[(long)r * nt + (okq[k] ? cb + s - k : park)] = (okq[k] ? 1.0 : 0.0) * (amask * cur[k] + xq[k]);
            }
        };
        // (without YLDS the forward sweep's y stores of the last steps may still be in flight: every wave re-reads values
        // written by OTHER waves of the workgroup, so drain them and make them visible first)
        if (!YLDS && phase == 3) { __threadfence_block(); __syncthreads(); }
        if (BLK) blockload(aA, ns - 1);
#pragma unroll
        for (int k = 0; k < RB; ++k)
            if (ns - 1 - k >= 0) load(k, ns - 1 - k);
        int s = ns - 1;
        for (; s - 3 * RB + 1 >= 0; s -= 2 * RB) {       // steady state
            half(aA, aB, s, std::false_type{});
            half(aB, aA, s - RB, std::false_type{});
        }
        for (; s >= 0; s -= 2 * RB) {
            half(aA, aB, s, std::true_type{});
            if (s - RB >= 0) half(aB, aA, s - RB, std::true_type{});
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Block-ILU(1) per tile (sub_1_sub_pc_factor_levels 1: pc_cprilu1_gmres, twophase.py:653-668).
// Level-1 fill on the 7-point cell graph in natural order adds three lower and three upper couplings per row:
// pattern = 6 lower offsets (in increasing global index), the diagonal, 6 upper offsets (oracle/linalg.py:TiledILU1).
// One wavefront per tile as for ILU(0), lane <-> column (j,k), but the step of a cell is s = i0 + 2j + 4k: every row a
// row depends on -- (i0-1,j,k), (i0,j-1,k), (i0+1,j-1,k), (i0,j,k-1), (i0+1,j,k-1), (i0,j+1,k-1) -- then has a smaller
// step (s-1, s-2, s-1, s-4, s-3, s-2), so the sweeps keep the last FOUR results of every lane in registers and take the
// neighbours' values by cross-lane shuffles.  Chunks are [tile][step][entry][64 lanes] doubles (512-byte rows).
// The factorisation runs once per Newton step and is latency-, not bandwidth-bound: one launch per step value, each
// thread eliminating one row against the rows of earlier launches (read back from the chunk arrays through L2).
__host__ __device__ constexpr int ilu1_off(int i, int a) {
    //            lower: -e2        -e2+e0       -e2+e1       -e1          -e1+e0       -e0        diag
    constexpr int O[13][3] = {{0, 0, -1}, {1, 0, -1}, {0, 1, -1}, {0, -1, 0}, {1, -1, 0}, {-1, 0, 0}, {0, 0, 0},
    //            upper: +e0        +e1-e0       +e1          +e2-e1       +e2-e0       +e2
                              {1, 0, 0},  {-1, 1, 0}, {0, 1, 0},  {0, -1, 1}, {-1, 0, 1}, {0, 0, 1}};
    return O[i][a];
}
__host__ __device__ constexpr int ilu1_find(int d0, int d1, int d2) {
    for (int i = 0; i < 13; ++i)
        if (ilu1_off(i, 0) == d0 && ilu1_off(i, 1) == d1 && ilu1_off(i, 2) == d2) return i;
    return -1;
}
// pattern entry of (lower ik) + (upper iu), -1 when the product falls outside the pattern (level-2 fill: dropped)
__host__ __device__ constexpr int ilu1_target(int ik, int iu) {
    return ilu1_find(ilu1_off(ik, 0) + ilu1_off(7 + iu, 0), ilu1_off(ik, 1) + ilu1_off(7 + iu, 1),
                     ilu1_off(ik, 2) + ilu1_off(7 + iu, 2));
}
// stencil slot of a pattern entry (-1: a fill entry, initially zero)
__host__ __device__ constexpr int ilu1_slot(int i) {
    const int d0 = ilu1_off(i, 0), d1 = ilu1_off(i, 1), d2 = ilu1_off(i, 2);
    if ((d0 != 0) + (d1 != 0) + (d2 != 0) > 1) return -1;
    return d0 ? (d0 < 0 ? 1 : 2) : d1 ? (d1 < 0 ? 3 : 4) : d2 ? (d2 < 0 ? 5 : 6) : 0;
}

__device__ __forceinline__ bool tile_cell1(const IluGeom &G, const TileInfo &t, int s, int &l0, long &c) {
    l0 = s - 2 * t.j - 4 * t.k;
    const bool ok = (t.k < G.t2) && (t.j < t.tj) && (t.k < t.tk) && (l0 >= 0) && (l0 < t.tt0);
    c = G.g.np + (long)(t.base0 + l0) + (long)G.g.n0 * (t.base1 + t.j) + G.g.np * (t.base2 + t.k);
    return ok;
}

template <int B, int I>
struct Ilu1Init {          // row entries from the Jacobian (compile-time slot per pattern entry)
    static __device__ __forceinline__ void run(double (&F)[13][B * B], const bool (&in)[13], const double *J, long nt, long c) {
        constexpr int SL = ilu1_slot(I);
#pragma unroll
        for (int e = 0; e < B * B; ++e) F[I][e] = (SL >= 0 && in[I]) ? J[(long)((SL < 0 ? 0 : SL) * B * B + e) * nt + c] : 0.0;
        if constexpr (I + 1 < 13) Ilu1Init<B, I + 1>::run(F, in, J, nt, c);
    }
};

template <int B, int IK, int IU>
struct Ilu1Upd {           // F[target(IK,IU)] -= L * U_k[IU]
    static __device__ __forceinline__ void run(double (&F)[13][B * B], const double (&Lck)[B * B], const double *bk) {
        constexpr int TG = ilu1_target(IK, IU);
        if constexpr (TG >= 0) {
            double U[B * B];
#pragma unroll
            for (int e = 0; e < B * B; ++e) U[e] = bk[(long)(IU * B * B + e) * 64];
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) {
                    double v = 0.0;
#pragma unroll
                    for (int m = 0; m < B; ++m) v += Lck[r * B + m] * U[m * B + q];
                    F[TG][r * B + q] -= v;
                }
        }
        if constexpr (IU + 1 < 6) Ilu1Upd<B, IK, IU + 1>::run(F, Lck, bk);
    }
};

// RING: the rows of the last four steps live in an LDS ring [step & 3][entry][64 lanes] (k_ilu1_factor_tile) instead of
// being read back from the chunk array
template <int B, int IK, bool RING = false>
struct Ilu1Elim {          // eliminate the lower entry IK of the row against row k = c + offset(IK)
    static __device__ __forceinline__ void run(double (&F)[13][B * B], const bool (&in)[13], const IluGeom &G,
                                               const double *bwd, long chunk0, int s, int lane) {
        if (in[IK]) {
            constexpr int d0 = ilu1_off(IK, 0), d1 = ilu1_off(IK, 1), d2 = ilu1_off(IK, 2);
            const long row = RING ? (long)((s + d0 + 2 * d1 + 4 * d2) & 3) : chunk0 + s + d0 + 2 * d1 + 4 * d2;
            const double *bk = bwd + row * (long)(7 * B * B * 64) + (lane + d1 + G.t1 * d2);
            double Dk[B * B], Lck[B * B];
#pragma unroll
            for (int e = 0; e < B * B; ++e) Dk[e] = bk[(long)(6 * B * B + e) * 64];
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int q = 0; q < B; ++q) {
                    double v = 0.0;
#pragma unroll
                    for (int m = 0; m < B; ++m) v += F[IK][r * B + m] * Dk[m * B + q];
                    Lck[r * B + q] = v;
                }
#pragma unroll
            for (int e = 0; e < B * B; ++e) F[IK][e] = Lck[e];
            Ilu1Upd<B, IK, 0>::run(F, Lck, bk);
        }
        if constexpr (IK + 1 < 6) Ilu1Elim<B, IK + 1, RING>::run(F, in, G, bwd, chunk0, s, lane);
    }
};

template <int B>
__global__ __launch_bounds__(64) void k_ilu1_level(IluGeom G, const double *__restrict__ J, double *fwd, double *bwd, int s) {
    constexpr int BB = B * B;
    const int tile = blockIdx.x, lane = threadIdx.x;
    const TileInfo ti = tile_info(G, tile, lane);
    const long chunk0 = (long)tile * G.nsteps;
    int l0;
    long c;
    const bool ok = tile_cell1(G, ti, s, l0, c);
    double *fch = fwd + (chunk0 + s) * (long)(6 * BB * 64) + lane;
    double *bch = bwd + (chunk0 + s) * (long)(7 * BB * 64) + lane;
    if (!ok) {                       // no row here: zero blocks, so that the sweeps need no masks
        for (int e = 0; e < 6 * BB; ++e) fch[(long)e * 64] = 0.0;
        for (int e = 0; e < 7 * BB; ++e) bch[(long)e * 64] = 0.0;
        return;
    }
    double F[13][BB];
    bool in[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int a0 = l0 + ilu1_off(i, 0), a1 = ti.j + ilu1_off(i, 1), a2 = ti.k + ilu1_off(i, 2);
        in[i] = a0 >= 0 && a0 < ti.tt0 && a1 >= 0 && a1 < ti.tj && a2 >= 0 && a2 < ti.tk;
    }
    Ilu1Init<B, 0>::run(F, in, J, G.g.ntot, c);
    Ilu1Elim<B, 0>::run(F, in, G, bwd, chunk0, s, lane);
    double D[B][B], Di[B][B];
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
        for (int q = 0; q < B; ++q) D[r][q] = F[6][r * B + q];
    inv_block<B>(D, Di);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int e = 0; e < BB; ++e) {
            fch[(long)(i * BB + e) * 64] = F[i][e];
            bch[(long)(i * BB + e) * 64] = F[7 + i][e];
        }
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
        for (int q = 0; q < B; ++q) bch[(long)(6 * BB + r * B + q) * 64] = Di[r][q];
}

// ILU(1) packed stream: does this lane's cell exist at step s in the NOMINAL tile, how many live lanes lie below it, how many
// there are (all lanes of the wave call this together)
__device__ __forceinline__ void packed_pos1(const IluGeom &G, const TileInfo &t, int lane, int s, bool &live, int &pos, int &cnt) {
    const int l0 = s - 2 * t.j - 4 * t.k;
    live = lane < G.nl && l0 >= 0 && l0 < G.t0;
    const unsigned long long m = __ballot(live);
    pos = live ? (int)__popcll(m & ((1ull << lane) - 1ull)) : 0;
    cnt = G.pref[s + 1] - G.pref[s];
}

// The whole factorisation of a tile by ONE workgroup of B wavefronts: steps in order, the rows of the last four steps (upper
// blocks and D~^-1: what later rows eliminate against) in an LDS ring, the Jacobian read once, the factor written once --
// straight into the packed rows the sweeps stream (PK) or into padded 64-lane rows.  Wave w owns block-row w of every block of
// the row being eliminated (L = F D_k^-1 and F -= L U act on block rows independently); only the diagonal block is exchanged
// (LDS) for its inverse.  k_ilu1_level does the same with one launch per step and reads every row back up to six times
// (C4: 3.2 ms + 0.5 ms repacking against 1.5 ms here).  What bounds this kernel is neither arithmetic nor latency (a one-wave
// version, no prefetch, full barriers: the same 1.5 ms; with eliminations AND stores switched off still 1.2 ms): it is the
// Jacobian read.  The lanes of a step hold cells of 54-64 different grid lines, so each of the 63 plane loads touches one
// 128-byte line per lane for 8 bytes, and the 16 steps that share a line are ~20 us apart with a working set of 435 KB per
// tile (13 MB per XCD): the lines are fetched again from the Infinity Cache / HBM every step, a 16x read amplification.  The
// ILU(0) path avoids it with a chunk-ordered copy of J (k_ilu_gather, one coalesced pass); the same pass here would bring this
// kernel to roughly 0.4 + 0.4 ms -- 2 % of a Newton step of pc_cprilu1_gmres, not done.
template <int B, int I>
struct Ilu1InitRow {       // one block row of the row's entries from a register copy of the cell's seven Jacobian block rows
    static __device__ __forceinline__ void run(double (&F)[13][B], const bool (&in)[13], const double (&Jc)[7][B]) {
        constexpr int SL = ilu1_slot(I);
#pragma unroll
        for (int q = 0; q < B; ++q) F[I][q] = (SL >= 0 && in[I]) ? Jc[SL < 0 ? 0 : SL][q] : 0.0;
        if constexpr (I + 1 < 13) Ilu1InitRow<B, I + 1>::run(F, in, Jc);
    }
};
template <int B, int IK, int IU>
struct Ilu1UpdRow {        // F[target(IK,IU)] -= L * U_k[IU]   (one block row)
    static __device__ __forceinline__ void run(double (&F)[13][B], const double (&L)[B], const double *bk) {
        constexpr int TG = ilu1_target(IK, IU);
        if constexpr (TG >= 0) {
#pragma unroll
            for (int q = 0; q < B; ++q) {
                double v = 0.0;
#pragma unroll
                for (int m = 0; m < B; ++m) v += L[m] * bk[(IU * B * B + m * B + q) * 64];
                F[TG][q] -= v;
            }
        }
        if constexpr (IU + 1 < 6) Ilu1UpdRow<B, IK, IU + 1>::run(F, L, bk);
    }
};
template <int B, int IK>
struct Ilu1ElimRow {       // eliminate the lower entry IK against row k = c + offset(IK), rows of earlier steps in the LDS ring
    static __device__ __forceinline__ void run(double (&F)[13][B], const bool (&in)[13], const IluGeom &G, const double *ring,
                                               int s, int lane) {
        if (in[IK]) {
            constexpr int d0 = ilu1_off(IK, 0), d1 = ilu1_off(IK, 1), d2 = ilu1_off(IK, 2);
            const double *bk = ring + ((s + d0 + 2 * d1 + 4 * d2) & 3) * (7 * B * B * 64) + (lane + d1 + G.t1 * d2);
            double L[B];
#pragma unroll
            for (int q = 0; q < B; ++q) {
                double v = 0.0;
#pragma unroll
                for (int m = 0; m < B; ++m) v += F[IK][m] * bk[(6 * B * B + m * B + q) * 64];
                L[q] = v;
            }
#pragma unroll
            for (int q = 0; q < B; ++q) F[IK][q] = L[q];
            Ilu1UpdRow<B, IK, 0>::run(F, L, bk);
        }
        if constexpr (IK + 1 < 6) Ilu1ElimRow<B, IK + 1>::run(F, in, G, ring, s, lane);
    }
};

// Memory ordering: vmcnt counts loads AND stores and retires in order, so (i) the Jacobian rows of step s+1 are requested BEFORE
// the factor stores of step s (J, fwd, bwd are deliberately not __restrict__: the compiler then cannot sink the loads below the
// stores), and (ii) every lane stores on every step (lanes without a packed row into a dump row behind the arrays) and the
// prefetch is unconditional, so that the wait at the end of a step is a static vmcnt(#stores) instead of vmcnt(0).
template <int B, bool PK>
__global__ __launch_bounds__(64 * B) void k_ilu1_factor_tile(IluGeom G, const double *J, double *fwd, double *bwd) {
    constexpr int BB = B * B, NFE = 6 * BB, NBE = 7 * BB, NBP = (NBE + 1) & ~1;
    extern __shared__ double ring[];               // [4][NBE][64] rows of the last four steps, then [BB][64] diagonal blocks
    double *dex = ring + 4 * NBE * 64;
    // the prefix table of the packed rows, in LDS: read from global memory it is a vector load (the scalar cache is not
    // coherent with the stores of this kernel) whose vmcnt(0) drains the prefetch in every step
    int *spref = reinterpret_cast<int *>(dex + BB * 64);
    if constexpr (PK) {
        for (int i = threadIdx.x; i <= G.nsteps; i += blockDim.x) spref[i] = G.pref[i];
        __syncthreads();
    }
    const int tile = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const TileInfo ti = tile_info(G, tile, lane);
    const long chunk0 = (long)tile * G.nsteps;
    // this wave's block row of the cell's seven Jacobian blocks, loaded one step ahead (a lane without a cell reads the
    // first owned cell and never uses it)
    double Jn[7][B];
    int l0n;
    long cn;
    bool okn = tile_cell1(G, ti, 0, l0n, cn);
    {
        const long cs = okn ? cn : G.g.np;
#pragma unroll
        for (int sl = 0; sl < 7; ++sl)
#pragma unroll
            for (int q = 0; q < B; ++q) Jn[sl][q] = J[(long)(sl * BB + w * B + q) * G.g.ntot + cs];
    }
    for (int s = 0; s < G.nsteps; ++s) {
        const int l0 = l0n;
        const bool ok = okn;
        double Jc[7][B];
#pragma unroll
        for (int sl = 0; sl < 7; ++sl)
#pragma unroll
            for (int q = 0; q < B; ++q) Jc[sl][q] = Jn[sl][q];
        {
            okn = tile_cell1(G, ti, min(s + 1, G.nsteps - 1), l0n, cn);
            const long cs = okn ? cn : G.g.np;
#pragma unroll
            for (int sl = 0; sl < 7; ++sl)
#pragma unroll
                for (int q = 0; q < B; ++q) Jn[sl][q] = J[(long)(sl * BB + w * B + q) * G.g.ntot + cs];
        }
        bool live = true;
        int pos = lane, cnt = 64;
        int prow = 0;
        if constexpr (PK) {                // (packed_pos1 with the table in LDS)
            const int lp = s - 2 * ti.j - 4 * ti.k;
            live = lane < G.nl && lp >= 0 && lp < G.t0;
            const unsigned long long m = __ballot(live);
            pos = live ? (int)__popcll(m & ((1ull << lane) - 1ull)) : 0;
            prow = spref[s];
            cnt = spref[s + 1] - prow;
        }
        double F[13][B];
        if (ok) {
            bool in[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                const int a0 = l0 + ilu1_off(i, 0), a1 = ti.j + ilu1_off(i, 1), a2 = ti.k + ilu1_off(i, 2);
                in[i] = a0 >= 0 && a0 < ti.tt0 && a1 >= 0 && a1 < ti.tj && a2 >= 0 && a2 < ti.tk;
            }
            Ilu1InitRow<B, 0>::run(F, in, Jc);
            Ilu1ElimRow<B, 0>::run(F, in, G, ring, s, lane);
        } else {
#pragma unroll
            for (int i = 0; i < 13; ++i)
#pragma unroll
                for (int q = 0; q < B; ++q) F[i][q] = 0.0;
        }
#pragma unroll
        for (int q = 0; q < B; ++q) dex[(w * B + q) * 64 + lane] = F[6][q];
        TP_LDS_BARRIER();                  // diagonal block complete; every wave is done reading the ring for this step
        double D[B][B], Di[B][B];
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int q = 0; q < B; ++q) D[r][q] = ok ? dex[(r * B + q) * 64 + lane] : (r == q ? 1.0 : 0.0);
        inv_block<B>(D, Di);
        double *slot = ring + (s & 3) * (NBE * 64) + lane;          // (slot s & 3 held step s - 4)
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int q = 0; q < B; ++q) slot[(i * BB + w * B + q) * 64] = F[7 + i][q];
#pragma unroll
        for (int q = 0; q < B; ++q) slot[(6 * BB + w * B + q) * 64] = ok ? Di[w][q] : 0.0;
        {                                  // (padded rows: every lane; lanes without a row store zero blocks)
            if constexpr (PK) {
                // lanes without a packed row store into the dump row behind the last tile (64-lane stride)
                const long row0 = live ? (long)tile * G.ptot + prow : (long)G.ntiles * G.ptot;
                double *fo = fwd + row0 * (long)NFE + 2 * (live ? pos : lane);
                double *bo = bwd + row0 * (long)NBP + 2 * (live ? pos : lane);
                const long ps = live ? 2 * cnt : 128;
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int q = 0; q < B; ++q) {
                        const int e = i * BB + w * B + q;
                        fo[(long)(e >> 1) * ps + (e & 1)] = F[i][q];
                        bo[(long)(e >> 1) * ps + (e & 1)] = F[7 + i][q];
                    }
#pragma unroll
                for (int q = 0; q < B; ++q) {
                    const int e = 6 * BB + w * B + q;
                    bo[(long)(e >> 1) * ps + (e & 1)] = ok ? Di[w][q] : 0.0;
                }
            } else {
                double *fch = fwd + (chunk0 + s) * (long)(NFE * 64) + lane;
                double *bch = bwd + (chunk0 + s) * (long)(NBE * 64) + lane;
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int q = 0; q < B; ++q) {
                        fch[(long)(i * BB + w * B + q) * 64] = F[i][q];
                        bch[(long)(i * BB + w * B + q) * 64] = F[7 + i][q];
                    }
#pragma unroll
                for (int q = 0; q < B; ++q) bch[(long)(6 * BB + w * B + q) * 64] = ok ? Di[w][q] : 0.0;
            }
        }
        TP_LDS_BARRIER();                  // the ring slot is complete before the next step reads it
    }
}

// padded factor chunks [tile][step][entry][64 lanes] -> packed [tile][step][entry PAIR][live lanes][2] (once per
// factorisation).  Pairs: a sweep step then is NE/2 16-byte loads per lane instead of NE 8-byte ones -- a wave can have 63
// vector-memory instructions in flight (vmcnt is 6 bits), so the bytes it keeps in flight double.  An odd NE is padded to the
// next even number of entries (the pad is never written: allocations are zeroed).
template <int NE>
__global__ __launch_bounds__(64) void k_ilu1_repack(IluGeom G, const double *__restrict__ src, double *__restrict__ dst) {
    constexpr int NEP = (NE + 1) & ~1;
    const int tile = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
    const TileInfo ti = tile_info(G, tile, lane);
    bool live;
    int pos, cnt;
    packed_pos1(G, ti, lane, s, live, pos, cnt);
    if (!live) return;
    const double *in = src + ((long)tile * G.nsteps + s) * (long)(NE * 64) + lane;
    double *out = dst + ((long)tile * G.ptot + G.pref[s]) * (long)NEP + 2 * pos;
#pragma unroll 9
    for (int e = 0; e < NE; ++e) out[(long)(e >> 1) * (2 * cnt) + (e & 1)] = in[(long)e * 64];
}

// x = addto + (L U)^-1 r for one tile per wavefront.  PF = chunks kept in registers (PF - 1 steps of loads in flight): one wave
// per CU has nothing to hide the HBM latency with but its own prefetch depth -- with PF = 2 a step lasts one memory round trip
// (1.5 us on C4, whatever the chunk holds)
template <int B, int PF, bool PK>
__global__ __launch_bounds__(64) void k_ilu1_solve(IluGeom G, const double *__restrict__ fwd, const double *__restrict__ bwd,
                                                   const double *__restrict__ rhs, double *__restrict__ ytmp, double *x,
                                                   const double *addto, int nadd) {
    constexpr int BB = B * B, NF = 6 * BB, NB = 7 * BB, NBP = (NB + 1) & ~1;
    static_assert(NF % 2 == 0, "forward rows are loaded in pairs");
    // the prefix table of the packed rows in LDS: read from global memory it is a VECTOR load (this kernel stores, so the
    // scalar cache may not be used) whose result is needed at once for the chunk address -- and vmcnt retires in order, so
    // that wait drained every prefetched chunk at every step
    extern __shared__ int spref[];
    if constexpr (PK) {
        for (int i = threadIdx.x; i <= G.nsteps; i += 64) spref[i] = G.pref[i];
        __syncthreads();
    }
    auto ppos = [&](const TileInfo &t, int ln, int st, int &pp, int &row0, int &cn) __attribute__((always_inline)) {
        const int lq = st - 2 * t.j - 4 * t.k;
        const bool lv = ln < G.nl && lq >= 0 && lq < G.t0;
        const unsigned long long m = __ballot(lv);
        pp = lv ? (int)__popcll(m & ((1ull << ln) - 1ull)) : 0;
        row0 = spref[st];
        cn = spref[st + 1] - row0;
    };
    const int tile = blockIdx.x, lane = threadIdx.x;
    const long nt = G.g.ntot;
    const TileInfo ti = tile_info(G, tile, lane);
    const long chunk0 = (long)tile * G.nsteps;
    const int ns = G.nsteps;
    const long park = min((long)lane, G.g.np - 1);     // an entry of the lower halo plane: what cell-less lanes read
    const int lane_dn = (lane - G.t1 + 1) & 63, lane_up = (lane + G.t1 - 1) & 63;
    int l0;
    long c;
    // ---- forward: y_c = r_c - sum_lower L_co y_(c+o) ---------------------------------------------------------------
    {
        double yh[4][B];                   // this lane's results of steps s-1 .. s-4
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < B; ++r) yh[d][r] = 0.0;
        struct Buf { double v[NF], rr[B]; bool ok; };
        Buf buf[3];
        auto load = [&](Buf &k, int step) {
            k.ok = tile_cell1(G, ti, step, l0, c);
            if constexpr (PK) {            // packed copy: rows of the live lanes only (dead lanes read slot 0: masked below)
                int pp, r0, cn;
                ppos(ti, lane, step, pp, r0, cn);
                const double2 *ch2 = reinterpret_cast<const double2 *>(fwd + ((long)tile * G.ptot + r0) * (long)NF) + pp;
#pragma unroll
                for (int e = 0; e < NF / 2; ++e) {
                    const double2 t2 = ch2[(long)e * cn];
                    k.v[2 * e] = t2.x;
                    k.v[2 * e + 1] = t2.y;
                }
            } else {
                const double *ch = fwd + (chunk0 + step) * (long)(NF * 64) + lane;
#pragma unroll
                for (int e = 0; e < NF; ++e) k.v[e] = ch[(long)e * 64];
            }
            const long cs = k.ok ? c : park;
#pragma unroll
            for (int r = 0; r < B; ++r) k.rr[r] = rhs[(long)r * nt + cs];
        };
        auto step = [&](const Buf &k, int s) {
            double yn[6][B], y[B];
#pragma unroll
            for (int r = 0; r < B; ++r) {
                yn[0][r] = __shfl_up(yh[3][r], G.t1, 64);      // (0,0,-1)  step s-4
                yn[1][r] = __shfl_up(yh[2][r], G.t1, 64);      // (1,0,-1)  step s-3
                yn[2][r] = __shfl(yh[1][r], lane_dn, 64);      // (0,1,-1)  step s-2
                yn[3][r] = __shfl_up(yh[1][r], 1, 64);         // (0,-1,0)  step s-2
                yn[4][r] = __shfl_up(yh[0][r], 1, 64);         // (1,-1,0)  step s-1
                yn[5][r] = yh[0][r];                           // (-1,0,0)  step s-1
                y[r] = k.rr[r];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)            // blocks of couplings that leave the tile are stored as zeros
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int q = 0; q < B; ++q) y[r] -= k.v[i * BB + r * B + q] * yn[i][q];
            double *ych = ytmp + (chunk0 + s) * (long)(B * 64) + lane;
#pragma unroll
            for (int r = 0; r < B; ++r) {
                y[r] = k.ok ? y[r] : 0.0;
                ych[r * 64] = y[r];
                yh[3][r] = yh[2][r]; yh[2][r] = yh[1][r]; yh[1][r] = yh[0][r]; yh[0][r] = y[r];
            }
        };
        load(buf[0], 0);
        if constexpr (PF == 2) {
            for (int s = 0; s < ns; s += 2) {
                if (s + 1 < ns) load(buf[1], s + 1);
                step(buf[0], s);
                if (s + 1 < ns) {
                    if (s + 2 < ns) load(buf[0], s + 2);
                    step(buf[1], s + 1);
                }
            }
        } else {
            // three whole steps per trip and no condition inside: every load is issued (step index clamped: the last trips
            // re-read the last chunk), so the waits are static vmcnt(N) that leave the two younger chunks in flight
            load(buf[1], min(1, ns - 1));
            int s = 0;
            for (; s + 3 <= ns; s += 3) {
                load(buf[2], min(s + 2, ns - 1));
                step(buf[0], s);
                load(buf[0], min(s + 3, ns - 1));
                step(buf[1], s + 1);
                load(buf[1], min(s + 4, ns - 1));
                step(buf[2], s + 2);
            }
            if (s < ns) {
                step(buf[0], s);
                if (s + 1 < ns) step(buf[1], s + 1);
            }
        }
    }
    // ---- backward: x_c = D~_c^-1 (y_c - sum_upper U_co x_(c+o)) ----------------------------------------------------
    {
        double xh[4][B];                   // this lane's results of steps s+1 .. s+4
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < B; ++r) xh[d][r] = 0.0;
        struct Buf { double v[NB], yy[B], aa[B]; bool ok; long c; };
        Buf buf[3];
        const double *asrc[B];
        double amask[B];
#pragma unroll
        for (int r = 0; r < B; ++r) {
            const bool use = addto && r < nadd;
            asrc[r] = (use ? addto : rhs) + (long)r * nt;
            amask[r] = use ? 1.0 : 0.0;
        }
        auto load = [&](Buf &k, int step) {
            k.ok = tile_cell1(G, ti, step, l0, c);
            k.c = k.ok ? c : park;
            if constexpr (PK) {
                int pp, r0, cn;
                ppos(ti, lane, step, pp, r0, cn);
                const double2 *ch2 = reinterpret_cast<const double2 *>(bwd + ((long)tile * G.ptot + r0) * (long)NBP) + pp;
#pragma unroll
                for (int e = 0; e < NBP / 2; ++e) {
                    const double2 t2 = ch2[(long)e * cn];
                    k.v[2 * e] = t2.x;
                    if (2 * e + 1 < NB) k.v[2 * e + 1] = t2.y;
                }
            } else {
                const double *ch = bwd + (chunk0 + step) * (long)(NB * 64) + lane;
#pragma unroll
                for (int e = 0; e < NB; ++e) k.v[e] = ch[(long)e * 64];
            }
            const double *ych = ytmp + (chunk0 + step) * (long)(B * 64) + lane;
#pragma unroll
            for (int r = 0; r < B; ++r) {
                k.yy[r] = ych[r * 64];
                k.aa[r] = asrc[r][k.c];
            }
        };
        auto step = [&](const Buf &k, int s) {
            double xn[6][B], t[B], xv[B];
#pragma unroll
            for (int r = 0; r < B; ++r) {
                xn[0][r] = xh[0][r];                           // (1,0,0)   step s+1
                xn[1][r] = __shfl_down(xh[0][r], 1, 64);       // (-1,1,0)  step s+1
                xn[2][r] = __shfl_down(xh[1][r], 1, 64);       // (0,1,0)   step s+2
                xn[3][r] = __shfl(xh[1][r], lane_up, 64);      // (0,-1,1)  step s+2
                xn[4][r] = __shfl_down(xh[2][r], G.t1, 64);    // (-1,0,1)  step s+3
                xn[5][r] = __shfl_down(xh[3][r], G.t1, 64);    // (0,0,1)   step s+4
                t[r] = k.yy[r];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int q = 0; q < B; ++q) t[r] -= k.v[i * BB + r * B + q] * xn[i][q];
#pragma unroll
            for (int r = 0; r < B; ++r) {
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < B; ++q) v += k.v[6 * BB + r * B + q] * t[q];
                xv[r] = k.ok ? v : 0.0;
            }
#pragma unroll
            for (int r = 0; r < B; ++r) {
                xh[3][r] = xh[2][r]; xh[2][r] = xh[1][r]; xh[1][r] = xh[0][r]; xh[0][r] = xv[r];
                // lanes without a cell write 0.0 to an entry of x's lower halo plane (as the ILU(0) sweep does)
                x[(long)r * nt + k.c] = k.ok ? amask[r] * k.aa[r] + xv[r] : 0.0;
            }
        };
        load(buf[0], ns - 1);
        if constexpr (PF == 2) {
            for (int s = ns - 1; s >= 0; s -= 2) {
                if (s - 1 >= 0) load(buf[1], s - 1);
                step(buf[0], s);
                if (s - 1 >= 0) {
                    if (s - 2 >= 0) load(buf[0], s - 2);
                    step(buf[1], s - 1);
                }
            }
        } else {
            load(buf[1], max(ns - 2, 0));
            int s = ns - 1;
            for (; s - 2 >= 0; s -= 3) {
                load(buf[2], max(s - 2, 0));
                step(buf[0], s);
                load(buf[0], max(s - 3, 0));
                step(buf[1], s - 1);
                load(buf[1], max(s - 4, 0));
                step(buf[2], s - 2);
            }
            if (s >= 0) {
                step(buf[0], s);
                if (s - 1 >= 0) step(buf[1], s - 1);
            }
        }
    }
}

// Rows as wide as the tile (t1*t2 double2) or as wide as the wave (64).  Compact rows save the padding bytes of tiles
// that do not fill a wave -- what the bandwidth-bound 3-D sweeps need -- but make the row stride a run-time value: the
// 14-18 loads of a step then need scalar address arithmetic instead of immediate offsets, and the small 2-D
// configurations, which are bound by the single wave's instruction issue, lose 15-25 % (C1: 0.24 -> 0.28 ms).
static bool ilu_compact(const tp_ctx *c) { return c->g.gn2 > 1 && c->ilu.t1 * c->ilu.t2 < 64; }

// ILU(1) factorisation: one wavefront per tile (k_ilu1_factor_tile, default) or one launch per step (k_ilu1_level)
static bool ilu1_per_tile() {
    static const bool v = !(getenv("TP_ILU1_FACTOR_TILE") && atoi(getenv("TP_ILU1_FACTOR_TILE")) == 0);
    return v;
}

static IluGeom geom_of(const tp_ctx *c) {
    IluGeom G;
    G.g = c->g;
    G.t0 = c->ilu.t0; G.t1 = c->ilu.t1; G.t2 = c->ilu.t2;
    G.nt0 = c->ilu.nt0; G.nt1 = c->ilu.nt1; G.nt2 = c->ilu.nt2;
    G.nsteps = c->ilu.nsteps;
    G.nl = G.t1 * G.t2;
    G.ntiles = c->ilu.ntiles;
    G.rs = ilu_compact(c) ? 2 * ((G.nl + ILU_ROW_ALIGN - 1) / ILU_ROW_ALIGN * ILU_ROW_ALIGN) : 128;
    G.ws = c->ilu.whole ? 1 : 0;
    G.pref = nullptr;           // (set by the ILU(1) sweeps and the repack kernel only: the factorisation uses padded rows)
    G.ptot = c->ilu.ptot;
    return G;
}

template <int B>
static void alloc_factor(IluData &d, bool compact) {
    using L = IluLayout<B>;
    const size_t chunks = (size_t)d.ntiles * d.nsteps,
                 rs = compact ? (size_t)2 * ((d.t1 * d.t2 + ILU_ROW_ALIGN - 1) / ILU_ROW_ALIGN * ILU_ROW_ALIGN) : 128;
    // (+ one dump chunk and 128 doubles of slack: where k_ilu_factor's column-less lanes store in the multi-wave layout)
    d.fwd.alloc((chunks + 1) * std::max((size_t)L::PF * rs, (size_t)B * ((3 * B + 1) / 2) * 2 * d.t1 * d.t2) + 128);   // (multi-wave layout: padded rows)
    d.bwd.alloc((chunks + 1) * L::PB * rs + 128);
    d.ytmp.alloc(chunks * L::PY * rs + (size_t)d.ntiles * 64 * B);     // + parking slots of idle lanes (branch-free stores)
    d.jt.alloc(chunks * L::PJ * rs);
}

void ilu_setup(tp_ctx *c) {
    IluData &d = c->ilu;
    const GridDev &g = c->g;
    int t0 = c->opt.ilu_t0, t1 = c->opt.ilu_t1, t2 = c->opt.ilu_t2;
    if (t0 <= 0) t0 = g.n0;
    if (t1 <= 0) t1 = (g.n2 == 1) ? 64 : 8;
    if (t2 <= 0) t2 = (g.n2 == 1) ? 1 : 8;
    t0 = std::min(t0, g.n0);
    t1 = std::min(t1, g.n1);
    t2 = std::min(t2, g.n2);
    TP_REQUIRE(t0 >= 1 && t1 >= 1 && t2 >= 1 && t1 * t2 <= 64,
               "ILU tile must satisfy t1*t2 <= 64 (one wavefront per tile)");
    d.t0 = t0; d.t1 = t1; d.t2 = t2;
    d.nt0 = (g.n0 + t0 - 1) / t0;
    d.nt1 = (g.n1 + t1 - 1) / t1;
    d.nt2 = (g.n2 + t2 - 1) / t2;
    d.ntiles = d.nt0 * d.nt1 * d.nt2;
    TP_REQUIRE(c->opt.ilu_levels == 0 || c->opt.ilu_levels == 1, "ilu_levels must be 0 or 1");
    TP_REQUIRE(!(c->opt.ilu_whole && c->opt.ilu_levels), "ilu_whole (one bjacobi block per rank) is implemented for block-ILU(0)");
    d.levels = c->opt.ilu_levels;
    static const bool mw_on = !(getenv("TP_ILU_MW") && atoi(getenv("TP_ILU_MW")) == 0);
    d.mw = mw_on && d.levels == 0;
    d.nsteps = d.levels ? t0 + 2 * (t1 - 1) + 4 * (t2 - 1) : t0 + t1 + t2 - 2;
    d.slots = (long)d.ntiles * d.nsteps * 64;
    c->graph_epoch++;            // new tile layout / factor buffers: captured pc_apply graphs are stale
    if (d.levels) {
        const size_t chunks = (size_t)d.ntiles * d.nsteps, bb = (size_t)c->b * c->b;
        static const bool pack1 = !(getenv("TP_ILU1_PACK") && atoi(getenv("TP_ILU1_PACK")) == 0);
        if (pack1 && ilu1_per_tile()) {     // the per-tile factorisation writes the packed rows itself: no padded arrays
            d.fwd.free(); d.bwd.free();
        } else {
            d.fwd.alloc(chunks * 6 * bb * 64);
            d.bwd.alloc(chunks * 7 * bb * 64);
        }
        d.ytmp.alloc(chunks * c->b * 64);
        d.jt.free();
        d.whole = false;
        d.ptot = 0;
        if (pack1) {
            std::vector<int> pf(d.nsteps + 1, 0);
            for (int s = 0; s < d.nsteps; ++s) {
                int cnt = 0;
                for (int k = 0; k < t2; ++k)
                    for (int j = 0; j < t1; ++j) cnt += (s - 2 * j - 4 * k >= 0 && s - 2 * j - 4 * k < t0);
                pf[s + 1] = pf[s] + cnt;
            }
            d.ptot = pf[d.nsteps];
            d.pref.alloc(pf.size());
            copy_sync(c, d.pref.p, pf.data(), sizeof(int) * pf.size(), hipMemcpyHostToDevice);
            // (rows of entry pairs: k_ilu1_repack; + one 64-lane dump row for k_ilu1_factor_tile's idle lanes)
            d.fwdp.alloc(((size_t)d.ntiles * d.ptot + 64) * 6 * bb);
            d.bwdp.alloc(((size_t)d.ntiles * d.ptot + 64) * ((7 * bb + 1) & ~1));
        } else {
            d.fwdp.free(); d.bwdp.free(); d.pref.free();
        }
        return;
    }
    if (c->b == 3) alloc_factor<3>(d, ilu_compact(c)); else alloc_factor<2>(d, ilu_compact(c));
    d.whole = c->opt.ilu_whole != 0;
    d.ndiag = 0;
    if (d.whole) {
        TP_REQUIRE(d.mw, "ilu_whole needs the multi-wave sweep kernel (TP_ILU_MW=0 is set)");
        // tile-diagonals: every lower neighbour tile of a tile on diagonal d lies on diagonal d-1
        d.ndiag = d.nt0 + d.nt1 + d.nt2 - 2;
        std::vector<std::vector<int>> by(d.ndiag);
        for (int t = 0; t < d.ntiles; ++t)
            by[t % d.nt0 + (t / d.nt0) % d.nt1 + t / (d.nt0 * d.nt1)].push_back(t);
        std::vector<int> flat;
        d.diag_off.assign(1, 0);
        for (auto &v : by) { flat.insert(flat.end(), v.begin(), v.end()); d.diag_off.push_back((int)flat.size()); }
        d.diag_tiles.alloc(flat.size());
        copy_sync(c, d.diag_tiles.p, flat.data(), sizeof(int) * flat.size(), hipMemcpyHostToDevice);
        d.xtmp.alloc(d.ytmp.n);
    } else {
        d.diag_tiles.free();
        d.xtmp.free();
    }
}

void ilu_factor(tp_ctx *c) {
    TP_REQUIRE(c->jac_ready, "Jacobian not assembled");
    if (c->ilu.slots == 0) ilu_setup(c);
    const IluGeom G = geom_of(c);
    if (c->ilu.levels) {
        if (ilu1_per_tile()) {                 // one wavefront per tile, rows of the last four steps in LDS, output in the sweeps' layout
            const bool pk = c->ilu.ptot > 0;
            IluGeom Gp = G;
            if (pk) Gp.pref = c->ilu.pref.p;
            double *fo = pk ? c->ilu.fwdp.p : c->ilu.fwd.p, *bo = pk ? c->ilu.bwdp.p : c->ilu.bwd.p;
            const size_t lds = (size_t)(4 * 7 + 1) * c->b * c->b * 64 * sizeof(double) + (size_t)(G.nsteps + 2) * sizeof(int);
            const dim3 gr(c->ilu.ntiles), bl(64 * c->b);
#define TP_ILU1_FT(BQ, PKQ)                                                                                               \
    do {                                                                                                                  \
        TP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ilu1_factor_tile<BQ, PKQ>),                          \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                \
        hipLaunchKernelGGL((k_ilu1_factor_tile<BQ, PKQ>), gr, bl, lds, c->stream, Gp, c->J.p, fo, bo);                    \
    } while (0)
            if (c->b == 3) { if (pk) TP_ILU1_FT(3, true); else TP_ILU1_FT(3, false); }
            else           { if (pk) TP_ILU1_FT(2, true); else TP_ILU1_FT(2, false); }
#undef TP_ILU1_FT
            TP_HIP(hipGetLastError());
            return;
        }
        for (int s = 0; s < G.nsteps; ++s) {
            if (c->b == 3) hipLaunchKernelGGL((k_ilu1_level<3>), dim3(c->ilu.ntiles), dim3(64), 0, c->stream, G, c->J.p, c->ilu.fwd.p, c->ilu.bwd.p, s);
            else           hipLaunchKernelGGL((k_ilu1_level<2>), dim3(c->ilu.ntiles), dim3(64), 0, c->stream, G, c->J.p, c->ilu.fwd.p, c->ilu.bwd.p, s);
        }
        if (c->ilu.ptot > 0) {          // the sweeps stream a packed copy
            IluGeom Gp = G;
            Gp.pref = c->ilu.pref.p;
            const dim3 gr(c->ilu.ntiles, G.nsteps);
            if (c->b == 3) {
                hipLaunchKernelGGL((k_ilu1_repack<54>), gr, dim3(64), 0, c->stream, Gp, c->ilu.fwd.p, c->ilu.fwdp.p);
                hipLaunchKernelGGL((k_ilu1_repack<63>), gr, dim3(64), 0, c->stream, Gp, c->ilu.bwd.p, c->ilu.bwdp.p);
            } else {
                hipLaunchKernelGGL((k_ilu1_repack<24>), gr, dim3(64), 0, c->stream, Gp, c->ilu.fwd.p, c->ilu.fwdp.p);
                hipLaunchKernelGGL((k_ilu1_repack<28>), gr, dim3(64), 0, c->stream, Gp, c->ilu.bwd.p, c->ilu.bwdp.p);
            }
        }
        TP_HIP(hipGetLastError());
        return;
    }
    const int nseg = (G.nsteps + ILU_SEG - 1) / ILU_SEG;
    const bool cp = ilu_compact(c);
#define TP_ILU_FACTOR(BB, CC)                                                                                          \
    do {                                                                                                               \
        hipLaunchKernelGGL((k_ilu_gather<BB, CC>), dim3(c->ilu.ntiles, (IluLayout<BB>::PJ + ILU_PPT - 1) / ILU_PPT, nseg), \
                           dim3(64 * ILU_SEG), 0, c->stream, G, c->J.p, c->ilu.jt.p);                                   \
        if (c->ilu.whole) {                                                                                            \
            for (int dg = 0; dg < c->ilu.ndiag; ++dg)      /* one launch per tile-diagonal: its lower neighbours are done */ \
                hipLaunchKernelGGL((k_ilu_factor<BB, CC, true, true>), dim3(c->ilu.diag_off[dg + 1] - c->ilu.diag_off[dg]), dim3(64), 0, \
                                   c->stream, G, c->ilu.jt.p, c->ilu.fwd.p, c->ilu.bwd.p,                              \
                                   (const int *)c->ilu.diag_tiles.p + c->ilu.diag_off[dg]);                            \
        } else if (c->ilu.mw) hipLaunchKernelGGL((k_ilu_factor<BB, CC, true>), dim3(c->ilu.ntiles), dim3(64), 0, c->stream, G, \
                                          c->ilu.jt.p, c->ilu.fwd.p, c->ilu.bwd.p, (const int *)nullptr);              \
        else hipLaunchKernelGGL((k_ilu_factor<BB, CC, false>), dim3(c->ilu.ntiles), dim3(64), 0, c->stream, G, c->ilu.jt.p, \
                                c->ilu.fwd.p, c->ilu.bwd.p, (const int *)nullptr);                                     \
    } while (0)
    if (c->b == 3) { if (cp) TP_ILU_FACTOR(3, true); else TP_ILU_FACTOR(3, false); }
    else           { if (cp) TP_ILU_FACTOR(2, true); else TP_ILU_FACTOR(2, false); }
#undef TP_ILU_FACTOR
    TP_HIP(hipGetLastError());
}

template <int BB, bool DD, bool CC>
static void ilu_solve_launch(tp_ctx *c, const IluGeom &G, bool ylds, size_t ybytes, const double *r, double *x,
                             const double *addto, int nadd) {
    if (ylds) {
        // set before every launch: the attribute is per device, and a process-wide "already set" flag would be wrong for a
        // second device and racy between slab threads (the call is a host-side table update, legal inside a capture)
        TP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ilu_solve<BB, DD, true, CC>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        hipLaunchKernelGGL((k_ilu_solve<BB, DD, true, CC>), dim3(c->ilu.ntiles), dim3(64), ybytes, c->stream, G, c->ilu.fwd.p,
                           c->ilu.bwd.p, r, c->ilu.ytmp.p, x, addto, nadd);
    } else {
        hipLaunchKernelGGL((k_ilu_solve<BB, DD, false, CC>), dim3(c->ilu.ntiles), dim3(64), 0, c->stream, G, c->ilu.fwd.p,
                           c->ilu.bwd.p, r, c->ilu.ytmp.p, x, addto, nadd);
    }
}

void ilu_solve(tp_ctx *c, const double *r, double *x, const double *addto, int nadd) {
    if (nadd < 0) nadd = c->b;
    TP_REQUIRE(c->ilu.slots > 0, "ILU not factored");
    const IluGeom G = geom_of(c);
    if (c->ilu.levels) {
        IluGeom G1 = G;
        const bool pk = c->ilu.ptot > 0;
        if (pk) G1.pref = c->ilu.pref.p;
        const double *ff = pk ? c->ilu.fwdp.p : c->ilu.fwd.p, *bbk = pk ? c->ilu.bwdp.p : c->ilu.bwd.p;
        static const int pf = getenv("TP_ILU1_PF") ? atoi(getenv("TP_ILU1_PF")) : 3;
        const dim3 gr(c->ilu.ntiles), bl(64);
#define TP_ILU1_LAUNCH(BQ, PFQ, PKQ) \
    hipLaunchKernelGGL((k_ilu1_solve<BQ, PFQ, PKQ>), gr, bl, (size_t)(G.nsteps + 2) * sizeof(int), c->stream, G1, ff, bbk, r, c->ilu.ytmp.p, x, addto, nadd)
        if (c->b == 3) {
            if (pk) { if (pf >= 3) TP_ILU1_LAUNCH(3, 3, true); else TP_ILU1_LAUNCH(3, 2, true); }
            else TP_ILU1_LAUNCH(3, 2, false);
        } else {
            if (pk) { if (pf >= 3) TP_ILU1_LAUNCH(2, 3, true); else TP_ILU1_LAUNCH(2, 2, true); }
            else TP_ILU1_LAUNCH(2, 2, false);
        }
#undef TP_ILU1_LAUNCH
        TP_HIP(hipGetLastError());
        return;
    }
    if (c->ilu.mw) {
        const size_t slot = (size_t)c->b * G.nl * sizeof(double);
        const size_t dump = (size_t)64 * c->b * sizeof(double);            // where idle lanes store
        const size_t full = ((size_t)G.nsteps + 1 + 2) * slot + dump, ring = 4 * slot + dump;
        if (c->ilu.whole) {
            // one block per rank: forward over the tile-diagonals in ascending order, backward in descending order
            static const int blk_env_w = getenv("TP_ILU_BLOCK") ? atoi(getenv("TP_ILU_BLOCK")) : -1;
            const bool blkw = (blk_env_w >= 0 ? blk_env_w == 1 : c->g.gn2 > 1) && c->g.np >= 8;
            const IluData &d = c->ilu;
#define TP_ILU_WS_LAUNCH(BB, KK, DG, PH)                                                                               \
            hipLaunchKernelGGL((k_ilu_solve_mw<BB, false, KK, true>), dim3(d.diag_off[(DG) + 1] - d.diag_off[DG]), dim3(64 * BB), \
                               ring, c->stream, G, d.fwd.p, d.bwd.p, r, d.ytmp.p, x, addto, nadd,                      \
                               (const int *)d.diag_tiles.p + d.diag_off[DG], PH, d.xtmp.p)
            for (int ph = 1; ph <= 2; ++ph)
                for (int i = 0; i < d.ndiag; ++i) {
                    const int dg = ph == 1 ? i : d.ndiag - 1 - i;
                    if (c->b == 3) { if (blkw) TP_ILU_WS_LAUNCH(3, true, dg, ph); else TP_ILU_WS_LAUNCH(3, false, dg, ph); }
                    else           { if (blkw) TP_ILU_WS_LAUNCH(2, true, dg, ph); else TP_ILU_WS_LAUNCH(2, false, dg, ph); }
                }
#undef TP_ILU_WS_LAUNCH
            TP_HIP(hipGetLastError());
            return;
        }
        static const bool ylds_mw = !(getenv("TP_ILU_YLDS") && atoi(getenv("TP_ILU_YLDS")) == 0);
        const bool yl = ylds_mw && full <= 156 * 1024;
#define TP_ILU_MW_LAUNCH(BB, YY, KK)                                                                                    \
        do {                                                                                                            \
            TP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ilu_solve_mw<BB, YY, KK>),                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));   /* per device: every launch */ \
            hipLaunchKernelGGL((k_ilu_solve_mw<BB, YY, KK>), dim3(c->ilu.ntiles), dim3(64 * BB), YY ? full : ring,      \
                               c->stream, G, c->ilu.fwd.p, c->ilu.bwd.p, r, c->ilu.ytmp.p, x, addto, nadd);             \
        } while (0)
        static const int blk_env = getenv("TP_ILU_BLOCK") ? atoi(getenv("TP_ILU_BLOCK")) : -1;
        // 3-D tiles: whole axis-0 lines, long sweeps.  The block loads clamp their start into the vector; that is harmless only
        // while every cell of a clamped block is a halo-plane cell, i.e. a plane holds at least one block (RF = 8 values)
        const bool blk = (blk_env >= 0 ? blk_env == 1 : c->g.gn2 > 1) && c->g.np >= 8;
        if (c->b == 3) {
            if (yl) { if (blk) TP_ILU_MW_LAUNCH(3, true, true); else TP_ILU_MW_LAUNCH(3, true, false); }
            else    { if (blk) TP_ILU_MW_LAUNCH(3, false, true); else TP_ILU_MW_LAUNCH(3, false, false); }
        } else {
            if (yl) { if (blk) TP_ILU_MW_LAUNCH(2, true, true); else TP_ILU_MW_LAUNCH(2, true, false); }
            else    { if (blk) TP_ILU_MW_LAUNCH(2, false, true); else TP_ILU_MW_LAUNCH(2, false, false); }
        }
#undef TP_ILU_MW_LAUNCH
        TP_HIP(hipGetLastError());
        return;
    }
    static const bool deep = !(getenv("TP_ILU_DEPTH") && atoi(getenv("TP_ILU_DEPTH")) == 1);
    static const bool ylds_on = !(getenv("TP_ILU_YLDS") && atoi(getenv("TP_ILU_YLDS")) == 0);
    // y in LDS when the tile's whole intermediate vector fits one CU's 160 KB (one workgroup per CU then)
    const size_t ybytes = (size_t)G.nsteps * c->b * 64 * sizeof(double);
    const bool ylds = ylds_on && ybytes <= 152 * 1024;
    const bool cp = ilu_compact(c);
#define TP_ILU_PICK(BB)                                                                                               \
    do {                                                                                                              \
        if (deep) { if (cp) ilu_solve_launch<BB, true, true>(c, G, ylds, ybytes, r, x, addto, nadd);                  \
                    else ilu_solve_launch<BB, true, false>(c, G, ylds, ybytes, r, x, addto, nadd); }                  \
        else      { if (cp) ilu_solve_launch<BB, false, true>(c, G, ylds, ybytes, r, x, addto, nadd);                 \
                    else ilu_solve_launch<BB, false, false>(c, G, ylds, ybytes, r, x, addto, nadd); }                 \
    } while (0)
    if (c->b == 3) TP_ILU_PICK(3); else TP_ILU_PICK(2);
#undef TP_ILU_PICK
    TP_HIP(hipGetLastError());
}

}  // namespace tp

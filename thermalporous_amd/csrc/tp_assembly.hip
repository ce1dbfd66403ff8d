// DG0/TPFA residual + exact block Jacobian (+ temperature convection-diffusion operator S~) on the
// structured slab, hand-written for gfx950.
//
// Replaces the TSFC/PyOP2-generated cell and interior-facet kernels the reference runs for
//   F        singlephase.py:60-165,167-273 / twophase.py:67-235,237-411
//   dF/du    thermalmodel.py:36 (derivative(F,u), assembled into MatAIJ)
//   S~       preconditioners.py:11-163 (ConvDiffSchurPC), :165-333 (ConvDiffSchurTwoPhasesPC)
// and the closure laws of physicalparameters.py:37-98.
//
// One thread per owned cell (coalesced: consecutive lanes = consecutive cells along axis 0).  The
// thread evaluates the closures of its own cell once and of each face neighbour on the fly
// (7 closure evaluations per cell; the neighbour states come from L2, each state byte is fetched
// from HBM once), forms the six face fluxes with their derivatives w.r.t. both sides, and writes
// its residual entries, its diagonal block and its six off-diagonal blocks -- every Jacobian
// plane is written exactly once, fully coalesced, no atomics.  HBM-bound: ~616 B per cell
// (SURVEY.md 8d); the 3 exp + 1 pow per closure evaluation stay under the memory time.
#include "tp_common.hpp"
#include "tp_closures.hpp"
#include <cstdlib>

namespace tp {

// ------------------------------------------------------------------------------------------------
template <int NPH>
struct Props {
    double p, T, S;
    double ro, ro_p, ro_T;
    double Lo[4];                 // kr_o rho_o / mu_o and d/d(p,T,S)
    double rw, rw_p, rw_T;
    double Lw[4];
    double kT, kT_S;
    double mo, mw;
};

template <int NPH>
__device__ __forceinline__ Props<NPH> eval_props(double p, double T, double S, double phi, double kTs,
                                                 const DevPrm &q) {
    Props<NPH> r;
    r.p = p; r.T = T; r.S = S;
    oil_rho(p, T, q, r.ro, r.ro_p, r.ro_T);
    double mo, mo_T;
    oil_mu(T, q, mo, mo_T);
    r.mo = mo;
    if (NPH == 2) {
        water_rho(p, T, r.rw, r.rw_p, r.rw_T);
        double mw, mw_T;
        water_mu(T, mw, mw_T);
        r.mw = mw;
        const double kw_ = 1.0 - S;
        r.Lw[0] = kw_ * r.rw / mw;
        r.Lw[1] = kw_ * r.rw_p / mw;
        r.Lw[2] = kw_ * (r.rw_T / mw - r.rw * mw_T / (mw * mw));
        r.Lw[3] = -r.rw / mw;
        r.Lo[0] = S * r.ro / mo;
        r.Lo[1] = S * r.ro_p / mo;
        r.Lo[2] = S * (r.ro_T / mo - r.ro * mo_T / (mo * mo));
        r.Lo[3] = r.ro / mo;
        r.kT = phi * (S * q.ko + (1.0 - S) * q.kw) + (1.0 - phi) * q.kr;   // twophase.py:135,311
        r.kT_S = phi * (q.ko - q.kw);
    } else {
        r.Lo[0] = r.ro / mo;
        r.Lo[1] = r.ro_p / mo;
        r.Lo[2] = r.ro_T / mo - r.ro * mo_T / (mo * mo);
        r.Lo[3] = 0.0;
        r.kT = kTs;
        r.kT_S = 0.0;
        r.rw = r.rw_p = r.rw_T = 0.0;
        r.Lw[0] = r.Lw[1] = r.Lw[2] = r.Lw[3] = 0.0;
        r.mw = 1.0;
    }
    return r;
}

// Flux through one face from the '+' cell P to the '-' cell M and its derivatives w.r.t. both
// states.  gam = g*Delta_h/2 on the gravity axis, else 0; Ga = |e|/Delta_h.
template <int NPH, bool SCHUR>
__device__ __forceinline__ void face_flux(const Props<NPH> &P, const Props<NPH> &M, double TK, double gam,
                                          double Ga, const DevPrm &q, double *f, double (*dP)[NPH + 1],
                                          double (*dM)[NPH + 1], double &sP, double &sM) {
    constexpr int B = NPH + 1;
#pragma unroll
    for (int r = 0; r < B; ++r) {
        f[r] = 0.0;
#pragma unroll
        for (int c = 0; c < B; ++c) { dP[r][c] = 0.0; dM[r][c] = 0.0; }
    }
    sP = 0.0; sM = 0.0;
    auto phase = [&](double ce, double c0, bool to2, double rP, double rP_p, double rP_T, double rM,
                     double rM_p, double rM_T, const double *LP, const double *LM) {
        const double Phi = P.p - M.p - gam * (rP + rM);
        const bool up = Phi > 0.0;                      // gt(flow, 0): strict, ties -> '-' side
        const double L = up ? LP[0] : LM[0];
        const double Tu = up ? P.T : M.T;
        const double F = TK * L * Phi;
        const double dPhiP[3] = {1.0 - gam * rP_p, -gam * rP_T, 0.0};
        const double dPhiM[3] = {-1.0 - gam * rM_p, -gam * rM_T, 0.0};
        f[0] += q.w0 * c0 * F;
        f[1] += ce * Tu * F;
        if (to2) f[B - 1] += q.w2 * F;
#pragma unroll
        for (int c = 0; c < B; ++c) {
            const double dFP = TK * (L * dPhiP[c] + (up ? LP[c + 1] : 0.0) * Phi);
            const double dFM = TK * (L * dPhiM[c] + (up ? 0.0 : LM[c + 1]) * Phi);
            dP[0][c] += q.w0 * c0 * dFP;
            dM[0][c] += q.w0 * c0 * dFM;
            dP[1][c] += ce * Tu * dFP;
            dM[1][c] += ce * Tu * dFM;
            if (to2) { dP[B - 1][c] += q.w2 * dFP; dM[B - 1][c] += q.w2 * dFM; }
        }
        const double adv = ce * F;
        if (up) dP[1][1] += adv; else dM[1][1] += adv;
        if (SCHUR) { if (up) sP += adv; else sM += adv; }
    };
    if (NPH == 2) {
        phase(q.c_v_w, q.c_v_w, false, P.rw, P.rw_p, P.rw_T, M.rw, M.rw_p, M.rw_T, P.Lw, M.Lw);
        phase(q.c_v_o, q.c_v_o, true, P.ro, P.ro_p, P.ro_T, M.ro, M.ro_p, M.ro_T, P.Lo, M.Lo);
    } else {
        phase(q.c_v_o, 1.0, false, P.ro, P.ro_p, P.ro_T, M.ro, M.ro_p, M.ro_T, P.Lo, M.Lo);
    }
    // conduction with harmonic kT (singlephase.py:103,125)
    const double s2 = P.kT + M.kT;
    const double Hk = s2 > 0.0 ? 2.0 * P.kT * M.kT / s2 : 0.0;
    const double dT = P.T - M.T;
    f[1] += Hk * Ga * dT;
    dP[1][1] += Hk * Ga;
    dM[1][1] -= Hk * Ga;
    if (SCHUR) { sP += Hk * Ga; sM -= Hk * Ga; }
    if (NPH == 2) {
        const double dHP = s2 > 0.0 ? 2.0 * M.kT * M.kT / (s2 * s2) : 0.0;
        const double dHM = s2 > 0.0 ? 2.0 * P.kT * P.kT / (s2 * s2) : 0.0;
        dP[1][2] += Ga * dT * dHP * P.kT_S;
        dM[1][2] += Ga * dT * dHM * M.kT_S;
    }
}

struct AsmArgs {
    GridDev g;
    DevPrm q;
    const double *u, *phi, *kTs, *TK[3], *acc_old;
    double Vdt;          // |E|/dt
    double gam[3];       // g*h_a/2 on the gravity axis, else 0
    double Ga[3];        // |e_a|/h_a
    double *R, *J, *Sm;
};

template <int NPH, bool JAC, bool SCHUR>
__global__ __launch_bounds__(256) void k_assemble(AsmArgs a) {
    constexpr int B = NPH + 1;
    const long tid = xcd_tid();
    const GridDev &g = a.g;
    if (tid >= g.nown) return;
    const long c = g.np + tid;
    const int i2 = (int)(tid / g.np);
    const int rem = (int)(tid - (long)i2 * g.np);
    const int i1 = rem / g.n0;
    const int i0 = rem - i1 * g.n0;
    const long nt = g.ntot;
    const DevPrm &q = a.q;

    const double *up_ = a.u, *uT_ = a.u + nt, *uS_ = a.u + 2 * nt;
    const Props<NPH> me = eval_props<NPH>(up_[c], uT_[c], NPH == 2 ? uS_[c] : 0.0, a.phi[c], a.kTs[c], q);

    double R[B], Jd[B][B];
    double sd = 0.0;
    // ---- accumulation (singlephase.py:120,123 ; twophase.py:162-176) -----------------------------
    {
        const double phi = a.phi[c];
        const double rock = (1.0 - phi) * q.rho_r * q.c_r;
        if (NPH == 2) {
            const double S = me.S, T = me.T;
            const double Mw = phi * me.rw * (1.0 - S), Mo = phi * me.ro * S;
            const double dMw[3] = {phi * me.rw_p * (1.0 - S), phi * me.rw_T * (1.0 - S), -phi * me.rw};
            const double dMo[3] = {phi * me.ro_p * S, phi * me.ro_T * S, phi * me.ro};
            const double e0 = q.c_v_w * Mw + q.c_v_o * Mo;
            R[0] = q.w0 * (e0 - a.acc_old[c]) * a.Vdt;
            R[1] = (e0 * T + rock * T - a.acc_old[nt + c]) * a.Vdt;
            R[2] = q.w2 * (Mo - a.acc_old[2 * nt + c]) * a.Vdt;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double de0 = q.c_v_w * dMw[k] + q.c_v_o * dMo[k];
                Jd[0][k] = q.w0 * de0 * a.Vdt;
                Jd[1][k] = de0 * T * a.Vdt;
                Jd[2][k] = q.w2 * dMo[k] * a.Vdt;
            }
            Jd[1][1] += (e0 + rock) * a.Vdt;
            if (SCHUR) sd = (phi * q.c_v_o * S * me.ro + phi * q.c_v_w * (1.0 - S) * me.rw + rock) * a.Vdt;
        } else {
            const double T = me.T;
            const double Mo = phi * me.ro;
            const double dMo[2] = {phi * me.ro_p, phi * me.ro_T};
            R[0] = q.w0 * (Mo - a.acc_old[c]) * a.Vdt;
            R[1] = (q.c_v_o * Mo * T + rock * T - a.acc_old[nt + c]) * a.Vdt;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                Jd[0][k] = q.w0 * dMo[k] * a.Vdt;
                Jd[1][k] = q.c_v_o * dMo[k] * T * a.Vdt;
            }
            Jd[1][1] += (q.c_v_o * Mo + rock) * a.Vdt;
            if (SCHUR) sd = (phi * q.c_v_o * me.ro + rock) * a.Vdt;
        }
    }
    // ---- faces -----------------------------------------------------------------------------------
    const long stride[3] = {1, (long)g.n0, g.np};
    const int idx[3] = {i0, i1, g.off2 + i2};
    const int ext[3] = {g.n0, g.n1, g.gn2};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {          // 0: neighbour at -ax (I am '-'), 1: at +ax (I am '+')
            const int slot = 1 + 2 * ax + dir;
            const bool exists = dir ? (idx[ax] < ext[ax] - 1) : (idx[ax] > 0);
            double f[B], dP[B][B], dM[B][B], sP = 0.0, sM = 0.0;
            if (exists) {
                const long nb = dir ? c + stride[ax] : c - stride[ax];
                const Props<NPH> ot =
                    eval_props<NPH>(up_[nb], uT_[nb], NPH == 2 ? uS_[nb] : 0.0, a.phi[nb], a.kTs[nb], q);
                if (dir) {
                    face_flux<NPH, SCHUR>(me, ot, a.TK[ax][c], a.gam[ax], a.Ga[ax], q, f, dP, dM, sP, sM);
#pragma unroll
                    for (int r = 0; r < B; ++r) {
                        R[r] += f[r];
#pragma unroll
                        for (int k = 0; k < B; ++k) Jd[r][k] += dP[r][k];
                    }
                    if (SCHUR) sd += sP;
                } else {
                    face_flux<NPH, SCHUR>(ot, me, a.TK[ax][nb], a.gam[ax], a.Ga[ax], q, f, dP, dM, sP, sM);
#pragma unroll
                    for (int r = 0; r < B; ++r) {
                        R[r] -= f[r];
#pragma unroll
                        for (int k = 0; k < B; ++k) Jd[r][k] -= dM[r][k];
                    }
                    if (SCHUR) sd -= sM;
                }
            }
            if (JAC) {
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int k = 0; k < B; ++k) {
                        double v = 0.0;
                        if (exists) v = dir ? dM[r][k] : -dP[r][k];
                        a.J[((long)(slot * B + r) * B + k) * nt + c] = v;
                    }
            }
            if (SCHUR) a.Sm[(long)slot * nt + c] = exists ? (dir ? sM : -sP) : 0.0;
        }
    }
#pragma unroll
    for (int r = 0; r < B; ++r) a.R[(long)r * nt + c] = R[r];
    if (JAC) {
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int k = 0; k < B; ++k) a.J[((long)(r)*B + k) * nt + c] = Jd[r][k];
    }
    if (SCHUR) a.Sm[c] = sd;
}

// ---- LDS-tiled variant (north_star: "LDS staging of per-face neighbour blocks"; TP_ASM_LDS=1) ----------------------
// A workgroup of 256 threads owns a 16 x 4 x 4 tile of cells.  Phase 1 evaluates the closures ONCE per cell of the tile and
// of its six face halos (544 evaluations for 256 cells: 2.1 per cell instead of the 7 of k_assemble) into an LDS image,
// structure-of-arrays [value][slot of the 18 x 6 x 6 box]; phase 2 is k_assemble's cell body with every Props read from the
// LDS.  Same functions on the same inputs: results are bitwise those of k_assemble.  Measured on C4.
constexpr int ATX = 16, ATY = 4, ATZ = 4, ABX = ATX + 2, ABY = ATY + 2, ABZ = ATZ + 2, ANS = ABX * ABY * ABZ;
template <int NPH> struct PropsLds {
    static constexpr int NV = NPH == 2 ? 19 : 9;
    static __device__ __forceinline__ void put(double *l, int s, const Props<NPH> &r) {
        int v = 0;
        auto w = [&](double x) { l[(v++) * ANS + s] = x; };
        w(r.p); w(r.T); w(r.ro); w(r.ro_p); w(r.ro_T); w(r.Lo[0]); w(r.Lo[1]); w(r.Lo[2]); w(r.kT);
        if (NPH == 2) { w(r.S); w(r.Lo[3]); w(r.rw); w(r.rw_p); w(r.rw_T); w(r.Lw[0]); w(r.Lw[1]); w(r.Lw[2]); w(r.Lw[3]); w(r.kT_S); }
    }
    static __device__ __forceinline__ Props<NPH> get(const double *l, int s) {
        Props<NPH> r;
        int v = 0;
        auto rd = [&]() { return l[(v++) * ANS + s]; };
        r.p = rd(); r.T = rd(); r.ro = rd(); r.ro_p = rd(); r.ro_T = rd(); r.Lo[0] = rd(); r.Lo[1] = rd(); r.Lo[2] = rd(); r.kT = rd();
        if (NPH == 2) {
            r.S = rd(); r.Lo[3] = rd(); r.rw = rd(); r.rw_p = rd(); r.rw_T = rd();
            r.Lw[0] = rd(); r.Lw[1] = rd(); r.Lw[2] = rd(); r.Lw[3] = rd(); r.kT_S = rd();
        } else {
            r.S = 0.0; r.Lo[3] = 0.0; r.kT_S = 0.0; r.rw = r.rw_p = r.rw_T = 0.0;
            r.Lw[0] = r.Lw[1] = r.Lw[2] = r.Lw[3] = 0.0;
        }
        r.mo = r.mw = 1.0;          // (not used by the fluxes)
        return r;
    }
};

template <int NPH, bool JAC, bool SCHUR>
__global__ __launch_bounds__(256) void k_assemble_lds(AsmArgs a) {
    constexpr int B = NPH + 1;
    extern __shared__ double lds[];
    const GridDev &g = a.g;
    const long nt = g.ntot;
    const DevPrm &q = a.q;
    const int X0 = blockIdx.x * ATX, Y0 = blockIdx.y * ATY, Z0 = blockIdx.z * ATZ;
    const double *up_ = a.u, *uT_ = a.u + nt, *uS_ = a.u + 2 * nt;
    // ---- phase 1: closures of the tile and its face halos, once each --------------------------------------------------
    for (int s = threadIdx.x; s < ANS; s += 256) {
        const int bx = s % ABX, by = (s / ABX) % ABY, bz = s / (ABX * ABY);
        const int nh = (bx == 0 || bx == ABX - 1) + (by == 0 || by == ABY - 1) + (bz == 0 || bz == ABZ - 1);
        const int i0 = X0 + bx - 1, i1 = Y0 + by - 1, i2 = Z0 + bz - 1;
        const int gz = g.off2 + i2;
        if (nh > 1 || i0 < 0 || i0 >= g.n0 || i1 < 0 || i1 >= g.n1 || i2 < -1 || i2 > g.n2 || gz < 0 || gz >= g.gn2) continue;
        const long cc = g.np + (long)i0 + (long)g.n0 * i1 + g.np * i2;
        PropsLds<NPH>::put(lds, s, eval_props<NPH>(up_[cc], uT_[cc], NPH == 2 ? uS_[cc] : 0.0, a.phi[cc], a.kTs[cc], q));
    }
    __syncthreads();
    // ---- phase 2: one thread per cell of the tile (k_assemble's body; Props from the LDS image) ------------------------
    const int lx = threadIdx.x & (ATX - 1), ly = (threadIdx.x / ATX) & (ATY - 1), lz = threadIdx.x / (ATX * ATY);
    const int i0 = X0 + lx, i1 = Y0 + ly, i2 = Z0 + lz;
    if (i0 >= g.n0 || i1 >= g.n1 || i2 >= g.n2) return;
    const long c = g.np + (long)i0 + (long)g.n0 * i1 + g.np * i2;
    const int sme = (lx + 1) + ABX * ((ly + 1) + ABY * (lz + 1));
    const Props<NPH> me = PropsLds<NPH>::get(lds, sme);
    double R[B], Jd[B][B];
    double sd = 0.0;
    {
        const double phi = a.phi[c];
        const double rock = (1.0 - phi) * q.rho_r * q.c_r;
        if (NPH == 2) {
            const double S = me.S, T = me.T;
            const double Mw = phi * me.rw * (1.0 - S), Mo = phi * me.ro * S;
            const double dMw[3] = {phi * me.rw_p * (1.0 - S), phi * me.rw_T * (1.0 - S), -phi * me.rw};
            const double dMo[3] = {phi * me.ro_p * S, phi * me.ro_T * S, phi * me.ro};
            const double e0 = q.c_v_w * Mw + q.c_v_o * Mo;
            R[0] = q.w0 * (e0 - a.acc_old[c]) * a.Vdt;
            R[1] = (e0 * T + rock * T - a.acc_old[nt + c]) * a.Vdt;
            R[2] = q.w2 * (Mo - a.acc_old[2 * nt + c]) * a.Vdt;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double de0 = q.c_v_w * dMw[k] + q.c_v_o * dMo[k];
                Jd[0][k] = q.w0 * de0 * a.Vdt;
                Jd[1][k] = de0 * T * a.Vdt;
                Jd[2][k] = q.w2 * dMo[k] * a.Vdt;
            }
            Jd[1][1] += (e0 + rock) * a.Vdt;
            if (SCHUR) sd = (phi * q.c_v_o * S * me.ro + phi * q.c_v_w * (1.0 - S) * me.rw + rock) * a.Vdt;
        } else {
            const double T = me.T;
            const double Mo = phi * me.ro;
            const double dMo[2] = {phi * me.ro_p, phi * me.ro_T};
            R[0] = q.w0 * (Mo - a.acc_old[c]) * a.Vdt;
            R[1] = (q.c_v_o * Mo * T + rock * T - a.acc_old[nt + c]) * a.Vdt;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                Jd[0][k] = q.w0 * dMo[k] * a.Vdt;
                Jd[1][k] = q.c_v_o * dMo[k] * T * a.Vdt;
            }
            Jd[1][1] += (q.c_v_o * Mo + rock) * a.Vdt;
            if (SCHUR) sd = (phi * q.c_v_o * me.ro + rock) * a.Vdt;
        }
    }
    const long stride[3] = {1, (long)g.n0, g.np};
    const int sstride[3] = {1, ABX, ABX * ABY};
    const int idx[3] = {i0, i1, g.off2 + i2};
    const int ext[3] = {g.n0, g.n1, g.gn2};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            const int slot = 1 + 2 * ax + dir;
            const bool exists = dir ? (idx[ax] < ext[ax] - 1) : (idx[ax] > 0);
            double f[B], dP[B][B], dM[B][B], sP = 0.0, sM = 0.0;
            if (exists) {
                const long nb = dir ? c + stride[ax] : c - stride[ax];
                const Props<NPH> ot = PropsLds<NPH>::get(lds, dir ? sme + sstride[ax] : sme - sstride[ax]);
                if (dir) {
                    face_flux<NPH, SCHUR>(me, ot, a.TK[ax][c], a.gam[ax], a.Ga[ax], q, f, dP, dM, sP, sM);
#pragma unroll
                    for (int r = 0; r < B; ++r) {
                        R[r] += f[r];
#pragma unroll
                        for (int k = 0; k < B; ++k) Jd[r][k] += dP[r][k];
                    }
                    if (SCHUR) sd += sP;
                } else {
                    face_flux<NPH, SCHUR>(ot, me, a.TK[ax][nb], a.gam[ax], a.Ga[ax], q, f, dP, dM, sP, sM);
#pragma unroll
                    for (int r = 0; r < B; ++r) {
                        R[r] -= f[r];
#pragma unroll
                        for (int k = 0; k < B; ++k) Jd[r][k] -= dM[r][k];
                    }
                    if (SCHUR) sd -= sM;
                }
            }
            if (JAC) {
#pragma unroll
                for (int r = 0; r < B; ++r)
#pragma unroll
                    for (int k = 0; k < B; ++k) {
                        double v = 0.0;
                        if (exists) v = dir ? dM[r][k] : -dP[r][k];
                        a.J[((long)(slot * B + r) * B + k) * nt + c] = v;
                    }
            }
            if (SCHUR) a.Sm[(long)slot * nt + c] = exists ? (dir ? sM : -sP) : 0.0;
        }
    }
#pragma unroll
    for (int r = 0; r < B; ++r) a.R[(long)r * nt + c] = R[r];
    if (JAC) {
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int k = 0; k < B; ++k) a.J[((long)(r)*B + k) * nt + c] = Jd[r][k];
    }
    if (SCHUR) a.Sm[c] = sd;
}

// ---- accumulation of the old state (once per time step) -------------------------------------------
template <int NPH>
__global__ void k_accum_old(GridDev g, DevPrm q, const double *u, const double *phi_, double *acc) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid, nt = g.ntot;
    const double p = u[c], T = u[nt + c];
    const double phi = phi_[c];
    const double rock = (1.0 - phi) * q.rho_r * q.c_r;
    double ro, ro_p, ro_T;
    oil_rho(p, T, q, ro, ro_p, ro_T);
    if (NPH == 2) {
        const double S = u[2 * nt + c];
        double rw, rw_p, rw_T;
        water_rho(p, T, rw, rw_p, rw_T);
        const double Mw = phi * rw * (1.0 - S), Mo = phi * ro * S;
        const double e0 = q.c_v_w * Mw + q.c_v_o * Mo;
        acc[c] = e0;
        acc[nt + c] = e0 * T + rock * T;
        acc[2 * nt + c] = Mo;
    } else {
        const double Mo = phi * ro;
        acc[c] = Mo;
        acc[nt + c] = q.c_v_o * Mo * T + rock * T;
    }
}

// ---- face transmissibilities T^K_f = H(K)|e|/Delta_h (singlephase.py:98-103) -----------------------
__global__ void k_trans(GridDev g, const double *K, int ax, double geom, double *TK) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid;
    const int i2 = (int)(tid / g.np);
    const int rem = (int)(tid - (long)i2 * g.np);
    const int i1 = rem / g.n0, i0 = rem - i1 * g.n0;
    const long stride = ax == 0 ? 1 : (ax == 1 ? g.n0 : g.np);
    const int idx = ax == 0 ? i0 : (ax == 1 ? i1 : g.off2 + i2);
    const int ext = ax == 0 ? g.n0 : (ax == 1 ? g.n1 : g.gn2);
    double t = 0.0;
    if (idx < ext - 1) {
        const double kp = K[c], km = K[c + stride];
        const double s = kp + km;
        t = s > 0.0 ? 2.0 * kp * km / s * geom : 0.0;
    }
    TK[c] = t;
}

// lower halo plane of TK[2]: the face between the last plane of the slab below and my first plane
__global__ void k_trans_halo(GridDev g, const double *K, double geom, double *TK) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.np) return;
    double v = 0.0;
    if (g.off2 > 0) {
        const double kp = K[t], km = K[t + g.np];
        const double s = kp + km;
        v = s > 0.0 ? 2.0 * kp * km / s * geom : 0.0;
    }
    TK[t] = v;
}

// ---- wells / heaters: tiny kernel, one thread per group of entries sharing a cell ------------------
// Rate laws: wellcase.py:171-266; contributions to F: singlephase.py:151-165, twophase.py:212-235;
// to S~: preconditioners.py:102-108,269-276.  Derivatives by forward-mode duals (3 partials).
template <int NPH>
__device__ void source_eval(const tp_source &e, const DevPrm &q, Dual3 p, Dual3 T, Dual3 S, Dual3 *out,
                            double *rates, double &schur_diag) {
    constexpr int B = NPH + 1;
    for (int r = 0; r < B; ++r) out[r] = Dual3(0.0);
    rates[0] = rates[1] = rates[2] = 0.0;
    schur_diag = 0.0;
    if (e.kind == 2) {                                  // heater: R_E -= U (T_inj - T) wt
        out[1] = (Dual3(q.T_inj) - T) * (q.U * e.wt);
        schur_diag = -q.U * e.wt;
        return;
    }
    const Dual3 mo = oil_mu_t(T, q);
    const Dual3 ro = oil_rho_t(p, T, q);
    const Dual3 ddr = Dual3(e.bhp) - p;
    Dual3 dd;
    if (e.kind == 0) dd = (ddr.v >= 0.0) ? Dual3(0.0) : ddr;   // conditional(ge(bhp-p,0),0,bhp-p)
    else             dd = (ddr.v <= 0.0) ? Dual3(0.0) : ddr;   // conditional(le(bhp-p,0),0,bhp-p)
    auto cap = [&](Dual3 rate) {
        if (fabs(rate.v) - fabs(e.max_rate) >= 0.0) rate = Dual3(e.max_rate);
        if (e.constant_rate) rate = Dual3(e.max_rate);
        return rate;
    };
    if (NPH == 1) {
        const Dual3 rate = cap(dd * (e.WI) / mo);
        rates[0] = rate.v;
        if (e.kind == 0) {
            out[0] = ro * rate * (q.w0 * e.wt);
            out[1] = ro * rate * T * (q.c_v_o * e.wt);
            schur_diag = ro.v * rate.v * q.c_v_o * e.wt;
        } else {
            const Dual3 roi = oil_rho_t(p, Dual3(q.T_inj), q);
            out[0] = roi * rate * (q.w0 * e.wt);
            out[1] = roi * rate * (q.c_v_o * q.T_inj * e.wt);
        }
    } else {
        const Dual3 mw = water_mu_t(T);
        const Dual3 rw = water_rho_t(p, T);
        if (e.kind == 0) {
            const Dual3 lam_t = S / mo + (Dual3(1.0) - S) / mw;
            const Dual3 rate = cap(dd * lam_t * e.WI);
            const Dual3 qw = (Dual3(1.0) - S) / mw / lam_t * rate;
            const Dual3 qo = S / mo / lam_t * rate;
            rates[0] = rate.v; rates[1] = qw.v; rates[2] = qo.v;
            out[0] = (rw * qw * q.c_v_w + ro * qo * q.c_v_o) * (q.w0 * e.wt);
            out[2] = ro * qo * (q.w2 * e.wt);
            out[1] = (rw * qw * q.c_v_w + ro * qo * q.c_v_o) * T * e.wt;
            schur_diag = (rw.v * qw.v * q.c_v_w + ro.v * qo.v * q.c_v_o) * e.wt;
        } else {
            const Dual3 rate = cap(dd * e.WI / mw);
            rates[0] = rate.v;
            const Dual3 rwi = water_rho_t(p, Dual3(q.T_inj));
            out[0] = rwi * rate * (q.w0 * q.c_v_w * e.wt);
            out[1] = rwi * rate * (q.c_v_w * q.T_inj * e.wt);
        }
    }
}

template <int NPH>
__global__ void k_sources(GridDev g, DevPrm q, const tp_source *src, const int *start, int ngroups,
                          const double *u, double *R, double *J, double *Sm, double *rates, int nsrc) {
    constexpr int B = NPH + 1;
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= ngroups) return;
    const long nt = g.ntot;
    const long c = src[start[gi]].cell;
    Dual3 p(u[c], 0), T(u[nt + c], 1), S(NPH == 2 ? u[2 * nt + c] : 0.0, 2);
    for (int k = start[gi]; k < start[gi + 1]; ++k) {
        Dual3 out[B];
        double rt[3], sd;
        source_eval<NPH>(src[k], q, p, T, S, out, rt, sd);
        if (rates) { rates[k] = rt[0]; rates[nsrc + k] = rt[1]; rates[2 * nsrc + k] = rt[2]; }
        if (R)
            for (int r = 0; r < B; ++r) R[(long)r * nt + c] -= out[r].v;
        if (J)
            for (int r = 0; r < B; ++r)
                for (int k2 = 0; k2 < B; ++k2) J[((long)r * B + k2) * nt + c] -= out[r].d[k2];
        if (Sm) Sm[c] -= sd;
    }
}

// ------------------------------------------------------------------------------------------------
static inline dim3 grid_for(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

void compute_trans(tp_ctx *c) {
    const GridDev &g = c->g;
    for (int ax = 0; ax < 3; ++ax) {
        const double h = c->grid.h[ax];
        const double geom = c->vol / (h * h);
        hipLaunchKernelGGL(k_trans, grid_for(g.nown), dim3(256), 0, c->stream, g, c->K[ax].p, ax, geom, c->TK[ax].p);
    }
    const double h2 = c->grid.h[2];
    hipLaunchKernelGGL(k_trans_halo, grid_for(g.np), dim3(256), 0, c->stream, g, c->K[2].p, c->vol / (h2 * h2),
                       c->TK[2].p);
    TP_HIP(hipGetLastError());
}

void accum_old(tp_ctx *c) {
    const GridDev &g = c->g;
    if (c->nph == 2)
        hipLaunchKernelGGL(k_accum_old<2>, grid_for(g.nown), dim3(256), 0, c->stream, g, c->dprm, c->u_old.p,
                           c->phi.p, c->acc_old.p);
    else
        hipLaunchKernelGGL(k_accum_old<1>, grid_for(g.nown), dim3(256), 0, c->stream, g, c->dprm, c->u_old.p,
                           c->phi.p, c->acc_old.p);
    TP_HIP(hipGetLastError());
}

void assemble(tp_ctx *c, bool want_jac, bool want_schur) {
    TP_REQUIRE(c->fields_ready, "fields not finalised (tp_finalize_fields)");
    TP_REQUIRE(c->have_old && c->dt > 0.0, "old state / dt not set");
    const GridDev &g = c->g;
    AsmArgs a;
    a.g = g; a.q = c->dprm;
    a.u = c->u.p; a.phi = c->phi.p; a.kTs = c->kTs.p;
    for (int ax = 0; ax < 3; ++ax) {
        a.TK[ax] = c->TK[ax].p;
        const double h = c->grid.h[ax];
        a.gam[ax] = (ax == c->grid.gaxis) ? c->prm.g * h * 0.5 : 0.0;
        a.Ga[ax] = c->vol / (h * h);
    }
    a.acc_old = c->acc_old.p;
    a.Vdt = c->vol / c->dt;
    a.R = c->R.p; a.J = c->J.p; a.Sm = c->Sm.p;
    if (want_schur) TP_REQUIRE(c->Sm.p, "S~ storage not allocated");
    const dim3 gr = xcd_grid(g.nown), bl(256);
    static const bool lds_tiled = getenv("TP_ASM_LDS") && atoi(getenv("TP_ASM_LDS")) == 1;
    const dim3 grl((g.n0 + ATX - 1) / ATX, (g.n1 + ATY - 1) / ATY, (g.n2 + ATZ - 1) / ATZ);
#define LAUNCH(NPH, JAC, SCH)                                                                                            \
    do {                                                                                                                 \
        if (lds_tiled) {                                                                                                 \
            const size_t bytes = (size_t)PropsLds<NPH>::NV * ANS * sizeof(double);                                        \
            TP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_assemble_lds<NPH, JAC, SCH>),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));                         \
            hipLaunchKernelGGL((k_assemble_lds<NPH, JAC, SCH>), grl, bl, bytes, c->stream, a);                            \
        } else hipLaunchKernelGGL((k_assemble<NPH, JAC, SCH>), gr, bl, 0, c->stream, a);                                 \
    } while (0)
    if (c->nph == 2) {
        if (!want_jac) LAUNCH(2, false, false);
        else if (want_schur) LAUNCH(2, true, true);
        else LAUNCH(2, true, false);
    } else {
        if (!want_jac) LAUNCH(1, false, false);
        else if (want_schur) LAUNCH(1, true, true);
        else LAUNCH(1, true, false);
    }
#undef LAUNCH
    if (c->nsrc_groups > 0) {
        double *J = want_jac ? c->J.p : nullptr;
        double *Sm = (want_jac && want_schur) ? c->Sm.p : nullptr;
        const dim3 g2 = grid_for(c->nsrc_groups, 64);
        if (c->nph == 2)
            hipLaunchKernelGGL(k_sources<2>, g2, dim3(64), 0, c->stream, g, c->dprm, c->src.p, c->src_start.p,
                               c->nsrc_groups, c->u.p, c->R.p, J, Sm, (double *)nullptr, c->nsrc);
        else
            hipLaunchKernelGGL(k_sources<1>, g2, dim3(64), 0, c->stream, g, c->dprm, c->src.p, c->src_start.p,
                               c->nsrc_groups, c->u.p, c->R.p, J, Sm, (double *)nullptr, c->nsrc);
    }
    TP_HIP(hipGetLastError());
    if (want_jac) c->jac_ready = true;
}

void well_rates(tp_ctx *c) {
    if (c->nsrc_groups == 0) return;
    const dim3 g2 = grid_for(c->nsrc_groups, 64);
    if (c->nph == 2)
        hipLaunchKernelGGL(k_sources<2>, g2, dim3(64), 0, c->stream, c->g, c->dprm, c->src.p, c->src_start.p,
                           c->nsrc_groups, c->u.p, (double *)nullptr, (double *)nullptr, (double *)nullptr,
                           c->rates.p, c->nsrc);
    else
        hipLaunchKernelGGL(k_sources<1>, g2, dim3(64), 0, c->stream, c->g, c->dprm, c->src.p, c->src_start.p,
                           c->nsrc_groups, c->u.p, (double *)nullptr, (double *)nullptr, (double *)nullptr,
                           c->rates.p, c->nsrc);
    TP_HIP(hipGetLastError());
}

}  // namespace tp

// oracle/cport -- C++/OpenMP restatement of the hot path.  TEST INFRASTRUCTURE ONLY: the measured CPU baseline of
// bench.py (cpu_baseline.kind = "port") and a second checker for tests/.  The product (thermalporous_amd) never
// loads this library.  PARITY UNPINNED like the numpy oracle it mirrors (oracle/__init__.py): the reference
// (tlroy/thermalporous) is Python on Firedrake/PETSc/hypre, not runnable here, and holds no fixtures.
//
// It is the SAME algorithm as oracle/tpfa.py + oracle/linalg.py + oracle/engine.py (which cite the reference
// lines they follow), function for function, in the same arithmetic order wherever that order is defined:
//   closures            physicalparameters.py:37-98 of the reference                (oracle/closures.py)
//   residual/Jacobian   singlephase.py:60-273, twophase.py:67-411, wellcase.py:171-266  (oracle/tpfa.py)
//   S~ (ConvDiffSchur)  preconditioners.py:11-333
//   decoupling QI/TI    preconditioners.py:684-711,785-808,1445-1543                (oracle/linalg.py decouple)
//   stage 2             bjacobi + ILU(0), singlephase.py:348-349                    (TiledILU0)
//   stage 1             one AMG V-cycle per application, singlephase.py:303-307     (SemiAMG: the build's own AMG)
//   PCFIELDSPLIT FULL   twophase.py:536-545                                         (TwoStagePC.stage1)
//   FGMRES / Newton     twophase.py:416-433, thermalmodel.py:36-42,165              (fgmres, OracleEngine.newton_solve)
// Layout: cell arrays (n2, n1, n0), axis 0 fastest; vectors field-major; Jacobian planes [slot][row][col][cell] with
// slots 0 diag, 1 (-a0), 2 (+a0), 3 (-a1), 4 (+a1), 5 (-a2), 6 (+a2).
#include <omp.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

namespace {

typedef std::vector<double> vec;
using cplx = std::complex<double>;

// ---------------------------------------------------------------- closures (oracle/closures.py)
constexpr double OIL_C = 5.5e-5, OIL_P0 = 1.01325, OIL_E1 = 2.5e-4, OIL_T0 = 15.5556 + 273.15;
constexpr double A1 = -0.8021, A2 = 23.8765, A3 = 0.31458, A4 = -9.21592;
constexpr double E0 = 999.83952, E1 = 16.955176, E2 = -7.987e-3, E3 = -46.170461e-6, E4 = 105.56302e-9,
                 E5 = -280.54353e-12, E6 = 16.87985e-3, E7 = 10.2, CW = 3.98854e-4;
constexpr double AW = 2.1850, BW = 0.04012, CWM = 5.1547e-6;

inline double re(double x) { return x; }
inline double re(const cplx &x) { return x.real(); }

template <class T>
inline void oil_rho(T p, T Tk, double API, T &rho, T &rp, T &rT) {
    const double SG = 141.5 / (API + 131.5), rho_ref = SG * 999.0;
    const T pbar = p * 1e1;
    rho = rho_ref * std::exp(OIL_C * (pbar - OIL_P0)) * std::exp(-OIL_E1 * (Tk - OIL_T0));
    rp = (10.0 * OIL_C) * rho;
    rT = (-OIL_E1) * rho;
}
template <class T>
inline void oil_mu(T Tk, double API, T &mu, T &muT) {
    const T Tf = 1.8 * (Tk - 273.15) + 32.0;
    const double ex = A3 * API + A4;
    mu = 1e-3 * std::pow(10.0, A1 * API + A2) * std::pow(Tf, ex);
    muT = mu * ex * 1.8 / Tf;
}
template <class T>
inline void water_rho(T p, T Tk, T &rho, T &rp, T &rT) {
    const T Tc = Tk - 272.15;       // sic (physicalparameters.py:80)
    const T P = E0 + Tc * (E1 + Tc * (E2 + Tc * (E3 + Tc * (E4 + Tc * E5))));
    const T dP = E1 + Tc * (2.0 * E2 + Tc * (3.0 * E3 + Tc * (4.0 * E4 + Tc * 5.0 * E5)));
    const T den = 1.0 + E6 * Tc;
    const T ex = std::exp(CW * (p - E7));
    rho = P * ex / den;
    rp = CW * rho;
    rT = (dP - P * E6 / den) * ex / den;
}
template <class T>
inline void water_mu(T Tk, T &mu, T &muT) {
    const T Tf = 1.8 * (Tk - 272.15) + 32.0;
    const T den = -1.0 + BW * Tf + CWM * Tf * Tf;
    mu = 1e-3 * AW / den;
    muT = -mu * (BW + 2.0 * CWM * Tf) * 1.8 / den;
}

inline double harmonic(double ap, double am) {
    const double s = ap + am;
    return s > 0.0 ? 2.0 * ap * am / s : 0.0;
}

struct Prm {
    double ko, kw, kr, c_v_w, c_v_o, c_r, rho_r, p_inj, p_prod, T_inj, T_prod, API, p_ref, g, S_o, U, rate;
};

struct Opts {
    int32_t pc;           // 0 cpr, 1 cptr, 2 fieldsplit_cd, 4 bilu (bjacobi + ILU alone)
    int32_t decoup;       // 0 No, 1 QI, 2 TI, 3 QI_temp, 4 TI_temp
    double ksp_rtol, ksp_atol;
    int32_t ksp_max_it, ksp_restart;
    double snes_rtol, snes_atol, snes_stol;
    int32_t snes_max_it;
    double amg_omega;
    int32_t amg_nu, amg_min_cells, amg_full_levels, amg_coarse_pre, amg_coarse_post, amg_mid_skip, amg_tail_post,
        amg_single, schur_a11;
    int32_t tile[3];
    int32_t nslabs;
    double amg_dom_tau;
    int32_t ilu_levels;   // 0 or 1 (sub_1_sub_pc_factor_levels)
    int32_t fs_additive;  // pc_kind 2: additive (block-diagonal) fieldsplit instead of Schur FULL (pc_fieldsplit_diag)
};

struct Info {
    int32_t nits, lits, reason, complete;
    double fnorm0, fnorm, seconds;
};

// ---------------------------------------------------------------- 7-point scalar stencil on a box
struct Box {
    int n[3];
    long N;
    long st[3];
    void set(int n0, int n1, int n2) {
        n[0] = n0; n[1] = n1; n[2] = n2;
        N = (long)n0 * n1 * n2;
        st[0] = 1; st[1] = n0; st[2] = (long)n0 * n1;
    }
};

// A scalar 7-point operator stored inside a cell-interleaved array: entry (slot s, cell c) = base[c*cs + s*ss].
// CPU layout: everything a cell needs sits in consecutive cache lines (the GPU build uses planes instead).
struct SView {
    const double *base = nullptr;
    long cs = 0, ss = 0;
    inline double at(int s, long c) const { return base[c * cs + s * ss]; }
};

// y = A0 x + sum_a (A[2+2a] x(+a) + A[1+2a] x(-a))   (order of oracle.linalg.spmv_scalar)
static void spmv_scalar(const Box &g, const SView &A, const double *x, double *y) {
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
#pragma omp parallel for collapse(2) schedule(static)
    for (int i2 = 0; i2 < n2; ++i2)
        for (int i1 = 0; i1 < n1; ++i1) {
            const long base = ((long)i2 * n1 + i1) * n0;
            for (int i0 = 0; i0 < n0; ++i0) {
                const long c = base + i0;
                const double *a = A.base + c * A.cs;
                double s = a[0] * x[c];
                if (n0 > 1) {
                    if (i0 + 1 < n0) s += a[2 * A.ss] * x[c + 1];
                    if (i0 > 0) s += a[A.ss] * x[c - 1];
                }
                if (n1 > 1) {
                    if (i1 + 1 < n1) s += a[4 * A.ss] * x[c + g.st[1]];
                    if (i1 > 0) s += a[3 * A.ss] * x[c - g.st[1]];
                }
                if (n2 > 1) {
                    if (i2 + 1 < n2) s += a[6 * A.ss] * x[c + g.st[2]];
                    if (i2 > 0) s += a[5 * A.ss] * x[c - g.st[2]];
                }
                y[c] = s;
            }
        }
}

// ---------------------------------------------------------------- SemiAMG (oracle.linalg.SemiAMG)
struct AmgLevel {
    Box g;
    vec A8;            // [cell][8]: the 7 stencil slots + omega/diag -- one cache line per cell
    vec wm, wp;
    int axis = -1;
    inline double &A(int s, long c) { return A8[c * 8 + s]; }
    inline double A(int s, long c) const { return A8[c * 8 + s]; }
    inline double &invd(long c) { return A8[c * 8 + 7]; }
    inline double invd(long c) const { return A8[c * 8 + 7]; }
    SView view() const { SView v; v.base = A8.data(); v.cs = 8; v.ss = 1; return v; }
};

struct SemiAMG {
    std::vector<int> sched;
    std::vector<AmgLevel> lv;
    double omega = 0.8;
    int nu = 1, full_levels = 99, coarse_pre = 1, coarse_post = 1, tail_post = 1;
    bool mid_skip = false, single = false;
    vec coarseLU;
    int trunc_level = -1;      // first strongly diagonally dominant V(nu,nu) level (relaxation only), -1: none
    double dom_tau = 0.0;
    std::vector<int> piv;
    int ncoarse = 0;
    std::vector<vec> wx, wr, wb, we, wt;       // per-level work vectors

    inline double store(double x) const { return single ? (double)(float)x : x; }

    void init(const int n_[3], const double strength[3], const Opts &o) {
        omega = o.amg_omega; nu = o.amg_nu; full_levels = o.amg_full_levels;
        coarse_pre = o.amg_coarse_pre < 0 ? nu : o.amg_coarse_pre;
        coarse_post = o.amg_coarse_post < 0 ? nu : o.amg_coarse_post;
        tail_post = o.amg_tail_post < 0 ? coarse_post : o.amg_tail_post;
        mid_skip = o.amg_mid_skip != 0;
        single = o.amg_single != 0;
        dom_tau = o.amg_dom_tau;
        int n[3] = {n_[0], n_[1], n_[2]};
        double s[3];
        for (int a = 0; a < 3; ++a) s[a] = n[a] > 1 ? strength[a] : -1.0;
        sched.clear();
        while ((long)n[0] * n[1] * n[2] > o.amg_min_cells && sched.size() < 40) {
            int best = -1;
            for (int a = 0; a < 3; ++a)
                if (n[a] > 1 && (best < 0 || s[a] > s[best])) best = a;     // ties -> lowest axis (key (s, -q))
            if (best < 0) break;
            sched.push_back(best);
            n[best] = (n[best] + 1) / 2;
            for (int q = 0; q < 3; ++q) s[q] = (q == best) ? s[q] * 0.5 : s[q] * 2.0;
        }
        lv.assign(sched.size() + 1, AmgLevel());
        int m[3] = {n_[0], n_[1], n_[2]};
        for (size_t l = 0; l <= sched.size(); ++l) {
            AmgLevel &L = lv[l];
            L.g.set(m[0], m[1], m[2]);
            L.A8.assign((size_t)8 * L.g.N, 0.0);
            if (l < sched.size()) {
                L.axis = sched[l];
                L.wm.assign(L.g.N, 0.0);
                L.wp.assign(L.g.N, 0.0);
                m[L.axis] = (m[L.axis] + 1) / 2;
            }
        }
        wx.resize(lv.size()); wr.resize(lv.size()); wb.resize(lv.size()); we.resize(lv.size()); wt.resize(lv.size());
        for (size_t l = 0; l < lv.size(); ++l) {
            wx[l].assign(lv[l].g.N, 0.0); wr[l].assign(lv[l].g.N, 0.0); wb[l].assign(lv[l].g.N, 0.0);
            we[l].assign(lv[l].g.N, 0.0); wt[l].assign(lv[l].g.N, 0.0);
        }
    }

    void setup(const SView &A0) {
        {
            AmgLevel &L = lv[0];
#pragma omp parallel for schedule(static)
            for (long c = 0; c < L.g.N; ++c)
                for (int s = 0; s < 7; ++s) L.A(s, c) = store(A0.at(s, c));
        }
        for (size_t l = 0; l + 1 < lv.size(); ++l) {
            AmgLevel &L = lv[l], &C = lv[l + 1];
            const int a = L.axis;
            // weights (SemiAMG.weights): c = A0 + sum(cross slots)
#pragma omp parallel for schedule(static)
            for (long c = 0; c < L.g.N; ++c) {
                double cs = 0.0;
                for (int s = 1; s < 7; ++s)
                    if ((s - 1) / 2 != a) cs += L.A(s, c);
                const double cc = L.A(0, c) + cs;
                L.wm[c] = store(-L.A(1 + 2 * a, c) / cc);
                L.wp[c] = store(-L.A(2 + 2 * a, c) / cc);
            }
            // coarse operator (SemiAMG.coarsen)
            const int nfa = L.g.n[a];
            const long stf = L.g.st[a];
            const int c0 = C.g.n[0], c1 = C.g.n[1], c2 = C.g.n[2];
#pragma omp parallel for collapse(2) schedule(static)
            for (int I2 = 0; I2 < c2; ++I2)
                for (int I1 = 0; I1 < c1; ++I1)
                    for (int I0 = 0; I0 < c0; ++I0) {
                        int I[3] = {I0, I1, I2};
                        int F[3] = {I0, I1, I2};
                        F[a] = 2 * I[a];
                        const long f = F[0] + (long)L.g.n[0] * (F[1] + (long)L.g.n[1] * F[2]);
                        const long cc = I0 + (long)c0 * (I1 + (long)c1 * I2);
                        const bool hm = F[a] - 1 >= 0, hp = F[a] + 1 < nfa;
                        const long gm = f - stf, gp = f + stf;
                        const double Pm = hm ? L.wp[gm] : 0.0, Pp = hp ? L.wm[gp] : 0.0;
                        const double Wm = hm ? L.wm[gm] : 0.0, Wp = hp ? L.wp[gp] : 0.0;
                        double rho_f = 0.0, rho_m = 0.0, rho_p = 0.0;
                        for (int s = 0; s < 7; ++s) {
                            rho_f += L.A(s, f);
                            if (hm) rho_m += L.A(s, gm);
                            if (hp) rho_p += L.A(s, gp);
                        }
                        double out[7], offsum = 0.0;
                        for (int s = 1; s < 7; ++s) {
                            double v;
                            if (s == 1 + 2 * a) v = L.A(s, f) * Wm;
                            else if (s == 2 + 2 * a) v = L.A(s, f) * Wp;
                            else v = L.A(s, f) + Pm * (hm ? L.A(s, gm) : 0.0) + Pp * (hp ? L.A(s, gp) : 0.0);
                            out[s] = v;
                            offsum += v;
                        }
                        out[0] = -offsum + rho_f + Pm * rho_m + Pp * rho_p;
                        for (int s = 0; s < 7; ++s) C.A(s, cc) = store(out[s]);
                    }
        }
        for (auto &L : lv) {
#pragma omp parallel for schedule(static)
            for (long c = 0; c < L.g.N; ++c) L.invd(c) = store(omega / L.A(0, c));
        }
        trunc_level = -1;
        if (dom_tau > 0.0) {
            const int lim = std::min(full_levels, (int)lv.size() - 1);
            for (int l = 0; l < lim; ++l) {
                const AmgLevel &L = lv[l];
                double mx = 0.0;
#pragma omp parallel for reduction(max : mx) schedule(static)
                for (long c = 0; c < L.g.N; ++c) {
                    double so = 0.0;
                    for (int s = 1; s < 7; ++s) so += std::fabs(L.A(s, c));
                    mx = std::max(mx, so / std::fabs(L.A(0, c)));
                }
                if (mx <= dom_tau) { trunc_level = l; break; }
            }
        }
        // coarsest grid: dense LU with partial pivoting (the oracle uses SuperLU on the same <= min_cells matrix)
        AmgLevel &Lc = lv.back();
        const int n = (int)Lc.g.N;
        ncoarse = n;
        coarseLU.assign((size_t)n * n, 0.0);
        piv.assign(n, 0);
        for (int r = 0; r < n; ++r) {
            const int i0 = r % Lc.g.n[0], i1 = (r / Lc.g.n[0]) % Lc.g.n[1], i2 = r / (Lc.g.n[0] * Lc.g.n[1]);
            const bool has[7] = {true, i0 > 0, i0 < Lc.g.n[0] - 1, i1 > 0, i1 < Lc.g.n[1] - 1, i2 > 0, i2 < Lc.g.n[2] - 1};
            const long off[7] = {0, -1, 1, -Lc.g.st[1], Lc.g.st[1], -Lc.g.st[2], Lc.g.st[2]};
            for (int s = 0; s < 7; ++s)
                if (has[s]) coarseLU[(size_t)r * n + (r + off[s])] += Lc.A(s, r);
        }
        for (int k = 0; k < n; ++k) {
            int p = k;
            for (int r = k + 1; r < n; ++r)
                if (std::fabs(coarseLU[(size_t)r * n + k]) > std::fabs(coarseLU[(size_t)p * n + k])) p = r;
            piv[k] = p;
            if (p != k)
                for (int q = 0; q < n; ++q) std::swap(coarseLU[(size_t)k * n + q], coarseLU[(size_t)p * n + q]);
            const double d = coarseLU[(size_t)k * n + k];
            for (int r = k + 1; r < n; ++r) {
                const double f = coarseLU[(size_t)r * n + k] / d;
                coarseLU[(size_t)r * n + k] = f;
                if (f != 0.0)
                    for (int q = k + 1; q < n; ++q) coarseLU[(size_t)r * n + q] -= f * coarseLU[(size_t)k * n + q];
            }
        }
    }

    void coarse_solve(const double *b, double *x) const {
        const int n = ncoarse;
        for (int i = 0; i < n; ++i) x[i] = b[i];
        for (int k = 0; k < n; ++k)          // P b first (whole rows were swapped during the factorisation, LAPACK style)
            if (piv[k] != k) std::swap(x[k], x[piv[k]]);
        for (int k = 0; k < n; ++k)
            for (int r = k + 1; r < n; ++r) x[r] -= coarseLU[(size_t)r * n + k] * x[k];
        for (int k = n - 1; k >= 0; --k) {
            double s = x[k];
            for (int q = k + 1; q < n; ++q) s -= coarseLU[(size_t)k * n + q] * x[q];
            x[k] = s / coarseLU[(size_t)k * n + k];
        }
    }

    // rc = R r  (SemiAMG.restrict)
    void restrict_(int l, const double *r, double *rc) const {
        const AmgLevel &L = lv[l], &C = lv[l + 1];
        const int a = L.axis, nfa = L.g.n[a];
        const long stf = L.g.st[a];
        const int c0 = C.g.n[0], c1 = C.g.n[1], c2 = C.g.n[2];
#pragma omp parallel for collapse(2) schedule(static)
        for (int I2 = 0; I2 < c2; ++I2)
            for (int I1 = 0; I1 < c1; ++I1)
                for (int I0 = 0; I0 < c0; ++I0) {
                    int F[3] = {I0, I1, I2};
                    const int Ia = a == 0 ? I0 : (a == 1 ? I1 : I2);
                    F[a] = 2 * Ia;
                    const long f = F[0] + (long)L.g.n[0] * (F[1] + (long)L.g.n[1] * F[2]);
                    double v = r[f];
                    if (F[a] + 1 < nfa) v += L.wm[f + stf] * r[f + stf];
                    if (F[a] - 1 >= 0) v += L.wp[f - stf] * r[f - stf];
                    rc[I0 + (long)c0 * (I1 + (long)c1 * I2)] = v;
                }
    }

    // x += P ec  (SemiAMG.prolong, added to x)
    void prolong_add(int l, const double *ec, double *x) const {
        const AmgLevel &L = lv[l], &C = lv[l + 1];
        const int a = L.axis, nca = C.g.n[a];
        const long stc = C.g.st[a];
        const int n0 = L.g.n[0], n1 = L.g.n[1], n2 = L.g.n[2];
#pragma omp parallel for collapse(2) schedule(static)
        for (int i2 = 0; i2 < n2; ++i2)
            for (int i1 = 0; i1 < n1; ++i1)
                for (int i0 = 0; i0 < n0; ++i0) {
                    const int Fa = a == 0 ? i0 : (a == 1 ? i1 : i2);
                    const int Ia = Fa >> 1;
                    int I[3] = {i0, i1, i2};
                    I[a] = Ia;
                    const long ci = I[0] + (long)C.g.n[0] * (I[1] + (long)C.g.n[1] * I[2]);
                    const long c = i0 + (long)n0 * (i1 + (long)n1 * i2);
                    double e;
                    if (Fa & 1) {
                        const double right = (Ia + 1 < nca) ? ec[ci + stc] : 0.0;
                        e = L.wm[c] * ec[ci] + L.wp[c] * right;
                    } else {
                        e = ec[ci];
                    }
                    x[c] = x[c] + e;
                }
    }

    void smooth(int l, const double *b, const double *x, double *out, double *tmp) const {
        const AmgLevel &L = lv[l];
        spmv_scalar(L.g, L.view(), x, tmp);
#pragma omp parallel for schedule(static)
        for (long c = 0; c < L.g.N; ++c) out[c] = x[c] + L.invd(c) * (b[c] - tmp[c]);
    }

    // x = V-cycle(b) on level l (SemiAMG.vcycle).  b is read-only; x has level size.
    void vcycle(const double *b, double *x, int l = 0) {
        AmgLevel &L = lv[l];
        const long N = L.g.N;
        if (trunc_level >= 0 && l == trunc_level) {      // relaxation-only level (SemiAMG dom_tau): two damped-Jacobi sweeps
            double *t = wt[l].data(), *x2 = wx[l].data();
#pragma omp parallel for schedule(static)
            for (long c = 0; c < N; ++c) x[c] = L.invd(c) * b[c];
            smooth(l, b, x, x2, t);
#pragma omp parallel for schedule(static)
            for (long c = 0; c < N; ++c) x[c] = x2[c];
            return;
        }
        if (l == (int)lv.size() - 1) {
            if (N == 1) { x[0] = b[0] / L.A(0, 0); return; }
            coarse_solve(b, x);
            return;
        }
        int pre, post;
        if (l < full_levels) { pre = nu; post = nu; }
        else {
            pre = coarse_pre;
            post = N <= 1024 ? tail_post : coarse_post;
            if (mid_skip && N > 1024 && ((l - full_levels) % 2 == 1)) { pre = 0; post = 0; }
        }
        if (l == 0) {      // experiment hook (cycle-shape studies only; unset in every test and in bench.py)
            static const char *e0 = getenv("CP_L0_PRE"), *e1 = getenv("CP_L0_POST");
            if (e0) pre = atoi(e0);
            if (e1) post = atoi(e1);
        }
        double *r = wr[l].data(), *t = wt[l].data(), *x2 = wx[l].data();
        const double *rr;
        if (pre == 0) {
#pragma omp parallel for schedule(static)
            for (long c = 0; c < N; ++c) x[c] = 0.0;
            rr = b;
        } else {
#pragma omp parallel for schedule(static)
            for (long c = 0; c < N; ++c) x[c] = L.invd(c) * b[c];
            for (int k = 0; k < pre - 1; ++k) {
                smooth(l, b, x, x2, t);
#pragma omp parallel for schedule(static)
                for (long c = 0; c < N; ++c) x[c] = x2[c];
            }
            spmv_scalar(L.g, L.view(), x, t);
#pragma omp parallel for schedule(static)
            for (long c = 0; c < N; ++c) r[c] = b[c] - t[c];
            rr = r;
        }
        double *bc = wb[l + 1].data(), *ec = we[l + 1].data();
        restrict_(l, rr, bc);
        vcycle(bc, ec, l + 1);
        prolong_add(l, ec, x);
        for (int k = 0; k < post; ++k) {
            smooth(l, b, x, x2, t);
#pragma omp parallel for schedule(static)
            for (long c = 0; c < N; ++c) x[c] = x2[c];
        }
    }
};

// ---------------------------------------------------------------- small dense blocks
template <int B>
inline void inv_block(const double *A, double *I) {      // row-major BxB, Gauss-Jordan with partial pivoting
    double M[B][2 * B];
    for (int r = 0; r < B; ++r)
        for (int q = 0; q < B; ++q) { M[r][q] = A[r * B + q]; M[r][B + q] = (r == q) ? 1.0 : 0.0; }
    for (int k = 0; k < B; ++k) {
        int p = k;
        for (int r = k + 1; r < B; ++r)
            if (std::fabs(M[r][k]) > std::fabs(M[p][k])) p = r;
        if (p != k)
            for (int q = 0; q < 2 * B; ++q) std::swap(M[k][q], M[p][q]);
        const double d = 1.0 / M[k][k];
        for (int q = 0; q < 2 * B; ++q) M[k][q] *= d;
        for (int r = 0; r < B; ++r) {
            if (r == k) continue;
            const double f = M[r][k];
            for (int q = 0; q < 2 * B; ++q) M[r][q] -= f * M[k][q];
        }
    }
    for (int r = 0; r < B; ++r)
        for (int q = 0; q < B; ++q) I[r * B + q] = M[r][B + q];
}

// ---------------------------------------------------------------- the engine
struct Ctx {
    int nph, b, gaxis;
    Box g;
    double h[3], V;
    Prm prm;
    Opts o;
    vec phi, K[3], kTs, TK[3];
    double G[3];
    double w0, w2;
    // sources
    int nsrc = 0;
    std::vector<int64_t> scell;
    std::vector<int32_t> skind, sconst;
    vec swt, sbhp, sqmax, sWI;
    // state
    vec u, u_old, old_acc;
    double dt = 0.0;
    // props cache
    vec pr_ro[3], pr_rw[3], pr_Lw[4], pr_Lo[4], pr_kT, pr_kTS;
    // system
    vec R, J, Sm, At, dcoef[2][2];
    bool have_d = false;
    // ILU
    int tile[3];
    std::vector<int> l0, l1, l2, td0, td1, td2;       // in-tile coordinates and tile extents per cell
    std::vector<std::vector<long>> tiles;             // cells of each tile in natural order
    vec F1;                                           // ILU(1): 13 blocks per cell (6 lower, D~^-1, 6 upper)
    vec Dinv, Bf, Cb;                                 // D~^-1 ; B_cm = A_cm D~_m^-1 (3/cell) ; C_cm = D~_c^-1 A_up (3/cell)
    // AMG
    SemiAMG amg_p, amg_T;
    vec sp_S7, sp_invd, sp_d00inv, sp_u, sp_v;         // selfp (oracle.linalg.SelfpSchur): collapse of Sp, w/diag(Sp), 1/diag(A00)
    bool amg_ready = false;
    // work
    vec w_r0, w_r1, w_t, w_y0, w_y1, w_res, w_il, w_yt;
    std::vector<vec> Vb, Zb;
    std::string err;

    // Jacobian, cell-interleaved (what a CPU CSR/BSR code would use): J[(cell*7 + slot)*b*b + row*b + col]
    inline double &Jat(int s, int r, int q, long c) { return J[((size_t)c * 7 + s) * (b * b) + r * b + q]; }
    inline double Jat(int s, int r, int q, long c) const { return J[((size_t)c * 7 + s) * (b * b) + r * b + q]; }
    SView Jview(int r, int q) const { SView v; v.base = J.data() + r * b + q; v.cs = 7L * b * b; v.ss = (long)b * b; return v; }
};

static void compute_props(Ctx &C, const double *u) {
    const long N = C.g.N;
    const double *p = u, *T = u + N, *S = C.nph == 2 ? u + 2 * N : nullptr;
    const Prm &P = C.prm;
#pragma omp parallel for schedule(static)
    for (long c = 0; c < N; ++c) {
        double ro, rop, roT, mo, moT;
        oil_rho(p[c], T[c], P.API, ro, rop, roT);
        oil_mu(T[c], P.API, mo, moT);
        C.pr_ro[0][c] = ro; C.pr_ro[1][c] = rop; C.pr_ro[2][c] = roT;
        if (C.nph == 2) {
            double rw, rwp, rwT, mw, mwT;
            water_rho(p[c], T[c], rw, rwp, rwT);
            water_mu(T[c], mw, mwT);
            C.pr_rw[0][c] = rw; C.pr_rw[1][c] = rwp; C.pr_rw[2][c] = rwT;
            const double s = S[c], kw_ = 1.0 - s;
            C.pr_Lw[0][c] = kw_ * rw / mw; C.pr_Lw[1][c] = kw_ * rwp / mw;
            C.pr_Lw[2][c] = kw_ * (rwT / mw - rw * mwT / (mw * mw)); C.pr_Lw[3][c] = -rw / mw;
            C.pr_Lo[0][c] = s * ro / mo; C.pr_Lo[1][c] = s * rop / mo;
            C.pr_Lo[2][c] = s * (roT / mo - ro * moT / (mo * mo)); C.pr_Lo[3][c] = ro / mo;
            const double phi = C.phi[c];
            C.pr_kT[c] = phi * (s * P.ko + (1 - s) * P.kw) + (1 - phi) * P.kr;
            C.pr_kTS[c] = phi * (P.ko - P.kw);
        } else {
            C.pr_Lo[0][c] = ro / mo; C.pr_Lo[1][c] = rop / mo;
            C.pr_Lo[2][c] = roT / mo - ro * moT / (mo * mo); C.pr_Lo[3][c] = 0.0;
            C.pr_kT[c] = C.kTs[c];
            C.pr_kTS[c] = 0.0;
        }
    }
}

static void accum(const Ctx &C, const double *u, double *out) {      // Problem.accum
    const long N = C.g.N;
    const double *T = u + N, *S = C.nph == 2 ? u + 2 * N : nullptr;
    const Prm &P = C.prm;
#pragma omp parallel for schedule(static)
    for (long c = 0; c < N; ++c) {
        const double phi = C.phi[c], rock = (1 - phi) * P.rho_r * P.c_r;
        if (C.nph == 2) {
            const double Mw = phi * C.pr_rw[0][c] * (1.0 - S[c]), Mo = phi * C.pr_ro[0][c] * S[c];
            out[c] = P.c_v_w * Mw + P.c_v_o * Mo;
            out[N + c] = (P.c_v_w * Mw + P.c_v_o * Mo) * T[c] + rock * T[c];
            out[2 * N + c] = Mo;
        } else {
            const double Mo = phi * C.pr_ro[0][c];
            out[c] = Mo;
            out[N + c] = P.c_v_o * Mo * T[c] + rock * T[c];
        }
    }
}

// flux of every equation through the face between lo cell cP and hi cell cM along axis a (Problem._face_flux)
static inline void face_flux(const Ctx &C, int a, long cP, long cM, const double *p, const double *T, double f[3]) {
    const double TK = C.TK[a][cP];
    const Prm &P = C.prm;
    f[0] = f[1] = f[2] = 0.0;
    const double gam = (a == C.gaxis) ? P.g * C.h[a] * 0.5 : 0.0;
    const int nphs = C.nph == 2 ? 2 : 1;
    for (int ph = 0; ph < nphs; ++ph) {
        const bool water = (C.nph == 2 && ph == 0);
        const vec *L = water ? C.pr_Lw : C.pr_Lo;
        const vec *rho = water ? C.pr_rw : C.pr_ro;
        const double ce = water ? P.c_v_w : P.c_v_o;
        const double c0 = C.nph == 2 ? ce : 1.0;
        double Phi = p[cP] - p[cM];
        if (a == C.gaxis) Phi = Phi - gam * (rho[0][cP] + rho[0][cM]);
        const bool up = Phi > 0.0;
        const double Lu = up ? L[0][cP] : L[0][cM], Tu = up ? T[cP] : T[cM];
        const double F = TK * Lu * Phi;
        f[0] = f[0] + C.w0 * c0 * F;
        f[1] = f[1] + ce * Tu * F;
        if (C.nph == 2 && ph == 1) f[2] = f[2] + C.w2 * F;
    }
    f[1] = f[1] + harmonic(C.pr_kT[cP], C.pr_kT[cM]) * C.G[a] * (T[cP] - T[cM]);
}

// per-entry source vector and rates (Problem.source_terms); T = double or complex (complex-step Jacobian)
template <class T>
static void source_entry(const Ctx &C, int e, T p, T Tk, T S, T out[3], T rates[3]) {
    const Prm &P = C.prm;
    const int kind = C.skind[e];
    const double wt = C.swt[e], bhp = C.sbhp[e], qmax = C.sqmax[e], WI = C.sWI[e];
    const bool cst = C.sconst[e] != 0;
    T mo, moT, ro, rop, roT;
    oil_mu(Tk, P.API, mo, moT);
    oil_rho(p, Tk, P.API, ro, rop, roT);
    const T dd_raw = bhp - p;
    T dd;
    if (kind == 0) dd = (re(dd_raw) >= 0.0) ? T(0.0) : dd_raw;
    else dd = (re(dd_raw) <= 0.0) ? T(0.0) : dd_raw;
    out[0] = out[1] = out[2] = T(0.0);
    rates[0] = rates[1] = rates[2] = T(0.0);
    const T Tinj = P.T_inj + 0.0 * Tk;
    if (C.nph == 1) {
        T rate = WI / mo * dd;
        if (std::fabs(re(rate)) - std::fabs(qmax) >= 0.0) rate = qmax;
        if (cst) rate = qmax;
        T roi, d1, d2;
        oil_rho(p, Tinj, P.API, roi, d1, d2);
        const double cv = P.c_v_o;
        const T m = kind == 0 ? ro * rate : (kind == 1 ? roi * rate : T(0.0));
        out[0] = C.w0 * m * wt;
        out[1] = (kind == 0 ? ro * rate * cv * Tk : (kind == 1 ? roi * rate * cv * P.T_inj : P.U * (P.T_inj - Tk))) * wt;
        rates[0] = kind == 2 ? T(0.0) : rate;
    } else {
        T mw, mwT, rw, rwp, rwT;
        water_mu(Tk, mw, mwT);
        water_rho(p, Tk, rw, rwp, rwT);
        const T lam_t = S / mo + (1.0 - S) / mw;
        T rate_p = WI * lam_t * dd;
        if (std::fabs(re(rate_p)) - std::fabs(qmax) >= 0.0) rate_p = qmax;
        if (cst) rate_p = qmax;
        const T qw = (1.0 - S) / mw / lam_t * rate_p, qo = S / mo / lam_t * rate_p;
        T rate_i = WI / mw * dd;
        if (std::fabs(re(rate_i)) - std::fabs(qmax) >= 0.0) rate_i = qmax;
        if (cst) rate_i = qmax;
        T rwi, d1, d2;
        water_rho(p, Tinj, rwi, d1, d2);
        const double cw = P.c_v_w, co = P.c_v_o;
        out[0] = C.w0 * (kind == 0 ? cw * rw * qw + co * ro * qo : (kind == 1 ? cw * rwi * rate_i : T(0.0))) * wt;
        out[2] = C.w2 * (kind == 0 ? ro * qo : T(0.0)) * wt;
        out[1] = (kind == 0 ? (rw * qw * cw + ro * qo * co) * Tk
                            : (kind == 1 ? rwi * rate_i * cw * P.T_inj : P.U * (P.T_inj - Tk))) * wt;
        rates[0] = kind == 0 ? rate_p : (kind == 1 ? rate_i : T(0.0));
        rates[1] = kind == 0 ? qw : T(0.0);
        rates[2] = kind == 0 ? qo : T(0.0);
    }
}

static void residual(Ctx &C, const double *u, double *R) {      // Problem.residual
    const long N = C.g.N;
    const int b = C.b;
    compute_props(C, u);
    accum(C, u, R);
    const double w[3] = {C.w0, 1.0, C.w2};
    const double Vdt = C.V / C.dt;
    const double *p = u, *T = u + N;
    const int n0 = C.g.n[0], n1 = C.g.n[1], n2 = C.g.n[2];
#pragma omp parallel for collapse(2) schedule(static)
    for (int i2 = 0; i2 < n2; ++i2)
        for (int i1 = 0; i1 < n1; ++i1)
            for (int i0 = 0; i0 < n0; ++i0) {
                const long c = i0 + (long)n0 * (i1 + (long)n1 * i2);
                double r[3];
                for (int q = 0; q < b; ++q) r[q] = (R[q * N + c] - C.old_acc[q * N + c]) * Vdt * w[q];
                const int I[3] = {i0, i1, i2};
                for (int a = 0; a < 3; ++a) {
                    if (C.g.n[a] == 1) continue;
                    double f[3];
                    if (I[a] + 1 < C.g.n[a]) {
                        face_flux(C, a, c, c + C.g.st[a], p, T, f);
                        for (int q = 0; q < b; ++q) r[q] += f[q];
                    }
                    if (I[a] > 0) {
                        face_flux(C, a, c - C.g.st[a], c, p, T, f);
                        for (int q = 0; q < b; ++q) r[q] -= f[q];
                    }
                }
                for (int q = 0; q < b; ++q) R[q * N + c] = r[q];
            }
    for (int e = 0; e < C.nsrc; ++e) {
        const long c = C.scell[e];
        double out[3], rates[3];
        source_entry<double>(C, e, u[c], u[N + c], C.nph == 2 ? u[2 * N + c] : 0.0, out, rates);
        for (int q = 0; q < b; ++q) R[q * N + c] -= out[q];
    }
}

// flux derivatives of the face (lo cP, hi cM) along a: dP[r][c] = d f_r / d u_c(cP), dM likewise for cM; S~ entries
static inline void face_derivs(const Ctx &C, int a, long cP, long cM, const double *p, const double *T, double dP[3][3],
                               double dM[3][3], double &sP, double &sM) {
    const Prm &P = C.prm;
    const int b = C.b;
    const double TK = C.TK[a][cP];
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) dP[r][q] = dM[r][q] = 0.0;
    sP = sM = 0.0;
    const double gam = (a == C.gaxis) ? P.g * C.h[a] * 0.5 : 0.0;
    const int nphs = C.nph == 2 ? 2 : 1;
    for (int ph = 0; ph < nphs; ++ph) {
        const bool water = (C.nph == 2 && ph == 0);
        const vec *L = water ? C.pr_Lw : C.pr_Lo;
        const vec *rho = water ? C.pr_rw : C.pr_ro;
        const double ce = water ? P.c_v_w : P.c_v_o;
        const double c0 = C.nph == 2 ? ce : 1.0;
        const bool to2 = (C.nph == 2 && ph == 1);
        double Phi = p[cP] - p[cM];
        if (a == C.gaxis) Phi = Phi - gam * (rho[0][cP] + rho[0][cM]);
        const bool up = Phi > 0.0;
        const double Lu = up ? L[0][cP] : L[0][cM], Tu = up ? T[cP] : T[cM];
        const double F = TK * Lu * Phi;
        const double dPhiP[3] = {1.0 - gam * rho[1][cP], -gam * rho[2][cP], 0.0};
        const double dPhiM[3] = {-1.0 - gam * rho[1][cM], -gam * rho[2][cM], 0.0};
        for (int c = 0; c < b; ++c) {
            const double dLP = up ? L[c + 1][cP] : 0.0, dLM = up ? 0.0 : L[c + 1][cM];
            const double dFP = TK * (Lu * dPhiP[c] + dLP * Phi), dFM = TK * (Lu * dPhiM[c] + dLM * Phi);
            dP[0][c] += C.w0 * c0 * dFP;
            dM[0][c] += C.w0 * c0 * dFM;
            dP[1][c] += ce * Tu * dFP;
            dM[1][c] += ce * Tu * dFM;
            if (to2) { dP[2][c] += C.w2 * dFP; dM[2][c] += C.w2 * dFM; }
        }
        dP[1][1] += up ? ce * F : 0.0;
        dM[1][1] += up ? 0.0 : ce * F;
        sP += up ? ce * F : 0.0;
        sM += up ? 0.0 : ce * F;
    }
    const double kP = C.pr_kT[cP], kM = C.pr_kT[cM];
    const double Hk = harmonic(kP, kM), Gk = C.G[a];
    dP[1][1] += Hk * Gk;
    dM[1][1] -= Hk * Gk;
    sP += Hk * Gk;
    sM -= Hk * Gk;
    if (C.nph == 2) {
        const double s2 = kP + kM;
        const double dHP = s2 > 0 ? 2 * kM * kM / (s2 * s2) : 0.0, dHM = s2 > 0 ? 2 * kP * kP / (s2 * s2) : 0.0;
        const double dT = T[cP] - T[cM];
        dP[1][2] += Gk * dT * dHP * C.pr_kTS[cP];
        dM[1][2] += Gk * dT * dHM * C.pr_kTS[cM];
    }
}

static void jacobian(Ctx &C, const double *u, bool want_schur) {      // Problem.jacobian
    const long N = C.g.N;
    const int b = C.b;
    const Prm &P = C.prm;
    compute_props(C, u);
    const double w[3] = {C.w0, 1.0, C.w2};
    const double Vdt = C.V / C.dt;
    const double *p = u, *T = u + N, *S = C.nph == 2 ? u + 2 * N : nullptr;
    const int n0 = C.g.n[0], n1 = C.g.n[1], n2 = C.g.n[2];
    if (want_schur && C.Sm.size() != (size_t)7 * N) C.Sm.assign((size_t)7 * N, 0.0);
#pragma omp parallel for collapse(2) schedule(static)
    for (int i2 = 0; i2 < n2; ++i2)
        for (int i1 = 0; i1 < n1; ++i1)
            for (int i0 = 0; i0 < n0; ++i0) {
                const long c = i0 + (long)n0 * (i1 + (long)n1 * i2);
                double Jb[7][3][3];
                double Sb[7];
                for (int s = 0; s < 7; ++s) {
                    Sb[s] = 0.0;
                    for (int r = 0; r < 3; ++r)
                        for (int q = 0; q < 3; ++q) Jb[s][r][q] = 0.0;
                }
                const double phi = C.phi[c], rock = (1 - phi) * P.rho_r * P.c_r;
                const double ro = C.pr_ro[0][c], rop = C.pr_ro[1][c], roT = C.pr_ro[2][c];
                if (C.nph == 2) {
                    const double rw = C.pr_rw[0][c], rwp = C.pr_rw[1][c], rwT = C.pr_rw[2][c], s = S[c];
                    const double cw = P.c_v_w, co = P.c_v_o;
                    const double Mw[4] = {phi * rw * (1 - s), phi * rwp * (1 - s), phi * rwT * (1 - s), -phi * rw};
                    const double Mo[4] = {phi * ro * s, phi * rop * s, phi * roT * s, phi * ro};
                    for (int q = 0; q < 3; ++q) {
                        const double e0 = cw * Mw[q + 1] + co * Mo[q + 1];
                        Jb[0][0][q] += w[0] * e0 * Vdt;
                        Jb[0][1][q] += e0 * T[c] * Vdt;
                        Jb[0][2][q] += w[2] * Mo[q + 1] * Vdt;
                    }
                    Jb[0][1][1] += (cw * Mw[0] + co * Mo[0] + rock) * Vdt;
                    Sb[0] += (phi * co * s * ro + phi * cw * (1 - s) * rw + rock) * Vdt;
                } else {
                    const double cv = P.c_v_o;
                    const double Mo[3] = {phi * ro, phi * rop, phi * roT};
                    for (int q = 0; q < 2; ++q) {
                        Jb[0][0][q] += w[0] * Mo[q + 1] * Vdt;
                        Jb[0][1][q] += cv * Mo[q + 1] * T[c] * Vdt;
                    }
                    Jb[0][1][1] += (cv * Mo[0] + rock) * Vdt;
                    Sb[0] += (phi * cv * ro + rock) * Vdt;
                }
                const int I[3] = {i0, i1, i2};
                for (int a = 0; a < 3; ++a) {
                    if (C.g.n[a] == 1) continue;
                    double dP[3][3], dM[3][3], sP, sM;
                    if (I[a] + 1 < C.g.n[a]) {          // c is the lo ('+') cell of its upper face
                        face_derivs(C, a, c, c + C.g.st[a], p, T, dP, dM, sP, sM);
                        for (int r = 0; r < b; ++r)
                            for (int q = 0; q < b; ++q) { Jb[0][r][q] += dP[r][q]; Jb[2 + 2 * a][r][q] += dM[r][q]; }
                        Sb[0] += sP; Sb[2 + 2 * a] += sM;
                    }
                    if (I[a] > 0) {                      // c is the hi ('-') cell of its lower face
                        face_derivs(C, a, c - C.g.st[a], c, p, T, dP, dM, sP, sM);
                        for (int r = 0; r < b; ++r)
                            for (int q = 0; q < b; ++q) { Jb[0][r][q] -= dM[r][q]; Jb[1 + 2 * a][r][q] -= dP[r][q]; }
                        Sb[0] -= sM; Sb[1 + 2 * a] -= sP;
                    }
                }
                for (int s = 0; s < 7; ++s) {
                    for (int r = 0; r < b; ++r)
                        for (int q = 0; q < b; ++q) C.Jat(s, r, q, c) = Jb[s][r][q];
                    if (want_schur) C.Sm[(size_t)c * 7 + s] = Sb[s];
                }
            }
    // sources: complex-step derivative of source_terms (Problem.source_jac), S~ diagonal (schur_source_diag)
    const double hstep = 1e-30;
    for (int e = 0; e < C.nsrc; ++e) {
        const long c = C.scell[e];
        const cplx base[3] = {cplx(u[c]), cplx(u[N + c]), cplx(C.nph == 2 ? u[2 * N + c] : 0.0)};
        for (int q = 0; q < b; ++q) {
            cplx arg[3] = {base[0], base[1], base[2]};
            arg[q] += cplx(0.0, hstep);
            cplx out[3], rates[3];
            source_entry<cplx>(C, e, arg[0], arg[1], arg[2], out, rates);
            for (int r = 0; r < b; ++r) C.Jat(0, r, q, c) -= out[r].imag() / hstep;
        }
        if (want_schur) {
            double out[3], rates[3];
            source_entry<double>(C, e, u[c], u[N + c], C.nph == 2 ? u[2 * N + c] : 0.0, out, rates);
            double ro, d1, d2;
            oil_rho(u[c], u[N + c], P.API, ro, d1, d2);
            double prod;
            if (C.nph == 2) {
                double rw;
                water_rho(u[c], u[N + c], rw, d1, d2);
                prod = rw * rates[1] * P.c_v_w + ro * rates[2] * P.c_v_o;
            } else {
                prod = ro * rates[0] * P.c_v_o;
            }
            const double sd = (C.skind[e] == 0 ? prod : (C.skind[e] == 2 ? -P.U : 0.0)) * C.swt[e];
            C.Sm[(size_t)c * 7] -= sd;
        }
    }
}

// y = J x  (oracle.linalg.spmv_block: sum over (r,c) of scalar stencil products)
static void spmv_block(const Ctx &C, const double *x, double *y, int ncols = -1) {
    const long N = C.g.N;
    const int b = C.b;
    if (ncols < 0) ncols = b;
    const int n0 = C.g.n[0], n1 = C.g.n[1], n2 = C.g.n[2];
#pragma omp parallel for collapse(2) schedule(static)
    for (int i2 = 0; i2 < n2; ++i2)
        for (int i1 = 0; i1 < n1; ++i1) {
            const long base = ((long)i2 * n1 + i1) * n0;
            for (int i0 = 0; i0 < n0; ++i0) {
                const long c = base + i0;
                const bool has[7] = {true, i0 > 0, i0 + 1 < n0, i1 > 0, i1 + 1 < n1, i2 > 0, i2 + 1 < n2};
                const long off[7] = {0, -1, 1, -C.g.st[1], C.g.st[1], -C.g.st[2], C.g.st[2]};
                const int order[7] = {0, 2, 1, 4, 3, 6, 5};
                for (int r = 0; r < b; ++r) {
                    double acc = 0.0;
                    for (int q = 0; q < ncols; ++q) {
                        double s = 0.0;
                        for (int k = 0; k < 7; ++k) {
                            const int sl = order[k];
                            if (!has[sl]) continue;
                            s += C.Jat(sl, r, q, c) * x[q * N + c + off[sl]];
                        }
                        acc += s;
                    }
                    y[r * N + c] = acc;
                }
            }
        }
}

// ---------------------------------------------------------------- tiled block ILU(0)  (oracle.linalg.TiledILU0)
static void ilu_layout(Ctx &C) {
    const int n0 = C.g.n[0], n1 = C.g.n[1], n2 = C.g.n[2];
    const int t0 = std::max(1, std::min(C.o.tile[0], n0)), t1 = std::max(1, std::min(C.o.tile[1], n1)),
              t2 = std::max(1, std::min(C.o.tile[2], n2));
    C.tile[0] = t0; C.tile[1] = t1; C.tile[2] = t2;
    const int nslabs = std::max(1, C.o.nslabs);
    std::vector<int> lo2(n2), hi2(n2);
    {
        const int base = n2 / nslabs, rem = n2 % nslabs;
        for (int r = 0; r < nslabs; ++r) {
            const int lo = r * base + std::min(r, rem), hi = lo + base + (r < rem ? 1 : 0);
            for (int k = lo; k < hi; ++k) { lo2[k] = lo; hi2[k] = hi; }
        }
    }
    const long N = C.g.N;
    C.l0.resize(N); C.l1.resize(N); C.l2.resize(N); C.td0.resize(N); C.td1.resize(N); C.td2.resize(N);
    // tiles in lexicographic order (slab, T2, T1, T0); cells of a tile in natural order
    C.tiles.clear();
    for (int r = 0; r < nslabs; ++r) {
        const int base = n2 / nslabs, rem = n2 % nslabs;
        const int lo = r * base + std::min(r, rem), hi = lo + base + (r < rem ? 1 : 0);
        for (int b2 = lo; b2 < hi; b2 += t2)
            for (int b1 = 0; b1 < n1; b1 += t1)
                for (int b0 = 0; b0 < n0; b0 += t0) {
                    std::vector<long> cells;
                    const int e2 = std::min(b2 + t2, hi), e1 = std::min(b1 + t1, n1), e0 = std::min(b0 + t0, n0);
                    for (int i2 = b2; i2 < e2; ++i2)
                        for (int i1 = b1; i1 < e1; ++i1)
                            for (int i0 = b0; i0 < e0; ++i0) {
                                const long c = i0 + (long)n0 * (i1 + (long)n1 * i2);
                                cells.push_back(c);
                                C.l0[c] = i0 - b0; C.l1[c] = i1 - b1; C.l2[c] = i2 - b2;
                                C.td0[c] = e0 - b0; C.td1[c] = e1 - b1; C.td2[c] = e2 - b2;
                            }
                    C.tiles.push_back(std::move(cells));
                }
    }
    const int bb = C.b * C.b;
    C.Dinv.assign((size_t)bb * N, 0.0);
    C.Bf.assign((size_t)3 * bb * N, 0.0);
    C.Cb.assign((size_t)3 * bb * N, 0.0);
}

template <int B>
static void ilu_factor_t(Ctx &C) {
    const int BB = B * B;
    const long nt = (long)C.tiles.size();
#pragma omp parallel for schedule(dynamic, 1)
    for (long t = 0; t < nt; ++t) {
        for (long c : C.tiles[t]) {
            double D[BB];
            for (int r = 0; r < B; ++r)
                for (int q = 0; q < B; ++q) D[r * B + q] = C.Jat(0, r, q, c);
            const bool has_lo[3] = {C.l0[c] > 0, C.l1[c] > 0, C.l2[c] > 0};
            for (int a = 0; a < 3; ++a) {
                double *Bm = &C.Bf[((size_t)c * 3 + a) * BB];
                if (!has_lo[a]) { for (int e = 0; e < BB; ++e) Bm[e] = 0.0; continue; }
                const long m = c - C.g.st[a];
                const double *Dm = &C.Dinv[(size_t)m * BB];
                // B_cm = A_cm D~_m^-1 ; D -= B_cm A_mc
                for (int r = 0; r < B; ++r)
                    for (int q = 0; q < B; ++q) {
                        double v = 0.0;
                        for (int k = 0; k < B; ++k) v += C.Jat(1 + 2 * a, r, k, c) * Dm[k * B + q];
                        Bm[r * B + q] = v;
                    }
                for (int r = 0; r < B; ++r)
                    for (int q = 0; q < B; ++q) {
                        double v = 0.0;
                        for (int k = 0; k < B; ++k) v += Bm[r * B + k] * C.Jat(2 + 2 * a, k, q, m);
                        D[r * B + q] -= v;
                    }
            }
            double *Di = &C.Dinv[(size_t)c * BB];
            inv_block<B>(D, Di);
            const bool has_hi[3] = {C.l0[c] < C.td0[c] - 1, C.l1[c] < C.td1[c] - 1, C.l2[c] < C.td2[c] - 1};
            for (int a = 0; a < 3; ++a) {
                double *Cm = &C.Cb[((size_t)c * 3 + a) * BB];
                for (int r = 0; r < B; ++r)
                    for (int q = 0; q < B; ++q) {
                        double v = 0.0;
                        if (has_hi[a])
                            for (int k = 0; k < B; ++k) v += Di[r * B + k] * C.Jat(2 + 2 * a, k, q, c);
                        Cm[r * B + q] = v;
                    }
            }
        }
    }
}

template <int B>
static void ilu_solve_t(const Ctx &C, const double *r, double *x, double *y) {
    const long N = C.g.N;
    const int BB = B * B;
    const long nt = (long)C.tiles.size();
#pragma omp parallel for schedule(dynamic, 1)
    for (long t = 0; t < nt; ++t) {
        const std::vector<long> &cells = C.tiles[t];
        for (long c : cells) {                               // (I + L_A D~^-1) y = r
            double v[B];
            for (int q = 0; q < B; ++q) v[q] = r[q * N + c];
            const bool has_lo[3] = {C.l0[c] > 0, C.l1[c] > 0, C.l2[c] > 0};
            for (int a = 0; a < 3; ++a) {
                if (!has_lo[a]) continue;
                const long m = c - C.g.st[a];
                const double *Bm = &C.Bf[((size_t)c * 3 + a) * BB];
                for (int q = 0; q < B; ++q)
                    for (int k = 0; k < B; ++k) v[q] -= Bm[q * B + k] * y[k * N + m];
            }
            for (int q = 0; q < B; ++q) y[q * N + c] = v[q];
        }
        for (long i = (long)cells.size() - 1; i >= 0; --i) {  // (D~ + U_A) x = y
            const long c = cells[i];
            double v[B];
            const double *Di = &C.Dinv[(size_t)c * BB];
            for (int q = 0; q < B; ++q) {
                double s = 0.0;
                for (int k = 0; k < B; ++k) s += Di[q * B + k] * y[k * N + c];
                v[q] = s;
            }
            const bool has_hi[3] = {C.l0[c] < C.td0[c] - 1, C.l1[c] < C.td1[c] - 1, C.l2[c] < C.td2[c] - 1};
            for (int a = 0; a < 3; ++a) {
                if (!has_hi[a]) continue;
                const long m = c + C.g.st[a];
                const double *Cm = &C.Cb[((size_t)c * 3 + a) * BB];
                for (int q = 0; q < B; ++q)
                    for (int k = 0; k < B; ++k) v[q] -= Cm[q * B + k] * x[k * N + m];
            }
            for (int q = 0; q < B; ++q) x[q * N + c] = v[q];
        }
    }
}

// ---------------------------------------------------------------- tiled block ILU(1)  (oracle.linalg.TiledILU1)
// factor pattern of a row: 6 lower offsets (increasing global index), the diagonal, 6 upper offsets; (d0, d1, d2)
static const int ILU1_OFF[13][3] = {{0, 0, -1}, {1, 0, -1}, {0, 1, -1}, {0, -1, 0}, {1, -1, 0}, {-1, 0, 0}, {0, 0, 0},
                                    {1, 0, 0},  {-1, 1, 0}, {0, 1, 0},  {0, -1, 1}, {-1, 0, 1}, {0, 0, 1}};
static int ilu1_find(int d0, int d1, int d2) {
    for (int i = 0; i < 13; ++i)
        if (ILU1_OFF[i][0] == d0 && ILU1_OFF[i][1] == d1 && ILU1_OFF[i][2] == d2) return i;
    return -1;
}
static int ilu1_slot(int i) {          // stencil slot of a pattern entry, -1 for the level-1 fill entries
    const int *o = ILU1_OFF[i];
    if (std::abs(o[0]) + std::abs(o[1]) + std::abs(o[2]) > 1) return -1;
    if (o[0]) return o[0] < 0 ? 1 : 2;
    if (o[1]) return o[1] < 0 ? 3 : 4;
    if (o[2]) return o[2] < 0 ? 5 : 6;
    return 0;
}
static bool ilu1_inside(const Ctx &C, long c, int i) {
    const int a0 = C.l0[c] + ILU1_OFF[i][0], a1 = C.l1[c] + ILU1_OFF[i][1], a2 = C.l2[c] + ILU1_OFF[i][2];
    return a0 >= 0 && a0 < C.td0[c] && a1 >= 0 && a1 < C.td1[c] && a2 >= 0 && a2 < C.td2[c];
}
static long ilu1_off(const Ctx &C, int i) {
    return ILU1_OFF[i][0] * C.g.st[0] + ILU1_OFF[i][1] * C.g.st[1] + ILU1_OFF[i][2] * C.g.st[2];
}

template <int B>
static void ilu1_factor_t(Ctx &C) {
    const int BB = B * B;
    const long nt = (long)C.tiles.size();
    C.F1.resize((size_t)13 * BB * C.g.N);
    int target[6][6];
    for (int ik = 0; ik < 6; ++ik)
        for (int iu = 0; iu < 6; ++iu)
            target[ik][iu] = ilu1_find(ILU1_OFF[ik][0] + ILU1_OFF[7 + iu][0], ILU1_OFF[ik][1] + ILU1_OFF[7 + iu][1],
                                       ILU1_OFF[ik][2] + ILU1_OFF[7 + iu][2]);
#pragma omp parallel for schedule(dynamic, 1)
    for (long t = 0; t < nt; ++t) {
        for (long c : C.tiles[t]) {
            double F[13][BB];
            bool in[13];
            for (int i = 0; i < 13; ++i) {
                in[i] = ilu1_inside(C, c, i);
                const int sl = ilu1_slot(i);
                for (int r = 0; r < B; ++r)
                    for (int q = 0; q < B; ++q) F[i][r * B + q] = (in[i] && sl >= 0) ? C.Jat(sl, r, q, c) : 0.0;
            }
            for (int ik = 0; ik < 6; ++ik) {
                if (!in[ik]) continue;
                const long k = c + ilu1_off(C, ik);
                const double *Fk = &C.F1[(size_t)k * 13 * BB];
                double Lck[BB];
                for (int r = 0; r < B; ++r)
                    for (int q = 0; q < B; ++q) {
                        double v = 0.0;
                        for (int m = 0; m < B; ++m) v += F[ik][r * B + m] * Fk[6 * BB + m * B + q];
                        Lck[r * B + q] = v;
                    }
                for (int e = 0; e < BB; ++e) F[ik][e] = Lck[e];
                for (int iu = 0; iu < 6; ++iu) {
                    const int tg = target[ik][iu];
                    if (tg < 0) continue;
                    const double *U = Fk + (size_t)(7 + iu) * BB;       // zero where (k, j) leaves the tile
                    for (int r = 0; r < B; ++r)
                        for (int q = 0; q < B; ++q) {
                            double v = 0.0;
                            for (int m = 0; m < B; ++m) v += Lck[r * B + m] * U[m * B + q];
                            F[tg][r * B + q] -= v;
                        }
                }
            }
            double *out = &C.F1[(size_t)c * 13 * BB];
            for (int i = 0; i < 13; ++i)
                if (i != 6) for (int e = 0; e < BB; ++e) out[i * BB + e] = F[i][e];
            inv_block<B>(F[6], out + 6 * BB);
        }
    }
}

template <int B>
static void ilu1_solve_t(const Ctx &C, const double *r, double *x, double *y) {
    const long N = C.g.N;
    const int BB = B * B;
    const long nt = (long)C.tiles.size();
    long off[13];
    for (int i = 0; i < 13; ++i) off[i] = ilu1_off(C, i);
#pragma omp parallel for schedule(dynamic, 1)
    for (long t = 0; t < nt; ++t) {
        const std::vector<long> &cells = C.tiles[t];
        for (long c : cells) {                               // L y = r
            const double *Fc = &C.F1[(size_t)c * 13 * BB];
            double v[B];
            for (int q = 0; q < B; ++q) v[q] = r[q * N + c];
            for (int i = 0; i < 6; ++i) {
                if (!ilu1_inside(C, c, i)) continue;
                const long m = c + off[i];
                for (int q = 0; q < B; ++q)
                    for (int k = 0; k < B; ++k) v[q] -= Fc[i * BB + q * B + k] * y[k * N + m];
            }
            for (int q = 0; q < B; ++q) y[q * N + c] = v[q];
        }
        for (long ii = (long)cells.size() - 1; ii >= 0; --ii) {  // U x = y
            const long c = cells[ii];
            const double *Fc = &C.F1[(size_t)c * 13 * BB];
            double v[B];
            for (int q = 0; q < B; ++q) v[q] = y[q * N + c];
            for (int i = 7; i < 13; ++i) {
                if (!ilu1_inside(C, c, i)) continue;
                const long m = c + off[i];
                for (int q = 0; q < B; ++q)
                    for (int k = 0; k < B; ++k) v[q] -= Fc[i * BB + q * B + k] * x[k * N + m];
            }
            for (int q = 0; q < B; ++q) {
                double sacc = 0.0;
                for (int k = 0; k < B; ++k) sacc += Fc[6 * BB + q * B + k] * v[k];
                x[q * N + c] = sacc;
            }
        }
    }
}

static void ilu_factor_any(Ctx &C) {
    if (C.o.ilu_levels == 1) { if (C.b == 3) ilu1_factor_t<3>(C); else ilu1_factor_t<2>(C); }
    else                     { if (C.b == 3) ilu_factor_t<3>(C); else ilu_factor_t<2>(C); }
}
static void ilu_solve_any(const Ctx &C, const double *r, double *x, double *y) {
    if (C.o.ilu_levels == 1) { if (C.b == 3) ilu1_solve_t<3>(C, r, x, y); else ilu1_solve_t<2>(C, r, x, y); }
    else                     { if (C.b == 3) ilu_solve_t<3>(C, r, x, y); else ilu_solve_t<2>(C, r, x, y); }
}

// ---------------------------------------------------------------- stage 1 (oracle.linalg.decouple / TwoStagePC)
static void colsum(const Ctx &C, int q, int s, double *out) {      // column sums of block (q, s) == (A^T 1)
    const long N = C.g.N;
    const int n0 = C.g.n[0], n1 = C.g.n[1], n2 = C.g.n[2];
#pragma omp parallel for collapse(2) schedule(static)
    for (int i2 = 0; i2 < n2; ++i2)
        for (int i1 = 0; i1 < n1; ++i1)
            for (int i0 = 0; i0 < n0; ++i0) {
                const long c = i0 + (long)n0 * (i1 + (long)n1 * i2);
                const int I[3] = {i0, i1, i2};
                double v = C.Jat(0, q, s, c);
                for (int a = 0; a < 3; ++a) {
                    if (C.g.n[a] == 1) continue;
                    if (I[a] > 0) v += C.Jat(2 + 2 * a, q, s, c - C.g.st[a]);          // row lo has an entry in column hi
                    if (I[a] + 1 < C.g.n[a]) v += C.Jat(1 + 2 * a, q, s, c + C.g.st[a]);
                }
                out[c] = v;
            }
    (void)N;
}

// primary block (i,j) of the (decoupled) stage-1 system as a strided view
static SView blk(const Ctx &C, int i, int j) {
    const int npri = C.o.pc == 0 ? 1 : 2;
    if (!C.have_d) return C.Jview(i, j);
    SView v;
    v.base = C.At.data() + i * npri + j; v.cs = 7L * npri * npri; v.ss = (long)npri * npri;
    return v;
}

// Sp = A11 - A10 diag(A00)^-1 A01 (pc_fieldsplit_schur_precondition selfp): exact diagonal and 7-point collapse
// (oracle.linalg.SelfpSchur.setup; same order of operations)
static void selfp_build(Ctx &C) {
    const long N = C.g.N;
    const int n[3] = {C.g.n[0], C.g.n[1], C.g.n[2]};
    const SView A00 = blk(C, 0, 0), A01 = blk(C, 0, 1), A10 = blk(C, 1, 0), A11 = blk(C, 1, 1);
    C.sp_S7.assign((size_t)7 * N, 0.0);
    C.sp_invd.assign(N, 0.0); C.sp_d00inv.assign(N, 0.0); C.sp_u.assign(N, 0.0); C.sp_v.assign(N, 0.0);
#pragma omp parallel for schedule(static)
    for (long c = 0; c < N; ++c) C.sp_d00inv[c] = 1.0 / A00.at(0, c);
#pragma omp parallel for collapse(2) schedule(static)
    for (int i2 = 0; i2 < n[2]; ++i2)
        for (int i1 = 0; i1 < n[1]; ++i1)
            for (int i0 = 0; i0 < n[0]; ++i0) {
                const long c = i0 + (long)n[0] * (i1 + (long)n[1] * i2);
                const int I[3] = {i0, i1, i2};
                double S[7];
                for (int s = 0; s < 7; ++s) S[s] = A11.at(s, c);
                const double t0 = A10.at(0, c) * C.sp_d00inv[c];
                S[0] -= t0 * A01.at(0, c);
                double lump = 0.0;
                for (int s = 1; s < 7; ++s) {
                    const int a = (s - 1) / 2;
                    const bool odd = s % 2 == 1;
                    if (n[a] == 1 || (odd ? I[a] == 0 : I[a] + 1 >= n[a])) continue;      // no neighbour m in direction s
                    const long m = c + (odd ? -C.g.st[a] : C.g.st[a]);
                    const int opp = odd ? s + 1 : s - 1;
                    S[s] -= t0 * A01.at(s, c);
                    const double w = A10.at(s, c) * C.sp_d00inv[m];
                    S[s] -= w * A01.at(0, m);
                    S[0] -= w * A01.at(opp, m);
                    int Im[3] = {i0, i1, i2};
                    Im[a] += odd ? -1 : 1;
                    for (int t = 1; t < 7; ++t) {
                        const int at = (t - 1) / 2;
                        if (t == opp || n[at] == 1) continue;
                        if (t % 2 == 1 ? Im[at] == 0 : Im[at] + 1 >= n[at]) continue;    // m has no neighbour in direction t
                        lump -= w * A01.at(t, m);
                    }
                }
                C.sp_invd[c] = C.o.amg_omega / S[0];
                S[0] += lump;
                for (int s = 0; s < 7; ++s) C.sp_S7[(size_t)c * 7 + s] = S[s];
            }
}

static void pc_setup(Ctx &C) {
    const long N = C.g.N;
    const int b = C.b;
    if (C.tiles.empty()) ilu_layout(C);
    ilu_factor_any(C);
    if (C.o.pc == 4) return;                  // pc_bilu: bjacobi + ILU alone
    const int npri = C.o.pc == 0 ? 1 : 2;
    C.have_d = C.o.decoup != 0;
    if (C.have_d) {
        C.At.assign((size_t)7 * npri * npri * N, 0.0);      // [cell][slot][i][j]
        const int sfield = b - 1;
        if (C.o.decoup == 1 || C.o.decoup == 2) {
            vec Dss(N), D0s(N);
            for (int i = 0; i < npri; ++i) {
                C.dcoef[i][0].assign(N, 0.0);
                if (C.o.decoup == 1) {
#pragma omp parallel for schedule(static)
                    for (long c = 0; c < N; ++c) C.dcoef[i][0][c] = C.Jat(0, i, sfield, c) / C.Jat(0, sfield, sfield, c);
                } else {
                    colsum(C, sfield, sfield, Dss.data());
                    colsum(C, i, sfield, D0s.data());
#pragma omp parallel for schedule(static)
                    for (long c = 0; c < N; ++c) C.dcoef[i][0][c] = D0s[c] / Dss[c];
                }
            }
#pragma omp parallel for schedule(static)
            for (long c = 0; c < N; ++c)
                for (int s = 0; s < 7; ++s)
                    for (int i = 0; i < npri; ++i)
                        for (int j = 0; j < npri; ++j)
                            C.At[((size_t)c * 7 + s) * npri * npri + i * npri + j] =
                                C.Jat(s, i, j, c) - C.dcoef[i][0][c] * C.Jat(s, sfield, j, c);
        } else {      // QI_temp / TI_temp: two-phase pressure-only, (T,S) decoupled per cell
            vec E[6];
            const int pairs[6][2] = {{1, 1}, {1, 2}, {2, 1}, {2, 2}, {0, 1}, {0, 2}};
            for (int k = 0; k < 6; ++k) {
                E[k].assign(N, 0.0);
                if (C.o.decoup == 3) {
                    for (long c = 0; c < N; ++c) E[k][c] = C.Jat(0, pairs[k][0], pairs[k][1], c);
                } else colsum(C, pairs[k][0], pairs[k][1], E[k].data());
            }
            C.dcoef[0][0].assign(N, 0.0);
            C.dcoef[0][1].assign(N, 0.0);
            for (long c = 0; c < N; ++c) {
                const double det = E[0][c] * E[3][c] - E[1][c] * E[2][c];
                C.dcoef[0][0][c] = (E[4][c] * E[3][c] - E[5][c] * E[2][c]) / det;
                C.dcoef[0][1][c] = (E[5][c] * E[0][c] - E[4][c] * E[1][c]) / det;
            }
            for (long c = 0; c < N; ++c)
                for (int s = 0; s < 7; ++s)
                    C.At[(size_t)c * 7 + s] = C.Jat(s, 0, 0, c) - C.dcoef[0][0][c] * C.Jat(s, 1, 0, c) - C.dcoef[0][1][c] * C.Jat(s, 2, 0, c);
        }
    }
    C.amg_p.setup(blk(C, 0, 0));
    if (C.o.pc >= 1) {
        SView S;
        if (C.o.schur_a11 == 2) { selfp_build(C); S.base = C.sp_S7.data(); S.cs = 7; S.ss = 1; }
        else if (C.o.schur_a11 || C.o.fs_additive) S = blk(C, 1, 1);
        else { S.base = C.Sm.data(); S.cs = 7; S.ss = 1; }
        C.amg_T.setup(S);
    }
}

static void stage1(Ctx &C, const double *x, double *y) {      // TwoStagePC.stage1
    const long N = C.g.N;
    const int b = C.b, sfield = b - 1;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)b * N; ++i) y[i] = 0.0;
    if (C.o.pc == 0) {
        const double *r = x;
        if (C.have_d) {
            double *rr = C.w_r0.data();
            if (C.o.decoup >= 3) {
#pragma omp parallel for schedule(static)
                for (long c = 0; c < N; ++c) rr[c] = x[c] - C.dcoef[0][0][c] * x[N + c] - C.dcoef[0][1][c] * x[2 * N + c];
            } else {
#pragma omp parallel for schedule(static)
                for (long c = 0; c < N; ++c) rr[c] = x[c] - C.dcoef[0][0][c] * x[sfield * N + c];
            }
            r = rr;
        }
        C.amg_p.vcycle(r, y);
        return;
    }
    const double *r0 = x, *r1 = x + N;
    if (C.have_d) {
        double *a = C.w_r0.data(), *bb = C.w_r1.data();
#pragma omp parallel for schedule(static)
        for (long c = 0; c < N; ++c) {
            a[c] = x[c] - C.dcoef[0][0][c] * x[sfield * N + c];
            bb[c] = x[N + c] - C.dcoef[1][0][c] * x[sfield * N + c];
        }
        r0 = a; r1 = bb;
    }
    const SView A10 = blk(C, 1, 0), A01 = blk(C, 0, 1);
    double *y0 = C.w_y0.data(), *t = C.w_t.data();
    if (C.o.fs_additive) {                 // PCFIELDSPLIT additive: one V-cycle per field, no coupling
        C.amg_p.vcycle(r0, y);
        C.amg_T.vcycle(r1, y + N);
        return;
    }
    C.amg_p.vcycle(r0, y0);
    spmv_scalar(C.g, A10, y0, t);
#pragma omp parallel for schedule(static)
    for (long c = 0; c < N; ++c) t[c] = r1[c] - t[c];
    C.amg_T.vcycle(t, y + N);
    if (C.o.schur_a11 == 2) {          // selfp: x += w D^-1 (b - Sp x) with the exact Sp = A11 - A10 diag(A00)^-1 A01
        double *x1 = y + N, *u = C.sp_u.data(), *v = C.sp_v.data();
        spmv_scalar(C.g, A01, x1, u);
#pragma omp parallel for schedule(static)
        for (long c = 0; c < N; ++c) u[c] *= C.sp_d00inv[c];
        spmv_scalar(C.g, A10, u, v);
        spmv_scalar(C.g, blk(C, 1, 1), x1, u);
#pragma omp parallel for schedule(static)
        for (long c = 0; c < N; ++c) x1[c] += C.sp_invd[c] * (t[c] - (u[c] - v[c]));
    }
    spmv_scalar(C.g, A01, y + N, t);
#pragma omp parallel for schedule(static)
    for (long c = 0; c < N; ++c) t[c] = r0[c] - t[c];
    C.amg_p.vcycle(t, y);
}

static void pc_apply(Ctx &C, const double *x, double *y) {      // TwoStagePC.apply
    const long N = C.g.N;
    const int b = C.b;
    if (C.o.pc == 4) {
        if (C.w_yt.size() != (size_t)b * N) C.w_yt.assign((size_t)b * N, 0.0);
        ilu_solve_any(C, x, y, C.w_yt.data());
        return;
    }
    stage1(C, x, y);
    if (C.o.pc == 2) return;
    double *r = C.w_res.data(), *z = C.w_il.data();
    spmv_block(C, y, r);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)b * N; ++i) r[i] = x[i] - r[i];
    if (C.w_yt.size() != (size_t)b * N) C.w_yt.assign((size_t)b * N, 0.0);
    ilu_solve_any(C, r, z, C.w_yt.data());
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)b * N; ++i) y[i] = y[i] + z[i];
}

static double dot(const double *a, const double *b, long n) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (long i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

// FGMRES from x0 = 0 (oracle.linalg.fgmres): classical Gram-Schmidt, Givens, recurrence residual test
static int fgmres(Ctx &C, const double *bvec, double *x, int *its_out, double *rn_out) {
    const long nv = (long)C.b * C.g.N;
    const int maxit = C.o.ksp_max_it, restart = std::max(1, std::min(C.o.ksp_restart, maxit));
#pragma omp parallel for schedule(static)
    for (long i = 0; i < nv; ++i) x[i] = 0.0;
    const double bnorm = std::sqrt(dot(bvec, bvec, nv));
    *its_out = 0; *rn_out = bnorm;
    if (bnorm == 0.0) return 2;
    if (!std::isfinite(bnorm)) return -9;
    const double tol = std::max(C.o.ksp_rtol * bnorm, C.o.ksp_atol);
    int its = 0;
    vec r(bvec, bvec + nv), w(nv);
    double beta = bnorm;
    while (true) {
        const int m = std::min(restart, maxit - its);
        std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m, 0.0), sn(m, 0.0), gv(m + 1, 0.0), h(m + 1, 0.0);
        gv[0] = beta;
        if ((int)C.Vb.size() < 1) { C.Vb.emplace_back(nv); }
#pragma omp parallel for schedule(static)
        for (long i = 0; i < nv; ++i) C.Vb[0][i] = r[i] / beta;
        int k = 0, reason = 0;
        double res = beta;
        for (int j = 0; j < m; ++j) {
            while ((int)C.Vb.size() < j + 2) C.Vb.emplace_back(nv);
            while ((int)C.Zb.size() < j + 1) C.Zb.emplace_back(nv);
            pc_apply(C, C.Vb[j].data(), C.Zb[j].data());
            spmv_block(C, C.Zb[j].data(), w.data());
            // classical Gram-Schmidt as PETSc does it: VecMDot (all dots in one pass over w) then VecMAXPY
            {
                const int k1 = j + 1;
                std::vector<const double *> vp(k1);
                for (int i = 0; i < k1; ++i) vp[i] = C.Vb[i].data();
                const long BLK = 2048;
                const long nblk = (nv + BLK - 1) / BLK;
                for (int i = 0; i < k1; ++i) h[i] = 0.0;
#pragma omp parallel
                {
                    std::vector<double> loc(k1, 0.0);
#pragma omp for schedule(static) nowait
                    for (long bi = 0; bi < nblk; ++bi) {
                        const long q0 = bi * BLK, q1 = std::min(nv, q0 + BLK);
                        for (int i = 0; i < k1; ++i) {
                            const double *vi = vp[i];
                            double acc = 0.0;
                            for (long q = q0; q < q1; ++q) acc += vi[q] * w[q];
                            loc[i] += acc;
                        }
                    }
#pragma omp critical
                    for (int i = 0; i < k1; ++i) h[i] += loc[i];
                }
#pragma omp parallel for schedule(static)
                for (long bi = 0; bi < nblk; ++bi) {
                    const long q0 = bi * BLK, q1 = std::min(nv, q0 + BLK);
                    for (int i = 0; i < k1; ++i) {
                        const double hi = h[i];
                        const double *vi = vp[i];
                        for (long q = q0; q < q1; ++q) w[q] = w[q] - hi * vi[q];
                    }
                }
            }
            const double hn = std::sqrt(dot(w.data(), w.data(), nv));
            for (int i = 0; i <= j; ++i) H[(size_t)i * m + j] = h[i];
            H[(size_t)(j + 1) * m + j] = hn;
            for (int i = 0; i < j; ++i) {
                const double t = cs[i] * H[(size_t)i * m + j] + sn[i] * H[(size_t)(i + 1) * m + j];
                H[(size_t)(i + 1) * m + j] = -sn[i] * H[(size_t)i * m + j] + cs[i] * H[(size_t)(i + 1) * m + j];
                H[(size_t)i * m + j] = t;
            }
            const double d = std::hypot(H[(size_t)j * m + j], H[(size_t)(j + 1) * m + j]);
            cs[j] = H[(size_t)j * m + j] / d;
            sn[j] = H[(size_t)(j + 1) * m + j] / d;
            H[(size_t)j * m + j] = d;
            H[(size_t)(j + 1) * m + j] = 0.0;
            gv[j + 1] = -sn[j] * gv[j];
            gv[j] = cs[j] * gv[j];
            ++its;
            k = j + 1;
            res = std::fabs(gv[j + 1]);
            if (!std::isfinite(res)) { reason = -9; break; }
            if (res <= tol) { reason = 2; break; }
            if (hn == 0.0) { reason = 2; break; }
            double *vn = C.Vb[j + 1].data();
#pragma omp parallel for schedule(static)
            for (long q = 0; q < nv; ++q) vn[q] = w[q] / hn;
        }
        std::vector<double> yk(k, 0.0);
        for (int i = k - 1; i >= 0; --i) {
            double s = gv[i];
            for (int q = i + 1; q < k; ++q) s -= H[(size_t)i * m + q] * yk[q];
            yk[i] = s / H[(size_t)i * m + i];
        }
        for (int i = 0; i < k; ++i) {
            const double a = yk[i];
            const double *zi = C.Zb[i].data();
#pragma omp parallel for schedule(static)
            for (long q = 0; q < nv; ++q) x[q] = x[q] + a * zi[q];
        }
        *its_out = its; *rn_out = res;
        if (reason) return reason;
        if (its >= maxit) return -3;
        spmv_block(C, x, w.data());
#pragma omp parallel for schedule(static)
        for (long q = 0; q < nv; ++q) r[q] = bvec[q] - w[q];
        beta = std::sqrt(dot(r.data(), r.data(), nv));
        *rn_out = beta;
        if (beta <= tol) return 2;
    }
}

static void newton(Ctx &C, double budget_s, Info *info) {      // OracleEngine.newton_solve
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    const long nv = (long)C.b * C.g.N;
    const bool schur = C.o.pc >= 1;
    vec dx(nv);
    residual(C, C.u.data(), C.R.data());
    double fnorm = std::sqrt(dot(C.R.data(), C.R.data(), nv));
    const double fnorm0 = fnorm;
    int nits = 0, lits = 0, reason = 0, complete = 1;
    if (!std::isfinite(fnorm)) reason = -4;
    else if (fnorm < C.o.snes_atol) reason = 2;
    while (reason == 0) {
        if (nits >= C.o.snes_max_it) { reason = -5; break; }
        if (budget_s > 0 && nits > 0 && elapsed() > budget_s) { complete = 0; break; }
        jacobian(C, C.u.data(), schur);
        pc_setup(C);
        int kits = 0;
        double rn = 0.0;
        const int kreason = fgmres(C, C.R.data(), dx.data(), &kits, &rn);
        lits += kits;
        if (kreason < 0) { reason = -3; break; }
#pragma omp parallel for schedule(static)
        for (long i = 0; i < nv; ++i) C.u[i] = C.u[i] - dx[i];
        residual(C, C.u.data(), C.R.data());
        fnorm = std::sqrt(dot(C.R.data(), C.R.data(), nv));
        ++nits;
        const double snorm = std::sqrt(dot(dx.data(), dx.data(), nv)), xnorm = std::sqrt(dot(C.u.data(), C.u.data(), nv));
        if (!std::isfinite(fnorm)) reason = -4;
        else if (fnorm < C.o.snes_atol) reason = 2;
        else if (fnorm <= C.o.snes_rtol * fnorm0) reason = 3;
        else if (snorm < C.o.snes_stol * xnorm) reason = 4;
    }
    info->nits = nits; info->lits = lits; info->reason = reason; info->complete = complete;
    info->fnorm0 = fnorm0; info->fnorm = fnorm; info->seconds = elapsed();
}

}  // namespace

// ---------------------------------------------------------------- C ABI (ctypes: oracle/cport/__init__.py)
extern "C" {

void *cp_create(int nphase, const int *n, const double *h, int gaxis, const double *prm17, const double *phi,
                const double *K0, const double *K1, const double *K2, const double *kT, int nsrc, const int64_t *cell,
                const int32_t *kind, const int32_t *cst, const double *wt, const double *bhp, const double *qmax,
                const double *WI, const Opts *opts) {
    Ctx *C = new Ctx();
    C->nph = nphase; C->b = nphase + 1; C->gaxis = gaxis;
    C->g.set(n[0], n[1], n[2]);
    for (int a = 0; a < 3; ++a) C->h[a] = h[a];
    C->V = h[0] * h[1] * h[2];
    std::memcpy(&C->prm, prm17, sizeof(Prm));
    C->o = *opts;
    const long N = C->g.N;
    C->phi.assign(phi, phi + N);
    C->K[0].assign(K0, K0 + N); C->K[1].assign(K1, K1 + N); C->K[2].assign(K2, K2 + N);
    C->kTs.assign(kT, kT + N);
    for (int a = 0; a < 3; ++a) {
        C->TK[a].assign(N, 0.0);
        C->G[a] = C->V / (h[a] * h[a]);
        if (C->g.n[a] > 1)
            for (long c = 0; c < N; ++c) {
                const int ia = (int)((c / C->g.st[a]) % C->g.n[a]);
                if (ia + 1 < C->g.n[a]) C->TK[a][c] = harmonic(C->K[a][c], C->K[a][c + C->g.st[a]]) * (C->V / (h[a] * h[a]));
            }
    }
    if (nphase == 2) {
        C->w0 = C->prm.T_prod;
        C->w2 = C->prm.T_prod * (C->prm.c_v_w * (1 - C->prm.S_o) + C->prm.c_v_o * C->prm.S_o);
    } else { C->w0 = 1.0; C->w2 = 0.0; }
    C->nsrc = nsrc;
    C->scell.assign(cell, cell + nsrc); C->skind.assign(kind, kind + nsrc); C->sconst.assign(cst, cst + nsrc);
    C->swt.assign(wt, wt + nsrc); C->sbhp.assign(bhp, bhp + nsrc); C->sqmax.assign(qmax, qmax + nsrc);
    C->sWI.assign(WI, WI + nsrc);
    const int b = C->b;
    C->u.assign((size_t)b * N, 0.0); C->u_old = C->u; C->old_acc = C->u; C->R = C->u;
    C->J.assign((size_t)7 * b * b * N, 0.0);
    for (int k = 0; k < 3; ++k) { C->pr_ro[k].assign(N, 0.0); C->pr_rw[k].assign(N, 0.0); }
    for (int k = 0; k < 4; ++k) { C->pr_Lw[k].assign(N, 0.0); C->pr_Lo[k].assign(N, 0.0); }
    C->pr_kT.assign(N, 0.0); C->pr_kTS.assign(N, 0.0);
    C->w_r0.assign(N, 0.0); C->w_r1.assign(N, 0.0); C->w_t.assign(N, 0.0); C->w_y0.assign(N, 0.0); C->w_y1.assign(N, 0.0);
    C->w_res.assign((size_t)b * N, 0.0); C->w_il.assign((size_t)b * N, 0.0);
    // AMG schedules (TwoStagePC.__init__): mean interior-face transmissibility per axis; S~: G[a]
    double st[3], sg[3];
    for (int a = 0; a < 3; ++a) {
        double acc = 0.0, cnt = 0.0;
        if (C->g.n[a] > 1)
            for (long c = 0; c < N; ++c) {
                const int ia = (int)((c / C->g.st[a]) % C->g.n[a]);
                if (ia + 1 < C->g.n[a]) { acc += C->TK[a][c]; cnt += 1.0; }
            }
        st[a] = cnt > 0 ? acc / cnt : 0.0;
        sg[a] = C->g.n[a] > 1 ? C->G[a] : 0.0;
    }
    C->amg_p.init(C->g.n, st, C->o);
    if (C->o.pc >= 1) {
        C->amg_T.init(C->g.n, sg, C->o);
    }
    return C;
}

void cp_destroy(void *c) { delete (Ctx *)c; }
void cp_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int cp_max_threads(void) { return omp_get_max_threads(); }
// ABI guard: the Python loader compares these with ctypes.sizeof of its mirrors (a stale .so from an older checkout fails loudly)
int cp_sizeof_opts(void) { return (int)sizeof(Opts); }
int cp_sizeof_info(void) { return (int)sizeof(Info); }

void cp_set_state(void *c, const double *u) { Ctx *C = (Ctx *)c; std::copy(u, u + C->u.size(), C->u.begin()); }
void cp_get_state(void *c, double *u) { Ctx *C = (Ctx *)c; std::copy(C->u.begin(), C->u.end(), u); }
void cp_get_old(void *c, double *u) { Ctx *C = (Ctx *)c; std::copy(C->u_old.begin(), C->u_old.end(), u); }
void cp_restore(void *c) { Ctx *C = (Ctx *)c; C->u = C->u_old; }
void cp_set_dt(void *c, double dt) { ((Ctx *)c)->dt = dt; }
void cp_set_old(void *c, const double *u) {
    Ctx *C = (Ctx *)c;
    if (u) std::copy(u, u + C->u_old.size(), C->u_old.begin()); else C->u_old = C->u;
    compute_props(*C, C->u_old.data());
    accum(*C, C->u_old.data(), C->old_acc.data());
}
void cp_sat_range(void *c, double *lo, double *hi) {
    Ctx *C = (Ctx *)c;
    const double *S = C->u.data() + 2 * C->g.N;
    *lo = *std::min_element(S, S + C->g.N);
    *hi = *std::max_element(S, S + C->g.N);
}
void cp_clamp(void *c) {
    Ctx *C = (Ctx *)c;
    double *S = C->u.data() + 2 * C->g.N;
    for (long i = 0; i < C->g.N; ++i) S[i] = std::min(1.0, std::max(0.0, S[i]));
}
void cp_residual(void *c, double *out) {
    Ctx *C = (Ctx *)c;
    residual(*C, C->u.data(), C->R.data());
    std::copy(C->R.begin(), C->R.end(), out);
}
void cp_jacobian(void *c, double *Jout, double *Smout) {
    Ctx *C = (Ctx *)c;
    jacobian(*C, C->u.data(), Smout != nullptr || C->o.pc >= 1);
    const long N = C->g.N;
    const int bb = C->b * C->b;
    if (Jout)             // plane order [slot][row][col][cell], like the numpy oracle
        for (long cc = 0; cc < N; ++cc)
            for (int e = 0; e < 7 * bb; ++e) Jout[(long)e * N + cc] = C->J[(size_t)cc * 7 * bb + e];
    if (Smout)
        for (long cc = 0; cc < N; ++cc)
            for (int e = 0; e < 7; ++e) Smout[(long)e * N + cc] = C->Sm[(size_t)cc * 7 + e];
}
void cp_pc_setup(void *c) { pc_setup(*(Ctx *)c); }
void cp_pc_apply(void *c, const double *x, double *y) { pc_apply(*(Ctx *)c, x, y); }
void cp_stage1(void *c, const double *x, double *y) { stage1(*(Ctx *)c, x, y); }
void cp_spmv(void *c, const double *x, double *y) { spmv_block(*(Ctx *)c, x, y); }
void cp_ilu_solve(void *c, const double *r, double *x) {
    Ctx *C = (Ctx *)c;
    vec y((size_t)C->b * C->g.N);
    ilu_solve_any(*C, r, x, y.data());
}
void cp_vcycle(void *c, int which, const double *b, double *x) {
    Ctx *C = (Ctx *)c;
    (which == 0 ? C->amg_p : C->amg_T).vcycle(b, x);
}
int cp_fgmres(void *c, const double *b, double *x, int *its, double *rn) { return fgmres(*(Ctx *)c, b, x, its, rn); }
void cp_newton(void *c, double budget_s, Info *info) { newton(*(Ctx *)c, budget_s, info); }
int cp_amg_levels(void *c, int which) { Ctx *C = (Ctx *)c; return (int)(which == 0 ? C->amg_p : C->amg_T).lv.size(); }
// debug/test access: operator planes (7*N), weights (2*N: wm, wp) and inverse diagonal (N) of AMG level l
long cp_amg_level(void *c, int which, int l, double *A, double *w, double *invd) {
    Ctx *C = (Ctx *)c;
    const AmgLevel &L = (which == 0 ? C->amg_p : C->amg_T).lv[l];
    for (long cc = 0; cc < L.g.N; ++cc) {
        for (int s = 0; s < 7 && A; ++s) A[(long)s * L.g.N + cc] = L.A(s, cc);
        if (invd) invd[cc] = L.invd(cc);
    }
    if (w && L.axis >= 0) { std::copy(L.wm.begin(), L.wm.end(), w); std::copy(L.wp.begin(), L.wp.end(), w + L.g.N); }
    return L.g.N;
}
int cp_amg_trunc(void *c, int which) { Ctx *C = (Ctx *)c; return (which == 0 ? C->amg_p : C->amg_T).trunc_level; }
int cp_ntiles(void *c) { Ctx *C = (Ctx *)c; if (C->tiles.empty()) ilu_layout(*C); return (int)C->tiles.size(); }

}  // extern "C"

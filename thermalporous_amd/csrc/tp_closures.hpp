// Device closure laws with analytic derivatives -- restates
// /root/reference/thermalporous/physicalparameters.py:37-57 (oil), :69-90 (water) for gfx950.
// Quirks preserved: water_* use 272.15 (not 273.15) as the Celsius offset (:80,:89).
#pragma once
#include "tp_common.hpp"

namespace tp {

// oil_rho = rho_ref * e^(c(10p - p0)) * e^(-e1 (T - T0))            (:37-46)
__device__ __forceinline__ void oil_rho(double p, double T, const DevPrm &q, double &r, double &r_p, double &r_T) {
    constexpr double c = 5.5e-5, p0 = 1.01325, e1 = 2.5e-4, T0 = 15.5556 + 273.15;
    r = q.rho_ref * exp(c * (p * 1e1 - p0)) * exp(-e1 * (T - T0));
    r_p = (10.0 * c) * r;
    r_T = (-e1) * r;
}

// oil_mu = 1e-3 * 10^(A1 API + A2) * Tf^(A3 API + A4), Tf = 1.8 (T - 273.15) + 32     (:48-57)
__device__ __forceinline__ void oil_mu(double T, const DevPrm &q, double &mu, double &mu_T) {
    const double Tf = 1.8 * (T - 273.15) + 32.0;
    mu = q.mu_o_coef * pow(Tf, q.mu_o_exp);
    mu_T = mu * q.mu_o_exp * 1.8 / Tf;
}

// water_rho: Trangenstein's modification of Kell's correlation         (:69-82)
__device__ __forceinline__ void water_rho(double p, double T, double &r, double &r_p, double &r_T) {
    constexpr double E0 = 999.83952, E1 = 16.955176, E2 = -7.987e-3, E3 = -46.170461e-6, E4 = 105.56302e-9,
                     E5 = -280.54353e-12, E6 = 16.87985e-3, E7 = 10.2, Cw = 3.98854e-4;
    const double Tc = T - 272.15;
    const double P = E0 + Tc * (E1 + Tc * (E2 + Tc * (E3 + Tc * (E4 + Tc * E5))));
    const double dP = E1 + Tc * (2 * E2 + Tc * (3 * E3 + Tc * (4 * E4 + Tc * 5 * E5)));
    const double den = 1.0 + E6 * Tc;
    const double ex = exp(Cw * (p - E7));
    r = P * ex / den;
    r_p = Cw * r;
    r_T = (dP - P * E6 / den) * ex / den;
}

// water_mu: Grabowski                                                   (:84-90)
__device__ __forceinline__ void water_mu(double T, double &mu, double &mu_T) {
    constexpr double Aw = 2.1850, Bw = 0.04012, Cw = 5.1547e-6;
    const double Tf = 1.8 * (T - 272.15) + 32.0;
    const double den = -1.0 + Bw * Tf + Cw * Tf * Tf;
    mu = 1e-3 * Aw / den;
    mu_T = -mu * (Bw + 2.0 * Cw * Tf) * 1.8 / den;
}

// ---- forward-mode dual number with three partials (p, T, S): used only by the well kernel ----------
struct Dual3 {
    double v;
    double d[3];
    __host__ __device__ Dual3() : v(0.0) { d[0] = d[1] = d[2] = 0.0; }
    __host__ __device__ explicit Dual3(double x) : v(x) { d[0] = d[1] = d[2] = 0.0; }
    __host__ __device__ Dual3(double x, int k) : v(x) { d[0] = d[1] = d[2] = 0.0; d[k] = 1.0; }
};
__host__ __device__ inline Dual3 operator+(const Dual3 &a, const Dual3 &b) {
    Dual3 r; r.v = a.v + b.v; for (int k = 0; k < 3; ++k) r.d[k] = a.d[k] + b.d[k]; return r;
}
__host__ __device__ inline Dual3 operator-(const Dual3 &a, const Dual3 &b) {
    Dual3 r; r.v = a.v - b.v; for (int k = 0; k < 3; ++k) r.d[k] = a.d[k] - b.d[k]; return r;
}
__host__ __device__ inline Dual3 operator*(const Dual3 &a, const Dual3 &b) {
    Dual3 r; r.v = a.v * b.v; for (int k = 0; k < 3; ++k) r.d[k] = a.d[k] * b.v + a.v * b.d[k]; return r;
}
__host__ __device__ inline Dual3 operator*(const Dual3 &a, double s) {
    Dual3 r; r.v = a.v * s; for (int k = 0; k < 3; ++k) r.d[k] = a.d[k] * s; return r;
}
__host__ __device__ inline Dual3 operator/(const Dual3 &a, const Dual3 &b) {
    Dual3 r; r.v = a.v / b.v;
    for (int k = 0; k < 3; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) / b.v;
    return r;
}
__host__ __device__ inline Dual3 operator/(const Dual3 &a, double s) { return a * (1.0 / s); }

__device__ inline Dual3 oil_rho_t(const Dual3 &p, const Dual3 &T, const DevPrm &q) {
    double r, rp, rT;
    oil_rho(p.v, T.v, q, r, rp, rT);
    Dual3 o; o.v = r;
    for (int k = 0; k < 3; ++k) o.d[k] = rp * p.d[k] + rT * T.d[k];
    return o;
}
__device__ inline Dual3 oil_mu_t(const Dual3 &T, const DevPrm &q) {
    double m, mT;
    oil_mu(T.v, q, m, mT);
    Dual3 o; o.v = m;
    for (int k = 0; k < 3; ++k) o.d[k] = mT * T.d[k];
    return o;
}
__device__ inline Dual3 water_rho_t(const Dual3 &p, const Dual3 &T) {
    double r, rp, rT;
    water_rho(p.v, T.v, r, rp, rT);
    Dual3 o; o.v = r;
    for (int k = 0; k < 3; ++k) o.d[k] = rp * p.d[k] + rT * T.d[k];
    return o;
}
__device__ inline Dual3 water_mu_t(const Dual3 &T) {
    double m, mT;
    water_mu(T.v, m, mT);
    Dual3 o; o.v = m;
    for (int k = 0; k < 3; ++k) o.d[k] = mT * T.d[k];
    return o;
}

}  // namespace tp

"""DG0/TPFA residual and exact block Jacobian on a structured box (oracle; test infrastructure).

Restates the UFL forms of the reference for a DG0 space on a rectangle/box mesh, where
``int f q dx -> f_i |E|`` and ``int g jump(q) dS -> +g|e|`` on the '+' cell, ``-g|e|`` on the
'-' cell:
  single-phase  /root/reference/thermalporous/singlephase.py:60-165 (2D), :167-273 (3D)
  two-phase     /root/reference/thermalporous/twophase.py:67-235 (2D), :237-411 (3D)
  sources       wellcase.py:171-266, heatercase.py, sourceterms.py:155-269
  ConvDiff S~   preconditioners.py:11-163 (1-phase), :165-333 (2-phase)

Layout ("internal" axes): cell arrays have shape (n2, n1, n0), internal axis 0 fastest.
``spec`` (plain data, produced by thermalporous_amd.problem.build_spec) holds
  nphase, n=(n0,n1,n2), h=(h0,h1,h2), gaxis (internal axis with gravity or -1),
  phi, K=[K0,K1,K2], kT (1-phase static conductivity), prm (dict), sources (dict of arrays).
Orientation: along every internal axis '+' is the lower-index cell; on the gravity axis
this is the lower cell in space (z up), so that hydrostatic equilibrium gives zero flux
(singlephase.py:215, twophase.py:317-318).

Stencil slots of the Jacobian: 0 diag, 1 (-a0), 2 (+a0), 3 (-a1), 4 (+a1), 5 (-a2), 6 (+a2).
Unknown/equation order: (p,T[,S_o]) <-> (mass|pressure-eq, energy[, oil]).
"""
import numpy as np

from . import closures as cl

PROD, INJ, HEATER = 0, 1, 2


def _ax(a):
    return 2 - a


def _lo(a):
    s = [slice(None)] * 3
    s[_ax(a)] = slice(None, -1)
    return tuple(s)


def _hi(a):
    s = [slice(None)] * 3
    s[_ax(a)] = slice(1, None)
    return tuple(s)


def harmonic(ap, am):
    """conditional(gt(avg(a),0), a('+')a('-')/avg(a), 0)  (singlephase.py:98-103)."""
    s = ap + am
    with np.errstate(divide="ignore", invalid="ignore"):
        h = np.where(s.real > 0.0, 2.0 * ap * am / np.where(s == 0, 1.0, s), 0.0)
    return h


class Problem:
    def __init__(self, spec):
        self.spec = spec
        self.nph = int(spec["nphase"])
        self.b = 2 if self.nph == 1 else 3
        self.n = tuple(int(v) for v in spec["n"])
        n0, n1, n2 = self.n
        self.shape = (n2, n1, n0)
        self.h = tuple(float(v) for v in spec["h"])
        self.gaxis = int(spec["gaxis"])
        self.prm = dict(spec["prm"])
        self.V = self.h[0] * self.h[1] * self.h[2]
        f = lambda x: np.broadcast_to(np.asarray(x, dtype=float), self.shape).copy()
        self.phi = f(spec["phi"])
        self.K = [f(k) for k in spec["K"]]
        self.kT_static = f(spec["kT"]) if spec.get("kT") is not None else None
        # face transmissibilities  T^K_f = H(K)|e|/Delta_h   
        self.TK = []
        self.G = []
        for a in range(3):
            t = np.zeros(self.shape)
            if self.n[a] > 1:
                t[_lo(a)] = harmonic(self.K[a][_lo(a)], self.K[a][_hi(a)]) * (self.V / self.h[a] ** 2)
            self.TK.append(t)
            self.G.append(self.V / self.h[a] ** 2)
        src = spec.get("sources") or {}
        self.src = {k: np.asarray(v) for k, v in src.items()} if len(src) else None
        if self.src is not None and len(self.src["cell"]) == 0:
            self.src = None
        p = self.prm
        if self.nph == 2:
            # weights (twophase.py:142-147, 321-326): scaled_eqns and pressure_eqn are hard True (:29-30)
            self.w0 = p["T_prod"]
            self.w2 = p["T_prod"] * (p["c_v_w"] * (1 - p["S_o"]) + p["c_v_o"] * p["S_o"])
        else:
            self.w0 = 1.0      # m_w = 1 because scaled_eqns = False (singlephase.py:26,112-115)
            self.w2 = 0.0
        self.dt = None
        self.old = None

    # ---------------------------------------------------------------- properties
    def props(self, p, T, S):
        """Per-cell phase densities/mobilities and their derivatives."""
        prm = self.prm
        d = {}
        ro, ro_p, ro_T = cl.oil_rho(p, T, prm["API"])
        mo, mo_T = cl.oil_mu(T, prm["API"])
        d["ro"] = (ro, ro_p, ro_T)
        if self.nph == 2:
            rw, rw_p, rw_T = cl.water_rho(p, T)
            mw, mw_T = cl.water_mu(T)
            d["rw"] = (rw, rw_p, rw_T)
            kw_ = 1.0 - S
            # L = kr*rho/mu and derivatives (p, T, S)
            d["Lw"] = (kw_ * rw / mw, kw_ * rw_p / mw, kw_ * (rw_T / mw - rw * mw_T / mw ** 2), -rw / mw)
            d["Lo"] = (S * ro / mo, S * ro_p / mo, S * (ro_T / mo - ro * mo_T / mo ** 2), ro / mo)
            phi = self.phi
            d["kT"] = (phi * (S * prm["ko"] + (1 - S) * prm["kw"]) + (1 - phi) * prm["kr"],   # twophase.py:135,311
                       phi * (prm["ko"] - prm["kw"]))
            d["mu"] = (mo, mo_T, mw, mw_T)
        else:
            d["Lo"] = (ro / mo, ro_p / mo, ro_T / mo - ro * mo_T / mo ** 2, 0.0 * ro)
            d["kT"] = (self.kT_static, 0.0 * self.kT_static)
            d["mu"] = (mo, mo_T)
        return d

    def phases(self):
        """(key, rho-key, c_energy, c_in_eq0, goes_to_eq2)."""
        prm = self.prm
        if self.nph == 2:
            return [("Lw", "rw", prm["c_v_w"], prm["c_v_w"], False),
                    ("Lo", "ro", prm["c_v_o"], prm["c_v_o"], True)]
        return [("Lo", "ro", prm["c_v_o"], 1.0, False)]

    # ---------------------------------------------------------------- accumulation
    def accum(self, p, T, S, pr=None):
        """Accumulation densities per cell (times phi etc.), shape (b,...).

        1-phase: singlephase.py:120,123.  2-phase: twophase.py:162,165,170,174."""
        prm = self.prm
        pr = pr or self.props(p, T, S)
        phi = self.phi
        rock = (1 - phi) * prm["rho_r"] * prm["c_r"]
        if self.nph == 2:
            Mw = phi * pr["rw"][0] * (1.0 - S)
            Mo = phi * pr["ro"][0] * S
            return np.array([prm["c_v_w"] * Mw + prm["c_v_o"] * Mo,
                             (prm["c_v_w"] * Mw + prm["c_v_o"] * Mo) * T + rock * T,
                             Mo])
        Mo = phi * pr["ro"][0]
        return np.array([Mo, prm["c_v_o"] * Mo * T + rock * T])

    def set_old(self, u_old):
        u_old = self.as_fields(u_old)
        self.u_old = u_old.copy()
        self.old = self.accum(*self.split(u_old))

    def set_dt(self, dt):
        self.dt = float(dt)

    def as_fields(self, u):
        return np.asarray(u).reshape((self.b,) + self.shape)

    def split(self, u):
        if self.nph == 2:
            return u[0], u[1], u[2]
        return u[0], u[1], None

    # ---------------------------------------------------------------- residual
    def residual(self, u):
        """F(u) with accumulation against the stored old state; returns array (b, n2, n1, n0)."""
        u = self.as_fields(u)
        p, T, S = self.split(u)
        pr = self.props(p, T, S)
        w = [self.w0, 1.0, self.w2]
        R = (self.accum(p, T, S, pr) - self.old) * (self.V / self.dt)
        R = R.astype(u.dtype, copy=False)
        for q in range(self.b):
            R[q] = R[q] * w[q]
        for a in range(3):
            if self.n[a] == 1:
                continue
            f = self._face_flux(a, p, T, pr)
            lo, hi = _lo(a), _hi(a)
            for q in range(self.b):
                R[q][lo] += f[q]
                R[q][hi] -= f[q]
        if self.src is not None:
            c = self.src["cell"]
            pc, Tc = p.reshape(-1)[c], T.reshape(-1)[c]
            Sc = S.reshape(-1)[c] if S is not None else None
            s = self.source_terms(pc, Tc, Sc)
            Rf = R.reshape(self.b, -1)
            for q in range(self.b):
                np.subtract.at(Rf[q], c, s[q])
        return R

    def _phi_face(self, a, p, rho):
        """Driving force Phi = jump(p) - g*Dh*avg(rho) on the gravity axis (singlephase.py:215)."""
        lo, hi = _lo(a), _hi(a)
        Phi = p[lo] - p[hi]
        if a == self.gaxis:
            Phi = Phi - (self.prm["g"] * self.h[a] * 0.5) * (rho[lo] + rho[hi])
        return Phi

    def _face_flux(self, a, p, T, pr):
        lo, hi = _lo(a), _hi(a)
        TK = self.TK[a][lo]
        f = [0.0] * self.b
        for (Lk, rk, ce, c0, to2) in self.phases():
            Phi = self._phi_face(a, p, pr[rk][0])
            up = Phi.real > 0.0                       # gt(flow, 0): strict, ties -> '-' side
            L = np.where(up, pr[Lk][0][lo], pr[Lk][0][hi])
            Tu = np.where(up, T[lo], T[hi])
            F = TK * L * Phi
            f[0] = f[0] + self.w0 * c0 * F
            f[1] = f[1] + ce * Tu * F
            if to2:
                f[2] = f[2] + self.w2 * F
        kT = pr["kT"][0]
        f[1] = f[1] + harmonic(kT[lo], kT[hi]) * self.G[a] * (T[lo] - T[hi])
        return f

    # ---------------------------------------------------------------- sources
    def source_terms(self, p, T, S, return_rates=False):
        """Per-entry source vector s (b, nent); the residual gets R[cell] -= s.

        Peaceman / constant rates: wellcase.py:171-266; assembly into F: singlephase.py:151-165,
        twophase.py:212-235; heaters: U*(T_inj-T)*delta."""
        prm = self.prm
        e = self.src
        kind, wt, bhp, qmax, WI, const = (e[k] for k in ("kind", "wt", "bhp", "max_rate", "WI", "const"))
        API = prm["API"]
        is_prod, is_inj, is_heat = kind == PROD, kind == INJ, kind == HEATER
        mo, _ = cl.oil_mu(T, API)
        ro, _, _ = cl.oil_rho(p, T, API)
        dd_raw = bhp - p
        dd = np.where(is_prod, np.where(dd_raw.real >= 0.0, 0.0, dd_raw), np.where(dd_raw.real <= 0.0, 0.0, dd_raw))
        out = np.zeros((self.b, len(kind)), dtype=np.result_type(p, T))
        rates = {}
        if self.nph == 1:
            rate = WI / mo * dd
            rate = np.where(np.abs(rate.real) - np.abs(qmax) >= 0.0, qmax, rate)
            rate = np.where(const != 0, qmax, rate)
            ro_inj, _, _ = cl.oil_rho(p, prm["T_inj"] + 0 * T, API)
            cv = prm["c_v_o"]
            m = np.where(is_prod, ro * rate, np.where(is_inj, ro_inj * rate, 0.0))
            out[0] = self.w0 * m * wt
            out[1] = np.where(is_prod, ro * rate * cv * T, np.where(is_inj, ro_inj * rate * cv * prm["T_inj"],
                              prm["U"] * (prm["T_inj"] - T))) * wt
            rates = {"rate": np.where(is_heat, 0.0, rate)}
        else:
            mw, _ = cl.water_mu(T)
            rw, _, _ = cl.water_rho(p, T)
            lam_t = S / mo + (1.0 - S) / mw           # 1/mu, wellcase.py:212
            # producers: total-mobility Peaceman, split by mobility fractions (:204-235)
            rate_p = WI * lam_t * dd
            rate_p = np.where(np.abs(rate_p.real) - np.abs(qmax) >= 0.0, qmax, rate_p)
            rate_p = np.where(const != 0, qmax, rate_p)
            qw = (1.0 - S) / mw / lam_t * rate_p
            qo = S / mo / lam_t * rate_p
            # injectors: water only, viscosity at the cell temperature (twophase.py:225)
            rate_i = WI / mw * dd
            rate_i = np.where(np.abs(rate_i.real) - np.abs(qmax) >= 0.0, qmax, rate_i)
            rate_i = np.where(const != 0, qmax, rate_i)
            rw_inj, _, _ = cl.water_rho(p, prm["T_inj"] + 0 * T)
            cw, co = prm["c_v_w"], prm["c_v_o"]
            out[0] = self.w0 * np.where(is_prod, cw * rw * qw + co * ro * qo,
                                        np.where(is_inj, cw * rw_inj * rate_i, 0.0)) * wt
            out[2] = self.w2 * np.where(is_prod, ro * qo, 0.0) * wt
            out[1] = np.where(is_prod, (rw * qw * cw + ro * qo * co) * T,
                              np.where(is_inj, rw_inj * rate_i * cw * prm["T_inj"],
                                       prm["U"] * (prm["T_inj"] - T))) * wt
            rates = {"rate": np.where(is_prod, rate_p, np.where(is_inj, rate_i, 0.0)),
                     "water_rate": np.where(is_prod, qw, 0.0), "oil_rate": np.where(is_prod, qo, 0.0)}
        if return_rates:
            return out, rates
        return out

    def source_jac(self, p, T, S):
        """d s / d(p,T,S) per entry by complex step (exact to rounding; branches frozen)."""
        hstep = 1e-30
        nent = len(self.src["cell"])
        J = np.zeros((self.b, self.b, nent))
        args = [p.astype(complex), T.astype(complex), None if S is None else S.astype(complex)]
        for c in range(self.b):
            a2 = [None if x is None else x.copy() for x in args]
            a2[c] = a2[c] + 1j * hstep
            J[:, c, :] = self.source_terms(*a2).imag / hstep
        return J

    # ---------------------------------------------------------------- Jacobian
    def jacobian(self, u, want_schur=False):
        """Exact dF/du with upwind/limiter branches frozen.

        Returns J of shape (7, b, b, n2, n1, n0) (and, if want_schur, the temperature
        convection-diffusion operator S~ of shape (7, n2, n1, n0), preconditioners.py:165-333)."""
        u = self.as_fields(u)
        p, T, S = self.split(u)
        pr = self.props(p, T, S)
        b = self.b
        prm = self.prm
        J = np.zeros((7, b, b) + self.shape)
        Sm = np.zeros((7,) + self.shape) if want_schur else None
        w = [self.w0, 1.0, self.w2]
        Vdt = self.V / self.dt
        phi = self.phi
        rock = (1 - phi) * prm["rho_r"] * prm["c_r"]
        # accumulation derivatives
        ro, ro_p, ro_T = pr["ro"]
        if self.nph == 2:
            rw, rw_p, rw_T = pr["rw"]
            cw, co = prm["c_v_w"], prm["c_v_o"]
            Mw = (phi * rw * (1 - S), phi * rw_p * (1 - S), phi * rw_T * (1 - S), -phi * rw)
            Mo = (phi * ro * S, phi * ro_p * S, phi * ro_T * S, phi * ro)
            for c in range(3):
                e0 = cw * Mw[c + 1] + co * Mo[c + 1]
                J[0, 0, c] += w[0] * e0 * Vdt
                J[0, 1, c] += e0 * T * Vdt
                J[0, 2, c] += w[2] * Mo[c + 1] * Vdt
            J[0, 1, 1] += (cw * Mw[0] + co * Mo[0] + rock) * Vdt
            if want_schur:   # preconditioners.py:235,250
                Sm[0] += (phi * co * S * ro + phi * cw * (1 - S) * rw + rock) * Vdt
        else:
            cv = prm["c_v_o"]
            Mo = (phi * ro, phi * ro_p, phi * ro_T)
            for c in range(2):
                J[0, 0, c] += w[0] * Mo[c + 1] * Vdt
                J[0, 1, c] += cv * Mo[c + 1] * T * Vdt
            J[0, 1, 1] += (cv * Mo[0] + rock) * Vdt
            if want_schur:   # preconditioners.py:73,85
                Sm[0] += (phi * cv * ro + rock) * Vdt
        # faces
        for a in range(3):
            if self.n[a] == 1:
                continue
            lo, hi = _lo(a), _hi(a)
            TK = self.TK[a][lo]
            fs = (b, b) + TK.shape
            dP = np.zeros(fs)
            dM = np.zeros(fs)
            sP = np.zeros(TK.shape)   # S~ entries: row P / col P, row P / col M
            sM = np.zeros(TK.shape)
            gam = prm["g"] * self.h[a] * 0.5 if a == self.gaxis else 0.0
            for (Lk, rk, ce, c0, to2) in self.phases():
                rho = pr[rk]
                Phi = self._phi_face(a, p, rho[0])
                up = Phi > 0.0
                Lt = pr[Lk]
                L = np.where(up, Lt[0][lo], Lt[0][hi])
                Tu = np.where(up, T[lo], T[hi])
                F = TK * L * Phi
                # dPhi/du on both sides (p, T; S does not enter)
                dPhiP = [1.0 - gam * rho[1][lo], -gam * rho[2][lo], 0.0]
                dPhiM = [-1.0 - gam * rho[1][hi], -gam * rho[2][hi], 0.0]
                for c in range(b):
                    dLP = np.where(up, np.broadcast_to(Lt[c + 1], self.shape)[lo], 0.0)
                    dLM = np.where(up, 0.0, np.broadcast_to(Lt[c + 1], self.shape)[hi])
                    dFP = TK * (L * dPhiP[c] + dLP * Phi)
                    dFM = TK * (L * dPhiM[c] + dLM * Phi)
                    dP[0, c] += w[0] * c0 * dFP
                    dM[0, c] += w[0] * c0 * dFM
                    dP[1, c] += ce * Tu * dFP
                    dM[1, c] += ce * Tu * dFM
                    if to2:
                        dP[2, c] += w[2] * dFP
                        dM[2, c] += w[2] * dFM
                dP[1, 1] += np.where(up, ce * F, 0.0)
                dM[1, 1] += np.where(up, 0.0, ce * F)
                if want_schur:
                    sP += np.where(up, ce * F, 0.0)
                    sM += np.where(up, 0.0, ce * F)
            kT, kT_S = pr["kT"]
            kP, kM = kT[lo], kT[hi]
            Hk = harmonic(kP, kM)
            Gk = self.G[a]
            dP[1, 1] += Hk * Gk
            dM[1, 1] -= Hk * Gk
            if want_schur:
                sP += Hk * Gk
                sM -= Hk * Gk
            if self.nph == 2:
                s2 = kP + kM
                with np.errstate(divide="ignore", invalid="ignore"):
                    dHP = np.where(s2 > 0, 2 * kM ** 2 / np.where(s2 == 0, 1, s2) ** 2, 0.0)
                    dHM = np.where(s2 > 0, 2 * kP ** 2 / np.where(s2 == 0, 1, s2) ** 2, 0.0)
                dT = T[lo] - T[hi]
                dP[1, 2] += Gk * dT * dHP * np.broadcast_to(kT_S, self.shape)[lo]
                dM[1, 2] += Gk * dT * dHM * np.broadcast_to(kT_S, self.shape)[hi]
            for r in range(b):
                for c in range(b):
                    J[0, r, c][lo] += dP[r, c]
                    J[2 + 2 * a, r, c][lo] += dM[r, c]
                    J[0, r, c][hi] -= dM[r, c]
                    J[1 + 2 * a, r, c][hi] -= dP[r, c]
            if want_schur:
                Sm[0][lo] += sP
                Sm[2 + 2 * a][lo] += sM
                Sm[0][hi] -= sM
                Sm[1 + 2 * a][hi] -= sP
        # sources
        if self.src is not None:
            cidx = self.src["cell"]
            pc, Tc = p.reshape(-1)[cidx], T.reshape(-1)[cidx]
            Sc = S.reshape(-1)[cidx] if S is not None else None
            dS = self.source_jac(pc, Tc, Sc)
            Jd = J[0].reshape(b, b, -1)
            for r in range(b):
                for c in range(b):
                    np.subtract.at(Jd[r, c], cidx, dS[r, c])
            if want_schur:
                # producers: a -= (rho_w q_w c_w + rho_o q_o c_o) T r delta ; heaters: a -= delta U (-T) r
                # (preconditioners.py:102-108, 269-276)
                sd = self.schur_source_diag(pc, Tc, Sc)
                np.subtract.at(Sm[0].reshape(-1), cidx, sd)
        if want_schur:
            return J, Sm
        return J

    def schur_source_diag(self, p, T, S):
        prm = self.prm
        kind, wt = self.src["kind"], self.src["wt"]
        _, rates = self.source_terms(p, T, S, return_rates=True)
        ro, _, _ = cl.oil_rho(p, T, prm["API"])
        if self.nph == 2:
            rw, _, _ = cl.water_rho(p, T)
            prod = (rw * rates["water_rate"] * prm["c_v_w"] + ro * rates["oil_rate"] * prm["c_v_o"])
        else:
            prod = ro * rates["rate"] * prm["c_v_o"]
        return np.where(kind == PROD, prod, np.where(kind == HEATER, -prm["U"], 0.0)) * wt

    # ---------------------------------------------------------------- checker
    def jacobian_complex_step(self, u):
        """Independent Jacobian by complex-step differentiation of residual() with a
        distance-2 colouring of the 7-point stencil ((i0 + 2 i1 + 3 i2) mod 7)."""
        u = self.as_fields(u).astype(float)
        b = self.b
        n0, n1, n2 = self.n
        i2, i1, i0 = np.meshgrid(np.arange(n2), np.arange(n1), np.arange(n0), indexing="ij")
        col = (i0 + 2 * i1 + 3 * i2) % 7
        J = np.zeros((7, b, b) + self.shape)
        hstep = 1e-30
        offs = [(0, 0, 0), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
        for c in range(b):
            for k in range(7):
                uc = u.astype(complex)
                uc[c][col == k] += 1j * hstep
                dR = self.residual(uc).imag / hstep     # (b, ...)
                for s, (d0, d1, d2) in enumerate(offs):
                    # row cell i gets column of neighbour j = i + off if colour(j) == k
                    j0, j1, j2 = i0 + d0, i1 + d1, i2 + d2
                    ok = (j0 >= 0) & (j0 < n0) & (j1 >= 0) & (j1 < n1) & (j2 >= 0) & (j2 < n2)
                    cj = (j0 + 2 * j1 + 3 * j2) % 7
                    m = ok & (cj == k)
                    for r in range(b):
                        J[s, r, c][m] = dR[r][m]
        return J

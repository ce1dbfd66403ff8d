"""Closure laws of the reference, restated with analytic derivatives (oracle; test infrastructure).

Follows /root/reference/thermalporous/physicalparameters.py:
  oil_rho   :37-46    oil_mu   :48-57
  water_rho :69-82    water_mu :84-90
  rel_perm_o:92-94    rel_perm_w:96-98
Units as in the reference: p in MPa, T in K, viscosity in Pa*s, density kg/m^3.
All functions accept real or complex numpy arrays (complex is used by the
complex-step Jacobian checker in tests).
"""
import numpy as np

# --- oil density (physicalparameters.py:37-46) -------------------------------
_OIL_C = 5.5e-5          # compressibility 1/bar            (:41)
_OIL_P0 = 1.01325        # reference pressure, bar          (:42)
_OIL_E1 = 2.5e-4         # thermal expansivity 1/K          (:43)
_OIL_T0 = 15.5556 + 273.15  #                              (:44)


def oil_rho_ref(API):
    SG = 141.5 / (API + 131.5)       # (:39)
    return SG * 999.0                # (:40)


def oil_rho(p, T, API):
    """rho_ref*e**(c*(10p-p0))*e**(-e1*(T-T0)); returns (rho, drho/dp, drho/dT)."""
    rho_ref = oil_rho_ref(API)
    pbar = p * 1e1                   # (:45)
    rho = rho_ref * np.exp(_OIL_C * (pbar - _OIL_P0)) * np.exp(-_OIL_E1 * (T - _OIL_T0))  # (:46)
    return rho, (10.0 * _OIL_C) * rho, (-_OIL_E1) * rho


# --- oil viscosity, Bennison (physicalparameters.py:48-57) -------------------
_A1, _A2, _A3, _A4 = -0.8021, 23.8765, 0.31458, -9.21592


def oil_mu(T, API):
    """1e-3*10**(A1*API+A2)*Tf**(A3*API+A4), Tf = 1.8(T-273.15)+32; returns (mu, dmu/dT)."""
    Tf = 1.8 * (T - 273.15) + 32.0   # (:56)
    ex = _A3 * API + _A4
    mu = 1e-3 * (10.0 ** (_A1 * API + _A2)) * Tf ** ex   # (:57)
    return mu, mu * ex * 1.8 / Tf


# --- water density, Trangenstein/Kell (physicalparameters.py:69-82) ----------
_E = (999.83952, 16.955176, -7.987e-3, -46.170461e-6, 105.56302e-9, -280.54353e-12)
_E6, _E7, _CW = 16.87985e-3, 10.2, 3.98854e-4


def water_rho(p, T):
    """(E0+..+E5 Tc^5)*e**(Cw(p-E7))/(1+E6 Tc), Tc = T-272.15 (sic, :80); returns (rho, d/dp, d/dT)."""
    Tc = T - 272.15                  # (:80) -- 272.15, not 273.15: preserved quirk
    P = _E[0] + Tc * (_E[1] + Tc * (_E[2] + Tc * (_E[3] + Tc * (_E[4] + Tc * _E[5]))))
    dP = _E[1] + Tc * (2 * _E[2] + Tc * (3 * _E[3] + Tc * (4 * _E[4] + Tc * 5 * _E[5])))
    den = 1.0 + _E6 * Tc
    ex = np.exp(_CW * (p - _E7))
    rho = P * ex / den               # (:82)
    return rho, _CW * rho, (dP - P * _E6 / den) * ex / den


# --- water viscosity, Grabowski (physicalparameters.py:84-90) ----------------
_AW, _BW, _CWM = 2.1850, 0.04012, 5.1547e-6


def water_mu(T):
    """1e-3*Aw/(-1+Bw Tf+Cw Tf^2), Tf = 1.8(T-272.15)+32 (:89); returns (mu, dmu/dT)."""
    Tf = 1.8 * (T - 272.15) + 32.0
    den = -1.0 + _BW * Tf + _CWM * Tf * Tf
    mu = 1e-3 * _AW / den            # (:90)
    return mu, -mu * (_BW + 2.0 * _CWM * Tf) * 1.8 / den


def rel_perm_o(S_o):                 # (:92-94)
    return S_o


def rel_perm_w(S_o):                 # (:96-98)
    return 1.0 - S_o

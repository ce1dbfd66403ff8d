"""Regenerates closure_kats.json: known-answer values of the closure laws, recomputed with plain
`math` from a line-by-line restatement of the reference's formulas
(/root/reference/thermalporous/physicalparameters.py:37-90, wellcase.py:182-191, twophase.py:142-147).

NOT outputs of the reference: the reference cannot be imported here (its modules start with
`from firedrake import *` and Firedrake is not installed), and it ships no golden vectors.  These
values pin the oracle against an independent scalar evaluation of the same formulas; they are the
values quoted in SURVEY.md section 8c.
"""
import json
import math
import os


def oil_rho(p, T, API=10.0):
    SG = 141.5/(API + 131.5)
    rho_ref = SG*999.0
    return rho_ref*math.e**(5.5e-5*(p*1e1 - 1.01325))*math.e**(-2.5e-4*(T - (15.5556 + 273.15)))


def oil_mu(T, API=10.0):
    Tf = 1.8*(T - 273.15) + 32.0
    return 1E-3*(10.0**(-0.8021*API + 23.8765)*Tf**(0.31458*API + -9.21592))


def water_rho(p, T):
    E = (999.83952, 16.955176, -7.987E-3, -46.170461E-6, 105.56302E-9, -280.54353E-12)
    Tc = T - 272.15
    return (E[0] + E[1]*Tc + E[2]*Tc**2 + E[3]*Tc**3 + E[4]*Tc**4 + E[5]*Tc**5)*math.e**(3.98854E-4*(p - 10.2))/(1 + 16.87985E-3*Tc)


def water_mu(T):
    Tf = 1.8*(T - 272.15) + 32
    return 1E-3*2.1850/(-1 + 0.04012*Tf + 5.1547E-6*Tf**2)


def main():
    pts = [(41.369, 288.706), (41.369, 422.039), (68.95, 373.15), (27.579, 320.0)]
    out = {"points": [{"p": p, "T": T, "oil_rho": oil_rho(p, T), "oil_mu": oil_mu(T), "water_rho": water_rho(p, T),
                       "water_mu": water_mu(T)} for p, T in pts]}
    K = 3e-7
    ro = 0.28*((K/K)**0.5*25.0 + (K/K)**0.5*25.0)**0.5/((K/K)**0.25 + (K/K)**0.25)
    out["peaceman"] = {"K": K, "ro": ro, "WI": 2*math.pi*5.0*K/math.log(ro/0.1)}
    out["kT_homogeneous"] = 0.2*0.15 + 0.8*1.7295772056
    out["weights"] = {"p_weight": 288.706, "o_weight_So1": 288.706*(4181.3*0.0 + 2093.4*1.0),
                      "o_weight_So09": 288.706*(4181.3*0.1 + 2093.4*0.9)}
    with open(os.path.join(os.path.dirname(__file__), "closure_kats.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

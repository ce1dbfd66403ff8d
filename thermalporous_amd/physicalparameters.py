"""Physical constants and closure laws (host-side mirror of the reference class).

Mirrors /root/reference/thermalporous/physicalparameters.py:3-98 attribute for attribute
(same names, same SI/MPa/mm^2 scaling), so driver scripts that mutate ``params.rate`` etc.
keep working.  The closure methods here are numpy restatements used on the host only for
diagnostics (oil in place, well rates); the device evaluates its own copy inside the HIP
assembly kernels (csrc/tp_closures.hpp).  Brooks-Corey / capillary-pressure members of the
reference (:100-152) are dead code behind ``if False:`` (twophase.py:103,277) and are not
carried over.
"""
import numpy as np


class PhysicalParameters():
    ko = 0.15              # conductivity of oil in W/m*K            (:9)
    kw = 0.6005638         # conductivity of water                   (:10)
    kr = 1.7295772056      # conductivity of rock (SPE4)             (:12)
    c_v_w = 4181.3         # specific heat of water J/(K*kg)         (:13)
    c_v_o = 2093.4         # specific heat of oil                    (:14)
    c_r = 920.0            # specific heat of sandstone              (:15)
    rho_r = 2650.0         # density of sandstone                    (:16)
    p_inj = 6.895e7*1e-6   # injection bhp, MPa                      (:17)
    p_prod = 2.7579e7*1e-6  # production bhp, MPa                    (:18)
    T_inj = 422.039        # 300F                                    (:20)
    T_prod = 288.706       # 60F                                     (:23)
    API = 10.0             #                                         (:24)
    p_ref = 4.1369e7*1e-6  # SPE10 reference pressure, MPa           (:25)
    T_ref = (T_inj+T_prod)/2.0
    g = 9.80665*1e-6       # gravity in MPa-consistent units         (:27)
    S_o = 1.0              # default initial oil saturation          (:28)
    U = 5.44409e6          # heater coefficient J/(s*K)              (:29)
    rate = 1.8e-3          # max inj/prod rate m^3/s                 (:30)
    well_radius = 0.1      #                                         (:35)

    def oil_rho(self, p, T):                                   # (:37-46)
        SG = 141.5/(self.API + 131.5)
        rho_ref = SG*999.0
        c = 5.5e-5
        p0 = 1.01325
        e1 = 2.5e-4
        T0 = 15.5556 + 273.15
        pbar = p*1e1
        return rho_ref*np.exp(c*(pbar-p0))*np.exp(-e1*(T-T0))

    def oil_mu(self, T):                                       # (:48-57)
        A1, A2, A3, A4 = -0.8021, 23.8765, 0.31458, -9.21592
        Tf = 1.8*(T - 273.15) + 32.0
        return 1E-3*(10.0**(A1*self.API + A2) * Tf**(A3*self.API + A4))

    def water_rho(self, p, T):                                 # (:69-82)
        E_0, E_1, E_2, E_3 = 999.83952, 16.955176, -7.987E-3, -46.170461E-6
        E_4, E_5, E_6, E_7 = 105.56302E-9, -280.54353E-12, 16.87985E-3, 10.2
        Cw = 3.98854E-4
        Tc = T - 272.15
        return (E_0 + E_1*Tc + E_2*Tc**2 + E_3*Tc**3 + E_4*Tc**4 + E_5*Tc**5)*np.exp(Cw*(p-E_7))/(1 + E_6*Tc)

    def water_mu(self, T):                                     # (:84-90)
        Aw, Bw, Cw = 2.1850, 0.04012, 5.1547E-6
        Tf = 1.8*(T - 272.15) + 32
        return 1E-3*Aw/(-1 + Bw*Tf + Cw*Tf**2)

    def rel_perm_o(self, S_o):                                 # (:92-94)
        return S_o

    def rel_perm_w(self, S_o):                                 # (:96-98)
        return 1.0 - S_o

    def as_dict(self):
        """Scalar parameters handed to the compute engine (tp_set_params)."""
        keys = ("ko", "kw", "kr", "c_v_w", "c_v_o", "c_r", "rho_r", "p_inj", "p_prod", "T_inj",
                "T_prod", "API", "p_ref", "g", "S_o", "U", "rate")
        return {k: float(getattr(self, k)) for k in keys}

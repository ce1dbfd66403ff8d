"""Problem builders shared by the CPU and GPU tests (synthetic data only; nothing is read from
/root/reference at run time)."""
import numpy as np

from thermalporous_amd.physicalparameters import PhysicalParameters
from thermalporous_amd.homogeneousgeo import HomogeneousGeo
from thermalporous_amd.homogeneousboxgeo import HomogeneousBoxGeo
from thermalporous_amd.SPE10model import SPE10Model
from thermalporous_amd.SPE10model3D import SPE10Model3D
from thermalporous_amd.wellcase import WellCase
from thermalporous_amd.wellheatercase import WellHeaterCase
from thermalporous_amd.problem import build_spec


def uniform_state(spec, p, T, S=None):
    sh = spec["phi"].shape
    u = [np.full(sh, p), np.full(sh, T)]
    if spec["nphase"] == 2:
        u.append(np.full(sh, S))
    return np.array(u)


def perturbed_state(spec, seed=0, amp=1.0):
    """A generic (non-equilibrium) state so that every upwind branch and derivative is exercised."""
    rng = np.random.default_rng(seed)
    sh = spec["phi"].shape
    u = [41.369 + 2.0*amp*rng.standard_normal(sh), 320.0 + 30.0*amp*rng.random(sh)]
    if spec["nphase"] == 2:
        u.append(np.clip(0.5 + 0.3*amp*rng.standard_normal(sh), 0.02, 0.98))
    return np.array(u)


def c1_homogeneous(N=16, nphase=1, constant_rate=True):
    """BASELINE config 1: tests/test_homo_wells.py of the reference (N x N, L=20, rate 1e-6, T_prod 320)."""
    p = PhysicalParameters()
    p.rate = 1e-6
    p.T_prod = 320.0
    if nphase == 2:
        p.S_o = 0.9
        p.T_inj = 373.15
    g = HomogeneousGeo(N, N, p, 20., 20.)
    c = WellCase(p, g, well_case="test0", constant_rate=constant_rate)
    spec = build_spec(g, c, p, nphase)
    return spec, uniform_state(spec, p.p_ref, p.T_prod, p.S_o), p, g, c


def c3_spe10_2d(Nx=20, Ny=30, nphase=2):
    """BASELINE configs 2/3 (reduced size): SPE10-like 2-D layer, Peaceman wells."""
    p = PhysicalParameters()
    p.rate = 2e-4 if nphase == 2 else 1e-3
    if nphase == 2:
        p.S_o = 0.9
    g = SPE10Model(Nx, Ny, p)
    L, Ly = g.Length, g.Length_y
    c = WellCase(p, g, prod_points=[[140.0/365.76*L, 210.0/670.56*Ly]], inj_points=[[265.0/365.76*L, 260.0/670.56*Ly]])
    spec = build_spec(g, c, p, nphase)
    return spec, uniform_state(spec, p.p_ref, p.T_prod, p.S_o), p, g, c


def c4_spe10_3d(Nx=12, Ny=22, Nz=10, nphase=2, homogeneous=False):
    """BASELINE config 4 (reduced size unless 60,220,85): two-phase 3-D, wells + heaters, gravity."""
    p = PhysicalParameters()
    p.rate = 2e-4
    p.S_o = 0.9
    p.T_inj = 373.15
    if homogeneous:
        g = HomogeneousBoxGeo(Nx, Ny, Nz, p, Length=Nx*6.096, Length_y=Ny*3.048, Length_z=Nz*0.6096)
    else:
        g = SPE10Model3D(Nx, Ny, Nz, p)
    L, Ly, Lz = g.Length, g.Length_y, g.Length_z
    prod = [[140.0/365.76*L, 210.0/670.56*Ly, 0.2*Lz]]
    inj = [[265.0/365.76*L, 260.0/670.56*Ly, 0.8*Lz]]
    c = WellHeaterCase(p, g, prod_points=prod, inj_points=inj)
    spec = build_spec(g, c, p, nphase)
    return spec, uniform_state(spec, p.p_ref, p.T_prod, p.S_o), p, g, c

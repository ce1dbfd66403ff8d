"""oracle.cport -- ctypes binding of the C++/OpenMP restatement (tp_cport.cpp).  TEST INFRASTRUCTURE ONLY.

Same interface as oracle.engine.OracleEngine / thermalporous_amd.engine.HipEngine.  Two users:
  * tests/: a second, faster checker (and the only one that finishes full-size configurations in seconds);
  * bench.py's ``cpu_baseline`` leg: the measured CPU baseline (kind "port"), timed on the GPU box's host cores.
The product (``thermalporous_amd``) never imports this package.  PARITY UNPINNED, like the numpy oracle it
mirrors function for function (oracle/__init__.py): the reference cannot run here and holds no fixtures.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_DIR, "libtp_cport.so")
_LIB = None

_PRM = ("ko", "kw", "kr", "c_v_w", "c_v_o", "c_r", "rho_r", "p_inj", "p_prod", "T_inj", "T_prod", "API", "p_ref", "g",
        "S_o", "U", "rate")
_PC = {"cpr": 0, "cptr": 1, "fieldsplit_cd": 2, "bilu": 4}
_DECOUP = {"No": 0, "QI": 1, "TI": 2, "QI_temp": 3, "TI_temp": 4}


class _Opts(C.Structure):
    _fields_ = [("pc", C.c_int32), ("decoup", C.c_int32), ("ksp_rtol", C.c_double), ("ksp_atol", C.c_double),
                ("ksp_max_it", C.c_int32), ("ksp_restart", C.c_int32), ("snes_rtol", C.c_double),
                ("snes_atol", C.c_double), ("snes_stol", C.c_double), ("snes_max_it", C.c_int32),
                ("amg_omega", C.c_double), ("amg_nu", C.c_int32), ("amg_min_cells", C.c_int32),
                ("amg_full_levels", C.c_int32), ("amg_coarse_pre", C.c_int32), ("amg_coarse_post", C.c_int32),
                ("amg_mid_skip", C.c_int32), ("amg_tail_post", C.c_int32), ("amg_single", C.c_int32),
                ("schur_a11", C.c_int32), ("tile", C.c_int32*3), ("nslabs", C.c_int32), ("amg_dom_tau", C.c_double),
                ("ilu_levels", C.c_int32), ("fs_additive", C.c_int32)]


class _Info(C.Structure):
    _fields_ = [("nits", C.c_int32), ("lits", C.c_int32), ("reason", C.c_int32), ("complete", C.c_int32),
                ("fnorm0", C.c_double), ("fnorm", C.c_double), ("seconds", C.c_double)]


def build():
    """Compile the library in-tree (gcc -O3 -fopenmp); called by __graft_entry__.build() and lazily by load()."""
    subprocess.check_call(["make", "-C", _DIR, "-s"])


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIBPATH) or os.path.getmtime(_LIBPATH) < os.path.getmtime(os.path.join(_DIR, "tp_cport.cpp")):
            build()
        # pin the OpenMP threads (read by libgomp when it is first loaded): unbound threads migrate between cores and a
        # parallel region then costs tens of milliseconds on shared hosts -- measured 59 ms vs 5 ms for a 1M-cell loop
        os.environ.setdefault("OMP_PROC_BIND", "true")
        lib = C.CDLL(_LIBPATH)
        lib.cp_create.restype = C.c_void_p
        lib.cp_max_threads.restype = C.c_int
        lib.cp_fgmres.restype = C.c_int
        lib.cp_amg_levels.restype = C.c_int
        lib.cp_ntiles.restype = C.c_int
        lib.cp_amg_trunc.restype = C.c_int
        # a stale library (older checkout, other struct layout) must not be compared against silently
        try:
            sizes = (lib.cp_sizeof_opts(), lib.cp_sizeof_info())
        except AttributeError:
            sizes = None
        if sizes != (C.sizeof(_Opts), C.sizeof(_Info)):
            raise RuntimeError("oracle/cport/libtp_cport.so does not match this checkout's ABI (%r vs %r): run `make -C "
                               "oracle/cport clean all`" % (sizes, (C.sizeof(_Opts), C.sizeof(_Info))))
        _LIB = lib
    return _LIB


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _usable_cpus():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


# measured at import time: once libgomp is loaded with OMP_PROC_BIND=true the calling thread is pinned to ONE core and
# sched_getaffinity would report 1
_NCPU = _usable_cpus()


def default_threads():
    """Threads for the OpenMP loops: the CPUs this process may run on, at most 16 (the CPU share of a one-GPU box:
    os.cpu_count() there reports the whole host, and 100+ pinned threads on a 16-CPU quota crawl); TP_CPU_THREADS
    overrides."""
    return int(os.environ.get("TP_CPU_THREADS", min(_NCPU, 16)))


class CPortEngine:
    def __init__(self, spec, opts=None):
        from ..engine import DEFAULT_OPTS
        self.lib = load()
        self.spec = spec
        self.opts = dict(DEFAULT_OPTS)
        self.opts.update(opts or {})
        o = self.opts
        if o.get("bjacobi_blocks") is not None:
            from ..engine import blocks_to_tile
            o["ilu_tile"] = blocks_to_tile(spec["n"], o["bjacobi_blocks"])
        if o.get("ilu_whole"):                   # one bjacobi block per slab: ILU(0) of the whole slab
            o["ilu_tile"] = (1 << 30, 1 << 30, 1 << 30)
        if o.get("ilu_tile") is None:
            from ..engine import default_ilu_tile
            o["ilu_tile"] = default_ilu_tile(spec["n"], nslabs=int(o.get("nslabs", 1)))
        self.nph = int(spec["nphase"])
        self.b = self.nph + 1
        n = tuple(int(v) for v in spec["n"])
        self.n = n
        self.shape = (n[2], n[1], n[0])
        self.N = n[0]*n[1]*n[2]
        f = lambda x: np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=float), self.shape)).reshape(-1).copy()
        src = spec.get("sources") or {}
        nsrc = len(src["cell"]) if len(src) else 0
        g = lambda k, dt: np.ascontiguousarray(np.asarray(src[k], dtype=dt)) if nsrc else np.zeros(1, dtype=dt)
        cell, kind, cst = g("cell", np.int64), g("kind", np.int32), g("const", np.int32)
        wt, bhp, qmax, WI = (g(k, np.float64) for k in ("wt", "bhp", "max_rate", "WI"))
        t = [int(min(v, 1 << 30)) for v in o["ilu_tile"]]
        neg = lambda v: -1 if v is None else int(v)
        self._o = _Opts(_PC[o["pc"]], _DECOUP[o["decoup"]], o["ksp_rtol"], o["ksp_atol"], o["ksp_max_it"], o["ksp_restart"],
                        o["snes_rtol"], o["snes_atol"], o["snes_stol"], o["snes_max_it"], o["amg_omega"], o["amg_nu"],
                        o["amg_min_cells"], int(o.get("amg_full_levels", 99)), neg(o.get("amg_coarse_pre")),
                        neg(o.get("amg_coarse_post")), int(bool(o.get("amg_mid_skip", False))), neg(o.get("amg_tail_post")),
                        int(bool(o.get("amg_single", False))), 2 if o.get("schur_selfp") else int(bool(o.get("schur_a11", False))), (C.c_int32*3)(*t),
                        int(o.get("nslabs", 1)), float(o.get("amg_dom_tau", 0.0)), int(o.get("ilu_levels", 0)), int(bool(o.get("fs_additive", False))))
        prm = np.array([float(spec["prm"][k]) for k in _PRM])
        kT = spec.get("kT")
        self.ctx = C.c_void_p(self.lib.cp_create(
            self.nph, (C.c_int*3)(*n), (C.c_double*3)(*[float(h) for h in spec["h"]]), int(spec["gaxis"]), _d(prm),
            _d(f(spec["phi"])), _d(f(spec["K"][0])), _d(f(spec["K"][1])), _d(f(spec["K"][2])),
            _d(f(kT if kT is not None else 0.0)), nsrc, cell.ctypes.data_as(C.POINTER(C.c_int64)),
            kind.ctypes.data_as(C.POINTER(C.c_int32)), cst.ctypes.data_as(C.POINTER(C.c_int32)), _d(wt), _d(bhp), _d(qmax),
            _d(WI), C.byref(self._o)))
        self.set_threads(default_threads())
        self.last = {}

    def set_threads(self, n):
        self.lib.cp_set_threads(int(n))

    def _vec(self, x):
        return np.ascontiguousarray(np.asarray(x, dtype=float).reshape(self.b, self.N))

    # state ------------------------------------------------------------------------------
    def set_state(self, u):
        self.lib.cp_set_state(self.ctx, _d(self._vec(u)))

    def get_state(self):
        out = np.empty((self.b,) + self.shape)
        self.lib.cp_get_state(self.ctx, _d(out))
        return out

    def set_old(self, u=None):
        self.lib.cp_set_old(self.ctx, None if u is None else _d(self._vec(u)))

    def set_dt(self, dt):
        self.lib.cp_set_dt(self.ctx, C.c_double(float(dt)))

    def get_old_state(self):
        out = np.empty((self.b,) + self.shape)
        self.lib.cp_get_old(self.ctx, _d(out))
        return out

    def restore_state(self):
        self.lib.cp_restore(self.ctx)

    def saturation_range(self):
        lo, hi = C.c_double(), C.c_double()
        self.lib.cp_sat_range(self.ctx, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def clamp_saturation(self):
        self.lib.cp_clamp(self.ctx)

    def well_rates(self):
        return {}

    # pieces ------------------------------------------------------------------------------
    def residual(self, u=None):
        if u is not None:
            self.set_state(u)
        out = np.empty((self.b,) + self.shape)
        self.lib.cp_residual(self.ctx, _d(out))
        return out

    def jacobian(self, u=None, want_schur=False):
        if u is not None:
            self.set_state(u)
        b = self.b
        J = np.empty((7, b, b) + self.shape)
        if want_schur:
            Sm = np.empty((7,) + self.shape)
            self.lib.cp_jacobian(self.ctx, _d(J), _d(Sm))
            return J, Sm
        self.lib.cp_jacobian(self.ctx, _d(J), None)
        return J

    def pc_setup(self):
        self.lib.cp_pc_setup(self.ctx)

    def _apply(self, fn, x):
        x = self._vec(x)
        y = np.empty((self.b,) + self.shape)
        fn(self.ctx, _d(x), _d(y))
        return y

    def pc_apply(self, x):
        return self._apply(self.lib.cp_pc_apply, x)

    def stage1(self, x):
        return self._apply(self.lib.cp_stage1, x)

    def spmv(self, x):
        return self._apply(self.lib.cp_spmv, x)

    def ilu_solve(self, x):
        return self._apply(self.lib.cp_ilu_solve, x)

    def vcycle(self, which, x):
        x = np.ascontiguousarray(np.asarray(x, dtype=float).reshape(-1))
        y = np.empty(self.shape)
        self.lib.cp_vcycle(self.ctx, int(which), _d(x), _d(y))
        return y

    def fgmres(self, bvec):
        bvec = self._vec(bvec)
        x = np.empty((self.b,) + self.shape)
        its, rn = C.c_int(), C.c_double()
        reason = self.lib.cp_fgmres(self.ctx, _d(bvec), _d(x), C.byref(its), C.byref(rn))
        return x, its.value, reason, rn.value

    def amg_levels(self, which=0):
        return self.lib.cp_amg_levels(self.ctx, which)

    def amg_trunc(self, which=0):
        """Level at which hierarchy `which` ends with relaxation only (amg_dom_tau), -1: full V-cycle."""
        return self.lib.cp_amg_trunc(self.ctx, which)

    def ntiles(self):
        return self.lib.cp_ntiles(self.ctx)

    # the hot path --------------------------------------------------------------------------
    def newton_solve(self, budget_s=0.0):
        """One nonlinear solve.  budget_s > 0: stop BETWEEN Newton iterations once that much time has been spent
        (at least one iteration is always completed); ``complete`` tells whether the solve ran to its end."""
        info = _Info()
        self.lib.cp_newton(self.ctx, C.c_double(float(budget_s)), C.byref(info))
        self.last = dict(nits=info.nits, lits=info.lits, reason=info.reason, fnorm=info.fnorm, fnorm0=info.fnorm0,
                         complete=info.complete, seconds=info.seconds, nits_done=info.nits)
        return self.last

    def close(self):
        if self.ctx:
            self.lib.cp_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

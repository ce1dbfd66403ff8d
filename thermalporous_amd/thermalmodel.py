"""Time-stepping driver: the *caller* of the hot path.

Mirrors /root/reference/thermalporous/thermalmodel.py ``ThermalModel`` (:7-413): same constructor
arguments, same ``solve()`` policy --
  * dt ramp from ``dt_init_fact*maxdt`` when ``small_dt_start`` (:97-102),
  * retry with dt/2 after a ConvergenceError, restoring u <- u_ (:162-181),
  * two-phase: one retry with dt/2 when S_o leaves [-1e-10, 1+1e-10], then clamp to [0,1] (:184-229;
    the reference's ``while`` is effectively an ``if`` because of the ``break`` at :218),
  * SPE10 adaptive dt heuristic on the Newton count (:337-345), clipping to ``end`` (:346-348),
  * per-step wall time of ``solver.solve()`` only (:164-167) and the results-file summary keys
    (:367-408): Newton steps/s = sum(nits)/sum(timings), FGMRES its/s = sum(lits)/sum(timings).
Everything inside ``self.solver.solve()`` runs on the GPU through the C ABI (engine.HipEngine).
``solve()`` = ``start()`` + ``step()`` until the end time + ``finish()``; bench.py drives the
same ``step()`` a fixed number of times.
"""
import os
from datetime import datetime

import numpy as np

from . import exceptions
from .function import Constant, Function
from .problem import build_spec, field_major_to_internal, internal_to_field_major, _flat
from .solver_options import engine_options

DAY = 24.0*3600.0


class _SNES():
    """The getters the time loop reads (thermalmodel.py:327-328)."""

    def __init__(self):
        self.nits = 0
        self.lits = 0
        self.reason = 0

    def getIterationNumber(self):
        return self.nits

    def getLinearSolveIterations(self):
        return self.lits

    def getConvergedReason(self):
        return self.reason


class NonlinearSolver():
    """Stand-in for Firedrake's NonlinearVariationalSolver: ``solve()`` = one Newton solve on the GPU."""

    def __init__(self, model):
        self.model = model
        self.snes = _SNES()

    def solve(self):
        m = self.model
        eng = m.engine
        m.u.flush()                   # no-ops unless the host wrote to the state since the last solve
        m.u_.flush()
        eng.set_dt(float(m.dt))
        info = eng.newton_solve()
        self.snes.nits, self.snes.lits, self.snes.reason = info["nits"], info["lits"], info["reason"]
        m.u.mark_device_result()
        if info["reason"] <= 0:
            raise exceptions.ConvergenceError("Nonlinear solve failed to converge after %d nonlinear iterations "
                                              "(SNES reason %d)" % (info["nits"], info["reason"]))
        return info


class ThermalModel:

    def __init__(self, end=1.0, maxdt=0.005, save=False, n_save=2, small_dt_start=True, checkpointing={},
                 filename="results/results.txt", dt_init_fact=2**(-10), verbosity=True):
        self.maxdt = maxdt
        self.dt_init_fact = dt_init_fact
        self.dt = Constant(maxdt*DAY)
        self.end = end  # in days
        self.verbosity = verbosity
        self.init_variational_form()
        self.init_solver()
        self.checkpointing = {"save": False, "load": False, "savename": "initial", "loadname": "initial"}
        self.checkpointing.update(checkpointing)
        self.filename = filename
        try:
            self.initial_condition = self.case.init_IC(phases=self.name)
        except AttributeError:
            self.initial_condition = self.init_IC_uniform()
        self.f = None

    # ---- what replaces the UFL form: the problem description for the compute engine ---------------
    def init_variational_form(self):
        nphase = 2 if self.name == "Two-phase" else 1
        self.spec = build_spec(self.geo, self.case, self.params, nphase)
        self.ncell = self.geo.Nx*self.geo.Ny*self.geo.Nz
        self.nfields = nphase + 1
        groups = [(0, 1), (2,)] if getattr(self, "vector", False) and nphase == 2 else None
        self.u = Function(self.nfields, self.ncell, groups=groups)
        self.u_ = Function(self.nfields, self.ncell, groups=groups)
        self.F = "DG0/TPFA residual assembled on the device (csrc/tp_assembly.hip)"

    def init_solver(self):
        self.engine_opts = engine_options(self.solver_parameters, self.name, self.decoup, vector=bool(getattr(self, "vector", False)))
        factory = getattr(self, "_engine_factory", None)
        if factory is None:
            from .engine import HipEngine
            factory = HipEngine      # no fallback: raises if the HIP library or the GPU is missing
        if self.comm.size > 1:
            # one process per GPU: this rank's slab of internal axis 2, RCCL bootstrapped through torch.distributed
            from . import parallel
            parallel.init()
            self.engine = factory(self.spec, self.engine_opts, rank=self.comm.rank, nranks=self.comm.size,
                                  comm_bootstrap=parallel.rccl_bootstrap)
        else:
            self.engine = factory(self.spec, self.engine_opts)
        self.u.bind(self, "u")
        self.u_.bind(self, "u_")
        self.solver = NonlinearSolver(self)
        if "ksp_monitor_residuals" in self.solver_parameters and hasattr(self.engine, "set_ksp_monitor"):
            # per-equation norms of ksp.buildResidual() at every Krylov iteration (thermalmodel.py:44-74)
            names = (("Pressure", "Energy", "Oil") if self.name == "Two-phase" else ("Mass", "Energy"))

            def my_monitor(its, rnorm, field_norms):
                if self.comm.rank == 0 and self.verbosity:
                    for nm, v in zip(names, field_norms):
                        print("  --> %s equation residual: %s" % (nm, v))
            self.engine.set_ksp_monitor(my_monitor)

    def _to_internal(self, f):
        return field_major_to_internal(f, self.geo, self.spec["axes"], self.nfields)

    def _from_internal(self, a):
        if self.comm.size > 1:      # engine returns this rank's slab: gather the slabs on every rank
            from . import parallel
            from .engine import slab_range
            gn2 = self.spec["n"][2]
            counts = [hi - lo for lo, hi in (slab_range(gn2, r, self.comm.size) for r in range(self.comm.size))]
            a = parallel.allgather_slabs(np.asarray(a), counts)
        return internal_to_field_major(a, self.geo, self.spec["axes"], self.nfields)

    def _saturation_range(self):
        smin, smax = self.engine.saturation_range()
        if self.comm.size > 1:      # thermalmodel.py:195-199 of the reference: comm.reduce(MAX) + bcast
            from . import parallel
            smin, smax = parallel.allreduce_minmax(smin, smax)
        return smin, smax

    def resultprint(self, *output):
        print(*output)
        if self.f is not None:
            print(*output, file=self.f)

    def _log(self, *a, **k):
        if self.comm.rank == 0 and self.verbosity:
            print(*a, **k)

    # ---- diagnostics (thermalmodel.py:190,232-294) ---------------------------------------------------
    def oil_mass(self):
        u = self.u._read()
        phi = _flat(self.geo.phi, self.geo)
        vol = self.geo.Dx*self.geo.Dy*(self.geo.Dz if self.geo.dim == 3 else 1.0)
        return float(np.sum(phi*u[2]*self.params.oil_rho(u[0], u[1]))*vol)

    def rates(self):
        """Total injection / production (/ water / oil) rates = sum over entries of delta_i|E_i| * rate_i."""
        r = self.engine.well_rates()
        if not r:
            return {}
        src = self.spec["sources"]
        idx = getattr(self.engine, "src_index", np.arange(len(src["cell"])))
        kind, wt = src["kind"][idx], src["wt"][idx]
        out = {"inj": float(np.sum((wt*r["rate"])[kind == 1])), "prod": float(np.sum((wt*r["rate"])[kind == 0]))}
        if "water_rate" in r:
            out["water"] = float(np.sum((wt*r["water_rate"])[kind == 0]))
            out["oil"] = float(np.sum((wt*r["oil_rate"])[kind == 0]))
        return out

    # ---- the time loop -------------------------------------------------------------------------------
    def start(self):
        """Initial condition, dt ramp and counters (thermalmodel.py:84-149)."""
        if self.filename:
            d = os.path.dirname(self.filename)
            if d:
                os.makedirs(d, exist_ok=True)
            self.f = open(self.filename, "a" if self.filename == "results/results.txt" else "w")
        u, u_ = self.u, self.u_
        if self.checkpointing["load"] is True:     # .npz instead of DumbCheckpoint (:87-91)
            self.resultprint("Using as initial solution checkpoint " + self.checkpointing["loadname"])
            chk = np.load(self.checkpointing["loadname"] + ".npz")
            u.assign(chk["solution"])
            u_.assign(chk["solution"])
        else:
            u.assign(self.initial_condition)
            u_.assign(self.initial_condition)
        if self.small_dt_start:
            self.dt.assign(self.dt_init_fact*self.maxdt*DAY)
        self.t = 0.0
        self.i_step = 0
        self.total_lits = 0
        self.total_nits = 0
        self.nits_vec, self.lits_vec, self.dt_vec, self.timings = [], [], [], []
        self.failed_solves = 0
        self._outfiles = None
        if self.save:           # (:113-133) initial fields
            self._write_fields()
        self._log("Solving time-dependent problem")

    def step(self):
        """One time step: the body of the reference's ``while (t < end)`` loop (:151-348)."""
        u, u_ = self.u, self.u_
        end = self.end*DAY
        dt_inj = self.maxdt*DAY
        self.i_step += 1
        i, t = self.i_step, self.t
        self._log("Time: ", t/DAY, " days. Time-step ", i, ". dt size: ", self.dt.values()[0]/DAY, flush=True)
        while True:
            try:
                old_cpu = datetime.now()
                self.solver.solve()
                now_cpu = datetime.now()
                self.timings.append((now_cpu-old_cpu).total_seconds())
            except exceptions.ConvergenceError:
                self.failed_solves += 1
                self.dt.assign(self.dt.values()[0]*0.5)
                self._log("Time: ", t/DAY, " days. Time-step ", i, ". New dt size: ", self.dt.values()[0]/DAY,
                          flush=True)
                u.assign(u_)
                if self.dt.values()[0] < 1e-12*DAY:
                    raise RuntimeError("time step underflow: the nonlinear solve keeps diverging")
                continue
            break

        # making sure 0 <= S_o <= 1 (thermalmodel.py:184-229)
        if self.name == "Two-phase":
            if self.verbosity:
                self._log("Total oil mass in reservoir: ", self.oil_mass(), flush=True)
            # min/max and the clamp run on the device (the reference does them on u.dat.data, :193-229)
            smin, smax = self._saturation_range()
            epsilon = 1e-10
            chop = bool(smax - 1.0 > epsilon or smin < -epsilon)
            if chop:       # single retry: the reference's `while` never re-evaluates (:218 break)
                self._log("------Negative saturation! Chopping time-step---------", flush=True)
                while True:
                    self.dt.assign(self.dt.values()[0]*0.5)
                    self._log("Time: ", t/DAY, " days. Time-step ", i, ". New dt size: ", self.dt.values()[0]/DAY,
                              flush=True)
                    u.assign(u_)
                    try:
                        self.solver.solve()
                    except exceptions.ConvergenceError:
                        # the reference halves dt here AND again at the top of the loop (:211-217): dt/4 per failed re-solve
                        self.failed_solves += 1
                        self.dt.assign(self.dt.values()[0]*0.5)
                        u.assign(u_)
                        if self.dt.values()[0] < 1e-12*DAY:
                            raise RuntimeError("time step underflow: the nonlinear solve keeps diverging")
                        continue
                    break
            self.engine.clamp_saturation()
            u.mark_device_result()

        r = self.rates() if self.verbosity else {}
        if "inj" in r:
            self._log("Total injection rate is ", r["inj"])
        if "water" in r:
            self._log("Total water production rate is ", r["water"])
            self._log("Total oil production rate is ", r["oil"])
        if "prod" in r:
            self._log("Total production rate is ", r["prod"])

        u_.assign(u)
        current_dt = float(self.dt.values()[0])
        self.t += current_dt
        self.dt_vec.append(current_dt)

        current_nits = self.solver.snes.getIterationNumber()
        current_lits = self.solver.snes.getLinearSolveIterations()
        self._log("Nonlinear iterations: ", current_nits)
        self._log("Linear iterations: ", current_lits)
        self.total_nits += current_nits
        self.total_lits += current_lits
        self.nits_vec.append(current_nits)
        self.lits_vec.append(current_lits)
        if self.geo.name.startswith("SPE10"):   # adaptive time-step heuristic (thermalmodel.py:337-345)
            if current_nits < 6:
                factor = 1 + min(1.0, (6 - current_nits)**2/3**2)
                self.dt.assign(min(dt_inj, current_dt*factor))
            elif current_nits > 9:
                factor = 1 - min(1.0, (current_nits - 9)**2/4**2)/2
                self.dt.assign(current_dt*factor)
            else:
                self.dt.assign(current_dt)
        current_dt = self.dt.values()[0]
        if current_dt > end-self.t and self.t < end:
            self.dt.assign(end-self.t)
        if self.save and i % self.n_save == 0:      # (:303-322)
            self._write_fields()
        return current_nits, current_lits

    def _write_fields(self):
        """pressure / temperature / saturation_o .pvd collections next to the results file (:113-133, :303-322).
        Reading the state is a collective on several ranks (slabs are gathered); rank 0 writes."""
        from .output import File
        d = self.u._read()
        fields = [d[f] for f in range(self.nfields)]
        if self.comm.rank != 0:
            return
        if self._outfiles is None:
            d = os.path.dirname(self.filename) if self.filename else "results"
            names = ["pressure", "temperature"] + (["saturation_o"] if self.nfields == 3 else [])
            self._outfiles = [File(os.path.join(d or ".", n + ".pvd")) for n in names]
        for f, out in enumerate(self._outfiles):
            out.write(os.path.basename(out.base), fields[f], self.geo, time=self.t/DAY)

    def finish(self):
        """Checkpoint and results-file summary (thermalmodel.py:361-412)."""
        u = self.u
        t = self.t
        nits_vec, lits_vec, dt_vec, timings = self.nits_vec, self.lits_vec, self.dt_vec, self.timings
        if self.checkpointing["save"] is True:
            np.savez(self.checkpointing["savename"] + ".npz", solution=u._read(), t=t, dt=self.dt.values()[0])
            self.resultprint("Saving checkpoint solution in " + self.checkpointing["savename"])
        dt_counter = len(dt_vec)
        if self.comm.rank == 0 and self.verbosity:
            self.resultprint("nits = ", nits_vec, ";")
            self.resultprint("lits = ", lits_vec, ";")
            self.resultprint("dts = ", dt_vec, ";")
            self.resultprint("timings = ", timings, ";")
            self.resultprint("----------------------------------------------------------------------")
            self.resultprint(self.name, "thermal model")
            self.resultprint("Geo model: ", self.geo.name)
            self.resultprint("Test case: ", self.case.name)
            self.resultprint("Max time-step: ", self.maxdt)
            self.resultprint("Final time: ", t/DAY)
            self.resultprint("Solver Parameters")
            self.resultprint("-----------------")
            for x in self.solver_parameters:
                self.resultprint(x, ':', self.solver_parameters[x])
            self.resultprint(" ")
            self.resultprint("Solver performance")
            self.resultprint("------------------")
            self.resultprint("Total CPU time (s):", sum(timings))
            avg_nitdt = self.total_nits/max(dt_counter, 1)
            avg_litdt = self.total_lits/max(dt_counter, 1)
            self.resultprint("Average Nonlinear iterations per time-step:", avg_nitdt)
            self.resultprint("Average Linear iterations per time-step: ", avg_litdt)
            self.resultprint("Average Linear iteration per Nonlinear iteration: ", avg_litdt/max(avg_nitdt, 1e-300))
            self.resultprint("Total Linear iterations: ", sum(lits_vec))
            self.resultprint("Total Nonlinear iterations: ", sum(nits_vec))
            self.resultprint("Number of time-steps: ", len(dt_vec))
            self.resultprint("Last Nonlinear iterations:", self.solver.snes.getIterationNumber())
            self.resultprint("Last Linear iterations: ", self.solver.snes.getLinearSolveIterations())
            self.resultprint("Newton steps per second: ", self.total_nits/max(sum(timings), 1e-300))
            self.resultprint("FGMRES iterations per second: ", self.total_lits/max(sum(timings), 1e-300))
            self.resultprint("----------------------------------------------------------------------")
            self.resultprint(" ")
        self.last_dt = dt_vec[-1] if dt_vec else None
        if self.f is not None:
            self.f.close()
            self.f = None

    def solve(self):
        self.start()
        while self.t < self.end*DAY:
            self.step()
            if getattr(self, "max_steps", None) and self.i_step >= self.max_steps:
                break
        self.finish()

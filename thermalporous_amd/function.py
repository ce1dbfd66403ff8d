"""Tiny stand-ins for the Firedrake objects the reference's time loop manipulates
(thermalmodel.py:13,93-94,174-180,193,229): ``Constant`` (assign/values) and a mixed ``Function``
whose per-field data is reachable as ``u.dat.data[i]`` in the reference's dof order
(here: flat cell index c = i + Nx*(j + Ny*k), x fastest).

The authoritative copy of the state (u) and of the previous time level (u_) lives in HBM inside
the compute engine.  The host array of a ``Function`` is a cache: refreshed lazily on first access
after a solve, pushed back only if the host wrote to it.  ``u_.assign(u)`` and ``u.assign(u_)`` --
the two copies the time loop performs every step -- are device-to-device copies, so a run that never
touches the state on the host moves no state over PCIe between time steps.
"""
import numpy as np


class Constant():
    def __init__(self, value):
        self._v = float(value)

    def assign(self, value):
        self._v = float(value.values()[0] if isinstance(value, Constant) else value)
        return self

    def values(self):
        return np.array([self._v])

    def __float__(self):
        return self._v


class _Dat():
    def __init__(self, owner):
        self._o = owner

    def _grouped(self, d):
        """The reference's mixed-space view: one array per sub-space.  With ``vector=True`` (twophase.py:19-21) that is
        [pT of shape (ncell, 2), S_o]: the interleaved block is a COPY of the two field planes (read-only: the device
        keeps field planes; S_o, the only entry the time loop writes, stays a live view)."""
        out = []
        for grp in self._o._groups:
            if len(grp) == 1:
                out.append(d[grp[0]])
            else:
                a = np.stack([d[i] for i in grp], axis=1)
                a.setflags(write=False)
                out.append(a)
        return out

    @property
    def data(self):
        d = self._o._data                  # read-write access: device copy becomes stale
        if self._o._groups is not None:
            return self._grouped(d)
        return [d[i] for i in range(d.shape[0])]

    @property
    def data_ro(self):
        d = self._o._read()
        if self._o._groups is not None:
            return self._grouped(d)
        return [d[i] for i in range(d.shape[0])]


class Function():
    """Mixed DQ0 function: ``nfields`` arrays of ncell doubles (field-major, like V*V*V)."""

    def __init__(self, nfields, ncell, name="solution", groups=None):
        self._store = np.zeros((nfields, ncell))
        self._groups = groups          # e.g. [(0, 1), (2,)] for VectorFunctionSpace(dim=2) x V; None: one array per field
        self.dat = _Dat(self)
        self._name = name
        self.host_stale = False         # device holds newer data than _store
        self.dev_stale = True           # host holds newer data than the device
        self._model = None              # owning model (engine + layout conversion)
        self._role = None               # "u" (current state) or "u_" (previous time level)

    def bind(self, model, role):
        self._model, self._role = model, role

    # -- cache protocol ------------------------------------------------------------------------------
    def _read(self):
        if self.host_stale:
            m = self._model
            a = m.engine.get_state() if self._role == "u" else m.engine.get_old_state()
            self._store[...] = m._from_internal(a)
            self.host_stale = False
        return self._store

    @property
    def _data(self):
        d = self._read()
        self.dev_stale = True           # caller may write through the returned view
        return d

    def flush(self):
        """Push host modifications (if any) to the device."""
        if self.dev_stale and self._model is not None:
            a = self._model._to_internal(self._store)
            if self._role == "u":
                self._model.engine.set_state(a)
            else:
                self._model.engine.set_old(a)
            self.dev_stale = False

    def mark_device_result(self):
        self.host_stale, self.dev_stale = True, False

    # -- Firedrake-like surface ------------------------------------------------------------------------
    def assign(self, other):
        if isinstance(other, Function):
            same_engine = self._model is not None and other._model is self._model and self._role != other._role
            if same_engine and not other.dev_stale:
                # device-to-device: u_ <- u (tp_set_old_state(NULL)) or u <- u_ (tp_restore_state)
                if self._role == "u_":
                    self._model.engine.set_old(None)
                else:
                    self._model.engine.restore_state()
                self.dev_stale = False
                if other.host_stale:
                    self.host_stale = True
                else:
                    self._store[...] = other._store
                    self.host_stale = False
                return self
            self._store[...] = other._read()
        else:
            self._store[...] = other
        self.host_stale = False
        self.dev_stale = True
        return self

    def sub(self, i):
        return _Sub(self, i)

    def split(self):
        d = self._read()
        return tuple(d[i] for i in range(d.shape[0]))

    def copy(self):
        f = Function(*self._store.shape)
        f._store[...] = self._read()
        return f


class _Sub():
    def __init__(self, f, i):
        self._f, self._i = f, i

    def assign(self, value):
        self._f._data[self._i][...] = float(value) if isinstance(value, Constant) or np.ndim(value) == 0 else value
        return self

    def vector(self):
        return self._f._data[self._i]

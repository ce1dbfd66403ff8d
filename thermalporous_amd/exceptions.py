"""``ConvergenceError`` -- raised where Firedrake raises ``firedrake.exceptions.ConvergenceError``
(caught by the time loop at thermalmodel.py:170,210 of the reference to halve dt)."""


class ConvergenceError(Exception):
    pass

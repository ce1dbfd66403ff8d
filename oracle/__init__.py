"""CPU oracle for the thermalporous hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain numpy restatement of the reference algorithm
(tlroy/thermalporous): closure laws, DG0/TPFA residual, exact block Jacobian,
CPR/CPTR two-stage preconditioner (decoupling -> aggregation-AMG V-cycle ->
block-Jacobi block-ILU(0)), FGMRES and the Newton loop.  Every function cites
the reference file:line it follows.

PARITY UNPINNED.  The reference is 100 % Python on top of Firedrake/PETSc/hypre,
none of which is installed here (``import firedrake`` raises ModuleNotFoundError),
it ships no golden vectors, no stored outputs and no assertion of any kind
(SURVEY.md section 4, 8c).  Nothing in this oracle has therefore been checked
against an output of the reference itself; it is pinned only by
  * known-answer values recomputed by hand from the reference's formulas
    (tests/golden/closure_kats.json),
  * identities the discrete equations must satisfy (conservation, hydrostatic
    equilibrium, x<->y symmetry), and
  * an independent complex-step differentiation of its own residual.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker.  The product
(``thermalporous_amd``) never imports it and fails loudly without its HIP library.
"""

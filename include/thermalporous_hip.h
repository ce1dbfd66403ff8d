/* thermalporous_hip.h -- C ABI of the MI355X hot path (libthermalporous_hip.so).
 *
 * Drop-in boundary for the reference's hot path (tlroy/thermalporous): everything below
 * `self.solver.solve()` (thermalporous/thermalmodel.py:165), which in the reference is executed by
 * TSFC/PyOP2-generated kernels, PETSc SNES/KSP/PC/Mat and hypre.  The reference is pure Python and
 * reaches that code through petsc4py / Firedrake; the binding a maintainer would add is the ctypes
 * stub shown in INTEGRATION.md (thermalporous_amd/engine.py is that stub).
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = usage/allocation/HIP/RCCL error (text from
 *     tp_last_error), >0 is never returned; solver outcomes are reported through out-parameters
 *     using PETSc's numbering of SNES/KSP converged (>0) / diverged (<0) reasons, so the Python
 *     side can raise ConvergenceError exactly where Firedrake does (thermalmodel.py:170,210).
 *   - all arithmetic is IEEE float64; indices int32/int64.
 *   - "internal" axis order: axis 0 fastest in memory, axis 2 is the slab axis of the 1-D
 *     multi-GPU decomposition.  Cell arrays passed through this API hold the rank's slab WITH
 *     one halo plane on each side along axis 2:  ntot = n0*n1*(n2+2), owned cell (i0,i1,i2) at
 *     i0 + n0*i1 + n0*n1*(i2+1).  Vectors are field-major: b planes of ntot doubles
 *     (p, T[, S_o]) -- the same field-major ordering as the reference's V*V*V mixed space.
 *   - Jacobian storage ("stencil-of-blocks"): 7*b*b planes of ntot doubles,
 *     plane ((s*b + r)*b + c), stencil slot s: 0 diag, 1 -a0, 2 +a0, 3 -a1, 4 +a1, 5 -a2, 6 +a2.
 *   - not thread-safe per context; one host thread drives one context (= one GPU).
 *   - host pointers are never retained after return.
 */
#ifndef THERMALPOROUS_HIP_H
#define THERMALPOROUS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tp_ctx tp_ctx;

/* Grid of this rank's slab.  Replaces geo.Nx/Ny/Nz/Dx/Dy/Dz + the DQ0 space
 * (rectanglegeo.py:28-34,64-65, boxgeo.py:31-44,85-86). */
typedef struct tp_grid {
    int32_t n0, n1, n2;      /* owned cells along internal axes (n2 = this rank's planes)     */
    int32_t gn2, off2;       /* global extent along axis 2 and this rank's first global plane */
    double  h[3];            /* cell sizes along internal axes                                */
    int32_t gaxis;           /* internal axis on which gravity acts (physical z), -1 = none   */
    int32_t nphase;          /* 1: unknowns (p,T)   2: unknowns (p,T,S_o)                     */
    int32_t rank, nranks;    /* slab index / number of slabs (one process per GPU)            */
} tp_grid;

/* physicalparameters.py:9-35 (scalars only; closure laws are compiled into the kernels). */
typedef struct tp_params {
    double ko, kw, kr, c_v_w, c_v_o, c_r, rho_r, p_inj, p_prod, T_inj, T_prod, API, p_ref, g,
           S_o, U, rate;
} tp_params;

/* One per-cell source entry: a well (wellcase.py:78-108,171-266) or heater (heatercase.py:63-77)
 * restricted to one cell; wt = delta_i*|E_i| (sums to 1 over a well).  cell = LOCAL index into
 * the slab-with-halo array. */
typedef struct tp_source {
    int64_t cell;
    int32_t kind;            /* 0 producer, 1 injector, 2 heater */
    int32_t constant_rate;   /* flow_rate_constant variants (wellcase.py:201-202,237-266) */
    double  wt, bhp, max_rate, WI;
} tp_source;

/* Solver options = the subset of the PETSc options dicts the hot path honours
 * (singlephase.py:289-354, twophase.py:416-433,531-597). */
typedef struct tp_options {
    int32_t pc_kind;         /* 0 = pc_cpr (CPRStage1PC + bjacobi/ILU0), 1 = pc_cptr (CPTRStage1PC
                                with fieldsplit Schur FULL, V(App), V(S~)), 2 = pc_fieldsplit_cd
                                (single-phase: fieldsplit Schur FULL on (p,T) with V(App) and the
                                ConvDiffSchurPC V(S~), no second stage; singlephase.py:309-319),
                                3 = pc_cptramg[_QI|_TI] (CPTRStage1PC with ONE system-AMG V-cycle on the 2x2-block
                                (p,T) operator Atilde_00 + bjacobi/ILU0; twophase.py:552-566),
                                4 = pc_bilu (bjacobi + ILU(ilu_levels) alone; twophase.py:758-762, singlephase.py:402-406) */
    int32_t decoup;          /* 0 "No", 1 "QI", 2 "TI", 3 "QI_temp", 4 "TI_temp" (option key sub_0_cpr_decoup;
                                the _temp variants decouple both T and S: two-phase pc_cpr only) */
    double  ksp_rtol, ksp_atol;
    int32_t ksp_max_it, ksp_restart;
    double  snes_rtol, snes_atol, snes_stol;
    int32_t snes_max_it;
    double  amg_omega;       /* damped-Jacobi weight of the AMG smoother */
    int32_t amg_nu;          /* pre/post smoothing sweeps */
    int32_t amg_min_cells;   /* coarsest-grid size (dense solve) */
    int32_t ilu_t1, ilu_t2;  /* bjacobi tile extent along axes 1,2 (t1*t2 <= 64: one wavefront per tile) */
    int32_t ilu_t0;          /* tile extent along axis 0 (<= 0: the whole line) */
    int32_t amg_full_levels; /* V(nu,nu) on the first amg_full_levels levels ... */
    int32_t amg_coarse_pre, amg_coarse_post;  /* ... V(coarse_pre, coarse_post) below (coarse_post >= 1) */
    int32_t amg_mid_skip;    /* 1: every second level between the full levels and the <= 1024-cell levels is a pure
                                transfer level (no smoothing): two coarsening directions per smoothing level there */
    int32_t amg_tail_post;   /* post-sweeps on the levels of <= 1024 cells (they run inside one workgroup, where a
                                sweep costs ~2 us: small grids live entirely there and want the stronger cycle) */
    int32_t amg_single;      /* 1: AMG operators/weights stored in fp32 (vectors and arithmetic stay fp64) */
    int32_t schur_a11;       /* what preconditions the Schur complement of pc_kind 1/2 (pc_fieldsplit_schur_precondition):
                                0 = the convection-diffusion operator S~ (ConvDiffSchurPC / ConvDiffSchurTwoPhasesPC);
                                1 = a11: A_11, the T-T block after decoupling (pc_fieldsplit_a11, pc_cptr_a11);
                                2 = selfp (pc_kind 2, one GPU): Sp = A11 - A10 diag(A00)^-1 A01 (pc_fieldsplit_selfp,
                                    singlephase.py:322-330): V-cycle of the hierarchy of Sp's 7-point collapse (far
                                    entries lumped onto the diagonal) + one damped-Jacobi sweep on the exact Sp */
    int32_t amg_gather_cells;/* multi-GPU: AMG levels with more cells than this stay distributed over the slabs
                                (halo exchange per sweep); smaller ones are gathered and replicated on every
                                rank.  < 0: replicate the whole hierarchy.  Ignored on one GPU. */
    double  amg_dom_tau;     /* relaxation-only truncation: the first V(nu,nu) level whose operator has
                                max_i sum_{j!=i}|a_ij| / |a_ii| <= amg_dom_tau ends the cycle with two damped-Jacobi
                                sweeps (no coarse-grid correction: damped Jacobi already contracts by
                                1 - omega (1 - tau) per sweep there; BoomerAMG's max_row_sum rule for diagonally dominant
                                rows).  Hits the temperature operator S~ of pc_cptr, never the pressure.  0: off. */
    int32_t ilu_levels;      /* stage 2 fill level: 0 = block-ILU(0) (sub_1_sub_pc_factor_levels 0, the presets' default),
                                1 = block-ILU(1) (pc_cprilu1_gmres, twophase.py:653-668): 13-block rows, the sweeps
                                take 4 steps of skew per axis-2 plane and 2 per axis-1 line instead of 1 and 1 */
    int32_t fs_additive;     /* pc_kind 2: 1 = PCFIELDSPLIT additive on (p,T) -- y_p = V(A_pp) x_p, y_T = V(A_TT) x_T, no coupling
                                (pc_fieldsplit_diag, singlephase.py:371-375) -- instead of Schur FULL */
    int32_t ilu_whole;       /* 1: ONE bjacobi block per rank = block-ILU(0) of the whole slab, PETSc's default bjacobi and the
                                reference's `sub_1_pc_bjacobi_blocks: 1` (tests/test_homo_wells.py:112, pc_cptr_a11
                                twophase.py:612): couplings between the tiles are kept; the tiles (ilu_t0 x ilu_t1 x ilu_t2,
                                now only the unit of the sweep) are swept one tile-diagonal T0+T1+T2 per launch.  ILU(0) only. */
} tp_options;

/* Result of one nonlinear solve (SNES iteration number / linear iterations / reason:
 * thermalmodel.py:327-336). */
typedef struct tp_solve_info {
    int32_t nits, lits, reason;     /* reason: SNES numbering; <0 diverged */
    int32_t last_ksp_reason;
    double  fnorm0, fnorm;
    int32_t vcycles;
} tp_solve_info;

const char *tp_last_error(void);
int tp_version(void);

/* lifetime ------------------------------------------------------------------------------------ */
int tp_create(const tp_grid *grid, const tp_params *prm, const tp_options *opt, int device, tp_ctx **out);
int tp_destroy(tp_ctx *ctx);
int tp_set_options(tp_ctx *ctx, const tp_options *opt);
/* multi-GPU: rank 0 calls tp_comm_unique_id, the 128 bytes are broadcast by the host launcher
 * (torch.distributed), then every rank calls tp_comm_init (RCCL ncclCommInitRank). */
int tp_comm_unique_id(void *id128);
int tp_comm_init(tp_ctx *ctx, const void *id128);
/* in-process slab group: N contexts driven by N host threads on ONE GPU exchange through device copies
 * instead of RCCL (same call sequence) -- validates the slab algorithm where only one GPU exists. */
int tp_local_group_create(int32_t nranks, void **group);
int tp_local_group_destroy(void *group);
int tp_comm_init_local(tp_ctx *ctx, void *group);

/* problem data: geo fields (homogeneousgeo.py:13-20, SPE10model*.py) -- arrays of ntot doubles
 * (slab + halo planes); name in {"phi","K0","K1","K2","kT"}.  tp_finalize_fields builds the face
 * transmissibilities H(K)|e|/Delta_h (singlephase.py:98-103). */
int tp_set_field(tp_ctx *ctx, const char *name, const double *host, int64_t n);
int tp_finalize_fields(tp_ctx *ctx);
int tp_set_sources(tp_ctx *ctx, int32_t n, const tp_source *entries);   /* wells/heaters: wellcase.py:78-108,171-266,
                                                                           heatercase.py:63-77, sourceterms.py:77-84,155-269 */

/* state u, old state u_ (thermalmodel.py:93-94,296), time step (thermalmodel.py:13). */
int tp_set_state(tp_ctx *ctx, const double *u_host);      /* b*ntot doubles */
int tp_get_state(tp_ctx *ctx, double *u_host);
int tp_set_old_state(tp_ctx *ctx, const double *u_host);  /* NULL: u_ <- u */
int tp_set_dt(tp_ctx *ctx, double dt);
int tp_get_old_state(tp_ctx *ctx, double *u_host);
int tp_restore_state(tp_ctx *ctx);                          /* u <- u_  (thermalmodel.py:179: u.assign(u_)) */
/* two-phase saturation guard of the time loop (thermalmodel.py:193-229): min/max of S_o over the owned
 * cells of this rank, and the clamp to [0,1]; both on the device. */
int tp_saturation_range(tp_ctx *ctx, double *smin, double *smax);
int tp_clamp_saturation(tp_ctx *ctx);

/* assembly: F(u) and J = dF/du (what TSFC/PyOP2 kernels + MatSetValues do in the reference for the forms
 * singlephase.py:60-273 / twophase.py:67-411 behind self.solver.solve(), thermalmodel.py:165). */
int tp_residual(tp_ctx *ctx, double *norm2);               /* R <- F(u); ||F||_2 over all ranks */
int tp_jacobian(tp_ctx *ctx);                              /* R, J (and S~ for pc_cptr) <- at u */
int tp_get_residual(tp_ctx *ctx, double *host);            /* b*ntot */
int tp_export_jacobian(tp_ctx *ctx, double *host);         /* 7*b*b*ntot */
int tp_export_schur(tp_ctx *ctx, double *host);            /* 7*ntot: ConvDiffSchur*PC operator */
int tp_well_rates(tp_ctx *ctx, double *rate, double *water_rate, double *oil_rate); /* per entry */

/* device vectors (work vectors for the PC plug-in API; ids are small ints) */
int tp_vec_create(tp_ctx *ctx, int32_t *id);
int tp_vec_set(tp_ctx *ctx, int32_t id, const double *host);
int tp_vec_get(tp_ctx *ctx, int32_t id, double *host);
int tp_vec_copy_residual(tp_ctx *ctx, int32_t id);        /* vec <- R */

/* operators (PETSc MatMult AIJ / PCApply in the reference: option dicts singlephase.py:303-354,
 * twophase.py:478-482,531-597) */
/* KSPMonitorSet analogue for the reference's per-field residual monitor (option key ksp_monitor_residuals,
 * thermalmodel.py:44-74: ksp.buildResidual() split by field).  When set, every FGMRES iteration builds the current
 * iterate x_j = Z y_j, the true residual b - J x_j (one extra SpMV) and its 2-norm per field, and calls
 * cb(its, recurrence_rnorm, field_norms[nfields], user).  Debug aid: costs about one Krylov iteration per call.
 * NULL removes the monitor. */
typedef void (*tp_ksp_monitor_fn)(int32_t its, double rnorm, const double *field_norms, int32_t nfields, void *user);
int tp_set_ksp_monitor(tp_ctx *ctx, tp_ksp_monitor_fn cb, void *user);

/* PETSc Vec kernels of one Krylov iteration (SURVEY.md 8b minimum list; KSP fgmres, twophase.py:426-432):
 *   tp_vec_create_batch  n vectors in ONE allocation (ids first..first+n-1): a Krylov basis
 *   tp_vec_dot_batch     out[i] = <v_{first+i}, w>, i < n   (VecMDot: one pass over w, one host sync)
 *   tp_vec_axpy_batch    w += sum_i coef[i] v_{first+i}     (VecMAXPY: one pass over w)
 *   tp_vec_norm2         ||x||_2                            (VecNorm)
 * Reductions run over owned cells only and are summed over the slabs (RCCL all-reduce) in multi-GPU runs. */
int tp_vec_create_batch(tp_ctx *ctx, int32_t n, int32_t *first_id);
int tp_vec_dot_batch(tp_ctx *ctx, int32_t first, int32_t n, int32_t w, double *out);
int tp_vec_axpy_batch(tp_ctx *ctx, int32_t first, int32_t n, const double *coef, int32_t w);
int tp_vec_norm2(tp_ctx *ctx, int32_t x, double *out);
int tp_spmv(tp_ctx *ctx, int32_t x, int32_t y);            /* y = J x */
int tp_pc_setup(tp_ctx *ctx);                              /* PCSetUp: decoupling, AMG setup, ILU factor */
int tp_pc_apply(tp_ctx *ctx, int32_t x, int32_t y);        /* composite multiplicative (stage1, ILU0) */
int tp_stage1_update(tp_ctx *ctx);                         /* CPRStage1PC/CPTRStage1PC.update (preconditioners.py:875,1545) */
int tp_stage1_apply(tp_ctx *ctx, int32_t x, int32_t y);    /* ....apply (preconditioners.py:881,1550) */
int tp_ilu0_factor(tp_ctx *ctx);                           /* sub_1: bjacobi + ILU(0) numeric factorisation (singlephase.py:348-349) */
int tp_ilu0_solve(tp_ctx *ctx, int32_t x, int32_t y);
int tp_amg_setup(tp_ctx *ctx, int32_t which);              /* v_cycle dict (singlephase.py:303-307); 0: pressure operator, 1: S~ */
/* which: 0 pressure hierarchy, 1 S~ hierarchy, 2 the (p,T) system hierarchy of pc_cptramg (fields 0,1 of b -> x) */
int tp_amg_vcycle(tp_ctx *ctx, int32_t which, int32_t field_b, int32_t b, int32_t field_x, int32_t x);
int tp_schur_apply(tp_ctx *ctx, int32_t x, int32_t y);     /* ConvDiffSchur*PC.apply: one V-cycle on S~, field 1 */

/* Krylov / Newton (PETSc KSP fgmres + SNES newtonls in the reference: twophase.py:416-433, singlephase.py:289-301;
 * reasons use PETSc's numbering so that the host raises ConvergenceError where Firedrake does, thermalmodel.py:170) */
int tp_fgmres(tp_ctx *ctx, int32_t b, int32_t x, int32_t *its, int32_t *reason, double *rnorm);
int tp_newton_solve(tp_ctx *ctx, tp_solve_info *info);

/* measurement hooks for bench.py: average device time (ms, HIP events on the context's stream)
 * of `reps` launches of one hot kernel.  which: 0 block SpMV, 1 ILU solve, 2 AMG V-cycle (pressure),
 * 3 assembly (residual+Jacobian), 4 full pc_apply, 5 pc_setup, 6 ILU factorisation, 7 one classical Gram-Schmidt
 * step against 16 basis vectors (VecMDot + VecMAXPY + VecNorm; needs a Krylov basis from an earlier solve). */
int tp_time_kernel(tp_ctx *ctx, int32_t which, int32_t reps, double *ms_avg);
int tp_amg_info(tp_ctx *ctx, int32_t which, int32_t *nlevels, double *op_complexity);
/* level at which hierarchy `which` ends with relaxation only (amg_dom_tau), -1: full V-cycle; ratio0 = the
 * dominance ratio measured on level 0 at the last set-up */
int tp_amg_trunc(tp_ctx *ctx, int32_t which, int32_t *level, double *ratio0);
/* coarsening axis of every level (internal axis numbering, 2 = slab axis) and how many of the top levels are
 * distributed over the slabs (0 on one GPU and when the hierarchy is replicated, see amg_gather_cells);
 * which: 0 pressure, 1 S~, 2 the (p,T) system hierarchy of pc_cptramg */
int tp_amg_layout(tp_ctx *ctx, int32_t which, int32_t *dist_levels, int32_t *axes, int32_t cap, int32_t *naxes);

#ifdef __cplusplus
}
#endif
#endif

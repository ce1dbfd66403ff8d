// Host orchestration of the hot path: PC set-up/apply, FGMRES, Newton.
//
//   pc_setup / pc_apply  = PCSetUp / PCApply of PCCOMPOSITE multiplicative("python,bjacobi")
//                          (singlephase.py:341-351, twophase.py:531-550,582-597) with the python stage
//                          being CPRStage1PC (preconditioners.py:335-906) or CPTRStage1PC (:1243-1571)
//   fgmres               = KSP fgmres, right PC, restart/max_it 200, classical Gram-Schmidt without
//                          refinement (twophase.py:426-432)
//   newton               = SNES newtonls with Firedrake's default `basic` line search
//                          (thermalmodel.py:36-42,165)
// The loops live here (C++) rather than in Python so that one Krylov iteration costs kernel time,
// not interpreter time; the Python PC classes call the same stage functions through the C ABI.
#include "tp_common.hpp"
#include <cmath>
#include <algorithm>
#include <cstdlib>

namespace tp {

// ------------------------------------------------------------------------------------------------
void ensure_work(tp_ctx *c) {
    const size_t nv = (size_t)c->b * c->g.ntot;
    // w3 holds the Schur stage's r0, r1 and t: three planes even for the 2-field single-phase system.
    // Every buffer is tested on its own size: nothing may assume that whoever allocated w1 also sized w3.
    const size_t n3 = std::max(nv, (size_t)3 * c->g.ntot);
    bool grew = false;
    if (c->w1.n < nv) { c->w1.alloc(nv); grew = true; }
    if (c->w2.n < nv) { c->w2.alloc(nv); grew = true; }
    if (c->w3.n < n3) { c->w3.alloc(n3); grew = true; }
    if (c->w4.n < nv) { c->w4.alloc(nv); grew = true; }
    if (c->dx.n < nv) { c->dx.alloc(nv); grew = true; }
    if (grew) c->graph_epoch++;          // captured pc_apply graphs hold the old addresses
}

// multi-GPU scratch on the gathered global grid: grown on demand, never shrunk; captured graphs hold the old address
static void ensure_global(tp_ctx *c, DBuf<double> &b, size_t n) {
    if (b.n >= n) return;
    b.alloc(n);
    c->graph_epoch++;
}

// mean interior-face transmissibility per axis over the GLOBAL grid (sum/count all-reduced over slabs)
static void face_strengths(tp_ctx *c, double st[3]) {
    const long nt = c->g.ntot;
    std::vector<double> h(nt);
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const int n[3] = {c->g.n0, c->g.n1, c->g.n2};
    for (int a = 0; a < 3; ++a) {
        copy_sync(c, h.data(), c->TK[a].p, nt * sizeof(double), hipMemcpyDeviceToHost);
        for (int i2 = 0; i2 < n[2]; ++i2)
            for (int i1 = 0; i1 < n[1]; ++i1)
                for (int i0 = 0; i0 < n[0]; ++i0) {
                    const int idx[3] = {i0, i1, c->g.off2 + i2};
                    const int ext[3] = {n[0], n[1], c->g.gn2};
                    if (idx[a] >= ext[a] - 1) continue;
                    acc[a] += h[c->g.np * (i2 + 1) + (long)c->g.n0 * i1 + i0];
                    acc[3 + a] += 1.0;
                }
    }
    if (c->dist) {
        if (c->red_out.n < 6) c->red_out.alloc(64);
        TP_HIP(hipMemcpyAsync(c->red_out.p, acc, sizeof(acc), hipMemcpyHostToDevice, c->stream));
        allreduce_sum(c, c->red_out.p, 6);
        TP_HIP(hipMemcpyAsync(acc, c->red_out.p, sizeof(acc), hipMemcpyDeviceToHost, c->stream));
        TP_HIP(hipStreamSynchronize(c->stream));
    }
    for (int a = 0; a < 3; ++a) st[a] = acc[3 + a] > 0 ? acc[a] / acc[3 + a] : 0.0;
}

// the captured pc_apply graphs bake in buffer addresses and options: invalidate them when any changes
static void refresh_pc_signature(tp_ctx *c) {
    const uintptr_t sig[] = {(uintptr_t)c->opA00.base, (uintptr_t)c->opA01.base, (uintptr_t)c->opA10.base,
                             (uintptr_t)c->Sm.p, (uintptr_t)c->ilu.fwd.p, (uintptr_t)c->ilu.fwdp.p, (uintptr_t)c->amg_p, (uintptr_t)c->amg_T, (uintptr_t)c->bamg,
                             (uintptr_t)c->w1.p, (uintptr_t)c->w3.p, (uintptr_t)c->w4.p, (uintptr_t)c->dcoef.p, (uintptr_t)c->spbuf.p,
                             (uintptr_t)c->opt.amg_nu, (uintptr_t)c->opt.pc_kind, (uintptr_t)c->opt.decoup,
                             (uintptr_t)c->opt.amg_single, (uintptr_t)c->opt.amg_gather_cells, (uintptr_t)c->opt.schur_a11, (uintptr_t)c->opt.fs_additive, (uintptr_t)c->opt.amg_full_levels, (uintptr_t)c->opt.amg_coarse_pre, (uintptr_t)c->opt.amg_coarse_post, (uintptr_t)c->opt.amg_tail_post, (uintptr_t)c->opt.amg_mid_skip,
                             (uintptr_t)c->ilu.ntiles, (uintptr_t)c->ilu.nsteps, (uintptr_t)c->ilu.whole};
    uintptr_t h = 1469598103934665603ull;
    for (uintptr_t v : sig) h = (h ^ v) * 1099511628211ull;
    if (h != c->pc_sig) { c->pc_sig = h; c->graph_epoch++; }
}

// pc_cptramg[_QI|_TI] (twophase.py:552-566): CPTRStage1PC.update with ONE system-AMG V-cycle as stage-1 solver
static void pc_setup_sysamg(tp_ctx *c) {
    TP_REQUIRE(c->b == 3, "pc_cptramg is a two-phase preconditioner");
    TP_REQUIRE(c->opt.decoup >= 0 && c->opt.decoup <= 2, "pc_cptramg: decoupling No, QI or TI");
    decouple(c);
    const GridDev gam = c->dist ? c->gfull : make_grid(c->g.n0, c->g.n1, c->g.n2, c->g.n2, 0);
    if (!c->bamg) {
        double st[3];
        face_strengths(c, st);             // the pressure's coarsening schedule
        bamg_build(c, c->bamg, gam, st);
    }
    const long nt = c->g.ntot;
    BStencil A0;
    if (c->opt.decoup == 0) { A0.base = c->J.p; A0.ss = (long)c->b * c->b * nt; A0.rs = (long)c->b * nt; A0.cs = nt; }
    else { A0.base = c->At.p; A0.ss = 4 * nt; A0.rs = 2 * nt; A0.cs = nt; }
    if (c->dist && bamg_dist_levels(c->bamg) == 0) {
        // small grids: the hierarchy lives on the gathered global grid, replicated on every rank (as small scalar hierarchies
        // do); larger ones keep their top levels on the slabs (tp_amg_block.hip) and work on the slab operator directly
        const size_t ng = (size_t)c->gfull.ntot;
        ensure_global(c, c->gAt, 28 * ng);
        ensure_global(c, c->gvec, 6 * ng);      // (sized for every pc kind: switching kinds never shrinks it)
        for (int s = 0; s < 7; ++s)
            for (int q = 0; q < 2; ++q)       // the two column planes of a block row are nt apart on both sides
                gather_slabs(c, A0.at(s, q, 0), A0.cs, c->gAt.p + ((size_t)(s * 2 + q) * 2) * ng, (long)ng, 2);
        A0.base = c->gAt.p; A0.ss = 4 * (long)ng; A0.rs = 2 * (long)ng; A0.cs = (long)ng;
    }
    bamg_setup(c, c->bamg, A0);
    ilu_factor(c);
    c->pc_ready = true;
    refresh_pc_signature(c);
}

void pc_setup(tp_ctx *c) {
    TP_REQUIRE(c->jac_ready, "pc_setup needs an assembled Jacobian");
    ensure_work(c);
    if (c->opt.pc_kind == 4) {               // pc_bilu (twophase.py:758-762): bjacobi + ILU is the whole preconditioner
        ilu_factor(c);
        c->pc_ready = true;
        refresh_pc_signature(c);
        return;
    }
    if (sysamg_of(c->opt)) { pc_setup_sysamg(c); return; }
    const bool cptr = schur_of(c->opt);       // fieldsplit-Schur stage on (p,T): pc_cptr and pc_fieldsplit_cd
    if (c->opt.pc_kind == 1) TP_REQUIRE(c->b == 3, "pc_cptr is a two-phase preconditioner");
    if (c->opt.pc_kind == 2) {
        TP_REQUIRE(c->b == 2, "pc_fieldsplit_cd is a single-phase preconditioner (singlephase.py:309-319)");
        TP_REQUIRE(c->opt.decoup == 0, "pc_fieldsplit_cd has no decoupling stage");
    }
    // stage 1: decoupling + AMG hierarchies (CPRStage1PC.update / CPTRStage1PC.update)
    decouple(c);
    // single GPU: the AMG works on the slab (= whole grid).  Multi-GPU: the hierarchy is that of the GLOBAL grid,
    // so the preconditioner (and the iteration counts) are those of the single-GPU run and only stage 2 is
    // bjacobi.  Grids above amg_gather_cells keep their top levels distributed over the slabs (tp_amg.hip);
    // smaller ones are replicated from the top: every rank gathers the scalar stage-1 operators.
    const GridDev gam = c->dist ? c->gfull : make_grid(c->g.n0, c->g.n1, c->g.n2, c->g.n2, 0);
    // selfp on several GPUs works on slab vectors (its exact-Sp sweep needs the slab's own Jacobian rows): the hierarchies
    // keep every level with >= 2 planes per rank distributed, whatever amg_gather_cells says
    c->gather_override = (c->dist && cptr && c->opt.schur_a11 == 2) ? 0 : -2;
    if (!c->amg_p) {
        double st[3];
        face_strengths(c, st);         // coarsening schedule decided once; structure is static
        amg_build(c, c->amg_p, gam, st);
        if (cptr) {
            double sg[3];
            const int n[3] = {gam.n0, gam.n1, gam.n2};
            for (int a = 0; a < 3; ++a) {
                const double hh = c->grid.h[a];
                sg[a] = n[a] > 1 ? c->vol / (hh * hh) : 0.0;
            }
            amg_build(c, c->amg_T, gam, sg);
            TP_REQUIRE((c->amg_p->dist_levels > 0) == (c->amg_T->dist_levels > 0), "stage-1 hierarchies disagree on distribution");
        }
    }
    Stencil Sl;
    Sl.base = c->Sm.p;
    Sl.slot_stride = c->g.ntot;
    if (c->opt.fs_additive) TP_REQUIRE(c->opt.pc_kind == 2 && c->opt.schur_a11 != 2, "fs_additive is the single-phase pc_fieldsplit_diag preset");
    const bool selfp = cptr && c->opt.schur_a11 == 2;
    if (selfp) {
        // pc_fieldsplit_schur_precondition selfp (pc_fieldsplit_selfp, singlephase.py:322-330)
        TP_REQUIRE(c->opt.pc_kind == 2, "selfp is the single-phase pc_fieldsplit_selfp preset's Schur preconditioner");
        TP_REQUIRE(!c->dist || c->amg_p->dist_levels > 0, "selfp on several GPUs needs slabs of at least two planes (its Schur "
                   "sweep works on slab vectors; the replicated global-grid stage 1 has no exact-Sp sweep)");
        if (c->spbuf.n < (size_t)10 * c->g.ntot) c->spbuf.alloc((size_t)10 * c->g.ntot);
        Sl.base = c->spbuf.p;                  // S7, filled by selfp_build on the stream of the S set-up below
        Sl.slot_stride = c->g.ntot;
    } else if (cptr && (c->opt.schur_a11 || c->opt.fs_additive)) {
        // pc_fieldsplit_schur_precondition a11 (singlephase.py:331-338, twophase.py:598-616): the T-T block of the
        // (decoupled) primary system stands in for the Schur complement
        Sl.base = c->opA00.base + 3 * (c->opA01.base - c->opA00.base);     // block (1,1) = 3 planes after (0,0)
        Sl.slot_stride = c->opA00.slot_stride;
        if (c->opt.decoup == 0) Sl.base = c->J.p + (long)(c->b + 1) * c->g.ntot;
    }
    if (cptr && !selfp) TP_REQUIRE(Sl.base, "pc_cptr needs the S~ operator (assemble with want_schur)");
    if (c->dist && c->amg_p->dist_levels == 0) {
        const size_t ng = (size_t)c->gfull.ntot;
        // every buffer is tested on its own size (an options switch cpr -> cptr, or cptr -> cptramg -> cptr, on a live
        // context must not find gvec shrunk or gA01/gA10/gSm missing because some OTHER buffer was already large enough)
        ensure_global(c, c->gA00, 7 * ng);
        if (cptr) { ensure_global(c, c->gA01, 7 * ng); ensure_global(c, c->gA10, 7 * ng); ensure_global(c, c->gSm, 7 * ng); }
        ensure_global(c, c->gvec, 6 * ng);
        gather_slabs(c, c->opA00.base, c->opA00.slot_stride, c->gA00.p, (long)ng, 7);
        Stencil G;
        G.slot_stride = (long)ng;
        G.base = c->gA00.p;
        amg_setup(c, c->amg_p, G);
        if (cptr) {
            gather_slabs(c, c->opA01.base, c->opA01.slot_stride, c->gA01.p, (long)ng, 7);
            gather_slabs(c, c->opA10.base, c->opA10.slot_stride, c->gA10.p, (long)ng, 7);
            gather_slabs(c, Sl.base, Sl.slot_stride, c->gSm.p, (long)ng, 7);
            G.base = c->gSm.p;
            amg_setup(c, c->amg_T, G);
        }
    } else if (c->dist) {
        amg_setup(c, c->amg_p, c->opA00);
        if (selfp) {
            // Sp of a boundary cell reads diag(A00) and the A01 row of its neighbour across the slab boundary: the Jacobian's
            // halo rows (singlephase.py:322-330 lets PETSc form Sp from the assembled parallel matrix)
            halo_exchange(c, c->g, c->J.p, 7 * c->b * c->b, c->g.ntot);
            selfp_build(c);
        }
        if (cptr) amg_setup(c, c->amg_T, Sl);
    } else {
        // one GPU: the AMG set-ups (2 x ~35 launch-latency-bound kernels) and the ILU factorisation are
        // independent -> three concurrent streams, joined before anything uses the preconditioner.
        // (not with RCCL in the set-up: one communicator must not be driven from two streams at once)
        hipStream_t main = c->stream;
        struct Restore { tp_ctx *c; hipStream_t s; ~Restore() { c->stream = s; } } restore{c, main};   // also on a throw
        TP_HIP(hipEventRecord(c->ev_fork, main));
        TP_HIP(hipStreamWaitEvent(c->aux[0], c->ev_fork, 0));
        c->stream = c->aux[0];
        amg_setup(c, c->amg_p, c->opA00);
        TP_HIP(hipEventRecord(c->ev_join[0], c->aux[0]));
        if (cptr) {
            TP_HIP(hipStreamWaitEvent(c->aux[1], c->ev_fork, 0));
            c->stream = c->aux[1];
            if (selfp) selfp_build(c);
            amg_setup(c, c->amg_T, Sl);
            TP_HIP(hipEventRecord(c->ev_join[1], c->aux[1]));
        }
        c->stream = main;
        ilu_factor(c);
        TP_HIP(hipStreamWaitEvent(main, c->ev_join[0], 0));
        if (cptr) TP_HIP(hipStreamWaitEvent(main, c->ev_join[1], 0));
        c->pc_ready = true;
    }
    // stage 2: numeric block-ILU(0) of every tile of this rank's slab
    if (c->dist) ilu_factor(c);
    c->pc_ready = true;
    refresh_pc_signature(c);
}

// the relaxation-only truncation level of each hierarchy is known once its set-up kernels have run
void resolve_cycle_shapes(tp_ctx *c) {
    bool changed = amg_resolve_trunc(c, c->amg_p);
    changed = amg_resolve_trunc(c, c->amg_T) || changed;
    if (changed) c->graph_epoch++;
}

// y = B1 x :  CPRStage1PC.apply (preconditioners.py:881-903) / CPTRStage1PC.apply (:1550-1567)
void stage1_apply(tp_ctx *c, const double *x, double *y, bool zero_secondary) {
    const GridDev &g = c->g;
    const long nt = g.ntot;
    ensure_work(c);
    resolve_cycle_shapes(c);
    double *r0 = c->w3.p, *r1 = c->w3.p + nt, *t = c->w3.p + 2 * nt;   // w3 has >= 3 planes
    // y_s = 0 for the non-primary fields (:902-903, :1566-1567)
    const int npri = npri_of(c->opt);
    if (zero_secondary)
        for (int f = npri; f < c->b; ++f) vec_zero(c, y + (long)f * nt, nt);
    if (c->opt.decoup == 0 && !c->dist) {
        // decoupling "No" (pc_cptr, pc_cpr, pc_fieldsplit_cd presets): the stage-1 right-hand sides ARE the
        // primary fields of x -- no copy (multi-GPU keeps the copy: the V-cycle's exchange writes b's halos)
        r0 = const_cast<double *>(x);
        r1 = const_cast<double *>(x) + nt;
    } else {
        stage1_rhs(c, x, 0, r0);                   // r_p = x_p - (D_ps D_ss^-1) x_s
        if (npri == 2) stage1_rhs(c, x, 1, r1);
    }
    if (sysamg_of(c->opt)) {
        // pc_cptramg: y_pT = K(Atilde_00) r_pT, one V-cycle of the 2x2-block system AMG (r0, r1 are adjacent planes)
        if (c->dist && bamg_dist_levels(c->bamg) == 0) {
            const long ng = c->gfull.ntot;
            gather_slabs(c, r0, nt, c->gvec.p, ng, 2);
            bamg_vcycle(c, c->bamg, c->gvec.p, c->gvec.p + 2 * ng);
            const long off = g.np * c->grid.off2;          // my slab INCLUDING its halo planes
            vec_copy(c, c->gvec.p + 2 * ng + off, y, nt);
            vec_copy(c, c->gvec.p + 3 * ng + off, y + nt, nt);
        } else {
            bamg_vcycle(c, c->bamg, r0, y);
        }
        return;
    }
    if (c->dist && c->amg_p->dist_levels == 0) {
        // gathered global system: work vectors gr0, gr1, gy0, gy1, gt, gw on the global grid
        const GridDev &G = c->gfull;
        const long ng = G.ntot;
        double *gr0 = c->gvec.p, *gr1 = gr0 + ng, *gy0 = gr0 + 2 * ng, *gy1 = gr0 + 3 * ng, *gt = gr0 + 4 * ng,
               *gw = gr0 + 5 * ng;
        gather_slabs(c, r0, nt, gr0, ng, npri);    // r0 (and r1: consecutive planes on both sides)
        if (npri == 1) {
            amg_vcycle(c, c->amg_p, gr0, gy0);
        } else {
            Stencil A10, A01;
            A10.base = c->gA10.p; A10.slot_stride = ng;
            A01.base = c->gA01.p; A01.slot_stride = ng;
            if (c->opt.fs_additive) {               // PCFIELDSPLIT additive: one V-cycle per field
                amg_vcycle(c, c->amg_p, gr0, gy0);
                amg_vcycle(c, c->amg_T, gr1, gy1);
            } else {
            amg_vcycle(c, c->amg_p, gr0, gw);
            spmv_scalar(c, G, A10, gw, gt, -1.0, gr1);
            amg_vcycle(c, c->amg_T, gt, gy1);
            spmv_scalar(c, G, A01, gy1, gt, -1.0, gr0);
            amg_vcycle(c, c->amg_p, gt, gy0);
            }
        }
        // my slab of the result INCLUDING its halo planes (global planes lo-1 .. hi), so y needs no exchange
        const long off = g.np * c->grid.off2;
        vec_copy(c, gy0 + off, y, nt);
        if (npri == 2) vec_copy(c, gy1 + off, y + nt, nt);
        return;
    }
    if (npri == 1) {
        amg_vcycle(c, c->amg_p, r0, y);
        return;
    }
    // PCFIELDSPLIT schur FULL on (p,T) (twophase.py:536-545): K(A00), K(S~) = one V-cycle each
    // (multi-GPU with distributed AMG levels: the V-cycles return owned cells, the couplings read halos)
    double *y0 = y, *y1 = y + nt;
    if (c->opt.fs_additive) {              // PCFIELDSPLIT additive (pc_fieldsplit_diag, singlephase.py:371-375)
        amg_vcycle(c, c->amg_p, r0, y0);
        amg_vcycle(c, c->amg_T, r1, y1);
        return;
    }
    amg_vcycle(c, c->amg_p, r0, c->w4.p);                                   // y0 = K(A00) r0
    if (c->dist) halo_exchange(c, g, c->w4.p, 1, nt);
    spmv_scalar(c, g, c->opA10, c->w4.p, t, -1.0, r1);                      // t = r1 - A10 y0
    if (c->opt.schur_a11 == 2) {                                            // selfp: V7 then one Jacobi sweep on the exact Sp
        double *xv = c->spbuf.p + 9 * nt;
        amg_vcycle(c, c->amg_T, t, xv);
        selfp_post(c, t, xv, y1);
    } else {
        amg_vcycle(c, c->amg_T, t, y1);                                     // y1 = K(S~) t
    }
    if (c->dist) halo_exchange(c, g, y1, 1, nt);
    spmv_scalar(c, g, c->opA01, y1, t, -1.0, r0);                           // t = r0 - A01 y1
    amg_vcycle(c, c->amg_p, t, y0);                                         // y0 = K(A00) t
}

// composite multiplicative: y = B1 x ; r = x - J y ; y += B2 r
static void pc_apply_body(tp_ctx *c, const double *x, double *y) {
    if (c->opt.pc_kind == 4) { ilu_solve(c, x, y, nullptr, 0); return; }
    const int npri = npri_of(c->opt);
    // (y's secondary fields are left untouched: the second stage below never reads them and overwrites them)
    stage1_apply(c, x, y, false);                 // multi-GPU, replicated stage 1: y comes back with live halo planes
    if (c->dist && ((c->amg_p && c->amg_p->dist_levels > 0) || (sysamg_of(c->opt) && bamg_dist_levels(c->bamg) > 0)))
        halo_exchange(c, c->g, y, npri, c->g.ntot);       // (slab-distributed hierarchies return owned cells only)
    if (c->opt.pc_kind == 2) return;                          // pc_fieldsplit_cd: the Schur stage IS the preconditioner
    resid_block_cols(c, c->J.p, x, y, npri, c->w1.p);        // secondary fields of y are zero
    ilu_solve(c, c->w1.p, y, y, npri);                       // y = y + M^-1 r  (y's secondary fields are zero: not read)
}

// ---- recording of multi-GPU pc_apply programs: one capture segment between two exchanges -------------------------------
void seg_begin(tp_ctx *c) {
    TP_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    c->rec_capturing = true;
}
// closes the segment; a non-empty one is instantiated, appended to the program and LAUNCHED (a capture executes nothing: the
// recording pass must still produce the result, and the exchange that follows reads what these kernels wrote)
void seg_end(tp_ctx *c) {
    hipGraph_t graph = nullptr;
    c->rec_capturing = false;
    TP_HIP(hipStreamEndCapture(c->stream, &graph));
    size_t nnodes = 0;
    TP_HIP(hipGraphGetNodes(graph, nullptr, &nnodes));
    if (nnodes > 0) {
        hipGraphExec_t exec = nullptr;
        const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { (void)hipGraphDestroy(graph); TP_HIP(e); }
        c->rec->steps.push_back({exec, nullptr});
        TP_HIP(hipGraphLaunch(exec, c->stream));
    }
    TP_HIP(hipGraphDestroy(graph));
}

// One preconditioner application is ~100 short kernels (the V-cycles' coarse levels); issued eagerly
// the host launch path (~3 us per kernel) is slower than the GPU executes them.  The whole sequence
// is therefore captured into a hipGraph per (input, output) address pair -- FGMRES uses the fixed pairs
// (V_j, Z_j) -- and replayed per Krylov iteration.
void pc_apply(tp_ctx *c, const double *x, double *y) {
    TP_REQUIRE(c->pc_ready, "pc_apply before pc_setup");
    static const bool use_graph = !(getenv("TP_GRAPH") && atoi(getenv("TP_GRAPH")) == 0);
    c->vcycles += c->opt.pc_kind == 4 ? 0 : c->opt.fs_additive ? 2 : schur_of(c->opt) ? 3 : 1;
    ensure_work(c);                      // never allocate inside a stream capture
    resolve_cycle_shapes(c);             // (waits for the last set-up's dominance ratios: not inside a capture either)
    if (!use_graph) {
        pc_apply_body(c, x, y);
        return;
    }
    if (c->pc_graph_epoch != c->graph_epoch || c->pc_graphs.size() > 512 || c->pc_programs.size() > 512) {      // stale (or runaway) cache
        for (auto &gph : c->pc_graphs) (void)hipGraphExecDestroy(gph.exec);
        c->pc_graphs.clear();
        for (auto &pr : c->pc_programs)
            for (auto &st : pr.steps)
                if (st.exec) (void)hipGraphExecDestroy(st.exec);
        c->pc_programs.clear();
        c->pc_graph_epoch = c->graph_epoch;
    }
    if (c->dist) {
        // several GPUs: graph segments between the exchanges, the exchanges themselves as host closures (tp_common.hpp)
        for (auto &pr : c->pc_programs)
            if (pr.x == x && pr.y == y) {
                for (auto &st : pr.steps) {
                    if (st.exec) TP_HIP(hipGraphLaunch(st.exec, c->stream));
                    else st.comm();
                }
                return;
            }
        c->pc_programs.push_back({x, y, {}});
        c->rec = &c->pc_programs.back();
        try {
            seg_begin(c);
            pc_apply_body(c, x, y);
            seg_end(c);
        } catch (...) {
            if (c->rec_capturing) {
                hipGraph_t g = nullptr;
                (void)hipStreamEndCapture(c->stream, &g);
                if (g) (void)hipGraphDestroy(g);
                c->rec_capturing = false;
            }
            for (auto &st : c->pc_programs.back().steps)
                if (st.exec) (void)hipGraphExecDestroy(st.exec);
            c->pc_programs.pop_back();
            c->rec = nullptr;
            throw;
        }
        c->rec = nullptr;
        return;
    }
    // TP_DEBUG=2: one line per HIP graph call (used to locate the profiler crash described in DESIGN.md 6)
    static const bool trace = getenv("TP_DEBUG") && atoi(getenv("TP_DEBUG")) >= 2;
    auto say = [&](const char *what) { if (trace) { fprintf(stderr, "[tp] pc_apply(%p,%p): %s\n", (const void *)x, (void *)y, what); fflush(stderr); } };
    hipGraphExec_t exec = nullptr;
    for (auto &gph : c->pc_graphs)
        if (gph.x == x && gph.y == y) { exec = gph.exec; break; }
    if (!exec) {
        hipGraph_t graph = nullptr;
        say("begin capture");
        TP_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        try {
            pc_apply_body(c, x, y);
        } catch (...) {
            (void)hipStreamEndCapture(c->stream, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        say("end capture");
        TP_HIP(hipStreamEndCapture(c->stream, &graph));
        say("instantiate");
        TP_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        TP_HIP(hipGraphDestroy(graph));
        c->pc_graphs.push_back({x, y, exec});
    }
    say("launch");
    TP_HIP(hipGraphLaunch(exec, c->stream));
    say("launched");
}

// ------------------------------------------------------------------------------------------------
// FGMRES(m) from x0 = 0.  Returns KSP reason (2 = CONVERGED_RTOL, 3 = CONVERGED_ATOL, -3 = DIVERGED_ITS).
int fgmres(tp_ctx *c, const double *bvec, double *x, int *its_out, double *rnorm_out) {
    const GridDev &g = c->g;
    const int B = c->b;
    const long nv = (long)B * g.ntot;
    const int maxit = c->opt.ksp_max_it;
    const int restart = std::max(1, std::min(c->opt.ksp_restart, maxit));
    ensure_work(c);
    // basis storage grows on demand (restart 200 x 2 vectors would be 10.8 GB on SPE10 3-D)
    auto ensure_basis = [&](int need) {
        if (c->gs_cap >= need) return;
        int cap = std::max(need, std::min(restart + 1, std::max(32, 2 * c->gs_cap)));
        DBuf<double> nV, nZ;
        nV.alloc((size_t)cap * nv);
        nZ.alloc((size_t)cap * nv);
        if (c->gs_cap > 0) {
            TP_HIP(hipMemcpyAsync(nV.p, c->V.p, sizeof(double) * c->gs_cap * nv, hipMemcpyDeviceToDevice, c->stream));
            TP_HIP(hipMemcpyAsync(nZ.p, c->Z.p, sizeof(double) * c->gs_cap * nv, hipMemcpyDeviceToDevice, c->stream));
            TP_HIP(hipStreamSynchronize(c->stream));
        }
        std::swap(c->V.p, nV.p); std::swap(c->V.n, nV.n);
        std::swap(c->Z.p, nZ.p); std::swap(c->Z.n, nZ.n);
        c->gs_cap = cap;
    };
    vec_zero(c, x, nv);
    const double bnorm = norm2(c, B, bvec);
    int its = 0;
    if (bnorm == 0.0) { *its_out = 0; *rnorm_out = 0.0; return 2; }
    if (!std::isfinite(bnorm)) { *its_out = 0; *rnorm_out = bnorm; return -9; }
    const double tol = std::max(c->opt.ksp_rtol * bnorm, c->opt.ksp_atol);
    double beta = bnorm;
    const double *rsrc = bvec;
    std::vector<double> H, cs, sn, gvec, hcol, yk;
    while (true) {
        const int m = std::min(restart, maxit - its);
        H.assign((size_t)(m + 1) * m, 0.0);
        cs.assign(m, 0.0); sn.assign(m, 0.0); gvec.assign(m + 1, 0.0);
        hcol.resize(m + 2);
        gvec[0] = beta;
        ensure_basis(2);
        vec_scale_to(c, B, 1.0 / beta, rsrc, c->V.p);                      // v0 = r/beta
        int k = 0, reason = 0;
        double res = beta;
        // The loop is PIPELINED.  The host needs h (Givens rotations, convergence test) once per iteration; waiting for
        // it with the stream empty leaves the GPU idle for a host wake-up plus a launch latency (~40 us of a 1.1 ms iteration).
        // Instead v_{j+1} = w / ||w|| is formed from the norm on the device and z_{j+1} = M^-1 v_{j+1}, J z_{j+1} are enqueued
        // BEFORE the host waits -- on an event behind the reductions, not on the stream.  Speculative: if iteration j turns out
        // to be the last, that work is discarded (~0.9 ms), so it is only issued while the residual, extrapolated with the last
        // reduction factor, stays 4x above the tolerance.  Same arithmetic either way (bit-identical iterates).
        static const bool pipe_on = !(getenv("TP_FGMRES_PIPE") && atoi(getenv("TP_FGMRES_PIPE")) == 0);
        static const double spec_margin = getenv("TP_SPEC_MARGIN") ? atof(getenv("TP_SPEC_MARGIN")) : 4.0;
        bool have_w = false;               // z_j, w = J z_j already enqueued by the previous iteration
        double res_prev = beta, rate = 1.0;
        for (int j = 0; j < m; ++j) {
            const bool pipe = pipe_on && !c->monitor && orthogonalize_can_split(c, j + 2);
            ensure_basis(j + (pipe ? 3 : 2));
            double *vj = c->V.p + (long)j * nv, *zj = c->Z.p + (long)j * nv, *w = c->V.p + (long)(j + 1) * nv;
            if (!have_w) {
                pc_apply(c, vj, zj);                                        // z_j = M^-1 v_j
                spmv_block_halo(c, c->J.p, zj, w);                          // w = J z_j (multi-GPU: exchange of z_j's halos overlapped)
            }
            have_w = false;
            bool spec = false;
            if (pipe) {
                orthogonalize_enqueue(c, B, c->V.p, nv, j + 1, w);          // h = V^T w ; w -= V h ; ||w||^2 (no host wait yet)
                spec = j + 1 < m && its + 1 < maxit && res_prev * std::min(rate, 1.0) > spec_margin * tol;      // (predicted res_j)
                if (spec) ++c->spec_issued; else ++c->spec_skipped;
                if (spec) {
                    vec_scale_dev_norm(c, B, orthogonalize_norm_dev(c, j + 1), w);          // v_{j+1} = w/||w||
                    pc_apply(c, w, c->Z.p + (long)(j + 1) * nv);
                    spmv_block_halo(c, c->J.p, c->Z.p + (long)(j + 1) * nv, c->V.p + (long)(j + 2) * nv);
                    have_w = true;
                }
                orthogonalize_wait(c, j + 1, hcol.data());
            } else {
                orthogonalize(c, B, c->V.p, nv, j + 1, w, hcol.data());     // h = V^T w ; w -= V h ; ||w||^2
            }
            const double hn = std::sqrt(hcol[j + 1]);
            for (int i = 0; i <= j; ++i) H[(size_t)i * m + j] = hcol[i];
            H[(size_t)(j + 1) * m + j] = hn;
            for (int i = 0; i < j; ++i) {                                   // previous Givens rotations
                const double t = cs[i] * H[(size_t)i * m + j] + sn[i] * H[(size_t)(i + 1) * m + j];
                H[(size_t)(i + 1) * m + j] = -sn[i] * H[(size_t)i * m + j] + cs[i] * H[(size_t)(i + 1) * m + j];
                H[(size_t)i * m + j] = t;
            }
            const double d = std::hypot(H[(size_t)j * m + j], H[(size_t)(j + 1) * m + j]);
            cs[j] = H[(size_t)j * m + j] / d;
            sn[j] = H[(size_t)(j + 1) * m + j] / d;
            H[(size_t)j * m + j] = d;
            H[(size_t)(j + 1) * m + j] = 0.0;
            gvec[j + 1] = -sn[j] * gvec[j];
            gvec[j] = cs[j] * gvec[j];
            ++its;
            k = j + 1;
            res = std::fabs(gvec[j + 1]);
            if (c->monitor) {
                // ksp.buildResidual() per field (thermalmodel.py:44-74): x_j = Z y_j, r = b - J x_j, ||r_f||
                std::vector<double> ym(k, 0.0), fn(B, 0.0);
                std::vector<double> gm(gvec.begin(), gvec.begin() + k);
                for (int i = k - 1; i >= 0; --i) {
                    double s = gm[i];
                    for (int q = i + 1; q < k; ++q) s -= H[(size_t)i * m + q] * ym[q];
                    ym[i] = s / H[(size_t)i * m + i];
                }
                double *xm = c->w4.p, *rm = c->w2.p;                        // free between pc_apply calls
                vec_copy(c, x, xm, nv);
                multi_axpy(c, B, c->Z.p, nv, k, ym.data(), 1.0, xm);
                if (c->dist) halo_exchange(c, g, xm, B, g.ntot);
                resid_block_cols(c, c->J.p, bvec, xm, B, rm);
                std::vector<const double *> fp(B);
                for (int f = 0; f < B; ++f) fp[f] = rm + (long)f * g.ntot;
                // (each field plane as a 1-field vector)
                for (int f = 0; f < B; ++f) { const double *one[1] = {fp[f]}; multi_norm2sq(c, 1, 1, one, &fn[f]); fn[f] = std::sqrt(fn[f]); }
                c->monitor(its, res, fn.data(), B, c->monitor_user);
            }
            rate = res_prev > 0.0 ? res / res_prev : 1.0;
            res_prev = res;
            if (!std::isfinite(res) || res <= tol || hn == 0.0) {
                if (have_w) ++c->spec_wasted;
                if (have_w) c->vcycles -= c->opt.pc_kind == 4 ? 0 : c->opt.fs_additive ? 2 : schur_of(c->opt) ? 3 : 1;   // (discarded application)
                reason = !std::isfinite(res) ? -9 : 2;                     // KSP_DIVERGED_NANORINF | converged (or happy breakdown)
                break;
            }
            if (!spec) vec_scale_to(c, B, 1.0 / hn, w, w);                  // v_{j+1} = w/||w||
        }
        // y = H^-1 g ; x += Z y
        yk.assign(k, 0.0);
        for (int i = k - 1; i >= 0; --i) {
            double s = gvec[i];
            for (int q = i + 1; q < k; ++q) s -= H[(size_t)i * m + q] * yk[q];
            yk[i] = s / H[(size_t)i * m + i];
        }
        multi_axpy(c, B, c->Z.p, nv, k, yk.data(), 1.0, x);
        if (reason) { *its_out = its; *rnorm_out = res; return reason; }
        if (its >= maxit) { *its_out = its; *rnorm_out = res; return -3; }
        // restart: r = b - J x
        if (c->dist) halo_exchange(c, g, x, B, g.ntot);
        resid_block_cols(c, c->J.p, bvec, x, B, c->w2.p);
        beta = norm2(c, B, c->w2.p);
        rsrc = c->w2.p;
        if (beta <= tol) { *its_out = its; *rnorm_out = beta; return 2; }
    }
}

// ------------------------------------------------------------------------------------------------
void newton(tp_ctx *c, tp_solve_info *info) {
    const GridDev &g = c->g;
    const int B = c->b;
    const long nv = (long)B * g.ntot;
    ensure_work(c);
    const bool schur = schur_of(c->opt);
    if (schur && c->Sm.n < (size_t)7 * g.ntot) c->Sm.alloc((size_t)7 * g.ntot);
    TP_REQUIRE(c->u.n > 0, "state not set");
    tp::DBuf<double> *dx = &c->dx;
    if (dx->n < (size_t)nv) dx->alloc(nv);

    if (c->dist) halo_exchange(c, g, c->u.p, B, g.ntot);
    assemble(c, true, schur);
    double fnorm = norm2(c, B, c->R.p);
    const double fnorm0 = fnorm;
    int nits = 0, lits = 0, reason = 0, kreason = 0;
    if (!std::isfinite(fnorm)) reason = -4;                      // SNES_DIVERGED_FNORM_NAN
    else if (fnorm < c->opt.snes_atol) reason = 2;               // SNES_CONVERGED_FNORM_ABS
    while (reason == 0) {
        if (nits >= c->opt.snes_max_it) { reason = -5; break; }  // SNES_DIVERGED_MAX_IT
        pc_setup(c);
        int kits = 0;
        double rn = 0.0;
        kreason = fgmres(c, c->R.p, dx->p, &kits, &rn);
        lits += kits;
        if (kreason < 0) { reason = -3; break; }                 // SNES_DIVERGED_LINEAR_SOLVE
        vec_axpy_owned(c, B, -1.0, dx->p, c->u.p);               // basic line search, lambda = 1
        if (c->dist) halo_exchange(c, g, c->u.p, B, g.ntot);
        assemble(c, true, schur);
        ++nits;
        double nrm[3];
        const double *nv3[3] = {c->R.p, dx->p, c->u.p};
        multi_norm2sq(c, B, 3, nv3, nrm);                       // one reduction, one all-reduce, one host sync
        fnorm = std::sqrt(nrm[0]);
        const double snorm = std::sqrt(nrm[1]), xnorm = std::sqrt(nrm[2]);
        if (!std::isfinite(fnorm)) reason = -4;
        else if (fnorm < c->opt.snes_atol) reason = 2;
        else if (fnorm <= c->opt.snes_rtol * fnorm0) reason = 3;  // SNES_CONVERGED_FNORM_RELATIVE
        else if (snorm < c->opt.snes_stol * xnorm) reason = 4;    // SNES_CONVERGED_SNORM_RELATIVE
    }
    info->nits = nits;
    info->lits = lits;
    info->reason = reason;
    info->last_ksp_reason = kreason;
    info->fnorm0 = fnorm0;
    info->fnorm = fnorm;
    info->vcycles = (int)c->vcycles;
}

}  // namespace tp

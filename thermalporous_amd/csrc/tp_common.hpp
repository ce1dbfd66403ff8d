// Shared host-side declarations of libthermalporous_hip (context, buffers, launch helpers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>
#include <functional>
#include "../../include/thermalporous_hip.h"

struct ncclComm;

namespace tp {
struct LocalGroup;

void set_error(const std::string &msg);

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define TP_HIP(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            throw tp::Error(std::string(#call) + " failed: " + hipGetErrorString(e_) + " at " +  \
                            __FILE__ + ":" + std::to_string(__LINE__));                          \
    } while (0)

#define TP_REQUIRE(cond, msg)                                                                    \
    do {                                                                                         \
        if (!(cond)) throw tp::Error(std::string(msg) + " [" #cond "]");                         \
    } while (0)

// Device view of the slab grid, passed by value to kernels.
struct GridDev {
    int n0, n1, n2;        // owned
    int gn2, off2;         // global extent/offset along axis 2
    long np;               // n0*n1 (plane)
    long nown;             // np*n2
    long ntot;             // np*(n2+2)
    int nb_lo, nb_hi;      // 1 if a neighbouring slab exists below/above (halo is live)
};

inline GridDev make_grid(int n0, int n1, int n2, int gn2, int off2) {
    GridDev g;
    g.n0 = n0; g.n1 = n1; g.n2 = n2; g.gn2 = gn2; g.off2 = off2;
    g.np = (long)n0 * n1; g.nown = g.np * n2; g.ntot = g.np * (n2 + 2);
    g.nb_lo = off2 > 0; g.nb_hi = off2 + n2 < gn2;
    return g;
}

// XCD-aware block -> cell-range mapping (cdna_hip_programming.md T1): workgroups are dealt round-robin over
// the 8 XCDs, each with its own L2.  Remapping block b to chunk (b % 8)*(nblocks/8) + b/8 gives every XCD
// one contiguous eighth of the cell range, so the stencil's neighbour reads (x at c +- 1, +- n0, +- n0*n1)
// hit in that XCD's L2 instead of being fetched once per XCD.  Pure performance: any placement is correct.
// Grids launched with xcd_grid() have a multiple of 8 blocks; the kernel bounds-checks the cell index.
#if defined(__HIPCC__)
// Every kernel that calls this is launched with TP_BLOCK (= 256) threads (xcd_grid's default): the block size is a compile-time
// constant here on purpose.  `blockDim.x` is a 16-bit field of the hidden kernel arguments that the compiler fetches with a
// VECTOR load (global_load_ushort) followed by s_waitcnt vmcnt(0) before the first useful instruction: one dependent memory round
// trip at the start of every kernel, ~0.5-1 us of the 4-6 us the small AMG levels' kernels take.
constexpr int TP_BLOCK = 256;
__device__ __forceinline__ long xcd_tid() {
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const unsigned rb = (b & 7u) * (nb >> 3) + (b >> 3);
    return (long)rb * TP_BLOCK + threadIdx.x;
}
#endif
inline dim3 xcd_grid(long n, int bs = 256) {
    const long nb = (n + bs - 1) / bs;
    return dim3((unsigned)(((nb + 7) / 8) * 8));
}

// pc_kind: 0 pc_cpr, 1 pc_cptr, 2 pc_fieldsplit_cd, 3 pc_cptramg.  1 and 2 run the fieldsplit-Schur-FULL stage on
// (p,T) and need the S~ operator from the assembly; 3 runs ONE system-AMG V-cycle on the 2x2-block (p,T) operator.
inline int npri_of(const tp_options &o) { return o.pc_kind >= 1 ? 2 : 1; }
inline bool schur_of(const tp_options &o) { return o.pc_kind == 1 || o.pc_kind == 2; }
inline bool sysamg_of(const tp_options &o) { return o.pc_kind == 3; }

// Scalar 7-point stencil operator: slot s lives at base + s*slot_stride (doubles).
struct Stencil {
    double *base = nullptr;
    long slot_stride = 0;
    __host__ __device__ const double *slot(int s) const { return base + (long)s * slot_stride; }
    __host__ __device__ double *slot(int s) { return base + (long)s * slot_stride; }
};

// Block 7-point stencil operator (system AMG): block (q,r) of slot s lives at base + s*ss + q*rs + r*cs.
struct BStencil {
    double *base = nullptr;
    long ss = 0, rs = 0, cs = 0;
    __host__ __device__ const double *at(int s, int q, int r) const { return base + s * ss + q * rs + r * cs; }
    __host__ __device__ double *at(int s, int q, int r) { return base + s * ss + q * rs + r * cs; }
};
struct BAmg;

// Derived closure constants (device copy of tp_params + precomputed factors).
struct DevPrm {
    double ko, kw, kr, c_v_w, c_v_o, c_r, rho_r, T_inj, g, U;
    double rho_ref, mu_o_coef, mu_o_exp;   // oil_rho / oil_mu (physicalparameters.py:37-57)
    double w0, w2;                          // equation weights (twophase.py:142-147)
};

template <class T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    bool owned = true;
    void alloc(size_t n_) {
        free();
        n = n_;
        if (n) {
            // (zeroed on the per-thread stream, complete on return.  Nothing in the library uses the LEGACY default stream: a
            // legacy-stream operation synchronises with every blocking stream of the device and is an error -- "would make the
            // legacy stream depend on a capturing blocking stream" -- while any other slab thread is capturing its pc_apply)
            TP_HIP(hipMalloc((void **)&p, n * sizeof(T)));
            TP_HIP(hipMemsetAsync(p, 0, n * sizeof(T), hipStreamPerThread));
            TP_HIP(hipStreamSynchronize(hipStreamPerThread));
        }
    }
    // a slice of somebody else's allocation (zeroed by its owner)
    void view(T *q, size_t n_) {
        free();
        p = q; n = n_; owned = false;
    }
    void free() {
        if (p && owned) (void)hipFree(p);
        p = nullptr;
        n = 0;
        owned = true;
    }
    ~DBuf() { free(); }
    DBuf() = default;
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
};

struct AmgLevel {
    GridDev g;
    DBuf<double> A;        // 7 planes (levels >= 1); level 0 uses an external stencil view
    Stencil op;
    DBuf<double> invd;     // omega / diag
    DBuf<double> wm, wp;   // interpolation weights of F points along `axis`
    DBuf<double> b, x, x2, r, e;
    int axis = -1;         // coarsening axis towards the next level (-1: coarsest)
};

struct Amg {
    // one allocation for every buffer of every level: the coarse levels are tiny, and with one hipMalloc each
    // (~150 of them) every small kernel of the cycle starts with TLB misses on half a dozen scattered pages
    DBuf<double> arena;
    std::vector<AmgLevel *> lv;
    DBuf<double> coarse_inv;   // dense inverse on the coarsest grid
    int ncoarse = 0;
    bool single = false;       // operators / weights / inverse diagonals stored in fp32
    int tail_level = 0;        // first level handled by the single-workgroup tail kernel
    int tail_lds = 0;          // doubles per LDS-resident vector set of the tail (0: tail vectors stay in global memory)
    long fuse_below = 200000;  // levels with fewer cells use the fused (launch-saving) kernels
    // multi-GPU: levels [0, dist_levels) live on this rank's slab (halo exchanges between sweeps), the levels
    // below on the gathered global grid, replicated on every rank.  0 = the whole hierarchy is replicated.
    int dist_levels = 0;
    std::vector<std::vector<std::pair<int, int>>> ranges;   // [level][rank] -> owned global planes along axis 2
    // relaxation-only truncation (tp_options.amg_dom_tau): dominance ratios of the V(nu,nu) levels, measured by the set-up
    // kernels into 64 slots per level (spread atomics), copied to pinned host memory behind `ev_ratio`
    DBuf<double> ratio_dev;
    double *ratio_host = nullptr;
    hipEvent_t ev_ratio = nullptr;
    bool ratio_pending = false;
    int trunc = -1;            // level that ends the cycle with two Jacobi sweeps (-1: none)
    bool dense_done = false;   // a coarse inverse exists (TP_EXP_SKIP_DENSE timing experiment)
    double ratio0 = 0.0;
    DBuf<char> lvdev;          // device array of level descriptors (LevelDev) for the tail kernel
    std::vector<char> lvhost;
    std::vector<int> sched;
    ~Amg() {
        for (auto *l : lv) delete l;
        if (ratio_host) (void)hipHostFree(ratio_host);
        if (ev_ratio) (void)hipEventDestroy(ev_ratio);
    }
};

struct IluData {
    int t0 = 0, t1 = 8, t2 = 8, nt0 = 0, nt1 = 0, nt2 = 0, ntiles = 0, nsteps = 0;
    DBuf<double> fwd, bwd, ytmp;   // streaming factor data in consumption order
    DBuf<double> jt;               // the Jacobian blocks re-ordered the same way (input of the factorisation)
    long slots = 0;                // ntiles*nsteps*64
    bool mw = false;               // ILU(0): factor stored in the row-major layout of the multi-wave sweep (tp_ilu.hip)
    int levels = 0;                // 0: ILU(0), 1: ILU(1) (tp_options.ilu_levels; other chunk layout, see tp_ilu.hip)
    // ILU(1): PACKED copy of the factor for the sweeps -- a (tile, step) chunk holds rows only for the lanes that have a cell
    // at that step (the s = l0 + 2j + 4k skew leaves a third of the padded stream zero, 54-lane tiles another 16 %);
    // pref[s] = slots before step s of a tile, ptot = slots per tile
    DBuf<double> fwdp, bwdp;
    DBuf<int> pref;
    int ptot = 0;
    // tp_options.ilu_whole: one block per rank.  Tiles keep their couplings; tile-diagonal d = T0+T1+T2 is one launch:
    // tiles diag_tiles[diag_off[d] .. diag_off[d+1]) (device array); xtmp: the raw backward-sweep result of every
    // (tile, step, lane) for the tiles of later launches (x itself may already hold addto + result)
    bool whole = false;
    int ndiag = 0;
    std::vector<int> diag_off;
    DBuf<int> diag_tiles;
    DBuf<double> xtmp;
};

}  // namespace tp

struct tp_ctx {
    tp_grid grid;
    tp_params prm;
    tp_options opt;
    tp::GridDev g;
    tp::DevPrm dprm;
    int nph = 1, b = 2, device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // pc_setup forks: the two AMG set-ups and the ILU factorisation are independent chains of small kernels
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    double dt = 0.0, vol = 0.0;
    bool fields_ready = false, have_old = false, jac_ready = false, pc_ready = false;
    // fields
    tp::DBuf<double> phi, K[3], kTs, TK[3];
    // state
    tp::DBuf<double> u, u_old, acc_old, R, J, Sm;
    // sources
    tp::DBuf<tp_source> src;
    tp::DBuf<int> src_start;      // groups of entries sharing a cell
    int nsrc = 0, nsrc_groups = 0;
    tp::DBuf<double> rates;       // 3*nsrc
    // linear algebra
    std::vector<tp::DBuf<double> *> vecs;
    tp::DBuf<double> At;          // decoupled primary block (QI/TI); planes as in J with b' = nprimary
    tp::DBuf<double> dcoef;       // decoupling coefficients d_q per cell (nprimary planes)
    tp::Stencil opA00, opA01, opA10;   // views used by stage 1
    tp::Amg *amg_p = nullptr, *amg_T = nullptr;
    tp::BAmg *bamg = nullptr;          // pc_cptramg: system AMG on the (p,T) blocks (tp_amg_block.hip)
    tp::DBuf<double> spbuf;            // selfp (schur_a11 == 2): S7 (7 planes), w/diag(Sp), two work planes
    tp::DBuf<double> gAt;              // multi-GPU pc_cptramg: the 28 operator planes gathered on the global grid
    tp::IluData ilu;
    // FGMRES workspace
    tp::DBuf<double> V, Z, gs_partial, gs_h, red_out;
    int gs_cap = 0;
    std::vector<double> hostbuf;
    // scratch vectors for PC apply
    tp::DBuf<double> w1, w2, w3, w4, dx;
    // captured preconditioner application (hipGraph on fixed staging buffers)
    // captured pc_apply graphs, one per (input, output) vector pair: FGMRES applies the preconditioner to basis
    // vector j into Z_j, a handful of fixed address pairs that recur in every solve
    struct PcGraph { const double *x; double *y; hipGraphExec_t exec; };
    std::vector<PcGraph> pc_graphs;
    // multi-GPU: a preconditioner application is recorded as a PROGRAM -- hipGraph segments (the kernel sequences between two
    // exchanges) alternating with the exchanges themselves (RCCL calls / in-process copies, replayed as host closures on the
    // same stream).  RCCL calls are never captured; only what lies between them is.
    struct PcStep { hipGraphExec_t exec; std::function<void()> comm; };
    struct PcProgram { const double *x; double *y; std::vector<PcStep> steps; };
    std::vector<PcProgram> pc_programs;
    PcProgram *rec = nullptr;          // program being recorded (comm calls split the capture), else null
    bool rec_capturing = false, rec_in_comm = false;
    uint64_t graph_epoch = 1, pc_graph_epoch = 0;
    uintptr_t pc_sig = 0;
    // comm: RCCL communicator (one process per GPU) or an in-process slab group (several contexts on one GPU,
    // used to validate the slab algorithm where only one GPU is available)
    ncclComm *comm = nullptr;
    tp::LocalGroup *lgroup = nullptr;
    bool dist = false;
    // multi-GPU stage 1: the pressure (and temperature) systems gathered on the global grid of every rank
    tp::GridDev gfull;
    tp::DBuf<double> gA00, gA01, gA10, gSm, gvec;   // operators: 7 planes each; gvec: work vectors
    long vcycles = 0;
    long spec_issued = 0, spec_wasted = 0, spec_skipped = 0;   // pipelined FGMRES loop: speculative applications issued / discarded / iterations run without one (TP_DEBUG)
    static constexpr int H_PIN = 1024;   // doubles of pinned, device-mapped host memory the reductions write their results to
    double *h_pin = nullptr;
    hipEvent_t ev_h = nullptr;           // recorded behind the reductions of an orthogonalisation (pipelined FGMRES loop)
    long gather_override = -2;     // != -2: replaces tp_options.amg_gather_cells in amg_build (selfp on several GPUs: 0)
    tp_ksp_monitor_fn monitor = nullptr;      // per-field true-residual monitor (ksp_monitor_residuals)
    void *monitor_user = nullptr;
    ~tp_ctx();
};

namespace tp {
// Host <-> device copy ordered on the context's stream (after everything queued there) and complete on return.  The context's
// streams are non-blocking and no call goes to the legacy default stream (see DBuf::alloc).
inline void copy_sync(tp_ctx *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    TP_HIP(hipMemcpyAsync(dst, src, bytes, kind, c->stream));
    TP_HIP(hipStreamSynchronize(c->stream));
}
// ---- launch wrappers implemented in the .hip files ------------------------------------------------
// assembly
void compute_trans(tp_ctx *c);
void accum_old(tp_ctx *c);
void assemble(tp_ctx *c, bool want_jac, bool want_schur);
void well_rates(tp_ctx *c);
// vectors / reductions (vectors are b field planes of ntot; reductions run over owned cells only)
void vec_zero(tp_ctx *c, double *x, long n);
void vec_copy(tp_ctx *c, const double *x, double *y, long n);
void vec_axpy_owned(tp_ctx *c, int nf, double a, const double *x, double *y);            // y += a x
void vec_scale_to(tp_ctx *c, int nf, double a, const double *x, double *y);             // y = a x (owned)
void multi_dot(tp_ctx *c, int nf, const double *V, long vstride, int k, const double *w, const double *w2,
               double *host_out);   // host_out[i] = <V_i, w>, i<k ; host_out[k] = <w2,w2> if w2
void multi_axpy(tp_ctx *c, int nf, const double *V, long vstride, int k, const double *hcoef_host, double sign,
                double *w);          // w += sign * sum_i h_i V_i
double norm2(tp_ctx *c, int nf, const double *x);
void multi_norm2sq(tp_ctx *c, int nf, int nvec, const double *const *x, double *host_out);   // host_out[i] = <x_i, x_i>
// h = V^T w ; w -= V h ; host_out[0..k-1] = h, host_out[k] = ||w||^2  (one host sync)
void orthogonalize(tp_ctx *c, int nf, const double *V, long vstride, int k, double *w, double *host_out);
bool orthogonalize_can_split(const tp_ctx *c, int k);
void orthogonalize_enqueue(tp_ctx *c, int nf, const double *V, long vstride, int k, double *w);
const double *orthogonalize_norm_dev(const tp_ctx *c, int k);
void orthogonalize_wait(tp_ctx *c, int k, double *host_out);
void vec_scale_dev_norm(tp_ctx *c, int nf, const double *n2_dev, double *x);   // x *= 1/sqrt(*n2_dev) (owned)
void field_minmax(tp_ctx *c, const double *x, double *lo, double *hi);
void field_clamp01(tp_ctx *c, double *x);
// stencil operators
void spmv_block(tp_ctx *c, const double *J, const double *x, double *y);
void spmv_block_halo(tp_ctx *c, const double *J, double *x, double *y);          // exchange x's halos, y = J x; interior overlaps the exchange                 // y = J x
void resid_block_cols(tp_ctx *c, const double *J, const double *x, const double *y, int ncols, double *r);  // r = x - J[:, :ncols] y
void spmv_scalar(tp_ctx *c, const GridDev &g, const Stencil &A, const double *x, double *y, double alpha, const double *z);  // y = z + alpha*A x (z may be null)
void decouple(tp_ctx *c);
void selfp_build(tp_ctx *c);                                                           // S7, w/diag(Sp) into spbuf
void selfp_post(tp_ctx *c, const double *b, const double *x, double *y);                  // y = x + w D^-1 (b - Sp x)
void stage1_rhs(tp_ctx *c, const double *x, int q, double *out);                        // out = x_q - d_q x_s
// ILU
void ilu_setup(tp_ctx *c);
void ilu_factor(tp_ctx *c);
// x = addto + M^-1 r ; only the first nadd fields of addto are read, the others count as zero (< 0: all fields)
void ilu_solve(tp_ctx *c, const double *r, double *x, const double *addto, int nadd = -1);
// AMG
void amg_build(tp_ctx *c, Amg *&amg, const GridDev &g0, const double strength[3]);
void amg_setup(tp_ctx *c, Amg *amg, const Stencil &A0);
void amg_vcycle(tp_ctx *c, Amg *amg, const double *b, double *x);
bool amg_resolve_trunc(tp_ctx *c, Amg *amg);      // waits for the set-up's dominance ratios; true if the cycle shape changed
// system AMG (2x2 blocks on (p,T))
void bamg_build(tp_ctx *c, BAmg *&amg, const GridDev &g0, const double strength[3]);
void bamg_setup(tp_ctx *c, BAmg *amg, const BStencil &A0);
void bamg_vcycle(tp_ctx *c, BAmg *amg, const double *b, double *x);     // b, x: 2 planes, stride = ntot of the grid
void bamg_destroy(BAmg *amg);
int bamg_levels(const BAmg *amg);
int bamg_dist_levels(const BAmg *amg);
const std::vector<int> &bamg_sched(const BAmg *amg);
// comm
void halo_exchange(tp_ctx *c, const GridDev &g, double *x, int nf, long fstride);
void halo_exchange_raw(tp_ctx *c, const GridDev &g, void *x, int nf, size_t fstride_bytes, size_t elem_bytes);
void gather_ranges(tp_ctx *c, void *global, long np, const std::vector<std::pair<int, int>> &ranges, int nslots,
                   size_t slot_stride_bytes, size_t elem_bytes);
void allreduce_sum(tp_ctx *c, double *dev, int n);
void allreduce_max(tp_ctx *c, double *dev, int n);
void slab_of(const tp_ctx *c, int rank, int &lo, int &hi);
// recording of pc_apply programs (tp_solver.hip): close / reopen the current stream-capture segment around an exchange
void seg_begin(tp_ctx *c);
void seg_end(tp_ctx *c);
// gather `nplanes` slab-distributed cell planes into arrays on the global grid (every rank gets all slabs)
void gather_slabs(tp_ctx *c, const double *local, long lstride, double *global, long gstride, int nplanes);
// solver
void resolve_cycle_shapes(tp_ctx *c);     // fix the AMG truncation levels after a set-up (host wait; never in a capture)
void ensure_work(tp_ctx *c);          // scratch vectors w1..w4, dx of the preconditioner / Krylov loops
void pc_setup(tp_ctx *c);
void stage1_apply(tp_ctx *c, const double *x, double *y, bool zero_secondary = true);
void pc_apply(tp_ctx *c, const double *x, double *y);
int fgmres(tp_ctx *c, const double *b, double *x, int *its, double *rnorm);
void newton(tp_ctx *c, tp_solve_info *info);
}  // namespace tp

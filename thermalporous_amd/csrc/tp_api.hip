// extern "C" surface of libthermalporous_hip.so (declared in include/thermalporous_hip.h) and the RCCL
// slab communication (halo exchange over xGMI point-to-point links, batched dot-product all-reduce).
#include "tp_common.hpp"
#include <rccl/rccl.h>
#include <cmath>
#include <algorithm>
#include <map>
#include <utility>
#include <mutex>
#include <condition_variable>

namespace tp {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

#define TP_NCCL(call)                                                                          \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess)                                                                 \
            throw tp::Error(std::string(#call) + " failed: " + ncclGetErrorString(r_));        \
    } while (0)

// ---- in-process slab group: N contexts (threads) on one GPU exchange through device-to-device copies ------
// Same call sequence and same buffer arithmetic as the RCCL path below; exists so that the slab
// algorithm can be validated end to end on a single GPU (tests/test_gpu_slabs.py).
struct LocalGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    std::vector<const void *> ptr;       // published buffer of every rank
    std::vector<std::vector<double>> red;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const long gen = generation;
        if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};

static void lg_publish(tp_ctx *c, const void *p) {
    TP_HIP(hipStreamSynchronize(c->stream));          // my data is complete before anybody reads it
    c->lgroup->ptr[c->grid.rank] = p;
    c->lgroup->barrier();
}
static void lg_done(tp_ctx *c) {
    TP_HIP(hipStreamSynchronize(c->stream));          // my copies are complete before sources may change
    c->lgroup->barrier();
}

// One 1-cell halo plane per side along axis 2; each plane of each field is contiguous, so the
// exchange is 2*nf send/recv pairs in one RCCL group on the compute stream (<= 2 neighbours, one
// xGMI link each).  Element size 8 (vectors, fp64 operators) or 4 (fp32 AMG operators); `g` is the
// grid of the array being exchanged (the slab itself, or one of its distributed AMG levels).
struct HaloPub { const char *p; size_t fstride; int n2; };

// While a pc_apply program is being recorded (tp_solver.hip) an exchange closes the current capture segment (which is
// instantiated and launched, so the recording pass computes the real result), runs eagerly, is remembered as a host
// closure, and a new segment is opened behind it.  `call` re-enters the same function with recording suspended.
#define TP_COMM_RECORD(c, call)                                                     \
    do {                                                                            \
        if ((c)->rec && !(c)->rec_in_comm) {                                        \
            seg_end(c);                                                             \
            (c)->rec->steps.push_back({nullptr, [=]() { call; }});                  \
            (c)->rec_in_comm = true;                                                \
            try { call; } catch (...) { (c)->rec_in_comm = false; throw; }          \
            (c)->rec_in_comm = false;                                               \
            seg_begin(c);                                                           \
            return;                                                                 \
        }                                                                           \
    } while (0)


void halo_exchange_raw(tp_ctx *c, const GridDev &g, void *x_, int nf, size_t fstride, size_t elem) {
    if (!c->dist) return;
    TP_REQUIRE(g.n2 >= 1, "halo exchange of an empty slab");
    TP_COMM_RECORD(c, halo_exchange_raw(c, g, x_, nf, fstride, elem));
    char *x = (char *)x_;
    const int lo = c->grid.rank - 1, hi = c->grid.rank + 1;
    const size_t pb = (size_t)g.np * elem;       // bytes per plane
    if (c->lgroup) {
        HaloPub mine{x, fstride, g.n2};
        lg_publish(c, &mine);
        for (int f = 0; f < nf; ++f) {
            char *p = x + (size_t)f * fstride;
            if (g.nb_lo) {       // neighbour's last owned plane -> my lower halo
                const HaloPub *q = (const HaloPub *)c->lgroup->ptr[lo];
                TP_HIP(hipMemcpyAsync(p, q->p + (size_t)f * q->fstride + pb * q->n2, pb, hipMemcpyDeviceToDevice, c->stream));
            }
            if (g.nb_hi) {       // neighbour's first owned plane -> my upper halo
                const HaloPub *q = (const HaloPub *)c->lgroup->ptr[hi];
                TP_HIP(hipMemcpyAsync(p + pb * (g.n2 + 1), q->p + (size_t)f * q->fstride + pb, pb, hipMemcpyDeviceToDevice, c->stream));
            }
        }
        lg_done(c);
        return;
    }
    ncclComm_t comm = (ncclComm_t)c->comm;
    const ncclDataType_t dt = elem == 8 ? ncclDouble : ncclFloat;
    TP_REQUIRE(elem == 8 || elem == 4, "halo element size");
    TP_NCCL(ncclGroupStart());
    for (int f = 0; f < nf; ++f) {
        char *p = x + (size_t)f * fstride;
        if (g.nb_lo) {
            TP_NCCL(ncclSend(p + pb, g.np, dt, lo, comm, c->stream));                      // first owned plane
            TP_NCCL(ncclRecv(p, g.np, dt, lo, comm, c->stream));                           // lower halo
        }
        if (g.nb_hi) {
            TP_NCCL(ncclSend(p + pb * g.n2, g.np, dt, hi, comm, c->stream));               // last owned plane
            TP_NCCL(ncclRecv(p + pb * (g.n2 + 1), g.np, dt, hi, comm, c->stream));         // upper halo
        }
    }
    TP_NCCL(ncclGroupEnd());
}

void halo_exchange(tp_ctx *c, const GridDev &g, double *x, int nf, long fstride) {
    halo_exchange_raw(c, g, x, nf, (size_t)fstride * sizeof(double), sizeof(double));
}

// In-place all-gather on a global-grid array every rank holds: rank r owns global planes
// [ranges[r].first, ranges[r].second) of each of the nslots planes-sets and broadcasts them to everybody.
void gather_ranges(tp_ctx *c, void *global_, long np, const std::vector<std::pair<int, int>> &ranges, int nslots,
                   size_t slot_stride, size_t elem) {
    TP_REQUIRE(c->dist, "gather_ranges without a communicator");
    TP_COMM_RECORD(c, gather_ranges(c, global_, np, ranges, nslots, slot_stride, elem));
    char *global = (char *)global_;
    const size_t pb = (size_t)np * elem;
    if (c->lgroup) {
        lg_publish(c, global);
        for (int r = 0; r < c->grid.nranks; ++r) {
            if (r == c->grid.rank) continue;
            const char *src = (const char *)c->lgroup->ptr[r];
            const size_t off = pb * (ranges[r].first + 1), len = pb * (ranges[r].second - ranges[r].first);
            for (int s = 0; s < nslots; ++s)
                TP_HIP(hipMemcpyAsync(global + s * slot_stride + off, src + s * slot_stride + off, len,
                                      hipMemcpyDeviceToDevice, c->stream));
        }
        lg_done(c);
        return;
    }
    ncclComm_t comm = (ncclComm_t)c->comm;
    const ncclDataType_t dt = elem == 8 ? ncclDouble : ncclFloat;
    TP_NCCL(ncclGroupStart());
    for (int s = 0; s < nslots; ++s)
        for (int r = 0; r < c->grid.nranks; ++r) {
            char *buf = global + s * slot_stride + pb * (ranges[r].first + 1);
            TP_NCCL(ncclBroadcast(buf, buf, (size_t)np * (ranges[r].second - ranges[r].first), dt, r, comm, c->stream));
        }
    TP_NCCL(ncclGroupEnd());
}

void slab_of(const tp_ctx *c, int rank, int &lo, int &hi) {
    const int base = c->grid.gn2 / c->grid.nranks, rem = c->grid.gn2 % c->grid.nranks;
    lo = rank * base + std::min(rank, rem);
    hi = lo + base + (rank < rem ? 1 : 0);
}

// Every rank broadcasts the owned part of each of its planes into the same place of everybody's global
// array (uneven slabs: one ncclBroadcast per (plane, root) inside one group).
void gather_slabs(tp_ctx *c, const double *local, long lstride, double *global, long gstride, int nplanes) {
    TP_REQUIRE(c->dist, "gather_slabs without a communicator");
    TP_COMM_RECORD(c, gather_slabs(c, local, lstride, global, gstride, nplanes));
    const long np = c->g.np;
    if (c->lgroup) {
        // NOTE: lstride is the same on every rank only when it is expressed in this rank's ntot; ranks
        // publish (pointer, stride) pairs
        struct Pub { const double *p; long stride; };
        Pub mine{local, lstride};
        lg_publish(c, &mine);
        for (int r = 0; r < c->grid.nranks; ++r) {
            int lo, hi;
            slab_of(c, r, lo, hi);
            const Pub *pr = (const Pub *)c->lgroup->ptr[r];
            for (int p = 0; p < nplanes; ++p)
                TP_HIP(hipMemcpyAsync(global + (long)p * gstride + np * (lo + 1), pr->p + (long)p * pr->stride + np,
                                      sizeof(double) * np * (hi - lo), hipMemcpyDeviceToDevice, c->stream));
        }
        lg_done(c);
        return;
    }
    ncclComm_t comm = (ncclComm_t)c->comm;
    TP_NCCL(ncclGroupStart());
    for (int p = 0; p < nplanes; ++p)
        for (int r = 0; r < c->grid.nranks; ++r) {
            int lo, hi;
            slab_of(c, r, lo, hi);
            double *dst = global + (long)p * gstride + np * (lo + 1);
            // sendbuff is only read on the root; elsewhere pass the (valid, large enough) receive buffer
            const double *src = (r == c->grid.rank) ? local + (long)p * lstride + np : dst;
            TP_NCCL(ncclBroadcast(src, dst, (size_t)np * (hi - lo), ncclDouble, r, comm, c->stream));
        }
    TP_NCCL(ncclGroupEnd());
}

void allreduce_sum(tp_ctx *c, double *dev, int n) {
    if (!c->dist || n <= 0) return;
    TP_COMM_RECORD(c, allreduce_sum(c, dev, n));
    if (c->lgroup) {
        LocalGroup *G = c->lgroup;
        std::vector<double> &mine = G->red[c->grid.rank];
        mine.resize(n);
        TP_HIP(hipMemcpyAsync(mine.data(), dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
        TP_HIP(hipStreamSynchronize(c->stream));
        G->barrier();
        std::vector<double> sum(n, 0.0);
        for (int r = 0; r < G->n; ++r)                 // fixed order: identical result on every rank
            for (int i = 0; i < n; ++i) sum[i] += G->red[r][i];
        G->barrier();                                  // everybody has read before anybody overwrites
        TP_HIP(hipMemcpyAsync(dev, sum.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        TP_HIP(hipStreamSynchronize(c->stream));
        return;
    }
    TP_NCCL(ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream));
}

// element-wise maximum over the ranks (non-negative doubles: the AMG dominance ratios)
void allreduce_max(tp_ctx *c, double *dev, int n) {
    if (!c->dist || n <= 0) return;
    TP_COMM_RECORD(c, allreduce_max(c, dev, n));
    if (c->lgroup) {
        LocalGroup *G = c->lgroup;
        std::vector<double> &mine = G->red[c->grid.rank];
        mine.resize(n);
        TP_HIP(hipMemcpyAsync(mine.data(), dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
        TP_HIP(hipStreamSynchronize(c->stream));
        G->barrier();
        std::vector<double> mx(n, 0.0);
        for (int r = 0; r < G->n; ++r)
            for (int i = 0; i < n; ++i) mx[i] = std::max(mx[i], G->red[r][i]);
        G->barrier();                                  // everybody has read before anybody overwrites
        TP_HIP(hipMemcpyAsync(dev, mx.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        TP_HIP(hipStreamSynchronize(c->stream));
        return;
    }
    TP_NCCL(ncclAllReduce(dev, dev, n, ncclDouble, ncclMax, (ncclComm_t)c->comm, c->stream));
}

}  // namespace tp

tp_ctx::~tp_ctx() {
    if (getenv("TP_DEBUG") && spec_issued + spec_skipped > 0)
        fprintf(stderr, "[tp] pipelined FGMRES: %ld speculative applications issued, %ld discarded, %ld iterations without one\n",
                spec_issued, spec_wasted, spec_skipped);
    for (auto *v : vecs) delete v;
    delete amg_p;
    delete amg_T;
    if (bamg) tp::bamg_destroy(bamg);
    for (auto &gph : pc_graphs) (void)hipGraphExecDestroy(gph.exec);
    for (auto &pr : pc_programs)
        for (auto &st : pr.steps)
            if (st.exec) (void)hipGraphExecDestroy(st.exec);
    if (comm) ncclCommDestroy((ncclComm_t)comm);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    for (int i = 0; i < 2; ++i) {
        if (aux[i]) (void)hipStreamDestroy(aux[i]);
        if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
    }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (stream) (void)hipStreamDestroy(stream);
    if (h_pin) (void)hipHostFree(h_pin);
    if (ev_h) (void)hipEventDestroy(ev_h);
}

using namespace tp;

#define TP_API_BEGIN try {
#define TP_API_END                                   \
    return 0;                                        \
    }                                                \
    catch (const std::exception &e) {                \
        tp::set_error(e.what());                     \
        return -1;                                   \
    }                                                \
    catch (...) {                                    \
        tp::set_error("unknown C++ exception");      \
        return -1;                                   \
    }

static void derive_params(tp_ctx *c) {
    const tp_params &p = c->prm;
    DevPrm &d = c->dprm;
    d.ko = p.ko; d.kw = p.kw; d.kr = p.kr; d.c_v_w = p.c_v_w; d.c_v_o = p.c_v_o; d.c_r = p.c_r;
    d.rho_r = p.rho_r; d.T_inj = p.T_inj; d.g = p.g; d.U = p.U;
    const double SG = 141.5 / (p.API + 131.5);               // physicalparameters.py:39-40
    d.rho_ref = SG * 999.0;
    d.mu_o_coef = 1e-3 * std::pow(10.0, -0.8021 * p.API + 23.8765);   // :52-57
    d.mu_o_exp = 0.31458 * p.API + (-9.21592);
    if (c->nph == 2) {                                        // twophase.py:142-147
        d.w0 = p.T_prod;
        d.w2 = p.T_prod * (p.c_v_w * (1.0 - p.S_o) + p.c_v_o * p.S_o);
    } else {                                                  // m_w = 1 (singlephase.py:26,112-115)
        d.w0 = 1.0;
        d.w2 = 0.0;
    }
}

static DBuf<double> &vec_of(tp_ctx *c, int id) {
    TP_REQUIRE(id >= 0 && id < (int)c->vecs.size(), "bad vector id");
    return *c->vecs[id];
}

extern "C" {

const char *tp_last_error(void) { return tp::g_err.c_str(); }
int tp_version(void) { return 100; }

int tp_create(const tp_grid *grid, const tp_params *prm, const tp_options *opt, int device, tp_ctx **out) {
    TP_API_BEGIN
    TP_REQUIRE(grid && prm && opt && out, "null argument");
    TP_REQUIRE(grid->n0 >= 1 && grid->n1 >= 1 && grid->n2 >= 1, "empty grid");
    TP_REQUIRE(grid->nphase == 1 || grid->nphase == 2, "nphase must be 1 or 2");
    TP_REQUIRE(grid->off2 >= 0 && grid->off2 + grid->n2 <= grid->gn2, "slab outside the global grid");
    TP_REQUIRE((long)grid->n0 * grid->n1 * (grid->n2 + 2) < (1L << 31), "slab too large for int32 block indices");
    int ndev = 0;
    TP_HIP(hipGetDeviceCount(&ndev));
    TP_REQUIRE(ndev > 0, "no HIP device: the thermalporous hot path has no CPU fallback");
    TP_REQUIRE(device >= 0 && device < ndev, "bad device ordinal");
    TP_HIP(hipSetDevice(device));
    tp_ctx *c = new tp_ctx();
    c->grid = *grid; c->prm = *prm; c->opt = *opt; c->device = device;
    c->nph = grid->nphase; c->b = grid->nphase + 1;
    c->g = make_grid(grid->n0, grid->n1, grid->n2, grid->gn2, grid->off2);
    c->gfull = make_grid(grid->n0, grid->n1, grid->gn2, grid->gn2, 0);
    c->vol = grid->h[0] * grid->h[1] * grid->h[2];
    derive_params(c);
    TP_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));      // (never tied to the legacy stream: tp_common.hpp)
    // coherent (fine-grained) host memory and a system-scope release at the event: the host reads what the reduction kernel
    // wrote after waiting for ev_h only, not for the stream
    TP_HIP(hipHostMalloc((void **)&c->h_pin, sizeof(double) * tp_ctx::H_PIN, hipHostMallocMapped | hipHostMallocCoherent));
    TP_HIP(hipEventCreateWithFlags(&c->ev_h, hipEventDisableTiming | hipEventReleaseToSystem));
    for (int i = 0; i < 2; ++i) {
        TP_HIP(hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking));
        TP_HIP(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    TP_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    TP_HIP(hipEventCreate(&c->ev0));
    TP_HIP(hipEventCreate(&c->ev1));
    const size_t nt = (size_t)c->g.ntot, B = (size_t)c->b;
    c->phi.alloc(nt); c->kTs.alloc(nt);
    for (int a = 0; a < 3; ++a) { c->K[a].alloc(nt); c->TK[a].alloc(nt); }
    c->u.alloc(B * nt); c->u_old.alloc(B * nt); c->acc_old.alloc(B * nt); c->R.alloc(B * nt);
    c->J.alloc(7 * B * B * nt);
    if (schur_of(*opt)) c->Sm.alloc(7 * nt);
    *out = c;
    TP_API_END
}

int tp_destroy(tp_ctx *ctx) {
    TP_API_BEGIN
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        // (this context's streams only: a device-wide synchronisation would wait for -- and, while one of them captures, fail on --
        // the other slab contexts of an in-process group)
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        for (int i = 0; i < 2; ++i)
            if (ctx->aux[i]) (void)hipStreamSynchronize(ctx->aux[i]);
        delete ctx;
    }
    TP_API_END
}

int tp_set_options(tp_ctx *c, const tp_options *opt) {
    TP_API_BEGIN
    TP_REQUIRE(c && opt, "null argument");
    TP_REQUIRE(opt->ilu_levels == 0 || opt->ilu_levels == 1, "ilu_levels must be 0 or 1");
    const bool tile_changed = opt->ilu_t1 != c->opt.ilu_t1 || opt->ilu_t2 != c->opt.ilu_t2 || opt->ilu_t0 != c->opt.ilu_t0 ||
                              opt->ilu_levels != c->opt.ilu_levels || opt->ilu_whole != c->opt.ilu_whole;
    const bool amg_changed = opt->amg_min_cells != c->opt.amg_min_cells || opt->pc_kind != c->opt.pc_kind ||
                             opt->amg_nu != c->opt.amg_nu || opt->amg_full_levels != c->opt.amg_full_levels ||
                             opt->amg_coarse_pre != c->opt.amg_coarse_pre || opt->amg_coarse_post != c->opt.amg_coarse_post ||
                             opt->amg_tail_post != c->opt.amg_tail_post || opt->amg_mid_skip != c->opt.amg_mid_skip ||
                             opt->amg_dom_tau != c->opt.amg_dom_tau ||
                             opt->amg_single != c->opt.amg_single || opt->amg_gather_cells != c->opt.amg_gather_cells ||
                             opt->schur_a11 != c->opt.schur_a11 || opt->fs_additive != c->opt.fs_additive;
    c->opt = *opt;
    if (tile_changed) c->ilu.slots = 0;
    if (amg_changed) {
        delete c->amg_p; c->amg_p = nullptr; delete c->amg_T; c->amg_T = nullptr;
        if (c->bamg) { bamg_destroy(c->bamg); c->bamg = nullptr; }
    }
    if (schur_of(*opt) && c->Sm.n == 0) c->Sm.alloc((size_t)7 * c->g.ntot);
    c->pc_ready = false;
    c->graph_epoch++;            // any option may change the captured kernel sequence: drop the pc_apply graphs
    TP_API_END
}

int tp_comm_unique_id(void *id128) {
    TP_API_BEGIN
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    TP_NCCL(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    TP_API_END
}

int tp_comm_init(tp_ctx *c, const void *id128) {
    TP_API_BEGIN
    TP_REQUIRE(c && id128, "null argument");
    TP_REQUIRE(c->grid.nranks >= 1, "bad nranks");     // a 1-rank communicator is legal (exercises the RCCL calls on one GPU)
    TP_HIP(hipSetDevice(c->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t comm;
    TP_NCCL(ncclCommInitRank(&comm, c->grid.nranks, id, c->grid.rank));
    c->comm = (ncclComm *)comm;
    c->dist = true;
    int lo, hi;
    slab_of(c, c->grid.rank, lo, hi);
    TP_REQUIRE(lo == c->grid.off2 && hi - lo == c->grid.n2, "slab of this rank does not follow the library's partition rule");
    TP_API_END
}

int tp_local_group_create(int32_t nranks, void **group) {
    TP_API_BEGIN
    TP_REQUIRE(nranks >= 2 && group, "bad arguments");
    LocalGroup *G = new LocalGroup();
    G->n = nranks;
    G->ptr.assign(nranks, nullptr);
    G->red.resize(nranks);
    *group = G;
    TP_API_END
}

int tp_local_group_destroy(void *group) {
    TP_API_BEGIN
    delete (LocalGroup *)group;
    TP_API_END
}

int tp_comm_init_local(tp_ctx *c, void *group) {
    TP_API_BEGIN
    TP_REQUIRE(c && group, "null argument");
    LocalGroup *G = (LocalGroup *)group;
    TP_REQUIRE(c->grid.nranks == G->n, "group size differs from the context's nranks");
    int lo, hi;
    slab_of(c, c->grid.rank, lo, hi);
    TP_REQUIRE(lo == c->grid.off2 && hi - lo == c->grid.n2, "slab of this rank does not follow the library's partition rule");
    c->lgroup = G;
    c->dist = true;
    TP_API_END
}

int tp_set_field(tp_ctx *c, const char *name, const double *host, int64_t n) {
    TP_API_BEGIN
    TP_REQUIRE(c && name && host, "null argument");
    TP_REQUIRE(n == c->g.ntot, "field must have n0*n1*(n2+2) entries (slab with halo planes)");
    DBuf<double> *dst = nullptr;
    const std::string s(name);
    if (s == "phi") dst = &c->phi;
    else if (s == "kT") dst = &c->kTs;
    else if (s == "K0") dst = &c->K[0];
    else if (s == "K1") dst = &c->K[1];
    else if (s == "K2") dst = &c->K[2];
    TP_REQUIRE(dst, "unknown field name (phi, kT, K0, K1, K2)");
    copy_sync(c, dst->p, host, sizeof(double) * n, hipMemcpyHostToDevice);
    c->fields_ready = false;
    TP_API_END
}

int tp_finalize_fields(tp_ctx *c) {
    TP_API_BEGIN
    compute_trans(c);
    TP_HIP(hipStreamSynchronize(c->stream));
    c->fields_ready = true;
    delete c->amg_p; c->amg_p = nullptr;
    delete c->amg_T; c->amg_T = nullptr;
    if (c->bamg) { bamg_destroy(c->bamg); c->bamg = nullptr; }
    c->pc_ready = false;
    c->graph_epoch++;
    TP_API_END
}

int tp_set_sources(tp_ctx *c, int32_t n, const tp_source *entries) {
    TP_API_BEGIN
    TP_REQUIRE(c && (n == 0 || entries), "null argument");
    std::vector<tp_source> v(entries, entries + n);
    const GridDev &g = c->g;
    for (auto &e : v) {
        TP_REQUIRE(e.cell >= g.np && e.cell < g.np + g.nown, "source cell is not an owned cell of this slab");
        TP_REQUIRE(e.kind >= 0 && e.kind <= 2, "source kind must be 0 (prod), 1 (inj) or 2 (heater)");
    }
    std::stable_sort(v.begin(), v.end(), [](const tp_source &a, const tp_source &b) { return a.cell < b.cell; });
    std::vector<int> start;
    for (int i = 0; i < n; ++i)
        if (i == 0 || v[i].cell != v[i - 1].cell) start.push_back(i);
    c->nsrc = n;
    c->nsrc_groups = (int)start.size();
    start.push_back(n);
    c->src.alloc(std::max(1, n));
    c->src_start.alloc(start.size());
    c->rates.alloc((size_t)3 * std::max(1, n));
    if (n) copy_sync(c, c->src.p, v.data(), sizeof(tp_source) * n, hipMemcpyHostToDevice);
    copy_sync(c, c->src_start.p, start.data(), sizeof(int) * start.size(), hipMemcpyHostToDevice);
    TP_API_END
}

int tp_set_state(tp_ctx *c, const double *u_host) {
    TP_API_BEGIN
    TP_REQUIRE(c && u_host, "null argument");
    copy_sync(c, c->u.p, u_host, sizeof(double) * c->u.n, hipMemcpyHostToDevice);
    TP_API_END
}

int tp_get_state(tp_ctx *c, double *u_host) {
    TP_API_BEGIN
    TP_REQUIRE(c && u_host, "null argument");
    TP_HIP(hipStreamSynchronize(c->stream));
    copy_sync(c, u_host, c->u.p, sizeof(double) * c->u.n, hipMemcpyDeviceToHost);
    TP_API_END
}

int tp_set_old_state(tp_ctx *c, const double *u_host) {
    TP_API_BEGIN
    TP_REQUIRE(c, "null argument");
    TP_REQUIRE(c->fields_ready, "fields not finalised");
    if (u_host) copy_sync(c, c->u_old.p, u_host, sizeof(double) * c->u.n, hipMemcpyHostToDevice);
    else vec_copy(c, c->u.p, c->u_old.p, (long)c->u.n);
    accum_old(c);
    c->have_old = true;
    TP_API_END
}

int tp_set_dt(tp_ctx *c, double dt) {
    TP_API_BEGIN
    TP_REQUIRE(c && dt > 0.0, "dt must be positive");
    c->dt = dt;
    TP_API_END
}

int tp_get_old_state(tp_ctx *c, double *u_host) {
    TP_API_BEGIN
    TP_REQUIRE(c && u_host, "null argument");
    TP_HIP(hipStreamSynchronize(c->stream));
    copy_sync(c, u_host, c->u_old.p, sizeof(double) * c->u_old.n, hipMemcpyDeviceToHost);
    TP_API_END
}

int tp_restore_state(tp_ctx *c) {
    TP_API_BEGIN
    TP_REQUIRE(c && c->have_old, "no old state to restore");
    vec_copy(c, c->u_old.p, c->u.p, (long)c->u.n);
    TP_API_END
}

int tp_saturation_range(tp_ctx *c, double *smin, double *smax) {
    TP_API_BEGIN
    TP_REQUIRE(c && smin && smax, "null argument");
    TP_REQUIRE(c->nph == 2, "saturation exists only in the two-phase model");
    field_minmax(c, c->u.p + 2 * c->g.ntot, smin, smax);
    TP_API_END
}

int tp_clamp_saturation(tp_ctx *c) {
    TP_API_BEGIN
    TP_REQUIRE(c && c->nph == 2, "saturation exists only in the two-phase model");
    field_clamp01(c, c->u.p + 2 * c->g.ntot);
    TP_API_END
}

int tp_residual(tp_ctx *c, double *norm2_out) {
    TP_API_BEGIN
    if (c->dist) halo_exchange(c, c->g, c->u.p, c->b, c->g.ntot);
    assemble(c, false, false);
    const double n = norm2(c, c->b, c->R.p);
    if (norm2_out) *norm2_out = n;
    TP_API_END
}

int tp_jacobian(tp_ctx *c) {
    TP_API_BEGIN
    if (c->dist) halo_exchange(c, c->g, c->u.p, c->b, c->g.ntot);
    assemble(c, true, schur_of(c->opt));
    c->pc_ready = false;
    TP_HIP(hipStreamSynchronize(c->stream));
    TP_API_END
}

int tp_get_residual(tp_ctx *c, double *host) {
    TP_API_BEGIN
    TP_HIP(hipStreamSynchronize(c->stream));
    copy_sync(c, host, c->R.p, sizeof(double) * c->R.n, hipMemcpyDeviceToHost);
    TP_API_END
}

int tp_export_jacobian(tp_ctx *c, double *host) {
    TP_API_BEGIN
    TP_REQUIRE(c->jac_ready, "Jacobian not assembled");
    TP_HIP(hipStreamSynchronize(c->stream));
    copy_sync(c, host, c->J.p, sizeof(double) * c->J.n, hipMemcpyDeviceToHost);
    TP_API_END
}

int tp_export_schur(tp_ctx *c, double *host) {
    TP_API_BEGIN
    TP_REQUIRE(c->Sm.n > 0 && c->jac_ready, "S~ not assembled (pc_cptr only)");
    TP_HIP(hipStreamSynchronize(c->stream));
    copy_sync(c, host, c->Sm.p, sizeof(double) * c->Sm.n, hipMemcpyDeviceToHost);
    TP_API_END
}

int tp_well_rates(tp_ctx *c, double *rate, double *water_rate, double *oil_rate) {
    TP_API_BEGIN
    if (c->nsrc == 0) return 0;
    well_rates(c);
    TP_HIP(hipStreamSynchronize(c->stream));
    const size_t n = c->nsrc;
    if (rate) copy_sync(c, rate, c->rates.p, sizeof(double) * n, hipMemcpyDeviceToHost);
    if (water_rate) copy_sync(c, water_rate, c->rates.p + n, sizeof(double) * n, hipMemcpyDeviceToHost);
    if (oil_rate) copy_sync(c, oil_rate, c->rates.p + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost);
    TP_API_END
}

int tp_vec_create(tp_ctx *c, int32_t *id) {
    TP_API_BEGIN
    auto *v = new DBuf<double>();
    v->alloc((size_t)c->b * c->g.ntot);
    c->vecs.push_back(v);
    *id = (int32_t)c->vecs.size() - 1;
    TP_API_END
}

int tp_set_ksp_monitor(tp_ctx *c, tp_ksp_monitor_fn cb, void *user) {
    TP_API_BEGIN
    TP_REQUIRE(c, "null argument");
    c->monitor = cb;
    c->monitor_user = user;
    TP_API_END
}

int tp_vec_create_batch(tp_ctx *c, int32_t n, int32_t *first_id) {
    TP_API_BEGIN
    TP_REQUIRE(c && n >= 1 && first_id, "bad arguments");
    const size_t nv = (size_t)c->b * c->g.ntot;
    auto *owner = new DBuf<double>();
    owner->alloc(nv * (size_t)n);                   // vector 0 of the batch owns the allocation
    DBuf<double> *first = new DBuf<double>();
    first->p = owner->p; first->n = nv; first->owned = true;
    owner->p = nullptr; owner->n = 0;
    delete owner;
    c->vecs.push_back(first);
    *first_id = (int32_t)c->vecs.size() - 1;
    for (int i = 1; i < n; ++i) {
        auto *v = new DBuf<double>();
        v->view(first->p + (size_t)i * nv, nv);
        c->vecs.push_back(v);
    }
    TP_API_END
}

// ids first..first+n-1 must be consecutive vectors of one batch (one allocation, stride b*ntot)
static const double *batch_base(tp_ctx *c, int32_t first, int32_t n) {
    const size_t nv = (size_t)c->b * c->g.ntot;
    const double *base = vec_of(c, first).p;
    for (int i = 1; i < n; ++i)
        TP_REQUIRE(vec_of(c, first + i).p == base + (size_t)i * nv, "vectors are not consecutive members of one batch "
                   "(tp_vec_create_batch)");
    return base;
}

int tp_vec_dot_batch(tp_ctx *c, int32_t first, int32_t n, int32_t w, double *out) {
    TP_API_BEGIN
    TP_REQUIRE(c && n >= 1 && out, "bad arguments");
    const double *V = batch_base(c, first, n);
    multi_dot(c, c->b, V, (long)c->b * c->g.ntot, n, vec_of(c, w).p, nullptr, out);
    TP_API_END
}

int tp_vec_axpy_batch(tp_ctx *c, int32_t first, int32_t n, const double *coef, int32_t w) {
    TP_API_BEGIN
    TP_REQUIRE(c && n >= 1 && coef, "bad arguments");
    const double *V = batch_base(c, first, n);
    TP_REQUIRE(w < first || w >= first + n, "w must not be a member of the batch");
    multi_axpy(c, c->b, V, (long)c->b * c->g.ntot, n, coef, 1.0, vec_of(c, w).p);
    TP_API_END
}

int tp_vec_norm2(tp_ctx *c, int32_t x, double *out) {
    TP_API_BEGIN
    TP_REQUIRE(c && out, "bad arguments");
    *out = norm2(c, c->b, vec_of(c, x).p);
    TP_API_END
}

int tp_vec_set(tp_ctx *c, int32_t id, const double *host) {
    TP_API_BEGIN
    DBuf<double> &v = vec_of(c, id);
    copy_sync(c, v.p, host, sizeof(double) * v.n, hipMemcpyHostToDevice);
    TP_API_END
}

int tp_vec_get(tp_ctx *c, int32_t id, double *host) {
    TP_API_BEGIN
    DBuf<double> &v = vec_of(c, id);
    TP_HIP(hipStreamSynchronize(c->stream));
    copy_sync(c, host, v.p, sizeof(double) * v.n, hipMemcpyDeviceToHost);
    TP_API_END
}

int tp_vec_copy_residual(tp_ctx *c, int32_t id) {
    TP_API_BEGIN
    vec_copy(c, c->R.p, vec_of(c, id).p, (long)c->R.n);
    TP_API_END
}

int tp_spmv(tp_ctx *c, int32_t x, int32_t y) {
    TP_API_BEGIN
    TP_REQUIRE(c->jac_ready, "Jacobian not assembled");
    TP_REQUIRE(x != y, "tp_spmv: x and y must differ");
    if (c->dist) halo_exchange(c, c->g, vec_of(c, x).p, c->b, c->g.ntot);
    spmv_block(c, c->J.p, vec_of(c, x).p, vec_of(c, y).p);
    TP_API_END
}

int tp_pc_setup(tp_ctx *c) {
    TP_API_BEGIN
    pc_setup(c);
    TP_API_END
}

int tp_pc_apply(tp_ctx *c, int32_t x, int32_t y) {
    TP_API_BEGIN
    TP_REQUIRE(x != y, "tp_pc_apply: x and y must differ");
    pc_apply(c, vec_of(c, x).p, vec_of(c, y).p);
    TP_API_END
}

int tp_stage1_update(tp_ctx *c) {
    TP_API_BEGIN
    // CPRStage1PC.update: assemble_blocks + create_decoup + pc_schur.setOperators (AMG set-up).
    // Here PCSetUp of both stages is one call; kept separate in the API for the PCBase mirror.
    pc_setup(c);
    TP_API_END
}

int tp_stage1_apply(tp_ctx *c, int32_t x, int32_t y) {
    TP_API_BEGIN
    TP_REQUIRE(c->pc_ready, "stage 1 not set up");
    TP_REQUIRE(x != y, "x and y must differ");
    stage1_apply(c, vec_of(c, x).p, vec_of(c, y).p);
    TP_API_END
}

int tp_ilu0_factor(tp_ctx *c) {
    TP_API_BEGIN
    ilu_factor(c);
    TP_API_END
}

int tp_ilu0_solve(tp_ctx *c, int32_t x, int32_t y) {
    TP_API_BEGIN
    TP_REQUIRE(x != y, "x and y must differ");
    ilu_solve(c, vec_of(c, x).p, vec_of(c, y).p, nullptr);
    TP_API_END
}

int tp_amg_setup(tp_ctx *c, int32_t which) {
    TP_API_BEGIN
    (void)which;
    pc_setup(c);
    TP_API_END
}

int tp_amg_vcycle(tp_ctx *c, int32_t which, int32_t field_b, int32_t b, int32_t field_x, int32_t x) {
    TP_API_BEGIN
    TP_REQUIRE(c->pc_ready, "AMG not set up");
    if (which == 2) {       // pc_cptramg: the system V-cycle on fields (0,1) of b -> fields (0,1) of x
        TP_REQUIRE(c->bamg && !c->dist, "system AMG not set up (pc_cptramg, single slab)");
        TP_REQUIRE(b != x && field_b == 0 && field_x == 0, "system V-cycle: fields (0,1) of two different vectors");
        bamg_vcycle(c, c->bamg, vec_of(c, b).p, vec_of(c, x).p);
        return 0;
    }
    Amg *amg = which == 0 ? c->amg_p : c->amg_T;
    TP_REQUIRE(amg, "this AMG hierarchy does not exist for the selected preconditioner");
    resolve_cycle_shapes(c);
    TP_REQUIRE(!c->dist || amg->dist_levels > 0, "tp_amg_vcycle works on slab vectors: not available when the hierarchy is replicated on the gathered global grid");
    TP_REQUIRE(field_b >= 0 && field_b < c->b && field_x >= 0 && field_x < c->b, "bad field index");
    TP_REQUIRE(!(b == x && field_b == field_x), "b and x must differ");
    amg_vcycle(c, amg, vec_of(c, b).p + (long)field_b * c->g.ntot, vec_of(c, x).p + (long)field_x * c->g.ntot);
    TP_API_END
}

int tp_schur_apply(tp_ctx *c, int32_t x, int32_t y) {
    TP_API_BEGIN
    TP_REQUIRE(c->pc_ready && c->amg_T, "S~ AMG not set up (pc_cptr only)");
    resolve_cycle_shapes(c);
    TP_REQUIRE(!c->dist || c->amg_T->dist_levels > 0, "tp_schur_apply works on slab vectors: not available when the hierarchy is replicated on the gathered global grid");
    TP_REQUIRE(x != y, "x and y must differ");
    amg_vcycle(c, c->amg_T, vec_of(c, x).p + c->g.ntot, vec_of(c, y).p + c->g.ntot);
    TP_API_END
}

int tp_fgmres(tp_ctx *c, int32_t b, int32_t x, int32_t *its, int32_t *reason, double *rnorm) {
    TP_API_BEGIN
    TP_REQUIRE(b != x, "b and x must differ");
    if (!c->pc_ready) pc_setup(c);
    int it = 0;
    double rn = 0.0;
    const int r = fgmres(c, vec_of(c, b).p, vec_of(c, x).p, &it, &rn);
    if (its) *its = it;
    if (reason) *reason = r;
    if (rnorm) *rnorm = rn;
    TP_API_END
}

int tp_newton_solve(tp_ctx *c, tp_solve_info *info) {
    TP_API_BEGIN
    TP_REQUIRE(c && info, "null argument");
    newton(c, info);
    TP_HIP(hipStreamSynchronize(c->stream));
    TP_API_END
}

int tp_time_kernel(tp_ctx *c, int32_t which, int32_t reps, double *ms_avg) {
    TP_API_BEGIN
    TP_REQUIRE(reps > 0 && ms_avg, "bad arguments");
    ensure_work(c);          // (w3 needs three planes even when b == 2: tp_solver.hip)
    if (which != 3) TP_REQUIRE(c->jac_ready, "Jacobian not assembled");
    if (which == 1 || which == 2 || which == 4) TP_REQUIRE(c->pc_ready, "preconditioner not set up");
    auto run = [&]() {
        switch (which) {
            case 0: spmv_block(c, c->J.p, c->R.p, c->w2.p); break;
            case 1: ilu_solve(c, c->R.p, c->w2.p, nullptr); break;
            case 2:
                resolve_cycle_shapes(c);
                if (sysamg_of(c->opt)) { TP_REQUIRE(!c->dist, "single slab only"); bamg_vcycle(c, c->bamg, c->R.p, c->w2.p); break; }
                if (c->dist && c->amg_p->dist_levels == 0) amg_vcycle(c, c->amg_p, c->gvec.p, c->gvec.p + 2 * c->gfull.ntot);   // global-grid buffers
                else amg_vcycle(c, c->amg_p, c->R.p, c->w2.p);
                break;
            case 3: assemble(c, true, schur_of(c->opt)); break;
            case 4: pc_apply(c, c->R.p, c->dx.p); break;
            case 5: pc_setup(c); break;
            case 6: ilu_factor(c); break;
            case 7: {       // one classical Gram-Schmidt step against 16 basis vectors (VecMDot + VecMAXPY + norm)
                TP_REQUIRE(c->gs_cap >= 17, "Krylov basis smaller than 17 vectors: run a solve first");
                std::vector<double> hh(18);
                orthogonalize(c, c->b, c->V.p, (long)c->b * c->g.ntot, 16, c->w2.p, hh.data());
                break;
            }
            default: throw Error("unknown kernel id");
        }
    };
    run();                                                      // warm-up
    TP_HIP(hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < reps; ++i) run();
    TP_HIP(hipEventRecord(c->ev1, c->stream));
    TP_HIP(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    TP_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *ms_avg = (double)ms / reps;
    TP_API_END
}

int tp_amg_info(tp_ctx *c, int32_t which, int32_t *nlevels, double *op_complexity) {
    TP_API_BEGIN
    if (which == 2) {
        TP_REQUIRE(c->bamg, "system AMG hierarchy not built");
        if (nlevels) *nlevels = bamg_levels(c->bamg);
        if (op_complexity) *op_complexity = 0.0;
        return 0;
    }
    Amg *amg = which == 0 ? c->amg_p : c->amg_T;
    TP_REQUIRE(amg, "AMG hierarchy not built");
    if (nlevels) *nlevels = (int)amg->lv.size();
    double s = 0.0, s0 = 0.0;       // global cells per level (a distributed level holds this rank's slab only)
    for (size_t l = 0; l < amg->lv.size(); ++l) {
        const GridDev &g = amg->lv[l]->g;
        const double cells = (double)g.np * g.gn2;
        s += cells;
        if (l == 0) s0 = cells;
    }
    if (op_complexity) *op_complexity = s / s0;
    TP_API_END
}

int tp_amg_trunc(tp_ctx *c, int32_t which, int32_t *level, double *ratio0) {
    TP_API_BEGIN
    Amg *amg = which == 0 ? c->amg_p : c->amg_T;
    TP_REQUIRE(amg, "AMG hierarchy not built");
    resolve_cycle_shapes(c);
    if (level) *level = amg->trunc;
    if (ratio0) *ratio0 = amg->ratio0;
    TP_API_END
}

int tp_amg_layout(tp_ctx *c, int32_t which, int32_t *dist_levels, int32_t *axes, int32_t cap, int32_t *naxes) {
    TP_API_BEGIN
    if (which == 2) {       // the (p,T) system hierarchy of pc_cptramg
        TP_REQUIRE(c->bamg, "system AMG hierarchy not built");
        const std::vector<int> &sc = bamg_sched(c->bamg);
        if (dist_levels) *dist_levels = bamg_dist_levels(c->bamg);
        if (naxes) *naxes = (int)sc.size();
        for (int i = 0; axes && i < cap && i < (int)sc.size(); ++i) axes[i] = sc[i];
        return 0;
    }
    Amg *amg = which == 0 ? c->amg_p : c->amg_T;
    TP_REQUIRE(amg, "AMG hierarchy not built");
    if (dist_levels) *dist_levels = amg->dist_levels;
    if (naxes) *naxes = (int)amg->sched.size();
    for (int i = 0; axes && i < cap && i < (int)amg->sched.size(); ++i) axes[i] = amg->sched[i];
    TP_API_END
}

}  // extern "C"

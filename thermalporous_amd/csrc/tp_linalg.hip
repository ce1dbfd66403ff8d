// Vector kernels, reductions and stencil mat-vecs (what PETSc Vec*/MatMult do in the reference:
// SURVEY.md 2.2 N3, N4, N9, N10).  All HBM-bound; one thread per owned cell, consecutive lanes on
// consecutive cells, every matrix plane streamed exactly once per product.
#include "tp_common.hpp"
#include <algorithm>
#include <cstdlib>

namespace tp {

static inline dim3 grid_for(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

// ---- elementwise ---------------------------------------------------------------------------------
__global__ void k_zero(double *x, long n) {
    const long i = (long)blockIdx.x * TP_BLOCK + threadIdx.x;
    if (i < n) x[i] = 0.0;
}
__global__ void k_copy(const double *x, double *y, long n) {
    const long i = (long)blockIdx.x * TP_BLOCK + threadIdx.x;
    if (i < n) y[i] = x[i];
}
// y[f][c] (+)= a*x[f][c] over owned cells of nf fields
template <bool ACC>
__global__ void k_axpy_owned(GridDev g, int nf, double a, const double *x, double *y) {
    const long t = (long)blockIdx.x * TP_BLOCK + threadIdx.x;
    if (t >= g.nown * nf) return;
    const long f = t / g.nown, i = t - f * g.nown;
    const long c = f * g.ntot + g.np + i;
    y[c] = ACC ? y[c] + a * x[c] : a * x[c];
}

void vec_zero(tp_ctx *c, double *x, long n) {
    hipLaunchKernelGGL(k_zero, grid_for(n), dim3(256), 0, c->stream, x, n);
}
void vec_copy(tp_ctx *c, const double *x, double *y, long n) {
    hipLaunchKernelGGL(k_copy, grid_for(n), dim3(256), 0, c->stream, x, y, n);
}
void vec_axpy_owned(tp_ctx *c, int nf, double a, const double *x, double *y) {
    hipLaunchKernelGGL(k_axpy_owned<true>, grid_for(c->g.nown * nf), dim3(256), 0, c->stream, c->g, nf, a, x, y);
}
// x[f][c] *= 1 / sqrt(*n2) over owned cells: v_{j+1} = w / ||w|| with the norm still on the device (the same IEEE operations
// as the host's 1.0 / sqrt(n2) followed by vec_scale_to: bit-identical)
__global__ void k_scale_dev_norm(GridDev g, int nf, const double *__restrict__ n2, double *x) {
    const long t = (long)blockIdx.x * TP_BLOCK + threadIdx.x;
    if (t >= g.nown * nf) return;
    const double a = 1.0 / sqrt(*n2);
    const long f = t / g.nown, i = t - f * g.nown;
    const long c = f * g.ntot + g.np + i;
    x[c] = a * x[c];
}
void vec_scale_dev_norm(tp_ctx *c, int nf, const double *n2_dev, double *x) {
    hipLaunchKernelGGL(k_scale_dev_norm, grid_for(c->g.nown * nf), dim3(256), 0, c->stream, c->g, nf, n2_dev, x);
}

void vec_scale_to(tp_ctx *c, int nf, double a, const double *x, double *y) {
    hipLaunchKernelGGL(k_axpy_owned<false>, grid_for(c->g.nown * nf), dim3(256), 0, c->stream, c->g, nf, a, x, y);
}

// ---- batched dot products (VecMDot + VecNorm of one FGMRES iteration in ONE pass over w) -----------
// Each wave owns a chunk of owned entries (CHUNK per lane kept in registers), loops over the k basis
// vectors and reduces with DPP shuffles; per-wave partials go to gs_partial[i][wave], a second tiny
// kernel sums them in a fixed order (deterministic).  The k-th extra output is <w2, w2>.

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Memory-level parallelism: the Gram-Schmidt kernels stream k basis vectors past a register-resident piece of w.
// Written one vector at a time (4 loads, then a 6-step cross-lane reduction, then the next vector) a wave keeps only
// 2 KB in flight and the pass ran at 2.5 TB/s (rocprofv3, round 2: 119 us at k ~ 11).  MD_U vectors are now handled
// together: MD_U*MD_CHUNK independent, unconditional loads per lane are issued before anything waits on them (tail
// lanes read a valid address and multiply by w = 0) and the MD_U reductions interleave.  Summation order per output is
// unchanged (bitwise identical results).
constexpr int MD_U = 4;

template <int CH>
__global__ __launch_bounds__(256) void k_multi_dot(GridDev g, int nf, const double *__restrict__ V, long vstride, int k,
                                                   const double *__restrict__ w, const double *__restrict__ w2,
                                                   double *__restrict__ partial, long nwaves) {
    constexpr int MD_CHUNK = CH;
    const long wave = ((long)blockIdx.x * TP_BLOCK + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= nwaves) return;
    const long nall = g.nown * nf;
    long idx[MD_CHUNK];
    double wv[MD_CHUNK];
    bool ok[MD_CHUNK];
#pragma unroll
    for (int j = 0; j < MD_CHUNK; ++j) {
        const long t = (wave * MD_CHUNK + j) * 64 + lane;
        ok[j] = t < nall;
        const long tt = ok[j] ? t : 0;
        const long f = tt / g.nown, i = tt - f * g.nown;
        idx[j] = f * g.ntot + g.np + i;              // (tail lanes: entry 0 -- a valid address, weight 0)
        wv[j] = ok[j] ? w[idx[j]] : 0.0;
    }
    int i = 0;
    for (; i + MD_U <= k; i += MD_U) {
        double v[MD_U][MD_CHUNK];
#pragma unroll
        for (int u = 0; u < MD_U; ++u)
#pragma unroll
            for (int j = 0; j < MD_CHUNK; ++j) v[u][j] = V[(long)(i + u) * vstride + idx[j]];
        double s[MD_U];
#pragma unroll
        for (int u = 0; u < MD_U; ++u) {
            s[u] = 0.0;
#pragma unroll
            for (int j = 0; j < MD_CHUNK; ++j) s[u] += v[u][j] * wv[j];
        }
#pragma unroll
        for (int u = 0; u < MD_U; ++u) s[u] = wave_sum(s[u]);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < MD_U; ++u) partial[(long)(i + u) * nwaves + wave] = s[u];
        }
    }
    for (; i < k; ++i) {
        const double *Vi = V + (long)i * vstride;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < MD_CHUNK; ++j) s += Vi[idx[j]] * wv[j];
        s = wave_sum(s);
        if (lane == 0) partial[(long)i * nwaves + wave] = s;
    }
    if (w2) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < MD_CHUNK; ++j) {
            const double t = ok[j] ? w2[idx[j]] : 0.0;
            s += t * t;
        }
        s = wave_sum(s);
        if (lane == 0) partial[(long)k * nwaves + wave] = s;
    }
}

// second stage of the deterministic two-stage sums: one workgroup per output.  Latency bound (a few thousand
// partials): 1024 threads with 4 independent loads in flight each, fixed summation order.
// out2 (optional): a second destination -- the pinned, device-mapped host buffer tp_ctx::h_pin, so that the host can read
// the sums after the stream synchronisation without a device-to-host copy in between (a blit kernel of ~4.5 us)
__global__ __launch_bounds__(1024) void k_reduce_partials(const double *__restrict__ partial, long nwaves,
                                                          double *__restrict__ out, double *__restrict__ out2 = nullptr) {
    __shared__ double sh[16];
    const double *p = partial + (long)blockIdx.x * nwaves;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    long i = threadIdx.x;
    for (; i + 3 * 1024 < nwaves; i += 4 * 1024) {
        s0 += p[i]; s1 += p[i + 1024]; s2 += p[i + 2 * 1024]; s3 += p[i + 3 * 1024];
    }
    for (; i < nwaves; i += 1024) s0 += p[i];
    double s = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += sh[w];
        out[blockIdx.x] = t;
        if (out2) out2[blockIdx.x] = t;
    }
}

// entries per lane in the Gram-Schmidt kernels: 4 or 8 (TP_MD_CHUNK; 8 halves the cross-lane reductions per byte)
static int md_chunk() {
    static const int ch = (getenv("TP_MD_CHUNK") && atoi(getenv("TP_MD_CHUNK")) == 4) ? 4 : 8;
    return ch;
}
// (an XCD-aware chunk order, as in the stencil kernels, was measured in round 3: Gram-Schmidt step at k = 16 0.181-0.183 ms
// against 0.180-0.181 ms -- pure streams have nothing to gain from it; plain block order kept.  So was a software-pipelined
// dot kernel -- the next four vectors' 32 loads issued before the reductions of the current four: 256 VGPRs instead of 140,
// 0.182-0.183 ms against 0.176 ms on the same box: the occupancy lost costs more than the overlap gains)
static dim3 md_grid(long nw) { return dim3((unsigned)((nw * 64 + 255) / 256)); }
static long md_nwaves(const tp_ctx *c, int nf) {
    const long nall = c->g.nown * nf;
    return (nall + 64L * md_chunk() - 1) / (64L * md_chunk());
}
// second stage of a reduction + hand-over to the host: n sums of nw partials -> red_out (device) and host_out.  One GPU: the
// kernel also writes to the pinned buffer and the host copies from there after the synchronisation; several GPUs: all-reduce,
// then a device-to-host copy.
static void reduce_to_host(tp_ctx *c, long nw, int n, double *host_out) {
    static const bool use_pin = !(getenv("TP_PIN") && atoi(getenv("TP_PIN")) == 0);
    double *pin = (use_pin && !c->dist && n <= tp_ctx::H_PIN) ? c->h_pin : nullptr;
    hipLaunchKernelGGL(k_reduce_partials, dim3(n), dim3(1024), 0, c->stream, c->gs_partial.p, nw, c->red_out.p, pin);
    TP_HIP(hipGetLastError());
    allreduce_sum(c, c->red_out.p, n);
    if (pin) {
        TP_HIP(hipStreamSynchronize(c->stream));
        memcpy(host_out, pin, sizeof(double) * n);
        return;
    }
    if (use_pin && n <= tp_ctx::H_PIN) {   // several GPUs: the all-reduced sums through the pinned buffer (a truly asynchronous copy)
        TP_HIP(hipMemcpyAsync(c->h_pin, c->red_out.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
        TP_HIP(hipStreamSynchronize(c->stream));
        memcpy(host_out, c->h_pin, sizeof(double) * n);
        return;
    }
    TP_HIP(hipMemcpyAsync(host_out, c->red_out.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    TP_HIP(hipStreamSynchronize(c->stream));
}

#define TP_MD_LAUNCH(KERNEL, ...)                                                                              \
    do {                                                                                                       \
        if (md_chunk() == 4) hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__);                                       \
        else hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__);                                                       \
    } while (0)

void multi_dot(tp_ctx *c, int nf, const double *V, long vstride, int k, const double *w, const double *w2,
               double *host_out) {
    const long nw = md_nwaves(c, nf);
    const int nout = k + (w2 ? 1 : 0);
    if ((long)c->gs_partial.n < (long)nout * nw) c->gs_partial.alloc((size_t)(nout + 32) * nw);
    if ((long)c->red_out.n < nout) c->red_out.alloc(nout + 64);
    TP_MD_LAUNCH(k_multi_dot, md_grid(nw), dim3(256), 0, c->stream, c->g, nf, V, vstride, k, w, w2,
                 c->gs_partial.p, nw);
    reduce_to_host(c, nw, nout, host_out);
}

// ||x_i||^2 of several vectors with ONE reduction launch, ONE all-reduce and ONE host sync (the three norms of SNES's
// convergence test: ||F||, ||dx||, ||u||; each used to cost its own sync and -- on several GPUs -- its own all-reduce)
void multi_norm2sq(tp_ctx *c, int nf, int nvec, const double *const *x, double *host_out) {
    const long nw = md_nwaves(c, nf);
    if ((long)c->gs_partial.n < (long)nvec * nw) c->gs_partial.alloc((size_t)(nvec + 32) * nw);
    if ((long)c->red_out.n < nvec) c->red_out.alloc(nvec + 64);
    for (int i = 0; i < nvec; ++i)
        TP_MD_LAUNCH(k_multi_dot, md_grid(nw), dim3(256), 0, c->stream, c->g, nf, x[i], 0L, 0, x[i], x[i],
                     c->gs_partial.p + (long)i * nw, nw);
    reduce_to_host(c, nw, nvec, host_out);
}

double norm2(tp_ctx *c, int nf, const double *x) {
    double s = 0.0;
    multi_dot(c, nf, x, 0, 0, x, x, &s);
    return sqrt(s);
}

// w += sign * sum_i h_i V_i  (VecMAXPY): one pass over w, k coalesced streams
__global__ __launch_bounds__(256) void k_multi_axpy(GridDev g, int nf, const double *__restrict__ V, long vstride, int k,
                                                    const double *__restrict__ h, double sign, double *w) {
    const long t = (long)blockIdx.x * TP_BLOCK + threadIdx.x;
    if (t >= g.nown * nf) return;
    const long f = t / g.nown, i = t - f * g.nown;
    const long c = f * g.ntot + g.np + i;
    double s = 0.0;
    int j = 0;
    for (; j + MD_U <= k; j += MD_U) {          // MD_U independent loads in flight; same summation order
        double v[MD_U];
#pragma unroll
        for (int u = 0; u < MD_U; ++u) v[u] = V[(long)(j + u) * vstride + c];
#pragma unroll
        for (int u = 0; u < MD_U; ++u) s += h[j + u] * v[u];
    }
    for (; j < k; ++j) s += h[j] * V[(long)j * vstride + c];
    w[c] += sign * s;
}

void multi_axpy(tp_ctx *c, int nf, const double *V, long vstride, int k, const double *hcoef_host, double sign,
                double *w) {
    if (k <= 0) return;
    if ((int)c->gs_h.n < k) c->gs_h.alloc(k + 64);
    TP_HIP(hipMemcpyAsync(c->gs_h.p, hcoef_host, sizeof(double) * k, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_multi_axpy, grid_for(c->g.nown * nf), dim3(256), 0, c->stream, c->g, nf, V, vstride, k,
                       c->gs_h.p, sign, w);
    TP_HIP(hipGetLastError());
    // the host buffer may be reused by the caller right away
    TP_HIP(hipStreamSynchronize(c->stream));
}

// ---- one Gram-Schmidt step with a single host sync -------------------------------------------------
// h = V^T w (k dots) ; w -= V h ; ||w||^2 -- the coefficients never leave the device between the dot and
// the update, so an FGMRES iteration pays one D2H copy + sync here instead of three.
template <int CH>
__global__ __launch_bounds__(256) void k_multi_axpy_norm(GridDev g, int nf, const double *__restrict__ V, long vstride,
                                                         int k, const double *__restrict__ h, double *w,
                                                         double *__restrict__ partial, long nwaves, int rev) {
    constexpr int MD_CHUNK = CH;
    const long wave_d = ((long)blockIdx.x * TP_BLOCK + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave_d >= nwaves) return;
    // REVERSE traversal (TP_GS_REVERSE): this pass re-reads the k basis vectors the dot pass has just streamed front to back;
    // walking back to front, the entries read first are the ones read last a moment ago -- what is still in the 256 MB
    // Infinity Cache -- instead of the ones evicted longest ago.  Same chunks, same per-chunk sums: results unchanged.
    const long wave = rev ? nwaves - 1 - wave_d : wave_d;
    const long nall = g.nown * nf;
    long idx[MD_CHUNK];
    bool ok[MD_CHUNK];
    double s[MD_CHUNK], w0[MD_CHUNK];
#pragma unroll
    for (int j = 0; j < MD_CHUNK; ++j) {
        const long t = (wave * MD_CHUNK + j) * 64 + lane;
        ok[j] = t < nall;
        const long tt = ok[j] ? t : 0;
        const long f = tt / g.nown, i = tt - f * g.nown;
        idx[j] = f * g.ntot + g.np + i;
        s[j] = 0.0;
        w0[j] = w[idx[j]];
    }
    // MD_U vectors x MD_CHUNK entries = 16 independent loads in flight per lane (see k_multi_dot); the sum over q keeps
    // its order, so the result is bitwise that of the one-vector-at-a-time loop
    int q = 0;
    for (; q + MD_U <= k; q += MD_U) {
        double v[MD_U][MD_CHUNK], hq[MD_U];
#pragma unroll
        for (int u = 0; u < MD_U; ++u) {
            hq[u] = h[q + u];
#pragma unroll
            for (int j = 0; j < MD_CHUNK; ++j) v[u][j] = V[(long)(q + u) * vstride + idx[j]];
        }
#pragma unroll
        for (int u = 0; u < MD_U; ++u)
#pragma unroll
            for (int j = 0; j < MD_CHUNK; ++j) s[j] += hq[u] * v[u][j];
    }
    for (; q < k; ++q) {
        const double hq = h[q];
#pragma unroll
        for (int j = 0; j < MD_CHUNK; ++j) s[j] += hq * V[(long)q * vstride + idx[j]];
    }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < MD_CHUNK; ++j) {
        if (ok[j]) {
            const double wn = w0[j] - s[j];
            w[idx[j]] = wn;
            acc += wn * wn;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) partial[wave] = acc;
}

void orthogonalize(tp_ctx *c, int nf, const double *V, long vstride, int k, double *w, double *host_out) {
    const long nw = md_nwaves(c, nf);
    if ((long)c->gs_partial.n < (long)(k + 1) * nw) c->gs_partial.alloc((size_t)(k + 33) * nw);
    if ((long)c->red_out.n < k + 1) c->red_out.alloc(k + 65);
    // Several GPUs: TWO all-reduces per Krylov iteration (the k dots, then the norm of the orthogonalised vector), ONE host
    // sync.  A single message per iteration was built in round 3 -- ||w - V h||^2 = ||w||^2 - sum h_i^2 with ||w||^2 riding in
    // the dot batch -- and is unstable inside classical Gram-Schmidt: with a good preconditioner J M^-1 v_j ~ v_j, so the new
    // direction carries 1e-4 .. 1e-8 of ||w||^2; the difference then amplifies the basis' loss of orthogonality by
    // ||w||^2 / h_{j+1,j}^2 per iteration, the mis-normalised v_{j+1} feeds that back, and FGMRES stalls (the 2-slab case of
    // tests/test_gpu_slabs.py: DIVERGED_ITS where two messages converge in 12 iterations).  The exact one-message form needs
    // a lagged normalisation (two more vector passes and one wasted iteration per solve) for ~10 us of ~700: not adopted.
    TP_MD_LAUNCH(k_multi_dot, md_grid(nw), dim3(256), 0, c->stream, c->g, nf, V, vstride, k, w,
                 (const double *)nullptr, c->gs_partial.p, nw);
    // one GPU: the sums also go straight to pinned host memory (no copy between the last kernel and the host's wake-up)
    static const bool use_pin = !(getenv("TP_PIN") && atoi(getenv("TP_PIN")) == 0);
    double *pin = (use_pin && !c->dist && k + 1 <= tp_ctx::H_PIN) ? c->h_pin : nullptr;
    hipLaunchKernelGGL(k_reduce_partials, dim3(k), dim3(1024), 0, c->stream, c->gs_partial.p, nw, c->red_out.p, pin);
    allreduce_sum(c, c->red_out.p, k);
    static const int gs_rev = !(getenv("TP_GS_REVERSE") && atoi(getenv("TP_GS_REVERSE")) == 0);
    TP_MD_LAUNCH(k_multi_axpy_norm, md_grid(nw), dim3(256), 0, c->stream, c->g, nf, V, vstride, k,
                 c->red_out.p, w, c->gs_partial.p, nw, gs_rev);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, c->stream, c->gs_partial.p, nw, c->red_out.p + k,
                       pin ? pin + k : (double *)nullptr);
    TP_HIP(hipGetLastError());
    allreduce_sum(c, c->red_out.p + k, 1);
    if (pin) {
        if (!host_out) {                   // split form (orthogonalize_enqueue / orthogonalize_wait): the caller goes on enqueueing
            TP_HIP(hipEventRecord(c->ev_h, c->stream));
            return;
        }
        TP_HIP(hipStreamSynchronize(c->stream));
        memcpy(host_out, pin, sizeof(double) * (k + 1));
        return;
    }
    if (!host_out) {                       // split form on several GPUs: the all-reduced sums -> pinned buffer (asynchronous), event
        TP_REQUIRE(k + 1 <= tp_ctx::H_PIN, "split orthogonalisation needs the pinned result buffer");
        TP_HIP(hipMemcpyAsync(c->h_pin, c->red_out.p, sizeof(double) * (k + 1), hipMemcpyDeviceToHost, c->stream));
        TP_HIP(hipEventRecord(c->ev_h, c->stream));
        return;
    }
    TP_HIP(hipMemcpyAsync(host_out, c->red_out.p, sizeof(double) * (k + 1), hipMemcpyDeviceToHost, c->stream));
    TP_HIP(hipStreamSynchronize(c->stream));
}

// The same in two halves for the pipelined FGMRES loop: enqueue the kernels (and all-reduces) and record an event; later wait
// for that event only -- whatever was enqueued behind it (the next iteration's preconditioner application) keeps running -- and
// read the k dots and ||w||^2 from the pinned buffer.  ||w||^2 also stays on the device at orthogonalize_norm_dev(c, k).
// Several GPUs: every rank holds the same all-reduced sums, takes the same speculation decision and therefore issues the same
// sequence of collectives.
bool orthogonalize_can_split(const tp_ctx *c, int k) {
    static const bool use_pin = !(getenv("TP_PIN") && atoi(getenv("TP_PIN")) == 0);
    return use_pin && k + 1 <= tp_ctx::H_PIN;
}
void orthogonalize_enqueue(tp_ctx *c, int nf, const double *V, long vstride, int k, double *w) {
    orthogonalize(c, nf, V, vstride, k, w, nullptr);
}
const double *orthogonalize_norm_dev(const tp_ctx *c, int k) { return c->red_out.p + k; }
void orthogonalize_wait(tp_ctx *c, int k, double *host_out) {
    TP_HIP(hipEventSynchronize(c->ev_h));
    memcpy(host_out, c->h_pin, sizeof(double) * (k + 1));
}

// ---- saturation guard (thermalmodel.py:193-229): min/max and clamp of one field over owned cells -----
__global__ __launch_bounds__(256) void k_minmax(GridDev g, const double *x, double *partial) {
    __shared__ double smin[4], smax[4];
    double lo = 1e300, hi = -1e300;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < g.nown; i += (long)gridDim.x * blockDim.x) {
        const double v = x[g.np + i];
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_down(lo, o, 64));
        hi = fmax(hi, __shfl_down(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
        partial[2 * blockIdx.x + 1] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    }
}
__global__ void k_clamp01(GridDev g, double *x) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.nown) return;
    const double v = x[g.np + i];
    x[g.np + i] = fmax(fmin(v, 1.0), 0.0);
}

void field_minmax(tp_ctx *c, const double *x, double *lo, double *hi) {
    const int nb = (int)std::min<long>(512, (c->g.nown + 255) / 256);
    if ((long)c->gs_partial.n < 2L * nb) c->gs_partial.alloc(4096);
    hipLaunchKernelGGL(k_minmax, dim3(nb), dim3(256), 0, c->stream, c->g, x, c->gs_partial.p);
    std::vector<double> h(2 * nb);
    TP_HIP(hipMemcpyAsync(h.data(), c->gs_partial.p, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, c->stream));
    TP_HIP(hipStreamSynchronize(c->stream));
    *lo = 1e300; *hi = -1e300;
    for (int i = 0; i < nb; ++i) { *lo = std::min(*lo, h[2 * i]); *hi = std::max(*hi, h[2 * i + 1]); }
}

void field_clamp01(tp_ctx *c, double *x) {
    hipLaunchKernelGGL(k_clamp01, grid_for(c->g.nown), dim3(256), 0, c->stream, c->g, x);
    TP_HIP(hipGetLastError());
}

// ---- block stencil mat-vec: y = J x  (MatMult) ----------------------------------------------------
// MODE 0: y = J x ; MODE 1: r = x0 - J[:, :NC] y   (stage-1 output has zero secondary fields)
// (first, count): the owned cells [first, first + count) -- the whole slab, or a range of planes when the boundary planes
// wait for a halo exchange that the interior overlaps (spmv_block_halo)
template <int B, int NS, int NC, int MODE>
__global__ __launch_bounds__(256) void k_spmv_block(GridDev g, const double *__restrict__ J,
                                                    const double *__restrict__ x, const double *__restrict__ x0,
                                                    double *__restrict__ y, long first, long count) {
    const long tid = xcd_tid();
    if (tid >= count) return;
    const long c = g.np + first + tid, nt = g.ntot;
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    double acc[B];
#pragma unroll
    for (int r = 0; r < B; ++r) acc[r] = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double xv[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) xv[k] = x[(long)k * nt + c + off[s]];
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
            for (int k = 0; k < NC; ++k) acc[r] += J[((long)(s * B + r) * B + k) * nt + c] * xv[k];
    }
#pragma unroll
    for (int r = 0; r < B; ++r) y[(long)r * nt + c] = MODE ? x0[(long)r * nt + c] - acc[r] : acc[r];
}

static void spmv_block_range(tp_ctx *c, hipStream_t st, const double *J, const double *x, double *y, long first, long count) {
    const GridDev &g = c->g;
    const dim3 gr = xcd_grid(count), bl(256);
    const bool d3 = g.gn2 > 1;
    if (c->b == 3) {
        if (d3) hipLaunchKernelGGL((k_spmv_block<3, 7, 3, 0>), gr, bl, 0, st, g, J, x, x, y, first, count);
        else    hipLaunchKernelGGL((k_spmv_block<3, 5, 3, 0>), gr, bl, 0, st, g, J, x, x, y, first, count);
    } else {
        if (d3) hipLaunchKernelGGL((k_spmv_block<2, 7, 2, 0>), gr, bl, 0, st, g, J, x, x, y, first, count);
        else    hipLaunchKernelGGL((k_spmv_block<2, 5, 2, 0>), gr, bl, 0, st, g, J, x, x, y, first, count);
    }
    TP_HIP(hipGetLastError());
}

void spmv_block(tp_ctx *c, const double *J, const double *x, double *y) {
    spmv_block_range(c, c->stream, J, x, y, 0, c->g.nown);
}

// y = J x on a slab whose halo planes of x are stale: the interior planes (which read no halo) run on a second stream while
// the halo exchange -- RCCL send/recv, always on the main stream: one communicator, one stream -- is in flight; the two
// boundary planes follow the exchange on the main stream, which then joins the interior (SURVEY.md 8e "overlap with
// interior rows").  Per-cell arithmetic unchanged: bitwise the result of exchange-then-SpMV.
void spmv_block_halo(tp_ctx *c, const double *J, double *x, double *y) {
    const GridDev &g = c->g;
    static const bool overlap = !(getenv("TP_HALO_OVERLAP") && atoi(getenv("TP_HALO_OVERLAP")) == 0);
    if (!c->dist) { spmv_block(c, J, x, y); return; }
    if (!overlap || g.n2 < 3) {
        halo_exchange(c, g, x, c->b, g.ntot);
        spmv_block(c, J, x, y);
        return;
    }
    TP_HIP(hipEventRecord(c->ev_fork, c->stream));                 // x is final on the main stream
    TP_HIP(hipStreamWaitEvent(c->aux[0], c->ev_fork, 0));
    spmv_block_range(c, c->aux[0], J, x, y, g.np, g.np * (g.n2 - 2));      // planes 1 .. n2-2
    TP_HIP(hipEventRecord(c->ev_join[0], c->aux[0]));
    halo_exchange(c, g, x, c->b, g.ntot);
    spmv_block_range(c, c->stream, J, x, y, 0, g.np);                      // plane 0 (reads the lower halo)
    spmv_block_range(c, c->stream, J, x, y, g.np * (g.n2 - 1), g.np);      // plane n2-1 (reads the upper halo)
    TP_HIP(hipStreamWaitEvent(c->stream, c->ev_join[0], 0));
}

void resid_block_cols(tp_ctx *c, const double *J, const double *x, const double *y, int ncols, double *r) {
    const GridDev &g = c->g;
    const dim3 gr = xcd_grid(g.nown), bl(256);
    const bool d3 = g.gn2 > 1;
#define RL(B, NS, NC) hipLaunchKernelGGL((k_spmv_block<B, NS, NC, 1>), gr, bl, 0, c->stream, g, J, y, x, r, 0L, g.nown)
    if (c->b == 3) {
        if (ncols == 1) { if (d3) RL(3, 7, 1); else RL(3, 5, 1); }
        else if (ncols == 2) { if (d3) RL(3, 7, 2); else RL(3, 5, 2); }
        else { if (d3) RL(3, 7, 3); else RL(3, 5, 3); }
    } else {
        if (ncols == 1) { if (d3) RL(2, 7, 1); else RL(2, 5, 1); }
        else { if (d3) RL(2, 7, 2); else RL(2, 5, 2); }
    }
#undef RL
    TP_HIP(hipGetLastError());
}

// ---- scalar stencil: y = z + alpha * A x ----------------------------------------------------------
__global__ __launch_bounds__(256) void k_spmv_scalar(GridDev g, Stencil A, const double *__restrict__ x,
                                                     double *__restrict__ y, double alpha,
                                                     const double *__restrict__ z) {
    const long tid = xcd_tid();
    if (tid >= g.nown) return;
    const long c = g.np + tid;
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += A.slot(k)[c] * x[c + off[k]];
    y[c] = (z ? z[c] : 0.0) + alpha * s;
}

void spmv_scalar(tp_ctx *c, const GridDev &g, const Stencil &A, const double *x, double *y, double alpha,
                 const double *z) {
    hipLaunchKernelGGL(k_spmv_scalar, xcd_grid(g.nown), dim3(256), 0, c->stream, g, A, x, y, alpha, z);
    TP_HIP(hipGetLastError());
}

// ---- selfp: Sp = A11 - A10 diag(A00)^-1 A01 (pc_fieldsplit_schur_precondition selfp, singlephase.py:322-330) ------
// PETSc forms Sp explicitly (13-point in 2-D, 25-point in 3-D).  Here (oracle/linalg.py:SelfpSchur): the AMG hierarchy
// is that of Sp's 7-point collapse S7 -- the 7-point entries of Sp exactly, the far entries (c -> m -> j, j neither c nor
// a stencil neighbour of c) lumped onto the diagonal -- and one damped-Jacobi sweep on the EXACT Sp follows the V-cycle,
// matrix-free: x += w D^-1 (b - A11 x + A10 (diag(A00)^-1 (A01 x))), D = diag(Sp) exactly.
__global__ __launch_bounds__(256) void k_selfp_build(GridDev g, Stencil A00, Stencil A01, Stencil A10, Stencil A11,
                                                     double omega, double *__restrict__ S7, double *__restrict__ invd) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid, nt = g.ntot;
    // (axis 2 in GLOBAL plane numbers: on a slab the neighbour across the slab boundary exists -- its rows of A00 / A01 are
    // in the halo planes of the Jacobian, exchanged before this kernel)
    const int n[3] = {g.n0, g.n1, g.gn2};
    const int I[3] = {(int)(tid % g.n0), (int)((tid / g.n0) % g.n1), (int)(tid / g.np) + g.off2};
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    double S[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) S[s] = A11.slot(s)[c];
    const double t0 = A10.slot(0)[c] * (1.0 / A00.slot(0)[c]);
    S[0] -= t0 * A01.slot(0)[c];
    double lump = 0.0;
#pragma unroll
    for (int s = 1; s < 7; ++s) {
        const int a = (s - 1) / 2;
        const bool odd = s & 1;
        if (n[a] == 1 || (odd ? I[a] == 0 : I[a] + 1 >= n[a])) continue;          // no neighbour m in direction s
        const long m = c + off[s];
        const int opp = odd ? s + 1 : s - 1;
        S[s] -= t0 * A01.slot(s)[c];                                               // c -> c -> m
        const double w = A10.slot(s)[c] * (1.0 / A00.slot(0)[m]);                  // c -> m
        S[s] -= w * A01.slot(0)[m];                                                // c -> m -> m
        S[0] -= w * A01.slot(opp)[m];                                              // c -> m -> c
#pragma unroll
        for (int t = 1; t < 7; ++t) {
            const int at = (t - 1) / 2;
            if (t == opp || n[at] == 1) continue;
            const int im = I[at] + (at == a ? (odd ? -1 : 1) : 0);
            if ((t & 1) ? im == 0 : im + 1 >= n[at]) continue;                    // m has no neighbour in direction t
            lump -= w * A01.slot(t)[m];                                            // c -> m -> far cell: lumped
        }
    }
    invd[c] = omega / S[0];
    S[0] += lump;
#pragma unroll
    for (int s = 0; s < 7; ++s) S7[(long)s * nt + c] = S[s];
}

// u = diag(A00)^-1 (A01 x)
__global__ __launch_bounds__(256) void k_selfp_u(GridDev g, Stencil A00, Stencil A01, const double *__restrict__ x,
                                                 double *__restrict__ u) {
    const long tid = xcd_tid();
    if (tid >= g.nown) return;
    const long c = g.np + tid;
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += A01.slot(k)[c] * x[c + off[k]];
    u[c] = s * (1.0 / A00.slot(0)[c]);
}

// y = x + invd (b - (A11 x - A10 u))
__global__ __launch_bounds__(256) void k_selfp_post(GridDev g, Stencil A10, Stencil A11, const double *__restrict__ invd,
                                                    const double *__restrict__ b, const double *__restrict__ x,
                                                    const double *__restrict__ u, double *__restrict__ y) {
    const long tid = xcd_tid();
    if (tid >= g.nown) return;
    const long c = g.np + tid;
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    double s11 = 0.0, s10 = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        s11 += A11.slot(k)[c] * x[c + off[k]];
        s10 += A10.slot(k)[c] * u[c + off[k]];
    }
    y[c] = x[c] + invd[c] * (b[c] - (s11 - s10));
}

static Stencil selfp_a11(const tp_ctx *c) {          // the T-T block of the undecoupled single-phase Jacobian
    Stencil A;
    A.base = c->J.p + (long)(c->b + 1) * c->g.ntot;
    A.slot_stride = c->opA00.slot_stride;
    return A;
}

void selfp_build(tp_ctx *c) {
    const GridDev &g = c->g;
    if (c->spbuf.n < (size_t)10 * g.ntot) c->spbuf.alloc((size_t)10 * g.ntot);     // S7 (7), invd, u, x
    hipLaunchKernelGGL(k_selfp_build, dim3((unsigned)((g.nown + 255) / 256)), dim3(256), 0, c->stream, g, c->opA00, c->opA01,
                       c->opA10, selfp_a11(c), c->opt.amg_omega, c->spbuf.p, c->spbuf.p + 7 * g.ntot);
    TP_HIP(hipGetLastError());
}

// y = x + w D^-1 (b - Sp x)     (several GPUs: x comes in with owned cells only; u = diag(A00)^-1 A01 x is needed one plane out)
void selfp_post(tp_ctx *c, const double *b, const double *x, double *y) {
    const GridDev &g = c->g;
    double *invd = c->spbuf.p + 7 * g.ntot, *u = c->spbuf.p + 8 * g.ntot;
    if (c->dist) halo_exchange(c, g, const_cast<double *>(x), 1, 0);
    hipLaunchKernelGGL(k_selfp_u, xcd_grid(g.nown), dim3(256), 0, c->stream, g, c->opA00, c->opA01, x, u);
    if (c->dist) halo_exchange(c, g, u, 1, 0);
    hipLaunchKernelGGL(k_selfp_post, xcd_grid(g.nown), dim3(256), 0, c->stream, g, c->opA10, selfp_a11(c), invd, b, x, u, y);
    TP_HIP(hipGetLastError());
}

// ---- Quasi-/True-IMPES decoupling (preconditioners.py:684-711,785-808,1445-1543) -------------------
// At[slot][i][j] = J[slot][q_i][q_j] - d_i * J[slot][s][q_j],  d_i = D_{q_i s} / D_ss per cell,
// D = diagonal entries (QI) or column sums (TI).  A per-cell row operation: no SpGEMM.
template <int B>
__global__ void k_decoup_coef(GridDev g, const double *J, int npri, int ti, double *d) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid, nt = g.ntot;
    const int s = B - 1;
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    auto colsum = [&](int q) {
        // sum over rows i of block(q,s)[i, j=c]: diag at c, slot(+a) of the row below, slot(-a) of the row above
        double v = J[((long)(0 * B + q) * B + s) * nt + c];
        if (ti) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                v += J[((long)((2 + 2 * a) * B + q) * B + s) * nt + c + off[1 + 2 * a]];   // row c-a, slot +a
                v += J[((long)((1 + 2 * a) * B + q) * B + s) * nt + c + off[2 + 2 * a]];   // row c+a, slot -a
            }
        }
        return v;
    };
    const double dss = colsum(s);
    for (int i = 0; i < npri; ++i) d[(long)i * nt + c] = colsum(i) / dss;
}

template <int B>
__global__ void k_decoup_apply(GridDev g, const double *J, int npri, const double *d, double *At) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid, nt = g.ntot;
    const int s = B - 1;
    for (int slot = 0; slot < 7; ++slot)
        for (int i = 0; i < npri; ++i) {
            const double di = d[(long)i * nt + c];
            for (int j = 0; j < npri; ++j)
                At[((long)(slot * npri + i) * npri + j) * nt + c] =
                    J[((long)(slot * B + i) * B + j) * nt + c] - di * J[((long)(slot * B + s) * B + j) * nt + c];
        }
}

// QI_temp / TI_temp (preconditioners.py:714-783, 810-873): two-phase pressure-only CPR where both non-pressure
// fields are decoupled -- D_ss is a 2x2 block per cell:  (d_T, d_S) = [D_pT D_pS] inv([[D_TT D_TS],[D_ST D_SS]]),
// Atilde_pp = A_pp - d_T A_Tp - d_S A_Sp,  r_p = x_p - d_T x_T - d_S x_S.
__global__ void k_decoup_temp(GridDev g, const double *J, int ti, double *d, double *At) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid, nt = g.ntot;
    constexpr int B = 3;
    const long off[7] = {0, -1, 1, -(long)g.n0, (long)g.n0, -g.np, g.np};
    auto entry = [&](int q, int s) {
        double v = J[((long)(0 * B + q) * B + s) * nt + c];
        if (ti) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                v += J[((long)((2 + 2 * a) * B + q) * B + s) * nt + c + off[1 + 2 * a]];   // row c-a, slot +a
                v += J[((long)((1 + 2 * a) * B + q) * B + s) * nt + c + off[2 + 2 * a]];   // row c+a, slot -a
            }
        }
        return v;
    };
    const double DTT = entry(1, 1), DTS = entry(1, 2), DST = entry(2, 1), DSS = entry(2, 2);
    const double DpT = entry(0, 1), DpS = entry(0, 2);
    const double det = DTT * DSS - DTS * DST;
    const double dT = (DpT * DSS - DpS * DST) / det, dS = (DpS * DTT - DpT * DTS) / det;
    d[c] = dT;
    d[nt + c] = dS;
#pragma unroll
    for (int slot = 0; slot < 7; ++slot)
        At[(long)slot * nt + c] = J[((long)(slot * B + 0) * B + 0) * nt + c] - dT * J[((long)(slot * B + 1) * B + 0) * nt + c] -
                                  dS * J[((long)(slot * B + 2) * B + 0) * nt + c];
}

void decouple(tp_ctx *c) {
    const GridDev &g = c->g;
    const int npri = npri_of(c->opt);
    const long nt = g.ntot;
    const int B = c->b;
    if (c->opt.decoup == 0) {
        // "No": Atilde is the primary block of J itself -- zero-copy views of the J planes
        c->opA00.base = c->J.p;                                  c->opA00.slot_stride = (long)B * B * nt;
        c->opA01.base = c->J.p + 1 * nt;                         c->opA01.slot_stride = (long)B * B * nt;
        c->opA10.base = c->J.p + (long)B * nt;                   c->opA10.slot_stride = (long)B * B * nt;
        return;
    }
    if (c->opt.decoup >= 3) {       // QI_temp (3) / TI_temp (4)
        TP_REQUIRE(B == 3 && c->opt.pc_kind == 0, "QI_temp/TI_temp are two-phase pc_cpr decouplings");
        if (c->At.n < (size_t)7 * nt) c->At.alloc((size_t)7 * nt);
        if (c->dcoef.n < (size_t)2 * nt) c->dcoef.alloc((size_t)2 * nt);
        const int tit = c->opt.decoup == 4;
        if (tit && c->dist) halo_exchange(c, g, c->J.p, 7 * B * B, nt);
        hipLaunchKernelGGL(k_decoup_temp, grid_for(g.nown), dim3(256), 0, c->stream, g, c->J.p, tit, c->dcoef.p, c->At.p);
        TP_HIP(hipGetLastError());
        c->opA00.base = c->At.p; c->opA00.slot_stride = nt;
        c->opA01 = c->opA00; c->opA10 = c->opA00;      // unused for pc_cpr
        return;
    }
    if (c->At.n < (size_t)7 * npri * npri * nt) c->At.alloc((size_t)7 * npri * npri * nt);
    if (c->dcoef.n < (size_t)npri * nt) c->dcoef.alloc((size_t)npri * nt);
    const int ti = c->opt.decoup == 2;
    if (ti && c->dist) halo_exchange(c, g, c->J.p, 7 * B * B, nt);   // column sums read neighbour rows
    const dim3 gr = grid_for(g.nown), bl(256);
    if (B == 3) {
        hipLaunchKernelGGL(k_decoup_coef<3>, gr, bl, 0, c->stream, g, c->J.p, npri, ti, c->dcoef.p);
        hipLaunchKernelGGL(k_decoup_apply<3>, gr, bl, 0, c->stream, g, c->J.p, npri, c->dcoef.p, c->At.p);
    } else {
        hipLaunchKernelGGL(k_decoup_coef<2>, gr, bl, 0, c->stream, g, c->J.p, npri, ti, c->dcoef.p);
        hipLaunchKernelGGL(k_decoup_apply<2>, gr, bl, 0, c->stream, g, c->J.p, npri, c->dcoef.p, c->At.p);
    }
    TP_HIP(hipGetLastError());
    const long ss = (long)npri * npri * nt;
    c->opA00.base = c->At.p;            c->opA00.slot_stride = ss;
    c->opA01.base = c->At.p + nt;       c->opA01.slot_stride = ss;
    c->opA10.base = c->At.p + 2 * nt;   c->opA10.slot_stride = ss;
}

// out = x_q - d_q x_s   (preconditioners.py:894-895, 1559-1560)
__global__ void k_stage1_rhs(GridDev g, const double *x, const double *d, int q, int s, double *out) {
    const long tid = (long)blockIdx.x * TP_BLOCK + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid, nt = g.ntot;
    out[c] = d ? x[(long)q * nt + c] - d[(long)q * nt + c] * x[(long)s * nt + c] : x[(long)q * nt + c];
}
// _temp variants: out = x_p - d_T x_T - d_S x_S
__global__ void k_stage1_rhs_temp(GridDev g, const double *x, const double *d, double *out) {
    const long tid = (long)blockIdx.x * TP_BLOCK + threadIdx.x;
    if (tid >= g.nown) return;
    const long c = g.np + tid, nt = g.ntot;
    out[c] = x[c] - d[c] * x[nt + c] - d[nt + c] * x[2 * nt + c];
}

void stage1_rhs(tp_ctx *c, const double *x, int q, double *out) {
    if (c->opt.decoup >= 3) {
        hipLaunchKernelGGL(k_stage1_rhs_temp, grid_for(c->g.nown), dim3(256), 0, c->stream, c->g, x, c->dcoef.p, out);
        TP_HIP(hipGetLastError());
        return;
    }
    const double *d = c->opt.decoup == 0 ? nullptr : c->dcoef.p;
    hipLaunchKernelGGL(k_stage1_rhs, grid_for(c->g.nown), dim3(256), 0, c->stream, c->g, x, d, q, c->b - 1, out);
    TP_HIP(hipGetLastError());
}

}  // namespace tp

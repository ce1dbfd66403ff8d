// Measures what a dependent phase boundary costs INSIDE one launch on MI355X, for the decision "persistent kernel for the AMG
// mid levels vs one launch per phase" (DESIGN.md 4.5):
//   flat     : one device-scope counter, lane-0 agent release fence before the arrive, acquire fence after (placement-independent)
//   nofence  : the same without the fences (synchronisation only: NOT a valid hand-off across XCDs; lower bound)
//   xcd      : XCD-hierarchical (per-XCC counter, last arriver of an XCC does the release + top counter; everybody acquires)
//   one-xcd  : only the workgroups that run on XCC 0 take part (the others leave at once); no L2 write-back / invalidate is
//              needed between CUs that share one L2 -- stores drained (vmcnt(0)), consumers would read with sc1 loads
// Every spin is bounded.  Build: hipcc -O3 --offload-arch=gfx950 barrier_probe.hip -o barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned ld_relaxed(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mode 0 flat, 1 nofence, 2 xcd-hierarchical, 3 one-xcd.  ctr: [0] top counter, [16 + 16*x] per-XCC counters, [200] fail flag,
// [208] registered one-xcd participants, [216] all-arrived counter.  work: per-WG dummy phase output.
__global__ void k_barrier(int mode, int nbar, unsigned *ctr, double *work, unsigned *nper_xcc) {
    const unsigned nwg = gridDim.x, lane0 = threadIdx.x == 0;
    const unsigned x = xcc_id();
    __shared__ unsigned s_n;
    if (mode == 3) {
        // registration: every workgroup reports where it runs; participants wait until all have reported
        if (lane0) {
            if (x == 0) atomicAdd(&ctr[208], 1u);
            __hip_atomic_fetch_add(&ctr[216], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (x != 0) return;
        if (lane0) {
            unsigned spins = 0;
            while (ld_relaxed(&ctr[216]) < nwg && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(2);
            if (spins >= (1u << 24)) ctr[200] = 1;
            s_n = ld_relaxed(&ctr[208]);
        }
        __syncthreads();
    }
    const unsigned np = mode == 3 ? s_n : nwg;
    double acc = 0.0;
    for (int b = 0; b < nbar; ++b) {
        acc += work[(blockIdx.x * 256 + threadIdx.x) & 4095] + b;        // a token phase
        work[blockIdx.x * 256 + threadIdx.x] = acc;
        __syncthreads();
        if (lane0) {
            unsigned spins = 0;
            if (mode == 0 || mode == 1 || mode == 3) {
                if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (mode == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = np * (unsigned)(b + 1);
                while (ld_relaxed(&ctr[0]) < target && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
                if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            } else {
                // hierarchical: arrive on my XCC's counter; the last arriver of the XCC releases and arrives on the top counter
                const unsigned mine = nper_xcc[x];
                const unsigned old = __hip_atomic_fetch_add(&ctr[16 + 16 * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == mine * (unsigned)(b + 1)) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    __hip_atomic_fetch_add(&ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const unsigned target = 8u * (unsigned)(b + 1);        // (all 8 XCCs hold workgroups in this probe)
                while (ld_relaxed(&ctr[0]) < target && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            if (spins >= (1u << 22)) ctr[200] = 2;
        }
        __syncthreads();
    }
    if (acc == 1.2345e300) work[0] = acc;
}

__global__ void k_census(unsigned *nper_xcc) { if (threadIdx.x == 0) atomicAdd(&nper_xcc[xcc_id()], 1u); }
__global__ void k_phase(double *work, int b) { work[blockIdx.x * 256 + threadIdx.x] += work[(blockIdx.x * 256 + threadIdx.x) & 4095] + b; }

int main() {
    unsigned *ctr, *nper;
    double *work;
    CK(hipMalloc(&ctr, 1024 * 4));
    CK(hipMalloc(&nper, 16 * 4));
    CK(hipMalloc(&work, 1024 * 256 * 8));
    CK(hipMemset(work, 0, 1024 * 256 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int nbar = 200;
    const char *names[4] = {"flat", "nofence", "xcd", "one-xcd"};
    for (int nwg : {32, 64, 128, 256}) {
        CK(hipMemset(nper, 0, 64));
        hipLaunchKernelGGL(k_census, dim3(nwg), dim3(256), 0, 0, nper);
        std::vector<unsigned> h(16);
        CK(hipMemcpy(h.data(), nper, 64, hipMemcpyDeviceToHost));
        printf("nwg %3d  per-XCC census:", nwg);
        for (int i = 0; i < 8; ++i) printf(" %u", h[i]);
        printf("\n");
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e30f;
            unsigned fail = 0, part = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(ctr, 0, 1024 * 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_barrier, dim3(nwg), dim3(256), 0, 0, mode, nbar, ctr, work, nper);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
                unsigned hc[256];
                CK(hipMemcpy(hc, ctr, 1024, hipMemcpyDeviceToHost));
                fail |= hc[200];
                part = hc[208];
            }
            printf("  %-8s %7.2f us per barrier+token phase%s%s\n", names[mode], 1e3 * best / nbar, fail ? "  (SPIN TIMEOUT)" : "",
                   mode == 3 ? (std::string("  participants ") + std::to_string(part)).c_str() : "");
        }
        // the same token phases as separate launches (eager, one stream): the kernel-boundary alternative
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int b = 0; b < nbar; ++b) hipLaunchKernelGGL(k_phase, dim3(nwg), dim3(256), 0, 0, work, b);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("  %-8s %7.2f us per launch (eager, dependent, same stream)\n", "launches", 1e3 * best / nbar);
    }
    return 0;
}

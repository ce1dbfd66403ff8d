// Development probe: semantics of global_load_lds_dwordx4 on gfx950 (lane i -> LDS base + 16*i ?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double *g, double *out) {
    __shared__ double lds[64 * 2 * 4];
    for (int r = 0; r < 4; ++r)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(g + r * 128 + threadIdx.x * 2),
                                         (void __attribute__((address_space(3))) *)(lds + r * 128), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int r = 0; r < 4; ++r) out[r * 64 + threadIdx.x] = lds[r * 128 + threadIdx.x * 2] * 1000.0 + lds[r * 128 + threadIdx.x * 2 + 1];
}
int main() {
    std::vector<double> h(512), o(256);
    for (int i = 0; i < 512; ++i) h[i] = i;
    double *d, *dout;
    hipMalloc(&d, 512 * 8); hipMalloc(&dout, 256 * 8);
    hipMemcpy(d, h.data(), 512 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
    hipMemcpy(o.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 4; ++r)
        for (int t = 0; t < 64; ++t) {
            const double want = (r * 128 + 2 * t) * 1000.0 + (r * 128 + 2 * t + 1);
            if (o[r * 64 + t] != want) { if (bad < 5) printf("mismatch r=%d t=%d got %g want %g\n", r, t, o[r * 64 + t], want); ++bad; }
        }
    printf("lds dma probe: %s (%d mismatches)\n", bad ? "DIFFERENT SEMANTICS" : "lane i -> base + 16*i confirmed", bad);
    return bad != 0;
}

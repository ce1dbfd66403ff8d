// Development probe: how fast can ONE wave per CU stream 16-byte-per-lane loads (the ILU solve's access pattern)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NP, int RING>
__global__ __launch_bounds__(64) void k_stream(const double2 *__restrict__ p, double *out, int steps) {
    const int lane = threadIdx.x;
    const double2 *base = p + (long)blockIdx.x * steps * NP * 64 + lane;
    double2 buf[RING][NP];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < RING; ++k)
#pragma unroll
        for (int i = 0; i < NP; ++i) buf[k][i] = base[((long)k * NP + i) * 64];
    for (int s = 0; s < steps; s += RING) {
#pragma unroll
        for (int k = 0; k < RING; ++k) {
            if (s + k < steps) {
#pragma unroll
                for (int i = 0; i < NP; ++i) acc += buf[k][i].x * 1.0000001 + buf[k][i].y;
                if (s + k + RING < steps) {
#pragma unroll
                    for (int i = 0; i < NP; ++i) buf[k][i] = base[((long)(s + k + RING) * NP + i) * 64];
                }
            }
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
}
template <int NP, int RING>
static void run(const double2 *d, double *o, int blocks, int steps, const char *tag) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_stream<NP, RING>), dim3(blocks), dim3(64), 0, 0, d, o, steps);
    hipEventRecord(a);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k_stream<NP, RING>), dim3(blocks), dim3(64), 0, 0, d, o, steps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 10;
    const double bytes = (double)blocks * steps * NP * 1024;
    printf("%-28s blocks %5d steps %4d  %.3f ms  %.2f TB/s  %.2f us/step\n", tag, blocks, steps, ms, bytes / ms / 1e9, ms * 1e3 / steps);
}
int main() {
    const size_t n = (size_t)2048 * 200 * 18 * 64;   // double2 elements
    double2 *d; double *o;
    hipMalloc(&d, n * sizeof(double2)); hipMalloc(&o, 2048 * 64 * 8);
    hipMemset(d, 0, n * sizeof(double2));
    run<14, 3>(d, o, 224, 198, "NP14 ring3, 224 waves");
    run<14, 2>(d, o, 224, 198, "NP14 ring2, 224 waves");
    run<18, 3>(d, o, 224, 198, "NP18 ring3, 224 waves");
    run<6, 3>(d, o, 224, 198, "NP6  ring3, 224 waves");
    run<14, 3>(d, o, 448, 198, "NP14 ring3, 448 waves");
    run<14, 3>(d, o, 1024, 198, "NP14 ring3, 1024 waves");
    run<14, 3>(d, o, 2048, 198, "NP14 ring3, 2048 waves");
    return 0;
}

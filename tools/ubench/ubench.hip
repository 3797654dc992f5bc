// Micro-benchmark of the f64 building blocks of the step kernel on gfx950 (throughput per wave,
// one wave per SIMD and 4 waves per SIMD).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/rs_detmath.h"

#define N_IT 2000
template <int OP>
__global__ void bench(double* out, unsigned long long* cyc, double seed) {
    double a = seed + threadIdx.x * 1e-3, b = 1.0000001 + threadIdx.x * 1e-9, c = 0.5, d = seed * 0.5 + 0.1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
    for (int i = 0; i < N_IT; ++i) {
        if (OP == 0) { a = __builtin_fma(a, b, c); d = __builtin_fma(d, b, c); }           // 2 independent fma chains
        if (OP == 1) { a = a / b; d = d / b; }                                              // 2 independent div chains
        if (OP == 2) { a = rs_exp(a * 1e-3) + c; d = rs_exp(d * 1e-3) + c; }
        if (OP == 3) { a = rs_log(a + 1.5) + c; d = rs_log(d + 1.5) + c; }
        if (OP == 4) { a = a + b; d = d + b; }
        if (OP == 5) { a = 1.0 / (1.0 + rs_exp(-0.3 * (a - 0.25))); d = 1.0 / (1.0 + rs_exp(-0.3 * (d - 0.25))); }
        if (OP == 6) { a = __builtin_rint(a * 1.37) * 0.7; d = __builtin_rint(d * 1.37) * 0.7; }
        if (OP == 7) { a = __builtin_sqrt(a + 2.0); d = __builtin_sqrt(d + 2.0); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int ops_per_iter) {
    double* out; unsigned long long* cyc;
    for (int wps : {1, 4}) {
        int blocks = 256 * wps, threads = 256;   // 256 CUs x wps blocks of 4 waves -> wps waves per SIMD
        hipMalloc(&out, sizeof(double) * blocks * threads);
        hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
        bench<OP><<<blocks, threads>>>(out, cyc, 1.25);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        bench<OP><<<blocks, threads>>>(out, cyc, 1.25);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        printf("%-22s waves/SIMD=%d : %8.1f cycles per op per wave (wall %.3f ms) -> SIMD issue cost %.1f cycles/op\n", name, wps,
               avg / (N_IT * (double)ops_per_iter), ms, avg / (N_IT * (double)ops_per_iter) / wps);
        hipFree(out); hipFree(cyc);
    }
}

int main() {
    run<0>("fma f64", 2);
    run<4>("add f64", 2);
    run<1>("div f64", 2);
    run<7>("sqrt f64", 2);
    run<6>("rint+mul f64", 2);
    run<2>("rs_exp (+mul,add)", 2);
    run<3>("rs_log (+2 add)", 2);
    run<5>("sigmoid (exp+div)", 2);
    return 0;
}

// Does ds_add_f64 resolve the lanes of ONE instruction that hit the same LDS address in a fixed (lane) order?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define NB 8
__global__ __launch_bounds__(1024) void probe(const int* a, const double* w, double* out, int iters) {
    __shared__ double W[16][NB * 4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        if (lane < NB * 4) W[wv][lane] = 0.0;
        const size_t e = ((size_t)(blockIdx.x * iters + it) * 16 + wv) * 64 + lane;
        unsafeAtomicAdd(&W[wv][a[e]], w[e]);
        if (lane < NB) out[((size_t)(blockIdx.x * iters + it) * 16 + wv) * NB + lane] = W[wv][lane];
    }
}
int main() {
    const int blocks = 512, iters = 64;
    const size_t n = (size_t)blocks * iters * 16 * 64, no = (size_t)blocks * iters * 16 * NB;
    std::vector<int> a(n);
    std::vector<double> w(n), out(no);
    srand(1);
    for (size_t i = 0; i < n; ++i) {
        a[i] = rand() % NB;
        w[i] = (double)rand() / RAND_MAX * ((rand() & 1) ? 1e-3 : 1e3) * ((rand() & 2) ? 1 : -1);
    }
    int* da; double *dw, *dout;
    hipMalloc(&da, n * 4); hipMalloc(&dw, n * 8); hipMalloc(&dout, no * 8);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), n * 8, hipMemcpyHostToDevice);
    std::vector<double> first;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(1024), 0, 0, da, dw, dout, iters);
        hipMemcpy(out.data(), dout, no * 8, hipMemcpyDeviceToHost);
        if (rep == 0) first = out;
        else printf("rep %d: %s the first launch\n", rep, memcmp(first.data(), out.data(), no * 8) ? "DIFFERS from" : "bitwise equal to");
    }
    size_t fwd = 0, rev = 0, tot = 0;
    for (size_t g = 0; g < no / NB; ++g) {
        double sf[NB] = {0}, sr[NB] = {0};
        for (int l = 0; l < 64; ++l) sf[a[g * 64 + l]] += w[g * 64 + l];
        for (int l = 63; l >= 0; --l) sr[a[g * 64 + l]] += w[g * 64 + l];
        for (int b = 0; b < NB; ++b) { ++tot; fwd += sf[b] == out[g * NB + b]; rev += sr[b] == out[g * NB + b]; }
    }
    printf("%zu bins: %zu equal the sum in increasing lane order, %zu the sum in decreasing lane order\n", tot, fwd, rev);
    return 0;
}

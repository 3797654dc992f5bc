// Developer probe for the round-1 observation that step-kernel instances with more than ~256 B of scratch per lane
// computed wrong values while <= 240 B ones never did.  ROCr sizes a queue's scratch as bytes/lane x 64 x wave slots
// (256 CUs x 32) and serves a dispatch that needs more than HSA_SCRATCH_SINGLE_LIMIT (default 140 MiB = 280 B/lane on
// this chip) from a "use-once" allocation instead.  This kernel keeps N doubles per lane in scratch (dynamic indexing),
// fills them with a lane-specific pattern, lets the other waves do the same, reads them back and counts mismatches;
// it is run below and above the limit, and again with the limit raised in the environment.
//   hipcc --offload-arch=gfx950 -O2 -o scratch_probe.bin scratch_probe.hip ; ./scratch_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
template <int N>
__global__ __launch_bounds__(256) void probe(const int* perm, unsigned long long* bad, int rounds) {
    double a[N];
    const unsigned long long id = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < N; ++i) a[perm[i] % N] = (double)(id * 1000ull + (unsigned)(perm[i] % N) + (unsigned)r);
        __builtin_amdgcn_s_sleep(64);
        unsigned long long miss = 0;
        for (int i = 0; i < N; ++i) {
            const int k = perm[(i * 7 + r) % N] % N;
            miss += a[k] != (double)(id * 1000ull + (unsigned)k + (unsigned)r);
        }
        if (miss) atomicAdd(bad, miss);
    }
}
template <int N>
static void run(const int* dperm, unsigned long long* dbad) {
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, (const void*)probe<N>);
    (void)hipMemset(dbad, 0, 8);
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(probe<N>, dim3(5120), dim3(256), 0, 0, dperm, dbad, 8);
    const hipError_t e = hipDeviceSynchronize();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 20;
    unsigned long long bad = 0;
    (void)hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
    std::printf("N=%3d doubles: scratch %4zu B/lane (x64x8192 = %6.1f MiB)  %s  mismatches %llu  %.3f ms per launch\n", N,
                (size_t)at.localSizeBytes, at.localSizeBytes * 64.0 * 8192 / 1048576.0, hipGetErrorString(e), bad, ms);
}
int main() {
    int hperm[64];
    for (int i = 0; i < 64; ++i) hperm[i] = (i * 37 + 11) % 64;
    int* dperm;
    unsigned long long* dbad;
    (void)hipMalloc(&dperm, sizeof hperm);
    (void)hipMalloc(&dbad, 8);
    (void)hipMemcpy(dperm, hperm, sizeof hperm, hipMemcpyHostToDevice);
    run<24>(dperm, dbad);
    run<30>(dperm, dbad);
    run<34>(dperm, dbad);
    run<36>(dperm, dbad);
    run<40>(dperm, dbad);
    run<48>(dperm, dbad);
    run<64>(dperm, dbad);
    run<24>(dperm, dbad);
    return 0;
}

// Developer probe: in which order does v_mfma_f64_16x16x4_f64 accumulate its four products?
// D = A(16x4) B(4x16) + C.  Compares the device result bit for bit with host models:
//   chain:    fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C))))
//   rchain:   the same with k = 3..0
//   tree:     C + ((a0 b0 + a1 b1) + (a2 b2 + a3 b3))  (products exact-then-rounded variants are not modelled)
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_order tools/experiments/mfma_order.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, const double* C, double* D) {
    const int lane = threadIdx.x, li = lane & 15, kq = lane >> 4;
    // A operand: lane holds A[li][kq]; B operand: lane holds B[kq][li]; C/D: register r of the lane holds row kq + 4 r, column li
    f64x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(kq + 4 * r) * 16 + li];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[li * 4 + kq], B[kq * 16 + li], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(kq + 4 * r) * 16 + li] = acc[r];
}
int main() {
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    std::uniform_int_distribution<int> e(-40, 40);
    double hA[64], hB[64], hC[256], hD[256];
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    int ok_chain = 0, ok_r = 0, ok_tree = 0, total = 0;
    for (int trial = 0; trial < 200; ++trial) {
        for (auto& v : hA) v = std::ldexp(u(g), e(g) / (trial < 100 ? 8 : 1));
        for (auto& v : hB) v = std::ldexp(u(g), e(g) / (trial < 100 ? 8 : 1));
        for (auto& v : hC) v = std::ldexp(u(g), e(g) / (trial < 100 ? 8 : 1));
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double c = hC[i * 16 + j];
                double ch = c, rc = c;
                for (int k = 0; k < 4; ++k) ch = std::fma(hA[i * 4 + k], hB[k * 16 + j], ch);
                for (int k = 3; k >= 0; --k) rc = std::fma(hA[i * 4 + k], hB[k * 16 + j], rc);
                double p[4];
                for (int k = 0; k < 4; ++k) p[k] = hA[i * 4 + k] * hB[k * 16 + j];
                double tr = c + ((p[0] + p[1]) + (p[2] + p[3]));
                const double d = hD[i * 16 + j];
                ok_chain += std::memcmp(&d, &ch, 8) == 0;
                ok_r += std::memcmp(&d, &rc, 8) == 0;
                ok_tree += std::memcmp(&d, &tr, 8) == 0;
                ++total;
            }
    }
    std::printf("elements %d: chain(k=0..3) %d  reverse chain %d  tree %d\n", total, ok_chain, ok_r, ok_tree);
    return 0;
}

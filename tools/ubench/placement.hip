// Where does the dispatcher put the workgroups of a fully co-resident launch?  (developer tool)
// hipcc --offload-arch=gfx950 -O2 -o placement placement.hip && ./placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
    __shared__ double pad[3900];  // ~31 KB like the step kernel
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    pad[threadIdx.x] = (double)hw;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (pad[threadIdx.x] < 0) out[0] = 0;
}
int main() {
    const int blocks = 1280;
    unsigned* d;
    hipMalloc(&d, blocks * 4 * 2 * sizeof(unsigned));
    std::vector<unsigned> h(blocks * 4 * 2);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, d, 200000);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::map<unsigned, int> per_simd;
        for (int b = 0; b < blocks; ++b) {
            for (int w = 0; w < 4; ++w) {
                unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 15;
                unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                per_simd[(xcc << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd] += 1;
                if (rep == 0 && (b < 40 || (b % 256) < 2) && w < 4)
                    printf("block %4d wave %d : xcc %u se %u sh %u cu %2u simd %u waveid %u\n", b, w, xcc, se, sh, cu, simd, hw & 15);
            }
        }
        std::map<int, int> hist;
        for (auto& kv : per_simd) hist[kv.second] += 1;
        printf("rep %d: %zu distinct SIMDs; waves-per-SIMD histogram:", rep, per_simd.size());
        for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second);
        printf("\n");
    }
    return 0;
}

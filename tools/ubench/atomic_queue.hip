// How long do N one-wave workgroups take when lane 0 of each takes a queue slot with ONE returning atomicAdd on one address?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void q(int* ctr, int* out, int every) {
    int slot = -1;
    if (threadIdx.x == 0 && (blockIdx.x % every) == 0) slot = atomicAdd(ctr, 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot >= 0 && threadIdx.x == 0) out[slot] = blockIdx.x;
}
int main() {
    int *ctr, *out;
    hipMalloc(&ctr, 4); hipMalloc(&out, 4 << 20);
    for (int waves : {20480}) for (int every : {1000000, 16, 4, 1}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int it = 0; it < 3; ++it) { hipMemsetAsync(ctr, 0, 4, 0); hipLaunchKernelGGL(q, dim3(waves), dim3(64), 0, 0, ctr, out, every); }
        hipEventRecord(a);
        for (int it = 0; it < 10; ++it) { hipMemsetAsync(ctr, 0, 4, 0); hipLaunchKernelGGL(q, dim3(waves), dim3(64), 0, 0, ctr, out, every); }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%d one-wave workgroups, every %d-th takes a slot (%d atomics): %.1f us per launch (memset included)\n", waves, every, every > waves ? 0 : waves / every, 100.0 * ms);
    }
    return 0;
}

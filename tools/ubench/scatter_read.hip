// How fast can one-wave workgroups read scattered 6-KB chunks (twelve 512-byte rows, 8 bytes per lane, the next chunk requested while
// the current one is summed -- select_bin_kernel's access pattern) as a function of the footprint they are scattered over?
//   hipcc --offload-arch=gfx950 -O3 -o scatter_read scatter_read.hip && ./scatter_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(64, 4) void scatter(const double* __restrict__ pool, const uint32_t* __restrict__ pages, int chunks,
                                                  size_t page_stride, double* out, int rows) {
    const int lane = threadIdx.x;
    const uint32_t* my = pages + (size_t)blockIdx.x * chunks;
    double acc = 0.0;
    double cur[12], nxt[12];
    const double* P = pool + (size_t)my[0] * page_stride;
#pragma unroll
    for (int r = 0; r < 12; ++r) nxt[r] = r < rows ? P[r * 64 + lane] : 0.0;
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int r = 0; r < 12; ++r) cur[r] = nxt[r];
        if (c + 1 < chunks) {
            const double* Q = pool + (size_t)my[c + 1] * page_stride;
#pragma unroll
            for (int r = 0; r < 12; ++r) nxt[r] = r < rows ? Q[r * 64 + lane] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 12; ++r) acc += cur[r] * cur[r];
    }
    out[(size_t)blockIdx.x * 64 + lane] = acc;
}

int main() {
    const int waves = 24576, chunks = 4, rows = 12;
    const size_t page_doubles = 30 * 64;  // a vector page
    double* out;
    hipMalloc(&out, sizeof(double) * waves * 64);
    for (double gb : {0.5, 1.0, 2.0, 4.0, 8.0, 16.0, 32.0, 64.0}) {
        // pages spread evenly over `gb` GB: stride between candidate pages chosen so that waves * chunks pages cover the span
        const size_t bytes = (size_t)(gb * (1ull << 30));
        const size_t n_pages = (size_t)waves * chunks;
        size_t stride = bytes / 8 / n_pages;          // doubles between page starts
        stride = stride / 64 * 64;
        if (stride < page_doubles) stride = page_doubles;
        double* pool;
        if (hipMalloc(&pool, stride * n_pages * 8 + 4096) != hipSuccess) { printf("%.1f GB: alloc failed\n", gb); break; }
        hipMemset(pool, 0, stride * n_pages * 8);
        std::vector<uint32_t> pg(n_pages);
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < n_pages; ++i) pg[i] = (uint32_t)i;
        for (size_t i = n_pages - 1; i > 0; --i) {  // shuffle: a wave's chunks are far apart, as shells of one dictionary are
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            std::swap(pg[i], pg[x % (i + 1)]);
        }
        uint32_t* dpg;
        hipMalloc(&dpg, 4 * n_pages);
        hipMemcpy(dpg, pg.data(), 4 * n_pages, hipMemcpyHostToDevice);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(scatter, dim3(waves), dim3(64), 0, 0, pool, dpg, chunks, stride, out, rows);
        hipEventRecord(a);
        const int reps = 10;
        for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(scatter, dim3(waves), dim3(64), 0, 0, pool, dpg, chunks, stride, out, rows);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double bytes_read = (double)n_pages * rows * 512;
        printf("footprint %5.1f GB (page stride %8zu B): %7.1f us per launch, %6.0f GB/s\n", gb, stride * 8, 1e3 * ms / reps, bytes_read / (ms / reps * 1e-3) / 1e9);
        hipFree(pool); hipFree(dpg);
    }
    return 0;
}

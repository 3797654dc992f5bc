// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this box: kernels that move a KNOWN number of bytes in the access shapes
// the step kernel and the KBRL kernels use, to be run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes;
// tools/calibrate_fetch.sh).  The guide (MI355X_MICROARCH.md, HBM) gives x 2 for 16-byte-per-lane streams on gfx950 and calls every
// other shape uncalibrated.  Every kernel prints its true byte count; the counter / true ratio per kernel goes to profiles/.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip && ./fetch_calib
// The buffer is 2 GiB (eight times the Infinity Cache) and every kernel touches each of its bytes once.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

// coalesced streams: lane i of the grid reads element i, i + stride, ...
__global__ void read_4B_per_lane(const float* p, size_t n, float* out) {
    float a = 0.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a == 12345.678f) out[0] = a;
}
__global__ void read_8B_per_lane(const double* p, size_t n, double* out) {
    double a = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a == 12345.678) out[0] = a;
}
__global__ void read_16B_per_lane(const f4* p, size_t n, float* out) {
    f4 a = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a.x + a.y + a.z + a.w == 12345.678f) out[0] = a.x;
}
// the state loads of the step kernel: a 16-lane group reads one contiguous 128-byte segment (8 bytes per lane) at a scattered,
// 128-byte-aligned place; every segment of the buffer is read exactly once (seg = a permutation by an odd multiplier)
__global__ void read_128B_segments(const double* p, size_t n_seg, double* out) {
    double a = 0.0;
    const size_t g0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4, ng = ((size_t)gridDim.x * blockDim.x) >> 4;
    for (size_t g = g0; g < n_seg; g += ng) {
        const size_t seg = (g * 2654435761ull) & (n_seg - 1);  // n_seg is a power of two
        a += p[seg * 16 + (threadIdx.x & 15)];
    }
    if (a == 12345.678) out[0] = a;
}
// round 5's fading reads: 33 consecutive doubles (264 bytes) starting at an arbitrary element of a 1,600-byte column, 8 bytes per
// lane, one span per 64-lane wave (lanes 0..32 active); columns visited once each.  True bytes = 264 per span.
__global__ void read_264B_spans_of_1600B_columns(const double* p, size_t n_col, double* out) {
    double a = 0.0;
    const size_t w0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (size_t w = w0; w < n_col; w += nw) {
        const size_t col = (w * 2654435761ull) & (n_col - 1);
        const int off = (int)((col * 40503ull) % 167);  // span start inside the 200-element column
        if (lane < 33) a += p[col * 200 + off + lane];
    }
    if (a == 12345.678) out[0] = a;
}
// round 6's reception sums: 40 consecutive floats (160 bytes) at an arbitrary element of an 800-byte float column, lane j of an
// 8-lane team takes 16 unaligned bytes (lanes 0..9 of a 16-lane half active); one span per 16 lanes
__global__ void read_160B_spans_16B_per_lane_unaligned(const float* p, size_t n_col, float* out) {
    f4 a = {0, 0, 0, 0};
    const size_t g0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4, ng = ((size_t)gridDim.x * blockDim.x) >> 4;
    const int j = threadIdx.x & 15;
    for (size_t g = g0; g < n_col; g += ng) {
        const size_t col = (g * 2654435761ull) & (n_col - 1);
        const int off = (int)((col * 40503ull) % 157);
        if (j < 10) a += *(const f4u*)(p + col * 200 + off + 4 * j);
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) out[0] = a.x;
}
// two scattered 8-byte loads per lane (the prefix-sum channel estimates): 16 bytes of 1,608-byte rows, every row once
__global__ void read_2x8B_scattered(const double* p, size_t n_row, double* out) {
    double a = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_row; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = (i * 2654435761ull) & (n_row - 1);
        const int lo = (int)((row * 40503ull) % 160);
        a += p[row * 201 + lo + 40] - p[row * 201 + lo];
    }
    if (a == 12345.678) out[0] = a;
}
__global__ void write_4B_per_lane(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)i;
}
__global__ void write_8B_per_lane(double* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
__global__ void write_16B_per_lane(f4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f4 v = {(float)i, 1.0f, 2.0f, 3.0f};
        p[i] = v;
    }
}
// scratch-like stores: each lane writes 4 bytes to its own 64-byte-strided... no: the back end swizzles scratch so that a wave's
// dword is 256 contiguous bytes; a spill is a coalesced 4-byte-per-lane store (write_4B_per_lane covers it)
__global__ void write_128B_segments(double* p, size_t n_seg) {
    const size_t g0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4, ng = ((size_t)gridDim.x * blockDim.x) >> 4;
    for (size_t g = g0; g < n_seg; g += ng) {
        const size_t seg = (g * 2654435761ull) & (n_seg - 1);
        p[seg * 16 + (threadIdx.x & 15)] = (double)g;
    }
}

int main() {
    const size_t BYTES = (size_t)2 << 30;
    void* buf = nullptr;
    void* out = nullptr;
    CHECK(hipMalloc(&buf, BYTES + 4096));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 0, BYTES + 4096));
    const dim3 grid(256 * 8), block(256);
    const int REP = 3;
    auto done = [&](const char* name, double bytes) {
        if (hipDeviceSynchronize() != hipSuccess) printf("kernel failed: %s\n", name);
        printf("%-44s true bytes per launch %.0f\n", name, bytes);
    };
    for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(read_4B_per_lane, grid, block, 0, 0, (const float*)buf, BYTES / 4, (float*)out);
    done("read_4B_per_lane", (double)BYTES);
    for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(read_8B_per_lane, grid, block, 0, 0, (const double*)buf, BYTES / 8, (double*)out);
    done("read_8B_per_lane", (double)BYTES);
    for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(read_16B_per_lane, grid, block, 0, 0, (const f4*)buf, BYTES / 16, (float*)out);
    done("read_16B_per_lane", (double)BYTES);
    for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(read_128B_segments, grid, block, 0, 0, (const double*)buf, BYTES / 128, (double*)out);
    done("read_128B_segments", (double)BYTES);
    {
        const size_t n_col = (size_t)1 << 20;  // 1,048,576 columns of 1,600 bytes = 1.56 GiB
        for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(read_264B_spans_of_1600B_columns, grid, block, 0, 0, (const double*)buf, n_col, (double*)out);
        done("read_264B_spans_of_1600B_columns", 264.0 * n_col);
    }
    {
        const size_t n_col = (size_t)1 << 21;  // 2,097,152 float columns of 800 bytes = 1.56 GiB
        for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(read_160B_spans_16B_per_lane_unaligned, grid, block, 0, 0, (const float*)buf, n_col, (float*)out);
        done("read_160B_spans_16B_per_lane_unaligned", 160.0 * n_col);
    }
    {
        const size_t n_row = (size_t)1 << 20;  // rows of 1,608 bytes
        for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(read_2x8B_scattered, grid, block, 0, 0, (const double*)buf, n_row, (double*)out);
        done("read_2x8B_scattered", 16.0 * n_row);
    }
    for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(write_4B_per_lane, grid, block, 0, 0, (float*)buf, BYTES / 4);
    done("write_4B_per_lane", (double)BYTES);
    for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(write_8B_per_lane, grid, block, 0, 0, (double*)buf, BYTES / 8);
    done("write_8B_per_lane", (double)BYTES);
    for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(write_16B_per_lane, grid, block, 0, 0, (f4*)buf, BYTES / 16);
    done("write_16B_per_lane", (double)BYTES);
    for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(write_128B_segments, grid, block, 0, 0, (double*)buf, BYTES / 128);
    done("write_128B_segments", (double)BYTES);
    hipFree(buf);
    hipFree(out);
    return 0;
}

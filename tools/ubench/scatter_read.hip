// How fast can one-wave workgroups read scattered 6-KB chunks (twelve 512-byte rows, 8 bytes per lane, the next chunk requested while
// the current one is summed -- select_bin_kernel's access pattern), and which of that kernel's other ingredients costs what?
//   hipcc --offload-arch=gfx950 -O3 -o scatter_read scatter_read.hip && ./scatter_read
// Variants: +stores (two 512-byte rows written back per chunk), +skew (chunks per wave drawn like config 3's dictionaries at step 3000:
// mean 3.5, a few of 12-26), +chain (three dependent loads before the first chunk), +exp (a dependent chain of 60 f64 operations per chunk),
// +lds (one ds_add_f64 per chunk)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>

struct Args {
    const double* pool;
    double* wpool;
    const uint32_t* pages;   // [waves][max_chunks]
    const int* nchunks;      // [waves]
    const int* chain;        // dependent-load chain table
    size_t page_stride;
    double* out;
    int max_chunks, stores, do_chain, do_exp, do_lds, layout, extra;  // extra: bit 0 a 2-KB result per wave, bit 1 a dependent state load + barriers
    const float* state; double* wg;  // layout 1: the vector page's real rows (0-9 coordinates, 16 coefficient, 21 index)
};

__global__ __launch_bounds__(64, 4) void scatter(Args A) {
    __shared__ double W[256];
    const int lane = threadIdx.x;
    int w = blockIdx.x;
    if (A.do_chain) {  // three dependent loads, as task -> m / shells -> page
        w = A.chain[w];
        w = A.chain[w];
        w = A.chain[w];
    }
    const uint32_t* my = A.pages + (size_t)w * A.max_chunks;
    const int chunks = A.nchunks[w];
    for (int k = lane; k < 256; k += 64) W[k] = 0.0;
    __shared__ double xs[16];
    if (A.extra & 2) {
        if (lane < 10) xs[lane] = (double)A.state[(size_t)(w / 5) * 50 + (w % 5) * 10 + lane];
        __syncthreads();
    }
    double acc = (A.extra & 2) ? xs[lane & 7] : 0.0;
    double cur[12], nxt[12];
    const double* P = A.pool + (size_t)my[0] * A.page_stride;
    const int rmap[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 21};
#pragma unroll
    for (int r = 0; r < 12; ++r) nxt[r] = P[(A.layout ? rmap[r] : r) * 64 + lane];
    for (int c = 0; c < chunks; ++c) {
        double* Pw = A.wpool + (size_t)my[c] * A.page_stride;
#pragma unroll
        for (int r = 0; r < 12; ++r) cur[r] = nxt[r];
        if (c + 1 < chunks) {
            const double* Q = A.pool + (size_t)my[c + 1] * A.page_stride;
#pragma unroll
            for (int r = 0; r < 12; ++r) nxt[r] = Q[(A.layout ? rmap[r] : r) * 64 + lane];
        }
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 12; ++r) s += cur[r] * cur[r];
        if (A.do_exp)
            for (int i = 0; i < 30; ++i) s = __builtin_fma(s, 0.999, 1e-3);
        if (A.stores) {
            Pw[(A.layout ? 17 : 12) * 64 + lane] = s;
            Pw[(A.layout ? 18 : 13) * 64 + lane] = s + 1.0;
        }
        if (A.do_lds) unsafeAtomicAdd(&W[(lane * 7 + c) & 255], s);
        acc += s;
    }
    if (A.extra & 2) __syncthreads();
    if (A.extra & 1)
        for (int k = 0; k < 4; ++k) A.wg[(size_t)w * 256 + lane + 64 * k] = W[lane + 64 * k] + acc;
    A.out[(size_t)blockIdx.x * 64 + lane] = acc + W[lane];
}

int main() {
    const int waves = 24576, max_chunks = 32;
    const size_t page_doubles = 30 * 64;
    const double gb = 8.0;
    const size_t n_pages = (size_t)waves * 4;
    size_t stride = (size_t)(gb * (1ull << 30)) / 8 / n_pages / 64 * 64;
    if (stride < page_doubles) stride = page_doubles;
    double *pool, *out;
    hipMalloc(&pool, stride * n_pages * 8 + 4096);
    hipMemset(pool, 0, stride * n_pages * 8);
    hipMalloc(&out, sizeof(double) * waves * 64);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    std::vector<uint32_t> pg((size_t)waves * max_chunks);
    for (auto& p : pg) p = (uint32_t)(rnd() % n_pages);
    uint32_t* dpg;
    hipMalloc(&dpg, 4 * pg.size());
    hipMemcpy(dpg, pg.data(), 4 * pg.size(), hipMemcpyHostToDevice);
    std::vector<int> chain(waves);
    for (int i = 0; i < waves; ++i) chain[i] = i;
    std::shuffle(chain.begin(), chain.end(), std::default_random_engine(7));
    int* dchain;
    hipMalloc(&dchain, 4 * waves);
    hipMemcpy(dchain, chain.data(), 4 * waves, hipMemcpyHostToDevice);
    // chunk counts: uniform 4, or skewed with the same total (most 1-4, a tail up to 26; the largest first, as the big list is launched)
    std::vector<int> uni(waves, 4), skew(waves);
    long tot = 0;
    for (int i = 0; i < waves; ++i) {
        const double u = (rnd() % 100000) / 100000.0;
        int c = u < 0.30 ? 1 : u < 0.55 ? 2 : u < 0.72 ? 3 : u < 0.83 ? 4 : u < 0.90 ? 6 : u < 0.95 ? 8 : u < 0.985 ? 12 : u < 0.998 ? 18 : 26;
        skew[i] = c;
        tot += c;
    }
    std::sort(skew.begin(), skew.end(), std::greater<int>());
    int *duni, *dskew;
    hipMalloc(&duni, 4 * waves); hipMalloc(&dskew, 4 * waves);
    hipMemcpy(duni, uni.data(), 4 * waves, hipMemcpyHostToDevice);
    hipMemcpy(dskew, skew.data(), 4 * waves, hipMemcpyHostToDevice);
    printf("skewed chunk counts: mean %.2f\n", (double)tot / waves);
    struct V { const char* name; int skewed, stores, chain, ex, lds, layout, extra; };
    float* dstate; double* dwg;
    hipMalloc(&dstate, 4 * 50 * 8192); hipMemset(dstate, 0, 4 * 50 * 8192);
    hipMalloc(&dwg, 8 * 256 * (size_t)waves);
    const V vs[] = {{"plain", 0, 0, 0, 0, 0, 0}, {"+stores", 0, 1, 0, 0, 0, 0}, {"+chain", 0, 0, 1, 0, 0, 0}, {"+exp", 0, 0, 0, 1, 0, 0}, {"+lds", 0, 0, 0, 0, 1, 0},
                    {"+skew", 1, 0, 0, 0, 0, 0}, {"+skew+stores", 1, 1, 0, 0, 0, 0}, {"all", 1, 1, 1, 1, 1, 0}, {"all but skew", 0, 1, 1, 1, 1, 0},
                    {"real rows", 0, 0, 0, 0, 0, 1}, {"real rows+stores", 0, 1, 0, 0, 0, 1}, {"real rows, all", 1, 1, 1, 1, 1, 1},
                    {"contiguous, all", 1, 1, 1, 1, 1, 0}, {"all + 2 KB result", 1, 1, 1, 1, 1, 1, 1}, {"all + state/barriers", 1, 1, 1, 1, 1, 1, 2},
                    {"all + both", 1, 1, 1, 1, 1, 1, 3}};
    for (const V& v : vs) {
        Args A = {pool, pool, dpg, v.skewed ? dskew : duni, dchain, stride, out, max_chunks, v.stores, v.chain, v.ex, v.lds, v.layout, v.extra, dstate, dwg};
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(scatter, dim3(waves), dim3(64), 0, 0, A);
        hipEventRecord(a);
        const int reps = 10;
        for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(scatter, dim3(waves), dim3(64), 0, 0, A);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double chunks = v.skewed ? (double)tot : 4.0 * waves;
        const double bytes = chunks * (12 * 512 + (v.stores ? 1024 : 0));
        printf("%-14s %7.1f us per launch, %6.0f GB/s (reads + writes)\n", v.name, 1e3 * ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
    }
    return 0;
}

// kb_kbrl.hip -- the KBRL agent's kernel machinery on gfx950, batched over env replicas (third version: round 3).
//
// What it computes (reference call tree):
//   GaussianKernel.k / predict       algorithms/kernel.py:8-28
//   Projectron.predict / update      algorithms/projectron.py:32-60
//   KBRL_Control.select_action / adjust_action / update_control   kbrl_control.py:41-114
//
// Storage.  A dictionary (landmarks, coefficients, Kinv: SVvariable + Projectron.Kinv, projectron.py:3-30) lives in
// a POOL shared by all dictionaries of the handle and grows 64 landmarks at a time, like the reference's np.vstack /
// np.append / np.column_stack do one landmark at a time (projectron.py:17-21,52-57): when landmark 64 b arrives the
// learner's own workgroup takes "shell" b from a bump allocator --
//     [ vector page: 30 rows x 64 doubles ]  coordinates (16 rows, coordinate-major), coefficient, scratch rows
//     [ b + 1 tiles of 64 x 64 doubles    ]  Kinv tiles (b,0) .. (b,b): the LOWER block triangle (round 4)
//     [ b + 1 times 128 doubles           ]  the tiles' partial sums of d* = Kinv K_f (matvec_tri_tiles)
// and records its pool offset in the dictionary's shell table.  Kinv is symmetric bit for bit (it only ever receives
// (d_i d_j) / delta), so the upper block triangle is never stored: the rank-1 update streams half of what it did and
// the pool holds twice the landmarks.  (Shared dictionaries -- five per handle -- keep both triangles, 2 b + 1 tiles per
// shell, and the round-3 kernels that walk them: KbDev.tri = 0.)  Nothing is ever moved or freed before kb_reset, so
// the capacity is bounded by the pool (device memory), not by a per-learner reservation: a handle of 4096 x 5
// learners costs what its dictionaries hold (48 KB each while below 64 landmarks; round 2 reserved 8 MB each).
//
// Scoring.  Both hot loops of the agent -- the augmentation loop of update_control and the scan of select_action --
// evaluate the classifier on candidates x_c = (state, c / n_prbs) that differ only in the last coordinate, and the
// last coordinate of every landmark they ever inserted is itself a / n_prbs for an integer a.  So
//     k(l_j, x_c) = exp(-g |l_j[:d-1] - state|^2) * exp(-g ((a_j - c) / n)^2) = E_j * G[|a_j - c|]
// with ONE exp per landmark (E_j) and a 256-entry table G that depends on (gamma, n_prbs) only:
//     f(c) = sum_j (coeff_j E_j) G[|a_j - c|]
// One wavefront owns one learner; lane = candidate (c = 64 g + lane), landmarks stream through in 64-chunks
// (coalesced rows of the vector pages), their (w_j, a_j) are broadcast with v_readlane and every lane walks the G
// table in LDS at consecutive addresses (conflict-free).  Per (landmark, 64 candidates): one LDS read and one FMA
// instead of 64 exps -- the round-2 form (one v_mfma_f64_16x16x4 distance tile + 256 exps) issued 6x more
// instructions and could not leave the exp bound.  Landmarks with an off-grid last coordinate (inserted through the
// N=1 Projectron.update entry point with an arbitrary x) take the direct exp, per landmark.
// The reference's sequentially dependent loop (predict, update, predict, ...) is reproduced exactly in order: score
// the whole remaining range, find the first mistake, apply that one Projectron update, rescore what follows (H4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rs_philox.h"

namespace kb {

#define KB_DMAX 16        // max len(x)
#define KB_NPRB_MAX 255   // candidates 0..n_prbs fit four 64-lane groups
#define KB_GTAB 264       // G[k], k = 0..255 (+ padding)
#define KB_CH 64          // landmarks per chunk
#define KB_TILE 4096      // doubles per Kinv tile
#define KB_ROW_F32 11     // eMBB dictionaries (ten state coordinates + the allocation: rows 0..10): rows 11..15 hold the ten state
                          //   coordinates again as float32, [coordinate][64 landmarks] -- they ARE float32 observations
                          //   (slice_ran.py:321-325), so the copy is exact and the binning pass of select_action reads 40 B per
                          //   landmark where the f64 rows are 80 (round 6; K.f32bad says when a dictionary cannot use it)
#define KB_ROW_CO 16      // rows of a vector page (64 doubles each): 0..15 coordinates, then
#define KB_ROW_D0 17      //   |l_j[:d-1] - state|^2 of the state being processed
#define KB_ROW_E 18       //   exp(-gamma D0)
#define KB_ROW_KF 19      //   kernel column K_f of the sample being learned (also Projectron's cached K_f, projectron.py:34)
#define KB_ROW_DS 20      //   d* = Kinv K_f
#define KB_ROW_IDX 21     //   64 x int32 grid index a_j of the last coordinate (-1: off the grid), 64 x int32 chain link
#define KB_ROW_PART 22    //   8 rows: the partial sums of d* over the row classes j mod 8 (matvec_colsum)
#define KB_VEC_ROWS 30
#define KB_VEC (KB_VEC_ROWS * KB_CH)
#ifndef KB_OCC
#ifndef KB_SCORE_UNROLL1
#define KB_SCORE_UNROLL1 8  // landmarks per unrolled block of the scoring loop when one group of candidates is scored
#endif
#define KB_OCC 4        // waves per SIMD the one-wave kernels are built for
#endif
#define KB_SMALL_M 192     // dictionaries below this repair their mistakes on one wave, larger ones on a workgroup
#define KB_HEAVY_THREADS 512
#define KB_BIG_M 320       // dictionaries from this size on are launched first by the one-wave kernels
#define KB_BIG_MAX 4096    // at most this many (the rest keep their place)
#define KB_GEMM_M 256      // shared dictionaries from this size on are scored for all replicas at once as F = E Q on MFMA
#define KB_GEMM_KS 8       // the landmarks are split in this many parts (one wave each per 16 replicas)
#define KB_E_TINY 1e-300   // below this E_j G[.] (G >= 0.19) would come near the end of the normal range (2.2e-308), where a product of
                           // two roundings is no longer the kernel value to a few ulp: direct evaluation (score_pass, bin_pass)
#define KB_DLIST 48        // landmarks of one scoring pass that take the direct evaluation, listed by bin_pass (more: the rows are walked again)
#define KB_HEAD 256       // ints per dictionary: newest landmark per grid index (chains through the link row)
#define KB_BIN_M 128      // a repair rescoring a dictionary of this many landmarks or more uses the binned form (score_binned)
#define KB_SEL_WAVES 16   // learners per workgroup of select_gemm_kernel: the N dimension of its v_mfma_f64_16x16x4 tiles

typedef double kb_f64x4 __attribute__((ext_vector_type(4)));
typedef double kb_f64x2 __attribute__((ext_vector_type(2)));

struct KbDev {
    int32_t n_envs, S, n_prbs, cap, nv;
    int32_t max_shells;  // shell-table entries per dictionary = ceil(cap / 64)
    int32_t dims[8];  // state variables per learner (len(x) = dims+1)
    int32_t off[8];   // first state variable of learner s
    double alfa, lo, hi, gamma, eta;
    int32_t shared;   // 1: one dictionary per slice shared by all replicas (build-defined extension)
    int32_t first_env; // global id of local replica 0 (shared mode proposals carry global ids)
    int32_t serial_apply; // shared mode: apply a full dictionary's proposals one by one as well (KBRL_SERIAL_APPLY, tests)
    int32_t heavy_m;      // dictionaries of this many landmarks repair their mistakes in update_heavy_kernel
    int32_t tri;          // 1: only the lower block triangle of Kinv is stored (one agent per replica); 0: both (shared dictionaries)
    uint64_t pool_doubles;
};

// dictionary a learner (task = env * S + s) reads and writes
__device__ __forceinline__ int dict_of(const KbDev& D, int task) { return D.shared ? task % D.S : task; }
struct KbState;
__device__ __forceinline__ bool gemm_applies(const KbDev& D, const KbState& K, int s, int m);

struct KbState {
    int32_t* m;        // [ND] landmarks per dictionary
    uint64_t* shell;   // [ND][max_shells] pool offset (in doubles) of shell b; 0 = not allocated yet
    int32_t* head;     // [ND][KB_HEAD] newest landmark with grid index a (-1 none)
    double* pool;      // the pool
    unsigned long long* pool_top;  // [1] next free double
    double* gtab;      // [KB_GTAB] G[k] = exp(-gamma (k / n_prbs)^2)
    double* f_last;    // [T]
    int32_t* m_last;   // [T] dictionary size the cached (f_last, K_f row) was computed against (Q12 guard)
    int32_t* kf_owner; // [ND] task whose predict filled the dictionary's K_f row
    uint32_t* tie_ctr; // [T] Philox counter of the tie-break stream (kernel.py:26-27)
    uint64_t* seeds;   // [n_envs]
    int32_t* action;   // [n_envs][S]
    int32_t* security; // [n_envs][S]
    int32_t* margins;  // [n_envs][S]
    int32_t* adjusted; // [n_envs]
    double* acc;       // [n_envs][S][n_prbs]
    int32_t* err;      // [n_envs]  bit 8: a dictionary is at its capacity, bit 16: the pool is exhausted
    uint64_t* stats;   // [T][4]: predicts, mistakes, grows, kernel evaluations (candidates x landmarks)
    int32_t* heavy;    // [4 + T]: large learners queued, next to take, small learners queued, -; then the task ids (large
                       // from the front, small from the back)
    // the queued large learners between the kernels of the repair rounds (heavy_*_kernel), by queue slot
    int32_t* hv_cfrom;  // [T] first candidate of the range not yet examined
    int32_t* hv_cstar;  // [T] the mistake whose kernel column is in the K_f row
    int32_t* hv_state;  // [T] 1: a repair is pending, 0: done
    int32_t* hv_grew;   // [T] 1: the round's update inserted a landmark: Kinv still needs its rank-1 update
    int32_t* hv_m;      // [T] dictionary size before that insertion
    int32_t* hv_pend;   // [T][2] predictions / exact ties of the segment up to hv_cstar, counted when it is applied
    double* hv_delta;   // [T]
    long long* hv_mvbase;  // [T + 1] prefix sums of the mat-vec work of the pending learners (heavy_plan_kernel)
    long long* hv_r1base;  // [T + 1] prefix sums of the rank-1 work of the learners that inserted
    unsigned long long* hv_work;  // [8] since kb_reset ([4], [5]: scoring passes that took direct exponentials, landmarks they evaluated): tile passes of the chip-wide mat-vec kernel (8 rows x 512 B each), 16-row units of
                                  // the rank-1 kernel (8 KB read + 8 KB written each), launches of either that had work
    double* hv_f;       // [T][256] the scores of the candidates
    // launch order of the one-wave kernels: learners with large dictionaries first (their waves are the long ones)
    int32_t* big;      // [2][1 + KB_BIG_MAX]: count, then the learners select_kernel found at KB_BIG_M landmarks or more
    int32_t* isbig;    // [2][T] membership of that list
    double* workb;     // shared mode: [S][2][budget_cap][cap rounded up to 64] kernel columns and d* of a proposal list
    int32_t* offgrid;  // [ND] landmarks whose last coordinate is off the candidate grid (inserted through kb_update)
    int32_t* f32bad;   // [ND] landmarks of an eMBB dictionary with a state coordinate that is not a float32 value (inserted through
                       //      kb_update): the binning pass then reads the f64 coordinate rows instead of their float32 copy
    // the scores select_action formed for every candidate of its state (select_gemm_kernel): update_control of the SAME
    // state against the SAME dictionary (the next call of KBRL_Control.run's loop, kbrl_control.py:129-134) starts from them
    double* F;         // [T][256]
    float* fstate;     // [T][16] the state F was computed for
    int32_t* fver;     // [T] K.ver[dict] when F was computed (-1: none)
    int32_t* ver;      // [ND] bumped by every Projectron.update that changed the dictionary (finish_update)
    double* Wg;        // [T][256] W[a] of select_action's state, from select_bin_kernel to select_gemm_kernel
    double* dlist;     // [T][KB_DLIST][3] (coeff, last coordinate, D0) of those landmarks (bin_pass)
    int32_t* fdirect;  // [T] bin_pass's flags: the learner has landmarks that take the direct evaluation for that state (1) / off the grid (2)
    double* workq;     // shared mode: [S][16][capr][16] Q[j][c] = coeff_j G[|a_j - c|] in MFMA B-operand tiles (shared_q_kernel)
    double* workF;     // shared mode: [S][KB_GEMM_KS][n_envs][256] partial scores F = E Q (shared_fgemm_kernel)
    double* workE;     // shared mode: [S][KB_GEMM_KS][n_envs] largest E_j a replica met in its part of the landmarks
    double* workg;     // shared mode: [S][budget_cap][budget_cap] the proposals' Gram block (shared_gram_kernel)
    double* workf;     // shared mode: [S][budget_cap] f_p^0
};

// a shared dictionary large enough (and on the candidate grid) to be scored for all replicas at once (shared_fgemm_kernel)
__device__ __forceinline__ bool gemm_applies(const KbDev& D, const KbState& K, int s, int m) {
    return D.shared && m >= KB_GEMM_M && K.offgrid[s] == 0;
}
__host__ __device__ inline int kb_capr(int cap) { return (cap + 63) & ~63; }
__host__ __device__ inline size_t kb_apply_lds_doubles(int cap, int budget) {  // shared_apply_kernel's dynamic LDS
    return (size_t)kb_capr(cap) + 4 * (size_t)budget;
}
__host__ __device__ inline uint64_t kb_shell_doubles(int b, int tri) {  // tri: b + 1 tiles and their 128 partial sums each
    return (uint64_t)KB_VEC + (tri ? (uint64_t)(b + 1) * (KB_TILE + 128) : (uint64_t)(2 * b + 1) * KB_TILE);
}

__device__ __forceinline__ const uint64_t* shells_of(const KbDev& D, const KbState& K, int dict) {
    return K.shell + (size_t)dict * D.max_shells;
}
__device__ __forceinline__ double* vec_page(const KbState& K, const uint64_t* sh, int b) { return K.pool + sh[b]; }
// tile holding Kinv[i][j] for i in row block bi, j in column block bj (row-major 64 x 64).  kinv_tile_lo: bj <= bi, the
// tiles every dictionary stores; kinv_tile: any (bi, bj) of a dictionary that stores both triangles (KbDev.tri == 0).
__device__ __forceinline__ double* kinv_tile_lo(const KbState& K, const uint64_t* sh, int bi, int bj) {
    return K.pool + sh[bi] + KB_VEC + (size_t)bj * KB_TILE;
}
__device__ __forceinline__ double* kinv_tile(const KbState& K, const uint64_t* sh, int bi, int bj) {
    return bj <= bi ? K.pool + sh[bi] + KB_VEC + (size_t)bj * KB_TILE
                    : K.pool + sh[bj] + KB_VEC + (size_t)(bj + 1 + bi) * KB_TILE;
}
// element j of row `row` of the dictionary's vector pages
__device__ __forceinline__ double* vec_at(const KbState& K, const uint64_t* sh, int row, int j) {
    return vec_page(K, sh, j >> 6) + row * KB_CH + (j & 63);
}
__device__ __forceinline__ int32_t* idx_at(const KbState& K, const uint64_t* sh, int j) {
    return (int32_t*)(vec_page(K, sh, j >> 6) + KB_ROW_IDX * KB_CH) + (j & 63);
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return u.d;
}

__device__ __forceinline__ int tie_draw(const KbState& K, int task, int env, int s) {
    uint64_t seed = K.seeds[env];
    rs_stream st = {(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)s, 0xFFFFFFFFu, K.tie_ctr[task]};
    int v = rs_stream_pm1(&st);
    K.tie_ctr[task] = st.ctr;
    return v;
}

// grid index of a last coordinate t: a with (double)a / n == t exactly, else -1
__device__ __forceinline__ int grid_index(double t, int n) {
    const double r = __builtin_rint(t * (double)n);
    if (!(r >= 0.0 && r <= (double)n)) return -1;
    const int a = (int)r;
    return (double)a / (double)n == t ? a : -1;
}

// small per-block staging area (static LDS; nothing in these kernels scales with the capacity)
struct Lds {
    double G2[512];  // the kernel's last-coordinate factor, two-sided: G2[256 + k] = G[|k|], |k| <= 255 (no abs in the hot loop)
    double x[KB_DMAX];
    double red[16];
    double fbuf[256];  // the scores of the 256 candidates, handed from wave 0 to the other waves of a multi-wave block
    double W[256];     // binned scoring: W[a] = sum of coeff_j E_j over the landmarks with grid index a (score_binned); the segment in
                       //   progress of a dictionary of more than KB_BIN_SEG chunks borrows fbuf (never live during a binning pass)
    double dl[KB_DLIST * 3];  //   and the (coeff, last coordinate, D0) of the landmarks that take the direct evaluation (bin_pass)
    int ired[8];
};

__device__ __forceinline__ void load_gtab(const KbState& K, Lds& sm) {
    for (int k = threadIdx.x; k < 512; k += blockDim.x) sm.G2[k] = K.gtab[k < 256 ? 256 - k : k - 256];
}

// sum over the block, any block size that is a multiple of 64: butterfly per wave, wave totals in order
__device__ __forceinline__ double block_sum(double v, Lds& sm) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm.red[w];
    __syncthreads();
    return t;
}

// Sum_j a[j] b[j] over two rows of the vector pages, formed as 256 strided partial sums (j = t, t + 256, ...), the xor
// butterfly within each group of 64, and the four group totals in order -- by ONE wave, whatever the block size, so that
// every kernel (64-thread per-replica learners, 256-thread entry points, 1024-thread shared apply) gets the same bits.
// All lanes of the calling wave return the value.
__device__ __forceinline__ double wave_dot256_rows(const KbState& K, const uint64_t* sh, int row_a, int row_b, int m) {
    const int lane = threadIdx.x & 63;
    double part[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        part[v] = 0.0;
        for (int j = 64 * v + lane; j < m; j += 256) {
            const double* P = vec_page(K, sh, j >> 6);
            part[v] += P[row_a * KB_CH + lane] * P[row_b * KB_CH + lane];
        }
    }
    for (int dd = 32; dd >= 1; dd >>= 1) {  // four independent butterflies, stepped together
#pragma unroll
        for (int v = 0; v < 4; ++v) part[v] += __shfl_xor(part[v], dd);
    }
    double t = 0.0;
#pragma unroll
    for (int v = 0; v < 4; ++v) t += __shfl(part[v], 0);
    return t;
}
// the same sum over two plain arrays
__device__ __forceinline__ double wave_dot256(const double* a, const double* b, int m) {
    const int lane = threadIdx.x & 63;
    double part[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        part[v] = 0.0;
        for (int j = 64 * v + lane; j < m; j += 256) part[v] += a[j] * b[j];
    }
    for (int dd = 32; dd >= 1; dd >>= 1) {
#pragma unroll
        for (int v = 0; v < 4; ++v) part[v] += __shfl_xor(part[v], dd);
    }
    double t = 0.0;
#pragma unroll
    for (int v = 0; v < 4; ++v) t += __shfl(part[v], 0);
    return t;
}

// |l_j[:d-1] - x|^2 for the landmark this lane holds in page P, summed in coordinate order
__device__ __forceinline__ double dist0(const double* P, int lane, int d, const double* x) {
    double d0 = 0.0;
    if (d - 1 == 10) {  // eMBB learners (scenario_creator.py:80-82): all ten loads in flight
        double v[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) v[q] = P[q * KB_CH + lane];
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const double t = v[q] - x[q];
            d0 += t * t;
        }
    } else {
        for (int q = 0; q < d - 1; ++q) {
            const double t = P[q * KB_CH + lane] - x[q];
            d0 += t * t;
        }
    }
    return d0;
}

// ---------------------------------------------------------------------------------------------- streaming scoring
// f[g] = f(c) for c = 64 (g0 + g) + lane, g < NG, against the m landmarks of the dictionary, by ONE wave.
// MODE 0: compute D0 / E for the state in sm.x and keep them in the dictionary's rows (the learner owns its dictionary)
// MODE 1: reuse the E row (same state, coefficients may have changed)
// MODE 2: compute E, store nothing (shared dictionaries: many learners read the same pages)
// the rows of one chunk a scoring pass needs, loaded one chunk ahead of their use
template <int MODE>
struct ChunkRows {
    double v[MODE == 1 ? 1 : 10];  // MODE 1: E; else the first ten coordinates (eMBB learners) for D0
    double co;
    int a;
};
template <int MODE>
__device__ __forceinline__ void load_chunk(const double* P, int lane, int d, ChunkRows<MODE>& R, bool f32 = false) {
    if (MODE == 1) {
        R.v[0] = P[KB_ROW_E * KB_CH + lane];
    } else if (d - 1 == 10) {
        if (f32) {  // (wave-uniform) the float32 copy of the ten state coordinates: the same numbers in half the bytes
            const float* F = (const float*)(P + KB_ROW_F32 * KB_CH);
#pragma unroll
            for (int q = 0; q < 10; ++q) R.v[q] = (double)F[q * KB_CH + lane];
        } else {
#pragma unroll
            for (int q = 0; q < 10; ++q) R.v[q] = P[q * KB_CH + lane];
        }
    }
    R.co = P[KB_ROW_CO * KB_CH + lane];
    R.a = ((const int32_t*)(P + KB_ROW_IDX * KB_CH))[lane];
}

template <int NG, int MODE>
__device__ __forceinline__ void score_pass(const KbDev& D, const KbState& K, const uint64_t* sh, int m, int d, const Lds& sm,
                                           int c_base, int ng, double (&f)[NG]) {
    const int lane = threadIdx.x & 63;
    int cb[NG], cs[NG];  // candidate x 8; (256 - candidate) x 8: the byte offset of G2[256 + a - c] is a8 + cs
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = c_base + 64 * g + lane;
        cb[g] = (c < D.n_prbs ? c : D.n_prbs) * 8;  // lanes past the last candidate are never looked at
        cs[g] = (256 - (c < D.n_prbs ? c : D.n_prbs)) * 8;
        f[g] = 0.0;
    }
    const int nch = (m + 63) >> 6;
    const char* Gb = (const char*)sm.G2;
    ChunkRows<MODE> R, Rn;
    load_chunk<MODE>(vec_page(K, sh, 0), lane, d, Rn);
    for (int b = 0; b < nch; ++b) {
        double* P = vec_page(K, sh, b);
        R = Rn;
        if (b + 1 < nch) load_chunk<MODE>(vec_page(K, sh, b + 1), lane, d, Rn);
        const int cnt = m - 64 * b < 64 ? m - 64 * b : 64;
        double E, d0 = 0.0;
        if (MODE == 1) {
            E = R.v[0];
        } else {
            if (d - 1 == 10) {  // eMBB learners (scenario_creator.py:80-82)
#pragma unroll
                for (int q = 0; q < 10; ++q) {
                    const double t = R.v[q] - sm.x[q];
                    d0 += t * t;
                }
            } else {
                d0 = dist0(P, lane, d, sm.x);
            }
            E = rs_exp_nonpos(-D.gamma * d0);
            if (MODE == 0) {
                P[KB_ROW_D0 * KB_CH + lane] = d0;
                P[KB_ROW_E * KB_CH + lane] = E;
            }
        }
        // The product E_j G[.] is the kernel value to a few ulp only while E_j is a normal number.  An outlier state (a
        // normalised queue of 60 against landmarks near 1) puts every E_j at 1e-300 and below, where the reference's
        // exp(-gamma (D0 + dl^2)) is a handful of subnormal quanta or exactly zero (its f == 0 ties, kernel.py:26-27):
        // such landmarks -- and those whose last coordinate is off the grid -- take the exponential itself, after the
        // chunk's table terms (E_j == 0 exactly means -gamma D0 < -745.2: zero for every candidate, the landmark drops out).
        const bool direct = lane < cnt && (R.a < 0 || (!(E >= KB_E_TINY) && E > 0.0));
        const bool table = lane < cnt && !direct && R.a >= 0;
        const double w = table ? R.co * E : 0.0;
        const int a8 = table ? R.a * 8 : 0;
        // lanes without a table term carry w = 0: the loop may run to the next multiple of the unroll factor (a single group
        // of candidates has one table read per landmark: more of them in flight)
        constexpr int UNR = NG == 1 ? KB_SCORE_UNROLL1 : 4;
        for (int jj0 = 0; jj0 < cnt; jj0 += UNR) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int jj = jj0 + u;
                const double ws = readlane_f64(w, jj);
                const int as8 = __builtin_amdgcn_readlane(a8, jj);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g < ng) {  // (wave-uniform: only the groups the window reaches)
                        f[g] = __builtin_fma(ws, *(const double*)(Gb + (as8 + cs[g])), f[g]);
                    }
                }
            }
        }
        unsigned long long dm = __ballot(direct);
        if (dm) {
            if (MODE == 1) d0 = P[KB_ROW_D0 * KB_CH + lane];
            const double lam = P[(d - 1) * KB_CH + lane];
            while (dm) {
                const int jj = __builtin_ctzll(dm);
                dm &= dm - 1ull;
                const double cs = readlane_f64(R.co, jj), ls = readlane_f64(lam, jj), d0s = readlane_f64(d0, jj);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const double dl = ls - (double)(cb[g] >> 3) / (double)D.n_prbs;
                    f[g] = __builtin_fma(cs, rs_exp_nonpos(-D.gamma * (d0s + dl * dl)), f[g]);
                }
            }
        }
    }
}

// the single-landmark dictionary: numpy keeps k and coeff in float32 (kernel.py:16, projectron.py:9)
template <int NG, int MODE>
__device__ __forceinline__ void score_single(const KbDev& D, const KbState& K, const uint64_t* sh, int d, const double* x, int c_base,
                                             double (&f)[NG]) {
    const int lane = threadIdx.x & 63;
    double* P = vec_page(K, sh, 0);
    double d0 = 0.0;
    for (int q = 0; q < d - 1; ++q) {
        const double t = P[q * KB_CH] - x[q];
        d0 += t * t;
    }
    if (MODE == 0 && lane == 0) {
        P[KB_ROW_D0 * KB_CH] = d0;
        P[KB_ROW_E * KB_CH] = rs_exp_nonpos(-D.gamma * d0);
    }
    const double lam = P[(d - 1) * KB_CH], co = P[KB_ROW_CO * KB_CH];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = c_base + 64 * g + lane;
        const double t = (double)(c < D.n_prbs ? c : D.n_prbs) / (double)D.n_prbs;
        const double dl = lam - t;
        const double k = rs_exp(-D.gamma * (d0 + dl * dl));
        f[g] = (double)(float)((float)k * (float)co);
    }
}

// f[g] = f(c_base + 64 g + lane) for the first ng (<= NG) groups of 64 candidates
template <int NG, int MODE>
__device__ __forceinline__ void score(const KbDev& D, const KbState& K, const uint64_t* sh, int m, int d, const Lds& sm, int c_base,
                                      int ng, double (&f)[NG]) {
    if (m == 0) {
#pragma unroll
        for (int g = 0; g < NG; ++g) f[g] = 0.0;
    } else if (m == 1) {
        score_single<NG, MODE>(D, K, sh, d, sm.x, c_base, f);
    } else {
        score_pass<NG, MODE>(D, K, sh, m, d, sm, c_base, ng, f);
    }
}


// ------------------------------------------------------------------------------------------------ binned scoring (round 4)
// f(c) = sum_j (coeff_j E_j) G[|a_j - c|] only depends on the landmarks through the sums per grid index,
//     W[a] = sum_{j : a_j = a} coeff_j E_j          (one pass over the landmarks: the exp, nothing per candidate)
//     f(c) = sum_a G[|a - c|] W[a]                   (a Toeplitz matrix times a vector: no trace of m left)
// so the cost of scoring EVERY candidate of a state stops depending on the dictionary's size, and the second line is a
// dense product T W, T[c][a] = G[|a - c|] the same for every learner of the handle: select_gemm_kernel forms it for
// sixteen learners at a time on the matrix cores (v_mfma_f64_16x16x4, N = learners).  The order of both sums is fixed so
// that every kernel produces the same bits:
//   * W[a] takes its landmarks in increasing j (bin_pass: one wave, chunk after chunk, a chunk's lanes with one ds_add_f64
//     that the LDS resolves in lane order -- measured, see bin_pass);
//   * f(c) is ONE chain of fused multiply-adds over a = 0, 1, ..., KA - 1 (KA = n_prbs + 1 rounded up to four) starting
//     from zero -- which is what consecutive v_mfma_f64_16x16x4 on one accumulator compute (tools/experiments/mfma_order.hip)
//     and what chain_scores spells out on the vector ALU;
//   * landmarks that cannot be binned (off the candidate grid, or E_j below KB_E_TINY: see score_pass) add their exact
//     exponentials afterwards, in increasing j (add_direct_terms).
// The reference's k @ coeff is a BLAS dot in no particular order (kernel.py:24); like score_pass's table form these
// sums agree with it to a few ulp of sum |coeff_j k_j| (tests: f within 1e-9 (1 + sum |w|), every decision exact).

// W[a] += coeff_j E_j over the dictionary, by ONE wave, for the state x.  MODE 0: D0 / E are computed and left in the
// dictionary's rows; MODE 1: the E row is reused.  W (LDS, 256 doubles) zeroed on entry.  The lanes of a chunk add their
// terms with ONE ds_add_f64: the LDS resolves the lanes of an instruction that hit the same address in increasing lane
// order (tools/experiments/lds_add_order.hip, profiles/r04_lds_add_order.txt: 4,194,304 of 4,194,304 bins bit for bit the
// lane-ordered sum, sixteen waves of a CU at it together, three launches identical), and the LDS executes a wave's
// instructions in order -- so every W[a] is the sum over its landmarks in increasing j, on every run and in every kernel.
// Returns bit 0: some landmark takes the direct evaluation; bit 1: some landmark is off the candidate grid; bits 8..: how
// many take it -- their (coeff, last coordinate, D0) are listed in dlist, in increasing j, the first KB_DLIST of them (a state
// far from everything the dictionary holds leaves a handful of landmarks in the band where E_j is 1e-300 .. 5e-324: common
// enough -- thousands of learners per step in BASELINE config 3 -- that walking the rows a second time for them showed).
#ifndef KB_BIN_DEEP
#define KB_BIN_DEEP 0  // chunks whose rows a wave of the select_bin kernels requests together (0: one ahead, rotating registers;
                       // 2 and 4 measured no faster: profiles/r05_kbrl_variants.txt)
#endif
#ifndef KB_F32_ROWS
#define KB_F32_ROWS 1  // the select_bin kernels read the float32 copy of the state coordinates (KB_ROW_F32)
#endif
#ifndef KB_BIN_SEG
#define KB_BIN_SEG 4  // chunks per segment of a binning pass (below); changes the bits of W for dictionaries beyond a segment
#endif
// the pool offsets of a dictionary's first 64 shells, one per lane, in ONE coalesced load: the address of chunk b's page then
// comes out of a register (readlane) and the rows of the next chunks are requested without a dependent pointer load per chunk
// (round 5: a wave of select_bin_kernel lived 28 us for 3.3 chunks -- a chain of dependent loads, 76 % of its cycles waiting)
__device__ __forceinline__ uint64_t shell_vector(const KbDev& D, const uint64_t* sh) {
    const int lane = threadIdx.x & 63;
    return lane < D.max_shells ? sh[lane] : 0ull;
}
__device__ __forceinline__ double* page_of(const KbState& K, const uint64_t* sh, uint64_t shv, int b) {
    if (b >= 64) return vec_page(K, sh, b);  // (dictionaries beyond 4,096 landmarks: the pointer load)
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)shv, b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(shv >> 32), b);
    return K.pool + (((uint64_t)hi << 32) | lo);
}

// chunks [b0, b1) of the dictionary, by ONE wave, into Wacc (LDS, 256 doubles, zeroed by the caller); the landmarks that take
// the direct evaluation are listed from position `pos0` of dlist on.  Returns flags | (how many of them) << 8.
// one chunk's landmarks: D0 / E (MODE 0: computed and left in the dictionary's rows), the direct-evaluation list, W[a] += coeff_j E_j
template <int MODE>
__device__ __forceinline__ void bin_one_chunk(const KbDev& D, const ChunkRows<MODE>& R, double* P, int lane, int cnt, int d, const double* x,
                                              double* Wacc, double* dlist, int pos0, int& flags, int& ndir) {
    double E, d0 = 0.0;
    if (MODE == 1) {
        E = R.v[0];
    } else {
        if (d - 1 == 10) {  // eMBB learners (scenario_creator.py:80-82)
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const double t = R.v[q] - x[q];
                d0 += t * t;
            }
        } else {
            d0 = dist0(P, lane, d, x);
        }
        E = rs_exp_nonpos(-D.gamma * d0);
        P[KB_ROW_D0 * KB_CH + lane] = d0;
        P[KB_ROW_E * KB_CH + lane] = E;
    }
    const bool offg = lane < cnt && R.a < 0;
    const bool direct = offg || (lane < cnt && !(E >= KB_E_TINY) && E > 0.0);
    const unsigned long long dmask = __ballot(direct);
    if (dmask) {
        flags |= 1 | (__ballot(offg) != 0ull ? 2 : 0);
        const int pos = pos0 + ndir + __builtin_popcountll(dmask & ((1ull << lane) - 1ull));
        if (direct && pos < KB_DLIST) {
            dlist[3 * pos] = R.co;
            dlist[3 * pos + 1] = P[(d - 1) * KB_CH + lane];
            dlist[3 * pos + 2] = MODE == 1 ? P[KB_ROW_D0 * KB_CH + lane] : d0;
        }
        ndir += __builtin_popcountll(dmask);
    }
    const double w = R.co * E;
    if (lane < cnt && !direct && R.a >= 0 && w != 0.0) unsafeAtomicAdd(Wacc + R.a, w);  // ds_add_f64
}

// DEEP > 0: the rows of DEEP chunks are requested together, every chunk in registers of its own (fully unrolled).  DEEP = 0: one
// chunk requested ahead in rotating registers -- the rotation at the end of an iteration waits for the loads issued at its top, so
// a chunk costs a full loaded memory latency (7.6 us at step 3000 of config 3: a wave of select_bin_kernel lived 32 us for 3.5
// chunks; tools/stamps_probe.sh, the kernel's ISA).  Round 5 measured both on the select_bin kernels: DEEP = 2 (128 registers) and
// 4 (168, three waves per SIMD) are no faster than 0 (profiles/r05_kbrl_variants.txt), so 0 stays everywhere.
template <int MODE, int DEEP>
__device__ __forceinline__ int bin_chunks(const KbDev& D, const KbState& K, const uint64_t* sh, uint64_t shv, int m, int d, const double* x,
                                           int b0, int b1, double* Wacc, double* dlist, int pos0, bool f32 = false) {
    const int lane = threadIdx.x & 63;
    int flags = 0, ndir = 0;
    if (DEEP > 0) {  // DEEP chunks' rows requested together, in registers of their own; the segment in groups of DEEP
        static_assert(DEEP == 0 || KB_BIN_SEG % (DEEP ? DEEP : 1) == 0, "groups tile the segment");
#pragma unroll
        for (int g0 = 0; g0 < KB_BIN_SEG; g0 += (DEEP ? DEEP : 1)) {
            if (b0 + g0 < b1) {
                ChunkRows<MODE> Rs[DEEP ? DEEP : 1];
#pragma unroll
                for (int i = 0; i < DEEP; ++i)
                    if (b0 + g0 + i < b1) load_chunk<MODE>(page_of(K, sh, shv, b0 + g0 + i), lane, d, Rs[i], f32);
#pragma unroll
                for (int i = 0; i < DEEP; ++i) {
                    if (b0 + g0 + i < b1) {  // (wave-uniform)
                        const int b = b0 + g0 + i;
                        bin_one_chunk<MODE>(D, Rs[i], page_of(K, sh, shv, b), lane, m - 64 * b < 64 ? m - 64 * b : 64, d, x, Wacc, dlist, pos0, flags,
                                            ndir);
                    }
                }
            }
        }
    } else {
        ChunkRows<MODE> R, Rn;
        load_chunk<MODE>(page_of(K, sh, shv, b0), lane, d, Rn, f32);
        for (int b = b0; b < b1; ++b) {
            double* P = page_of(K, sh, shv, b);
            R = Rn;
            if (b + 1 < b1) load_chunk<MODE>(page_of(K, sh, shv, b + 1), lane, d, Rn, f32);
            bin_one_chunk<MODE>(D, R, P, lane, m - 64 * b < 64 ? m - 64 * b : 64, d, x, Wacc, dlist, pos0, flags, ndir);
        }
    }
    return flags | (ndir << 8);
}

// The whole dictionary by ONE wave.  The order of W[a]'s sum is fixed for every kernel: the landmarks of a SEGMENT of
// KB_BIN_SEG chunks add up in increasing j (as above), and the segments' sums are added in increasing order, starting from
// zero -- W[a] = ((0 + S_0[a]) + S_1[a]) + ... .  A dictionary of at most one segment (256 landmarks) is summed exactly as
// until round 5; a larger one can be walked by several waves at once, a segment each (select_bin_big_kernel), and comes out
// with the same bits as from this one wave.  Wseg: 256 doubles of LDS scratch (used beyond one segment).
template <int MODE, int DEEP = 0>
__device__ __forceinline__ int bin_pass(const KbDev& D, const KbState& K, const uint64_t* sh, int m, int d, const double* x,
                                         double* W, double* Wseg, double* dlist, uint64_t shv, bool f32 = false) {
    const int lane = threadIdx.x & 63;
    const int nch = (m + 63) >> 6;
    if (nch <= KB_BIN_SEG) return bin_chunks<MODE, DEEP>(D, K, sh, shv, m, d, x, 0, nch, W, dlist, 0, f32);
    int flags = 0, ndir = 0;
    for (int b0 = 0; b0 < nch; b0 += KB_BIN_SEG) {
#pragma unroll
        for (int k = 0; k < 4; ++k) Wseg[lane + 64 * k] = 0.0;
        const int b1 = b0 + KB_BIN_SEG < nch ? b0 + KB_BIN_SEG : nch;
        const int r = bin_chunks<MODE, DEEP>(D, K, sh, shv, m, d, x, b0, b1, Wseg, dlist, ndir, f32);
        flags |= r & 3;
        ndir += r >> 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) W[lane + 64 * k] += Wseg[lane + 64 * k];  // (the LDS executes a wave's instructions in order)
    }
    return flags | (ndir << 8);
}

// f[g] = sum_a G[|a - c|] W[a], c = c_base + 64 g + lane: one chain of fused multiply-adds over a = 0 .. KA - 1 per candidate
template <int NG>
__device__ __forceinline__ void chain_scores(const KbDev& D, const double* G2, const volatile double* W, int c_base, int ng,
                                             double (&f)[NG]) {
    const int lane = threadIdx.x & 63;
    const int KA = (D.n_prbs + 4) & ~3;
    int cs[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = c_base + 64 * g + lane;
        cs[g] = 256 - (c < D.n_prbs ? c : D.n_prbs);
        f[g] = 0.0;
    }
    for (int a0 = 0; a0 < KA; a0 += 4) {
        double wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wv[u] = W[a0 + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if (g < ng) f[g] = __builtin_fma(G2[a0 + u + cs[g]], wv[u], f[g]);
        }
    }
}

// The landmarks bin_pass left out (off the grid, or E_j below KB_E_TINY) add coeff_j exp(-gamma (D0_j + (l_j - c/n)^2)), in
// increasing j (D0 / E are in the dictionary's rows).  With off_grid == false the left-out terms are all below
// |coeff_j| 1e-280 in magnitude, and a candidate whose binned sum already stands at 1e-240 or more keeps it: the terms could
// not move its sign, nor its value by more than one part in 1e30 (65,536 landmarks with coefficients of a thousand sum to
// 1e-272) -- only a candidate whose binned sum is (next to) nothing, the reference's f == 0 ties of an outlier state
// (kernel.py:26-27), needs them, and then it needs them exactly.  The rule looks at the candidate's own binned sum only, so
// it is the same in every kernel.
#define KB_F_SETTLED 1e-240
template <int NG>
__device__ __forceinline__ void add_direct_terms(const KbDev& D, const KbState& K, const uint64_t* sh, int m, int d, int c_base, int ng,
                                                 int flags, const double* dlist, double (&f)[NG]) {
    const int lane = threadIdx.x & 63;
    const int nch = (m + 63) >> 6;
    const bool off_grid = (flags & 2) != 0;
    const int ndir = flags >> 8;
    double tc[NG];
    bool open_[NG], any = false;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = c_base + 64 * g + lane;
        tc[g] = (double)(c < D.n_prbs ? c : D.n_prbs) / (double)D.n_prbs;
        open_[g] = g < ng && (off_grid || !(__builtin_fabs(f[g]) >= KB_F_SETTLED));
        any = any || open_[g];
    }
    if (!__ballot(any)) return;
    auto term = [&](double cs, double ls, double d0s) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (__ballot(open_[g])) {
                const double dl = ls - tc[g];
                const double v = __builtin_fma(cs, rs_exp_nonpos(-D.gamma * (d0s + dl * dl)), f[g]);
                f[g] = open_[g] ? v : f[g];
            }
        }
    };
    if (ndir <= KB_DLIST) {  // bin_pass listed them all: the list comes in with one round of loads, its entries by readlane
        static_assert(KB_DLIST * 3 <= 192, "three registers hold the list");
        const int ne = 3 * ndir;
        const double e0 = lane < ne ? dlist[lane] : 0.0;
        const double e1 = 64 + lane < ne ? dlist[64 + lane] : 0.0;
        const double e2 = 128 + lane < ne ? dlist[128 + lane] : 0.0;
        auto entry = [&](int i) -> double {  // (i is wave-uniform)
            const double r = i < 64 ? e0 : (i < 128 ? e1 : e2);
            return readlane_f64(r, i & 63);
        };
        for (int q = 0; q < ndir; ++q) term(entry(3 * q), entry(3 * q + 1), entry(3 * q + 2));
    } else {                 // more than the list holds: the rows again
        for (int b = 0; b < nch; ++b) {
            const double* P = vec_page(K, sh, b);
            const int cnt = m - 64 * b < 64 ? m - 64 * b : 64;
            const double E = P[KB_ROW_E * KB_CH + lane];
            const int a = ((const int32_t*)(P + KB_ROW_IDX * KB_CH))[lane];
            unsigned long long dm = __ballot(lane < cnt && (a < 0 || (!(E >= KB_E_TINY) && E > 0.0)));
            if (!dm) continue;
            const double d0 = P[KB_ROW_D0 * KB_CH + lane], lam = P[(d - 1) * KB_CH + lane], co = P[KB_ROW_CO * KB_CH + lane];
            while (dm) {
                const int jj = __builtin_ctzll(dm);
                dm &= dm - 1ull;
                term(readlane_f64(co, jj), readlane_f64(lam, jj), readlane_f64(d0, jj));
            }
        }
    }
#ifdef KB_COUNT_DIRECT  // (how often, and how much: kb_get_repair_work[4], [5].  A developer build only: thousands of passes a step
                        // adding to ONE address cost select_gemm_kernel 0.1 ms per step, profiles/r04_m_*)
    if (lane == 0) {
        atomicAdd(&K.hv_work[4], 1ull);
        atomicAdd(&K.hv_work[5], (unsigned long long)ndir);
    }
#endif
}

// the binned scores of the first ng (<= NG) groups of 64 candidates from c_base on, by ONE wave using the block's sm.W / sm.tag
template <int NG, int MODE>
__device__ __forceinline__ void score_binned(const KbDev& D, const KbState& K, const uint64_t* sh, int m, int d, Lds& sm, int c_base,
                                             int ng, double (&f)[NG]) {
    if (m == 0) {
#pragma unroll
        for (int g = 0; g < NG; ++g) f[g] = 0.0;
    } else if (m == 1) {
        score_single<NG, MODE>(D, K, sh, d, sm.x, c_base, f);
    } else {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < 4; ++k) ((volatile double*)sm.W)[lane + 64 * k] = 0.0;
        const int direct = bin_pass<MODE>(D, K, sh, m, d, sm.x, sm.W, sm.fbuf, sm.dl, shell_vector(D, sh));
        chain_scores<NG>(D, sm.G2, sm.W, c_base, ng, f);
        if (direct) add_direct_terms<NG>(D, K, sh, m, d, c_base, ng, direct, (const double*)sm.dl, f);
    }
}

// the window of candidates a learner's scores cover: c_base + 64 g + lane, g < ng -- the augmentation range of the step
// (kbrl_control.py:102-112: [a_i, n] after a fulfilled SLA, [0, a_i] after a violation), not all 256
struct Win {
    int base, ng;
};
__device__ __forceinline__ Win window_of(int c_from, int c_to) {
    Win w = {c_from, (c_to - c_from) / 64 + 1};
    return w;
}

// f of candidate c out of the four group registers (c's lane holds it): broadcast to the wave
__device__ __forceinline__ double f_of(const double (&f)[4], Win w, int c) {
    const int g = (c - w.base) >> 6, l = (c - w.base) & 63;
    double v = g == 0 ? f[0] : (g == 1 ? f[1] : (g == 2 ? f[2] : f[3]));
    return readlane_f64(v, l);
}

// first candidate c in [c_from, c_to] (in order) with f(c) * y <= 0, or -1; *zeros = candidates with f == 0 among
// [c_from, min(c_to, found)]
__device__ __forceinline__ int first_mistake(const double (&f)[4], Win w, int y, int c_from, int c_to, int* zeros) {
    const int lane = threadIdx.x & 63;
    int found = -1, nz = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = w.base + 64 * g + lane;
        const bool in = c >= c_from && c <= c_to;
        const unsigned long long bad = __ballot(in && f[g] * (double)y <= 0.0);
        const unsigned long long zer = __ballot(in && f[g] == 0.0);
        if (found < 0) {
            if (bad) {
                const int l = __builtin_ctzll(bad);
                found = w.base + 64 * g + l;
                nz += __builtin_popcountll(zer & ((l == 63) ? ~0ull : ((2ull << l) - 1ull)));
            } else {
                nz += __builtin_popcountll(zer);
            }
        }
    }
    *zeros = nz;
    return found;
}

// ------------------------------------------------------------------------------------------ Projectron.update pieces
// kernel column of x = (state the D0 row was computed for, t) against the dictionary -> K_f row, in the exact form
// exp(-gamma (D0 + (lam - t)^2)) (kernel.py:18-19)
__device__ __forceinline__ void kernel_column_from_d0(const KbDev& D, const KbState& K, const uint64_t* sh, int m, int d, double t) {
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        double* P = vec_page(K, sh, j >> 6);
        const int l = j & 63;
        const double dl = P[(d - 1) * KB_CH + l] - t;
        const double k = rs_exp(-D.gamma * (P[KB_ROW_D0 * KB_CH + l] + dl * dl));
        P[KB_ROW_KF * KB_CH + l] = m == 1 ? (double)(float)k : k;
    }
    __syncthreads();
}
// the same for an arbitrary x (all d coordinates given): D0 is formed in the same coordinate order
__device__ __forceinline__ void kernel_column_full(const KbDev& D, const KbState& K, const uint64_t* sh, int m, int d,
                                                   const double* x, double t) {
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        double* P = vec_page(K, sh, j >> 6);
        const int l = j & 63;
        double d0 = 0.0;
        for (int q = 0; q < d - 1; ++q) {
            const double u = P[q * KB_CH + l] - x[q];
            d0 += u * u;
        }
        const double dl = P[(d - 1) * KB_CH + l] - t;
        const double k = rs_exp(-D.gamma * (d0 + dl * dl));
        P[KB_ROW_KF * KB_CH + l] = m == 1 ? (double)(float)k : k;
    }
    __syncthreads();
}

// d* = Kinv K_f -> DS row.  Kinv is symmetric bit for bit (it only ever receives (d_i d_j) / delta), so output i is
// summed down COLUMN i: no cross-lane reduction, loads coalesced along the tile rows.  The sum of a column is formed in a
// fixed shape that does not depend on the block size or on how many columns a lane owns: eight partial sums over the row
// classes j mod 8 (each in increasing j, fused multiply-adds), then ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)).
// A work unit is (64 W columns, one row class); the waves of the block share the units, eight rows are in flight per
// lane (W doubles each), and the partial sums meet in eight rows of the vector pages.
template <int W>
__device__ __forceinline__ void matvec_partials(const KbState& K, const uint64_t* sh, int m, int u0, int ustride, int u_end = 0x7fffffff) {
    static_assert(W == 1 || W == 2 || W == 4, "columns per lane");
    const int nb = (m + 63) >> 6, lane = threadIdx.x & 63;
    const int ncg = (nb + W - 1) / W;  // groups of W column blocks
    const int bsub = lane / (64 / W), coff = (lane % (64 / W)) * W;  // this lane's column block within the group, column in it
    for (int u = u0; u < ncg * 8 && u < u_end; u += ustride) {
        const int cg = u >> 3, sg = u & 7;
        const int bi = cg * W + bsub;
        const bool on = bi < nb;
        double acc[W];
#pragma unroll
        for (int c = 0; c < W; ++c) acc[c] = 0.0;
        for (int bj = 0; bj < nb; ++bj) {
            const double kfv = vec_page(K, sh, bj)[KB_ROW_KF * KB_CH + lane];
            const double* tp = kinv_tile(K, sh, bj, on ? bi : 0) + coff;
            const int rows = m - 64 * bj < 64 ? m - 64 * bj : 64;
            double v[8][W];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = sg + 8 * k;
#pragma unroll
                for (int c = 0; c < W; ++c) v[k][c] = (on && r < rows) ? tp[r * 64 + c] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = sg + 8 * k;
                if (r < rows) {
                    const double kfr = readlane_f64(kfv, r);
#pragma unroll
                    for (int c = 0; c < W; ++c) acc[c] = __builtin_fma(v[k][c], kfr, acc[c]);
                }
            }
        }
        if (on) {
            double* o = vec_page(K, sh, bi) + (KB_ROW_PART + sg) * KB_CH + coff;
#pragma unroll
            for (int c = 0; c < W; ++c) o[c] = acc[c];
        }
    }
}

__device__ __forceinline__ void matvec_combine(const KbState& K, const uint64_t* sh, int m) {
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        double* P = vec_page(K, sh, i >> 6) + (i & 63);
        const double p0 = P[(KB_ROW_PART + 0) * KB_CH], p1 = P[(KB_ROW_PART + 1) * KB_CH], p2 = P[(KB_ROW_PART + 2) * KB_CH],
                     p3 = P[(KB_ROW_PART + 3) * KB_CH], p4 = P[(KB_ROW_PART + 4) * KB_CH], p5 = P[(KB_ROW_PART + 5) * KB_CH],
                     p6 = P[(KB_ROW_PART + 6) * KB_CH], p7 = P[(KB_ROW_PART + 7) * KB_CH];
        P[KB_ROW_DS * KB_CH] = ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7));
    }
    __syncthreads();
}

template <int W>
__device__ __forceinline__ void matvec_colsum(const KbState& K, const uint64_t* sh, int m) {
    matvec_partials<W>(K, sh, m, threadIdx.x >> 6, blockDim.x >> 6);
    __syncthreads();
    matvec_combine(K, sh, m);
}

// tile t of the lower block triangle in the order (0,0) (1,0) (1,1) (2,0) ...
__device__ __forceinline__ void tri_tile_of(int t, int* bi, int* bj) {
    int b = (int)((__builtin_sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((b + 1) * (b + 2) / 2 <= t) ++b;
    while (b * (b + 1) / 2 > t) --b;
    *bi = b;
    *bj = t - b * (b + 1) / 2;
}

// ---- triangle storage (KbDev.tri): d* = Kinv K_f when only the tiles (b_i, b_j), b_j <= b_i, exist -- every tile read ONCE.
// Tile (bi, bj) holds Kinv[64 bi + r][64 bj + c].  It contributes to the outputs of block bj down its columns,
//     dpart[c] = sum_r T[r][c] K_f[64 bi + r],
// and, Kinv being symmetric, to the outputs of block bi along its rows (off-diagonal tiles only),
//     tpart[r] = sum_c T[r][c] K_f[64 bj + c].
// One wave forms both from one pass over the tile, eight rows at a time: the rows come in coalesced for dpart (lane =
// column; the eight rows of a slab are one row of each class r mod 8, so eight accumulators step once per slab), and the
// same 4 KB -- still in the CU's L1 -- come in again dealt as lane = (row 8 x + (lane >> 3), column class lane & 7) for
// tpart: the eight lanes of a row read 64 consecutive bytes, chain their eight products and meet in a three-step butterfly.
// Both partial sums have one shape in every kernel: eight chains of fused multiply-adds over the classes (index mod 8, in
// increasing index) and ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)).  The tile's 128 partial sums go to its slot of
// the shell's partial area; matvec_tri_combine adds, per output, the dpart of its column of tiles (increasing bi) and the
// tpart of its row of tiles (increasing bj), plain additions in that order.  Tiles are independent work units (the
// chip-wide rounds lay them end to end), and d* costs one read of the stored triangle -- half of what reading every tile
// once per orientation cost (the first round-4 version), a quarter of a square Kinv's row walk AND column walk.
__device__ __forceinline__ double* tri_part(const KbState& K, const uint64_t* sh, int bi, int bj) {
    return K.pool + sh[bi] + KB_VEC + (size_t)(bi + 1) * KB_TILE + (size_t)bj * 128;
}

__device__ __forceinline__ void matvec_tri_tiles(const KbState& K, const uint64_t* sh, int m, int t0, int tstride,
                                                 int t_end = 0x7fffffff) {
    const int nb = (m + 63) >> 6, lane = threadIdx.x & 63;
    const int cl = lane >> 3, sg = lane & 7;
    const int nt = nb * (nb + 1) / 2;
    for (int t = t0; t < nt && t < t_end; t += tstride) {
        int bi, bj;
        tri_tile_of(t, &bi, &bj);
        const double* Tp = kinv_tile_lo(K, sh, bi, bj);
        const int rows = m - 64 * bi < 64 ? m - 64 * bi : 64;
        const bool off = bi != bj;
        const double kfr = vec_page(K, sh, bi)[KB_ROW_KF * KB_CH + lane];
        double kc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) kc[k] = off ? vec_page(K, sh, bj)[KB_ROW_KF * KB_CH + sg + 8 * k] : 0.0;
        double* part = tri_part(K, sh, bi, bj);
        double acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.0;
        for (int x = 0; x < 8; ++x) {
            if (8 * x >= rows) break;  // (wave-uniform; the last row block of the dictionary)
            double v[8], tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = 8 * x + u < rows ? Tp[(8 * x + u) * 64 + lane] : 0.0;
            if (off) {
#pragma unroll
                for (int k = 0; k < 8; ++k) tv[k] = Tp[(8 * x + cl) * 64 + sg + 8 * k];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (8 * x + u < rows) acc[u] = __builtin_fma(v[u], readlane_f64(kfr, 8 * x + u), acc[u]);
            if (off) {
                double ta = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) ta = __builtin_fma(tv[k], kc[k], ta);
                ta += __shfl_xor(ta, 1);
                ta += __shfl_xor(ta, 2);
                ta += __shfl_xor(ta, 4);
                if (sg == 0) part[64 + 8 * x + cl] = ta;
            }
        }
        part[lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
}

// The same tiles for heavy_matvec_kernel, whose waves do nothing else: TWO slabs of eight rows are requested together and the
// row-wise operand is read back out of LDS (`slab`: 8 x KB_SLAB_LD doubles per WAVE, the wave's own copy of the slab) instead of a
// second time from memory -- twice the bytes in flight per wave at four waves per SIMD (round 5: the kernel moves one loaded
// memory latency per slab and a wave held 4 KB in flight; 0.24 -> 0.21 ms per step at step 3000 of config 3).  Same products,
// same sums, same order as above: same bits (rows past the dictionary's end contribute zeros here, whatever the tile holds
// there above: those partial sums are never read).  A function of its own: as one function with a flag the per-learner
// kernels that use the version above spilled 608 B per lane.
#ifndef KB_MV_SLABS
#define KB_MV_SLABS 2  // slabs of eight rows a wave of heavy_matvec_kernel requests together
#endif
#define KB_SLAB_LD 72  // doubles between the rows of the LDS copy (64 + 8: the row-wise reads of eight rows spread over the banks)
__device__ __forceinline__ void matvec_tri_tiles_lds(const KbState& K, const uint64_t* sh, int m, int t0, int t_end, double* slab) {
    const int nb = (m + 63) >> 6, lane = threadIdx.x & 63;
    const int cl = lane >> 3, sg = lane & 7;
    const int nt = nb * (nb + 1) / 2;
    for (int t = t0; t < nt && t < t_end; ++t) {
        int bi, bj;
        tri_tile_of(t, &bi, &bj);
        const double* Tp = kinv_tile_lo(K, sh, bi, bj);
        const int rows = m - 64 * bi < 64 ? m - 64 * bi : 64;
        const bool off = bi != bj;
        const double kfr = vec_page(K, sh, bi)[KB_ROW_KF * KB_CH + lane];
        double kc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) kc[k] = off ? vec_page(K, sh, bj)[KB_ROW_KF * KB_CH + sg + 8 * k] : 0.0;
        double* part = tri_part(K, sh, bi, bj);
        double acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.0;
        for (int x = 0; x < 8; x += KB_MV_SLABS) {
            if (8 * x >= rows) break;  // (wave-uniform; the last row block of the dictionary)
            double v[KB_MV_SLABS][8];
#pragma unroll
            for (int h = 0; h < KB_MV_SLABS; ++h) {
#pragma unroll
                for (int u = 0; u < 8; ++u) v[h][u] = 8 * (x + h) + u < rows ? Tp[(8 * (x + h) + u) * 64 + lane] : 0.0;
            }
#pragma unroll
            for (int h = 0; h < KB_MV_SLABS; ++h) {
                if (8 * (x + h) < rows) {
                    const int xs = x + h;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (8 * xs + u < rows) acc[u] = __builtin_fma(v[h][u], readlane_f64(kfr, 8 * xs + u), acc[u]);
                    if (off) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) slab[u * KB_SLAB_LD + lane] = v[h][u];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        double ta = 0.0;
#pragma unroll
                        for (int k = 0; k < 8; ++k) ta = __builtin_fma(slab[cl * KB_SLAB_LD + sg + 8 * k], kc[k], ta);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();  // (read before the next slab overwrites it)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        ta += __shfl_xor(ta, 1);
                        ta += __shfl_xor(ta, 2);
                        ta += __shfl_xor(ta, 4);
                        if (sg == 0) part[64 + 8 * xs + cl] = ta;
                    }
                }
            }
        }
        part[lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
}

__device__ __forceinline__ void matvec_tri_combine_wide(const KbState& K, const uint64_t* sh, int m) {
    // (heavy_finish_kernel's form of matvec_tri_combine below: same sums in the same order; the per-learner kernels keep the plain
    // loop -- with this one inlined they spilled 608 B per lane.)
    // Per output: its column of tiles (increasing bi), then its row of tiles, added one after the other in that order.  The partial
    // sums come from up to 2 n_b scattered places: their loads are issued eight at a time before the (ordered) adds, and the shell
    // offsets they need come out of a register (one coalesced load per wave) -- one dependent pair of loads per term, as until round
    // 5, made this loop the duration of heavy_finish_kernel for a dictionary of 26 blocks (~50 us).
    const int nb = (m + 63) >> 6, lane = threadIdx.x & 63;
    const uint64_t shv = lane < nb ? sh[lane] : 0ull;  // (the first 64 shells; a wave's outputs share their block b)
    auto part_of = [&](int bi, int bj) -> const double* {
        uint64_t off;
        if (bi < 64) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)shv, bi);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(shv >> 32), bi);
            off = ((uint64_t)hi << 32) | lo;
        } else {
            off = sh[bi];
        }
        return K.pool + off + KB_VEC + (size_t)(bi + 1) * KB_TILE + (size_t)bj * 128;
    };
    for (int i0 = (int)(threadIdx.x & ~63u); i0 < m; i0 += blockDim.x) {  // (whole waves: b is wave-uniform)
        const int i = i0 + lane;
        const int b = i0 >> 6, c = lane;
        double dsum = part_of(b, b)[c];
        for (int br0 = b + 1; br0 < nb; br0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = br0 + u < nb ? part_of(br0 + u, b)[c] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (br0 + u < nb) dsum += v[u];
        }
        double tsum = 0.0;
        for (int br0 = 0; br0 < b; br0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = br0 + u < b ? part_of(b, br0 + u)[64 + c] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (br0 + u < b) tsum += v[u];
        }
        if (i < m) vec_page(K, sh, b)[KB_ROW_DS * KB_CH + c] = dsum + tsum;
    }
    __syncthreads();
}

__device__ __forceinline__ void matvec_tri_combine(const KbState& K, const uint64_t* sh, int m) {
    const int nb = (m + 63) >> 6;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const int b = i >> 6, c = i & 63;
        double dsum = tri_part(K, sh, b, b)[c];
        for (int br = b + 1; br < nb; ++br) dsum += tri_part(K, sh, br, b)[c];
        double tsum = 0.0;
        for (int br = 0; br < b; ++br) tsum += tri_part(K, sh, b, br)[64 + c];
        vec_page(K, sh, b)[KB_ROW_DS * KB_CH + c] = dsum + tsum;
    }
    __syncthreads();
}

__device__ __forceinline__ void matvec_tri_colsum(const KbState& K, const uint64_t* sh, int m) {
    matvec_tri_tiles(K, sh, m, threadIdx.x >> 6, blockDim.x >> 6);
    __syncthreads();
    matvec_tri_combine(K, sh, m);
}

// Kinv <- [[Kinv,0],[0,0]] + outer([d*,-1],[d*,-1]) / delta (projectron.py:54-58) over the m1 = m + 1 landmarks, d* (with
// its -1) in the DS row.  A work unit is 16 rows of a tile: all 16 loads of a lane are in flight before the first store
// (d_i broadcast from the row block's lanes, d_j in this lane).  Every entry is formed as old + (d_i d_j) / delta, the
// reference's own expression (a true division: the kernel is bound by the tiles it streams, not by the divide), and
// d_i d_j = d_j d_i keeps Kinv symmetric bit for bit.  tri: only the tiles (b_i, b_j), b_j <= b_i, exist (units count
// along the lower block triangle); diagonal tiles hold both of their halves.
__device__ __forceinline__ void rank1_units(const KbState& K, const uint64_t* sh, int m, double delta, int tri, int u0, int ustride,
                                            int u_end = 0x7fffffff) {
    const int m1 = m + 1, nb = (m1 + 63) >> 6, lane = threadIdx.x & 63;
    const int ntile = tri ? nb * (nb + 1) / 2 : nb * nb;
    for (int u = u0; u < ntile * 4 && u < u_end; u += ustride) {
        const int tb = u >> 2, r0 = (u & 3) * 16;
        int bi, bj;
        if (tri) {
            tri_tile_of(tb, &bi, &bj);
        } else {
            bi = tb / nb;
            bj = tb - bi * nb;
        }
        const int rows = m1 - 64 * bi < 64 ? m1 - 64 * bi : 64;
        if (r0 >= rows) continue;  // (wave-uniform)
        // 16-byte accesses: a lane owns two neighbouring columns of one row, a half-wave a whole 512-byte row, one load
        // instruction two rows (the tile's rows are 16-byte aligned: every pool offset is even)
        const int h = lane >> 5, cp = (lane & 31) * 2;
        const int j0 = 64 * bj + cp;
        kb_f64x2* T = (kb_f64x2*)((tri ? kinv_tile_lo(K, sh, bi, bj) : kinv_tile(K, sh, bi, bj)) + cp);
        const double dsi_v = vec_page(K, sh, bi)[KB_ROW_DS * KB_CH + lane];
        const kb_f64x2 dsj = *(const kb_f64x2*)(vec_page(K, sh, bj) + KB_ROW_DS * KB_CH + cp);
        kb_f64x2 old[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = r0 + 2 * k + h, i = 64 * bi + r;
            const bool in = r < rows && i < m;
            kb_f64x2 v = {0.0, 0.0};
            if (in && j0 < m) v = T[r * 32];  // (a pair straddling m: its second entry is unwritten memory, masked below)
            old[k][0] = v[0];
            old[k][1] = (in && j0 + 1 < m) ? v[1] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = r0 + 2 * k + h;
            const double da = readlane_f64(dsi_v, r0 + 2 * k), db = readlane_f64(dsi_v, r0 + 2 * k + 1);
            const double dsi = h ? db : da;
            kb_f64x2 nv;
            nv[0] = old[k][0] + (dsi * dsj[0]) / delta;
            nv[1] = old[k][1] + (dsi * dsj[1]) / delta;
            if (r < rows) {
                if (j0 + 1 < m1)
                    T[r * 32] = nv;
                else if (j0 < m1)
                    ((double*)&T[r * 32])[0] = nv[0];
            }
        }
    }
}

// take shell b for the dictionary (thread 0; everybody learns the outcome).  false: capacity or pool exhausted.  The top
// only moves when the request fits (compare-and-swap): a large shell that does not fit leaves the space to the smaller
// ones of other dictionaries instead of pushing the top past the end for everybody.
__device__ __forceinline__ bool take_shell(const KbDev& D, const KbState& K, int dict, int b, Lds& sm) {
    if (threadIdx.x == 0) {
        int ok = 0;
        if (b < D.max_shells) {
            const unsigned long long need = kb_shell_doubles(b, D.tri);
            unsigned long long at = *(volatile unsigned long long*)K.pool_top;
            while (at + need <= D.pool_doubles) {
                const unsigned long long seen = atomicCAS(K.pool_top, at, at + need);
                if (seen == at) {
                    K.shell[(size_t)dict * D.max_shells + b] = at;
                    ok = 1;
                    break;
                }
                at = seen;
            }
        }
        sm.ired[6] = ok;
    }
    __syncthreads();
    const bool ok = sm.ired[6] != 0;
    __syncthreads();
    return ok;
}

// The part of Projectron.update (projectron.py:41-60) that follows d* = Kinv K_f (DS row): delta, the decision, and the
// projection or the insertion.  With defer_rank1 the O(m^2) update of Kinv is left to the caller (rank1_units with the
// returned delta).  Returns the new m.  branch: 1 = projection onto the dictionary, 2 = dictionary grew.
__device__ int finish_update(const KbDev& D, const KbState& K, int dict, int err_env, int m, int d, const double* x, double t_last,
                             int a_last, int y, Lds& sm, int* branch, double* delta_out, bool* saturated, bool defer_rank1) {
    const uint64_t* sh = shells_of(D, K, dict);
    double dot;
    if (m <= 1) {
        const float kf0 = m == 0 ? 0.0f : (float)*vec_at(K, sh, KB_ROW_KF, 0);
        const float ds = m == 0 ? 0.0f : (float)*vec_at(K, sh, KB_ROW_DS, 0);
        dot = (double)(float)(ds * kf0);
    } else {
        if (threadIdx.x < 64) {
            const double v = wave_dot256_rows(K, sh, KB_ROW_DS, KB_ROW_KF, m);
            if (threadIdx.x == 0) sm.red[15] = v;
        }
        __syncthreads();
        dot = sm.red[15];
        __syncthreads();
    }
    double delta = 1.0 - dot;  // Kii = k(x, x) = 1
    delta = delta > 0.0 ? delta : 0.0;
    *delta_out = delta;
    if (threadIdx.x == 0) atomicAdd(&K.ver[dict], 1);  // the dictionary changes (projection or insertion): stored scores are stale
                                                       // (no return value: nothing waits for it)
    // A dictionary that cannot grow (capacity reached, or the pool has no shell left) projects every further sample
    // onto its span -- the fixed-budget reading of Projectron -- instead of growing as the reference's unbounded
    // SVvariable would: learning goes on, nothing is dropped, and the replica is flagged (err bit 8 "saturated", bit 16
    // "pool exhausted": reported by kb_get_sizes / kb_get_pool and by the host classes as a warning).
    bool full = m >= D.cap;
    if (delta > D.eta && !full && (m & 63) == 0) {
        if (!take_shell(D, K, dict, m >> 6, sm)) {
            full = true;
            if (threadIdx.x == 0) atomicOr(&K.err[err_env], 16);
        }
    }
    if (delta > D.eta && full && threadIdx.x == 0) atomicOr(&K.err[err_env], 8);
    if (saturated) *saturated = delta > D.eta && full;
    if (delta <= D.eta || full) {
        *branch = 1;
        for (int j = threadIdx.x; j < m; j += blockDim.x) {
            double* P = vec_page(K, sh, j >> 6);
            const int l = j & 63;
            double nc = P[KB_ROW_CO * KB_CH + l] + (double)y * P[KB_ROW_DS * KB_CH + l];
            if (m == 1) nc = (double)(float)nc;
            P[KB_ROW_CO * KB_CH + l] = nc;
        }
        __syncthreads();
        return m;
    }
    *branch = 2;
    // SVvariable.extend / insert
    {
        double* P = vec_page(K, sh, m >> 6);
        const int l = m & 63;
        if ((int)threadIdx.x < d - 1) P[threadIdx.x * KB_CH + l] = x[threadIdx.x];
        if (d - 1 == 10 && (int)threadIdx.x < 10) {  // the float32 copy (KB_ROW_F32); exact for observations, flagged otherwise
            const float xf = (float)x[threadIdx.x];
            ((float*)(P + KB_ROW_F32 * KB_CH))[threadIdx.x * KB_CH + l] = xf;
            if ((double)xf != x[threadIdx.x]) K.f32bad[dict] = 1;
        }
        if (threadIdx.x == 0) {
            P[(d - 1) * KB_CH + l] = t_last;
            P[KB_ROW_CO * KB_CH + l] = (double)y;
            P[KB_ROW_D0 * KB_CH + l] = 0.0;  // the new landmark shares the state being processed
            P[KB_ROW_E * KB_CH + l] = 1.0;
            P[KB_ROW_DS * KB_CH + l] = -1.0;
            int32_t* ix = (int32_t*)(P + KB_ROW_IDX * KB_CH);
            ix[l] = a_last;
            if (a_last < 0) K.offgrid[dict] += 1;
            if (a_last >= 0) {  // chain of the landmarks with this grid index, newest first
                int32_t* hd = K.head + (size_t)dict * KB_HEAD;
                ix[64 + l] = hd[a_last];
                hd[a_last] = m;
            } else {
                ix[64 + l] = -1;
            }
        }
    }
    __syncthreads();
    if (m == 0) {
        if (threadIdx.x == 0) kinv_tile_lo(K, sh, 0, 0)[0] = 1.0;
    } else if (!defer_rank1) {
        rank1_units(K, sh, m, delta, D.tri, threadIdx.x >> 6, blockDim.x >> 6);
    }
    __syncthreads();
    return m + 1;
}

// Projectron.update (projectron.py:39-60) for x = (x[0..d-2], t_last) whose kernel column is in the K_f row.
// Returns the new m.  Runs on the whole block (64, 256, 512 or 1024 threads); every sum is formed in an order that does
// not depend on the block size.
__device__ int apply_update(const KbDev& D, const KbState& K, int dict, int err_env, int m, int d, const double* x, double t_last,
                            int a_last, int y, Lds& sm, int* branch, double* delta_out, bool* saturated = nullptr) {
    const uint64_t* sh = shells_of(D, K, dict);
    if (m == 1) {
        // Kinv is the 1-element float32 array [1 / Kii]; K_f is float32 (projectron.py:29,59)
        __syncthreads();
        if (threadIdx.x == 0) *vec_at(K, sh, KB_ROW_DS, 0) = (double)(1.0f * (float)*vec_at(K, sh, KB_ROW_KF, 0));
        __syncthreads();
    } else if (m >= 2) {
        if (D.tri)
            matvec_tri_colsum(K, sh, m);
        else if (blockDim.x >= 256)
            matvec_colsum<4>(K, sh, m);
        else
            matvec_colsum<2>(K, sh, m);
    }
    return finish_update(D, K, dict, err_env, m, d, x, t_last, a_last, y, sm, branch, delta_out, saturated, false);
}

struct CtlArgs {
    KbDev D;
    KbState K;
    const float* state;     // [n_envs][nv] state the action was taken in
    const int32_t* action;  // [n_envs][S]
    const int32_t* labels;  // [n_envs][S]
    int32_t* hits;          // [n_envs][S]
    int32_t big_par;        // which of the two large-learner lists orders this launch (-1: none, block index = learner)
};

// The learner of launch slot `slot` (one-wave kernels: slot = workgroup; select_gemm_kernel: a wave).  With a list: the
// first KB_BIG_MAX slots take the listed (large) learners, the others their own index unless it is listed; -1: nothing to
// do.  The list is the one the select kernel wrote at the end of the previous step (dictionaries only grow, and not between
// that select and these launches).
__device__ __forceinline__ int learner_of_slot(const KbState& K, int T, int par, int slot) {
    if (par < 0) return slot < T ? slot : -1;
    if (slot < KB_BIG_MAX) {
        const int32_t* L = K.big + (size_t)par * (1 + KB_BIG_MAX);
        return slot < L[0] ? L[1 + slot] : -1;
    }
    const int t = slot - KB_BIG_MAX;
    return t < T && !K.isbig[(size_t)par * T + t] ? t : -1;
}
__device__ __forceinline__ int learner_of_block(const KbState& K, int T, int par) { return learner_of_slot(K, T, par, (int)blockIdx.x); }

// y_pred of update_control's first predict, the accuracy table and the security factor (kbrl_control.py:88-101).
// Called by one-wave kernels only (64 threads).
__device__ __forceinline__ int control_bookkeeping(const KbDev& D, const KbState& K, int task, int env, int s, int m, double f0, int y,
                                                   int32_t* hits, Lds& sm) {
    const int n = D.n_prbs;
    int y_pred = 0;
    if (m > 0) {
        y_pred = f0 > 0.0 ? 1 : (f0 < 0.0 ? -1 : 0);
        if (y_pred == 0) {
            if (threadIdx.x == 0) sm.ired[0] = tie_draw(K, task, env, s);
            __syncthreads();
            y_pred = sm.ired[0];
            __syncthreads();
        }
    }
    const int hit = y == y_pred;
    int margin = K.margins[env * D.S + s];
    margin = margin > 0 ? margin : 0;
    double* acc = K.acc + ((size_t)env * D.S + s) * n;
    // (one wave: a lane holds the table entries c = lane, lane + 64, ... -- one fetch for the update and for the search of
    // the first entry above the threshold, no barrier and no second trip through memory between them)
    const bool adj = K.adjusted[env] != 0;
    if (y_pred == 1 || !adj) {
        const int lane = threadIdx.x & 63;
        double a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = lane + 64 * k < n ? acc[lane + 64 * k] : 0.0;
        if (y_pred == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = lane + 64 * k;
                if (c < n) {
                    if (!hit) {
                        if (c < margin + 1) {
                            a[k] = (1 - D.alfa) * a[k];
                            acc[c] = a[k];
                        }
                    } else if (c >= margin) {
                        a[k] = (1 - D.alfa) * a[k] + D.alfa;
                        acc[c] = a[k];
                    }
                }
            }
        }
        if (!adj) {
            int first = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long above = __ballot(lane + 64 * k < n && a[k] > D.lo);
                if (first == 0x7fffffff && above) first = 64 * k + __builtin_ctzll(above);
            }
            if (threadIdx.x == 0) K.security[env * D.S + s] = first == 0x7fffffff ? 0 : first;
        }
    }
    if (threadIdx.x == 0) hits[env * D.S + s] = hit;
    return hit;
}

__device__ __forceinline__ void stage_state(const KbDev& D, const float* state, int env, int s, int d, Lds& sm) {
    if ((int)threadIdx.x < d - 1) sm.x[threadIdx.x] = (double)state[(size_t)env * D.nv + D.off[s] + threadIdx.x];
}

// The augmentation loop of update_control (kbrl_control.py:102-112) from candidate c_from on, given the scores f of all
// candidates: find the next mistake in order, apply that one Projectron update, rescore, repeat.  Runs on the whole
// block: the scoring passes are wave 0's (its lanes are the candidates) and are handed to the other waves through LDS,
// the Projectron update (mat-vec, rank-1 update) is shared by all threads.
struct LoopStats {
    uint64_t n_pred, n_mist, n_grow, n_eval;
};

__device__ __forceinline__ void share_scores(double (&f)[4], Lds& sm) {
    if (blockDim.x == 64) return;
    __syncthreads();
    if (threadIdx.x < 64) {
#pragma unroll
        for (int g = 0; g < 4; ++g) sm.fbuf[g * 64 + threadIdx.x] = f[g];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) f[g] = sm.fbuf[g * 64 + (threadIdx.x & 63)];
}

// the scores of the window again (same state: the E row is reused).  A candidate's score does not depend on which other
// candidates are computed with it, so a workgroup of four waves or more gives each of the window's (up to four) groups of
// 64 candidates to a wave of its own -- a quarter of the scoring chain per wave -- and the groups meet in LDS; a single
// wave does all of them.  Same sums either way.  Dictionaries of KB_BIN_M landmarks and more are scored in the binned
// form (wave 0 forms W, every wave chains its group), smaller ones landmark by landmark (score_pass): which of the two
// depends on m alone, never on the kernel that asks.
__device__ __forceinline__ void rescore(const KbDev& D, const KbState& K, const uint64_t* sh, int m, int d, Lds& sm, Win w,
                                        double (&f)[4]) {
    const bool binned = m >= KB_BIN_M;
    if (blockDim.x == 64) {
        if (binned)
            score_binned<4, 1>(D, K, sh, m, d, sm, w.base, w.ng, f);
        else
            score<4, 1>(D, K, sh, m, d, sm, w.base, w.ng, f);
        return;
    }
    const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double f1[1] = {0.0};
    if (binned) {
        __syncthreads();
        if (threadIdx.x < 64) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ((volatile double*)sm.W)[lane + 64 * k] = 0.0;
            const int direct = bin_pass<1>(D, K, sh, m, d, sm.x, sm.W, sm.fbuf, sm.dl, shell_vector(D, sh));
            if (lane == 0) sm.ired[4] = direct;
        }
        __syncthreads();
        const int any_direct = sm.ired[4];
        if (blockDim.x >= 256) {
            if (g < w.ng) {
                chain_scores<1>(D, sm.G2, sm.W, w.base + 64 * g, 1, f1);
                if (any_direct) add_direct_terms<1>(D, K, sh, m, d, w.base + 64 * g, 1, any_direct, (const double*)sm.dl, f1);
            }
        } else if (threadIdx.x < 64) {
            chain_scores<4>(D, sm.G2, sm.W, w.base, w.ng, f);
            if (any_direct) add_direct_terms<4>(D, K, sh, m, d, w.base, w.ng, any_direct, (const double*)sm.dl, f);
        }
    } else if (blockDim.x >= 256) {
        if (g < w.ng) score<1, 1>(D, K, sh, m, d, sm, w.base + 64 * g, 1, f1);  // (w.ng <= 4: waves beyond it idle)
    } else if (threadIdx.x < 64) {
        score<4, 1>(D, K, sh, m, d, sm, w.base, w.ng, f);
    }
    __syncthreads();
    if (blockDim.x >= 256) {
        if (g < 4) sm.fbuf[g * 64 + lane] = g < w.ng ? f1[0] : 0.0;
    } else if (threadIdx.x < 64) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sm.fbuf[q * 64 + lane] = f[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = sm.fbuf[q * 64 + lane];
}

__device__ __forceinline__ int augment_loop(const KbDev& D, const KbState& K, int task, int env, int dict, int m, int d, int y,
                                            int c_from, int c_to, Win w, double (&f)[4], Lds& sm, LoopStats& st) {
    const uint64_t* sh = shells_of(D, K, dict);
    const int n = D.n_prbs;
    while (c_from <= c_to) {
        int zeros;
        const int cstar = first_mistake(f, w, y, c_from, c_to, &zeros);
        const int last = cstar < 0 ? c_to : cstar;
        st.n_pred += (uint64_t)(last - c_from + 1);
        // the predictions made on the way each consume a tie-break draw when f == 0 (Q11)
        if (m > 0 && zeros > 0 && threadIdx.x == 0) K.tie_ctr[task] += (uint32_t)zeros;
        if (cstar < 0) break;
        st.n_mist += 1;
        kernel_column_from_d0(D, K, sh, m, d, (double)cstar / (double)n);
        int branch;
        double delta;
        bool saturated;
        const int m_new = apply_update(D, K, dict, env, m, d, sm.x, (double)cstar / (double)n, cstar, y, sm, &branch, &delta, &saturated);
        // A FULL dictionary that met a sample it would have added cannot represent this region: the remaining
        // candidates would meet the same wall one O(m^2) projection at a time, so the augmentation of this learner
        // stops for this step (build-defined; the reference's dictionary is unbounded; the oracle does the same)
        if (saturated) break;
        c_from = cstar + 1;
        const uint64_t left = (uint64_t)(c_to - c_from + 1 > 0 ? c_to - c_from + 1 : 0);
        if (branch == 2 && m_new > m && m >= 2) {
            st.n_grow += 1;
            // The dictionary grew by the landmark (state, c*/n) with coefficient y and nothing else changed: in the
            // order the scores are summed it is one more term, E = 1 (same state).  The float32 regime of a
            // single-landmark dictionary (kernel.py:16) keeps the full pass.
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int o = cstar - (w.base + 64 * g + lane);
                o = o < 0 ? -o : o;
                f[g] = __builtin_fma((double)y, sm.G2[256 + (o < 255 ? o : 255)], f[g]);
            }
            st.n_eval += left;
        } else {  // projection (every coefficient moved), or the first two landmarks
            if (branch == 2 && m_new > m) st.n_grow += 1;
            rescore(D, K, sh, m_new, d, sm, w, f);
            st.n_eval += left * (uint64_t)m_new;
        }
        m = m_new;
    }
    return m;
}

__device__ __forceinline__ void flush_stats(const KbState& K, int task, int dict, int m, const LoopStats& st) {
    if (threadIdx.x == 0) {
        K.m[dict] = m;
        if (st.n_mist) K.kf_owner[dict] = -1;  // the K_f row no longer holds a Projectron.predict's cache (Q12 guard)
        // (atomics without a return value: the wave does not wait for four loads at its very end; a learner's counters are
        // only ever touched by one workgroup at a time)
        unsigned long long* o = (unsigned long long*)(K.stats + (size_t)task * 4);
        if (st.n_pred) atomicAdd(&o[0], (unsigned long long)st.n_pred);
        if (st.n_mist) atomicAdd(&o[1], (unsigned long long)st.n_mist);
        if (st.n_grow) atomicAdd(&o[2], (unsigned long long)st.n_grow);
        if (st.n_eval) atomicAdd(&o[3], (unsigned long long)st.n_eval);
    }
}

// KBRL_Control.update_control for one learner (kbrl_control.py:83-112): one wave scores every candidate of the state,
// does the accuracy bookkeeping and looks for the first mistake of the augmentation range.  Nine learners in ten have none
// and are done.  The others are queued for update_heavy_kernel (a whole workgroup per learner for the O(m^2) Projectron
// updates), which resumes exactly here; the INLINE instance keeps learners below D.heavy_m landmarks and repairs them
// itself (one wave; KBRL_HEAVY_M, tests).
template <bool INLINE>
__global__ __launch_bounds__(64, INLINE ? 2 : KB_OCC) void update_control_kernel(CtlArgs A) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    __shared__ Lds sm;
    const int task = learner_of_block(K, D.n_envs * D.S, A.big_par);
    if (task < 0) return;
    const int env = task / D.S, s = task - env * D.S;
    const int d = D.dims[s] + 1, n = D.n_prbs;
    const int dict = dict_of(D, task);
    const uint64_t* sh = shells_of(D, K, dict);
    int m = K.m[dict];
    load_gtab(K, sm);
    stage_state(D, A.state, env, s, d, sm);
    __syncthreads();
    const int a_i = A.action[env * D.S + s];
    const int y = A.labels[env * D.S + s];
    LoopStats st = {1, 0, 0, (uint64_t)m};

    // ---- the classifier on every candidate of this state (the augmentation range contains a_i)
    const int c_from = y == 1 ? a_i : 0;
    const int c_to = y == 1 ? n : a_i;
    const Win w = window_of(c_from, c_to);
    double f[4];
    {
        // select_action scored every candidate of the state it selected for (select_gemm_kernel) and left the D0 / E rows of
        // that state in the dictionary.  KBRL_Control.run hands the same state back (kbrl_control.py:129-134) and nothing
        // learns in between, so those scores ARE update_control's first predictions; otherwise (another state, or the
        // dictionary changed since: kb_update, another update_control) they are formed here, by the same sums.
        const int lane = threadIdx.x & 63;
        const float* st = A.state + (size_t)env * D.nv + D.off[s];
        const bool differs = lane < d - 1 && __float_as_uint(st[lane]) != __float_as_uint(K.fstate[(size_t)task * 16 + lane]);
        const bool stored = K.fver[task] == K.ver[dict] && __ballot(differs) == 0ull;
        if (stored) {
            const double* F = K.F + (size_t)task * 256;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = w.base + 64 * g + lane;
                f[g] = g < w.ng ? F[c < 255 ? c : 255] : 0.0;
            }
        } else {
            score_binned<4, 0>(D, K, sh, m, d, sm, w.base, w.ng, f);
            if (threadIdx.x == 0) K.fver[task] = -1;  // (the D0 / E rows now belong to this state, not to the stored scores')
        }
    }
    control_bookkeeping(D, K, task, env, s, m, f_of(f, w, a_i), y, A.hits, sm);

    // ---- sample augmentation (kbrl_control.py:102-112), in the reference's order
    st.n_eval += (uint64_t)(c_to - c_from + 1) * (uint64_t)m;
    if (!INLINE || m >= D.heavy_m) {
        int zeros;
        const int cst = first_mistake(f, w, y, c_from, c_to, &zeros);
        if (cst >= 0) {
            // queued with its scores: large dictionaries from the front of the queue array (the repair rounds), small ones
            // from its far end (update_small_kernel)
            const bool large = m >= KB_SMALL_M || INLINE;
            const int T = D.n_envs * D.S;
            int slot = 0;
            if (threadIdx.x == 0) {
                if (large) {
                    slot = atomicAdd(&K.heavy[0], 1);
                    const int nbq = (m + 63) >> 6;
                    atomicAdd(&K.heavy[3], nbq * nbq);  // tiles of Kinv a repair of this learner walks (the rounds' gate)
                } else {
                    slot = T - 1 - atomicAdd(&K.heavy[2], 1);
                }
            }
            slot = __builtin_amdgcn_readfirstlane(slot);
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int g = 0; g < 4; ++g) K.hv_f[(size_t)slot * 256 + 64 * g + lane] = f[g];
            if (large) {
                // the first repair is prepared here: range, the mistake and its kernel column
                kernel_column_from_d0(D, K, sh, m, d, (double)cst / (double)n);
                if (threadIdx.x == 0) {
                    K.heavy[4 + slot] = task;
                    K.hv_cfrom[slot] = c_from;
                    K.hv_cstar[slot] = cst;
                    K.hv_pend[2 * slot] = cst - c_from + 1;
                    K.hv_pend[2 * slot + 1] = zeros;
                    K.hv_grew[slot] = 0;
                    K.hv_state[slot] = 1;
                }
            } else if (threadIdx.x == 0) {
                K.heavy[4 + slot] = task;
            }
            flush_stats(K, task, dict, m, st);
            return;
        }
    }
    if (INLINE) {
        m = augment_loop(D, K, task, env, dict, m, d, y, c_from, c_to, w, f, sm, st);
    } else {  // no mistake anywhere in the range: the predictions of the whole range were made (their ties draw, Q11)
        int zeros;
        (void)first_mistake(f, w, y, c_from, c_to, &zeros);
        st.n_pred += (uint64_t)(c_to - c_from + 1);
        if (m > 0 && zeros > 0 && threadIdx.x == 0) K.tie_ctr[task] += (uint32_t)zeros;
    }
    flush_stats(K, task, dict, m, st);
}

// the learners update_control_kernel queued.  Dictionaries below KB_SMALL_M landmarks: a workgroup of four waves each, all at
// once (update_small_kernel; workgroups beyond the queue leave at once); larger ones: the repair rounds below.
__device__ __forceinline__ void repair_learner(const CtlArgs& A, int qslot, Lds& sm) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    const int task = K.heavy[4 + qslot];
    const int env = task / D.S, s = task - env * D.S;
    const int d = D.dims[s] + 1, n = D.n_prbs;
    const int dict = dict_of(D, task);
    int m = K.m[dict];
    stage_state(D, A.state, env, s, d, sm);
    __syncthreads();
    const int a_i = A.action[env * D.S + s];
    const int y = A.labels[env * D.S + s];
    LoopStats st = {0, 0, 0, 0};
    const int c_from = y == 1 ? a_i : 0, c_to = y == 1 ? n : a_i;
    const Win w = window_of(c_from, c_to);
    double f[4];  // the window's scores as update_control_kernel had them (the D0 / E rows are this state's too)
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int g = 0; g < 4; ++g) f[g] = K.hv_f[(size_t)qslot * 256 + 64 * g + lane];
    m = augment_loop(D, K, task, env, dict, m, d, y, c_from, c_to, w, f, sm, st);
    flush_stats(K, task, dict, m, st);
}

#ifndef KB_SMALL_OCC
#define KB_SMALL_OCC 2  // waves per SIMD update_small_kernel is built for
#endif
__global__ __launch_bounds__(256, KB_SMALL_OCC) void update_small_kernel(CtlArgs A) {
    const KbState& K = A.K;
    const int count = K.heavy[2];
    if ((int)blockIdx.x >= count) return;
    __shared__ Lds sm;
    load_gtab(K, sm);
    for (int slot = blockIdx.x; slot < count; slot += gridDim.x) {
        __syncthreads();
        repair_learner(A, A.D.n_envs * A.D.S - 1 - slot, sm);
    }
}

// ---- the repair rounds of the large learners.  One Projectron update of a dictionary of m landmarks reads m^2 doubles
// of Kinv (d* = Kinv K_f) and, when it inserts, rewrites them (rank-1 update): against a dictionary of a thousand
// landmarks that is 8 + 16 MB, far more than one workgroup should stream on its own while the rest of the chip waits.  So
// the queued learners advance together, one repair per round, in three kernels as wide as the chip:
//   heavy_matvec_kernel   the partial sums of d* of every pending learner, its work units spread over gridDim.y blocks
//   heavy_finish_kernel   per learner: d*, delta, the decision, projection or insertion; then the scores of what is left
//                         of the range, the next mistake and its kernel column -- or done
//   heavy_rank1_kernel    Kinv's rank-1 update of the learners that inserted, units spread as above
// After KB rounds (kb_api.hip) whatever is still pending is finished learner by learner by update_heavy_kernel.
// The work of a round is spread over the launch by COST, not by learner: a dictionary of 1,755 landmarks has sixty times
// the mat-vec and rank-1 work of one of 229, and a fixed number of workgroups per learner left the launch waiting for its
// largest one.  heavy_plan_kernel (one workgroup) lays the learners' work end to end -- mat-vec: n_b^2 x 8 tile-row
// passes (a unit = one column block and row class, n_b passes), rank-1: n_b^2 x 4 units -- and every wave of the two wide
// kernels takes an equal stretch of that line (whole units; one binary search for where its stretch begins).
__global__ __launch_bounds__(1024) void heavy_plan_kernel(KbDev D, KbState K) {
    __shared__ long long sc[2][1024];
    const int count = K.heavy[0], t = threadIdx.x;
    long long carry_mv = 0, carry_r1 = 0;
    for (int base = 0; base < count; base += 1024) {
        const int slot = base + t;
        long long wmv = 0, wr1 = 0;
        if (slot < count) {
            if (K.hv_state[slot] == 1) {
                const long long nb = (K.m[dict_of(D, K.heavy[4 + slot])] + 63) >> 6;
                wmv = D.tri ? nb * (nb + 1) / 2 : nb * nb * 8;  // tri: tiles; else units of nb passes (below)
            }
            if (K.hv_grew[slot] == 1) {
                const long long nb1 = (K.hv_m[slot] + 1 + 63) >> 6;
                wr1 = (D.tri ? nb1 * (nb1 + 1) / 2 : nb1 * nb1) * 4;
            }
        }
        __syncthreads();
        sc[0][t] = wmv;
        sc[1][t] = wr1;
        __syncthreads();
        long long imv = wmv, ir1 = wr1;
        for (int dd = 1; dd < 1024; dd <<= 1) {  // Hillis-Steele inclusive scans
            const long long a = t >= dd ? sc[0][t - dd] : 0, b = t >= dd ? sc[1][t - dd] : 0;
            __syncthreads();
            imv += a;
            ir1 += b;
            sc[0][t] = imv;
            sc[1][t] = ir1;
            __syncthreads();
        }
        if (slot < count) {
            K.hv_mvbase[slot] = carry_mv + imv - wmv;
            K.hv_r1base[slot] = carry_r1 + ir1 - wr1;
        }
        carry_mv += sc[0][1023];
        carry_r1 += sc[1][1023];
    }
    if (t == 0) {
        K.hv_mvbase[count] = carry_mv;
        K.hv_r1base[count] = carry_r1;
    }
}

// the stretch [lo, hi) of wave w of the launch on a work line of `total`; the slot whose work contains `lo`
__device__ __forceinline__ int stretch_of(const long long* base, int count, long long* lo, long long* hi) {
    const long long total = base[count];
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6), w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    *lo = total * w / nwaves;
    *hi = total * (w + 1) / nwaves;
    if (*lo >= *hi) return -1;
    int a = 0, b = count;  // last slot with base[slot] <= lo
    while (b - a > 1) {
        const int mid = (a + b) >> 1;
        if (base[mid] <= *lo) a = mid; else b = mid;
    }
    return a;
}

#ifndef KB_MV_OCC
#define KB_MV_OCC 4  // waves per SIMD heavy_matvec_kernel is built for (102 registers unconstrained: four)
#endif
#ifndef KB_MV_LDS
#define KB_MV_LDS 1  // two slabs in flight per wave, the row-wise operand through LDS (matvec_tri_tiles)
#endif
__global__ __launch_bounds__(256, KB_MV_OCC) void heavy_matvec_kernel(KbDev D, KbState K) {
    __shared__ double slabs[KB_MV_LDS ? 4 : 1][KB_MV_LDS ? 8 * KB_SLAB_LD : 1];
    const int count = K.heavy[0];
    if (count == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0 && K.hv_mvbase[count] > 0) {  // (the roofline's byte count: kb_get_repair_work)
        K.hv_work[0] += (unsigned long long)K.hv_mvbase[count];
        K.hv_work[2] += 1;
    }
    long long lo, hi;
    int slot = stretch_of(K.hv_mvbase, count, &lo, &hi);
    if (slot < 0) return;
    for (; slot < count && lo < hi; ++slot) {
        const long long sb = K.hv_mvbase[slot], se = K.hv_mvbase[slot + 1];
        if (se <= lo) continue;  // (no work of its own, or entirely before the stretch)
        const int dict = dict_of(D, K.heavy[4 + slot]);
        const int m = K.m[dict];
        const long long nb = (m + 63) >> 6;
        const long long a = lo > sb ? lo - sb : 0, b = (hi < se ? hi : se) - sb;
        if (D.tri) {  // the work line counts tiles
            if (a < b) {
                if (KB_MV_LDS) matvec_tri_tiles_lds(K, shells_of(D, K, dict), m, (int)a, (int)b, slabs[threadIdx.x >> 6]);
                else matvec_tri_tiles(K, shells_of(D, K, dict), m, (int)a, 1, (int)b);
            }
        } else {      // units whose first pass lies in [lo, hi)
            const int u0 = (int)((a + nb - 1) / nb), u1 = (int)((b + nb - 1) / nb);
            if (u0 < u1) matvec_partials<1>(K, shells_of(D, K, dict), m, u0, 1, u1);
        }
        lo = se;
    }
}

__global__ __launch_bounds__(256) void heavy_rank1_kernel(KbDev D, KbState K) {
    const int count = K.heavy[0];
    if (count == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0 && K.hv_r1base[count] > 0) {
        K.hv_work[1] += (unsigned long long)K.hv_r1base[count];
        K.hv_work[3] += 1;
    }
    long long lo, hi;
    int slot = stretch_of(K.hv_r1base, count, &lo, &hi);
    if (slot < 0) return;
    for (; slot < count && lo < hi; ++slot) {
        const long long sb = K.hv_r1base[slot], se = K.hv_r1base[slot + 1];
        if (se <= lo) continue;
        const int dict = dict_of(D, K.heavy[4 + slot]);
        const int u0 = (int)(lo > sb ? lo - sb : 0), u1 = (int)((hi < se ? hi : se) - sb);
        if (u0 < u1) rank1_units(K, shells_of(D, K, dict), K.hv_m[slot], K.hv_delta[slot], D.tri, u0, 1, u1);
        lo = se;
    }
}

__global__ __launch_bounds__(256) void heavy_finish_kernel(CtlArgs A) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    __shared__ Lds sm;
    const int count = K.heavy[0];
    if ((int)blockIdx.x >= count) return;
    load_gtab(K, sm);
    const int lane = threadIdx.x & 63;
    for (int slot = blockIdx.x; slot < count; slot += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) K.hv_grew[slot] = 0;  // the previous round's rank-1 update has been made
        if (K.hv_state[slot] != 1) continue;
        const int task = K.heavy[4 + slot], env = task / D.S, s = task - env * D.S;
        const int d = D.dims[s] + 1, n = D.n_prbs;
        const int dict = dict_of(D, task);
        const uint64_t* sh = shells_of(D, K, dict);
        const int m = K.m[dict];
        stage_state(D, A.state, env, s, d, sm);
        __syncthreads();
        const int a_i = A.action[env * D.S + s];
        const int y = A.labels[env * D.S + s];
        const int c_to = y == 1 ? n : a_i;
        const Win w = window_of(y == 1 ? a_i : 0, c_to);
        const int cstar = K.hv_cstar[slot];
        LoopStats st = {(uint64_t)K.hv_pend[2 * slot], 1, 0, 0};
        if (threadIdx.x == 0 && K.hv_pend[2 * slot + 1] > 0) K.tie_ctr[task] += (uint32_t)K.hv_pend[2 * slot + 1];  // Q11
        if (D.tri)
            matvec_tri_combine_wide(K, sh, m);
        else
            matvec_combine(K, sh, m);
        int branch;
        double delta;
        bool saturated;
        const int m_new = finish_update(D, K, dict, env, m, d, sm.x, (double)cstar / (double)n, cstar, y, sm, &branch, &delta,
                                        &saturated, true);
        int state = 1;
        if (saturated) {
            state = 0;  // (augment_loop's rule: a full dictionary stops augmenting for the step)
        } else {
            double f[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) f[g] = K.hv_f[(size_t)slot * 256 + 64 * g + lane];
            const int c_from = cstar + 1;
            const uint64_t left = (uint64_t)(c_to - c_from + 1 > 0 ? c_to - c_from + 1 : 0);
            if (branch == 2 && m_new > m) {
                st.n_grow = 1;
                if (threadIdx.x == 0) {
                    K.hv_grew[slot] = 1;
                    K.hv_m[slot] = m;
                    K.hv_delta[slot] = delta;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {  // one more term, E = 1 (augment_loop)
                    int o = cstar - (w.base + 64 * g + lane);
                    o = o < 0 ? -o : o;
                    f[g] = __builtin_fma((double)y, sm.G2[256 + (o < 255 ? o : 255)], f[g]);
                }
                st.n_eval += left;
            } else {
                rescore(D, K, sh, m_new, d, sm, w, f);
                st.n_eval += left * (uint64_t)m_new;
            }
            int zeros = 0;
            const int next = c_from <= c_to ? first_mistake(f, w, y, c_from, c_to, &zeros) : -1;
            if (next < 0) {
                st.n_pred += left;
                if (zeros > 0 && threadIdx.x == 0) K.tie_ctr[task] += (uint32_t)zeros;
                state = 0;
            } else {
                kernel_column_from_d0(D, K, sh, m_new, d, (double)next / (double)n);
                if (threadIdx.x < 64) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) K.hv_f[(size_t)slot * 256 + 64 * g + lane] = f[g];
                }
                if (threadIdx.x == 0) {
                    K.hv_cfrom[slot] = c_from;
                    K.hv_cstar[slot] = next;
                    K.hv_pend[2 * slot] = next - c_from + 1;
                    K.hv_pend[2 * slot + 1] = zeros;
                }
            }
        }
        if (threadIdx.x == 0) K.hv_state[slot] = state;
        flush_stats(K, task, dict, m_new, st);
    }
}

// what the rounds left pending (and, without rounds, every queued large learner): persistent workgroups take the
// learners one at a time and run the loop to its end
__global__ __launch_bounds__(KB_HEAVY_THREADS) void update_heavy_kernel(CtlArgs A) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    __shared__ Lds sm;
    const int count = K.heavy[0];
    if (count == 0) return;
    load_gtab(K, sm);
    const int lane = threadIdx.x & 63;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) sm.ired[7] = atomicAdd(&K.heavy[1], 1);
        __syncthreads();
        const int slot = sm.ired[7];
        if (slot >= count) break;
        if (K.hv_state[slot] != 1) continue;
        const int task = K.heavy[4 + slot], env = task / D.S, s = task - env * D.S;
        const int d = D.dims[s] + 1, n = D.n_prbs;
        const int dict = dict_of(D, task);
        int m = K.m[dict];
        stage_state(D, A.state, env, s, d, sm);
        __syncthreads();
        const int a_i = A.action[env * D.S + s];
        const int y = A.labels[env * D.S + s];
        LoopStats st = {0, 0, 0, 0};
        double f[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) f[g] = K.hv_f[(size_t)slot * 256 + 64 * g + lane];
        m = augment_loop(D, K, task, env, dict, m, d, y, K.hv_cfrom[slot], y == 1 ? n : a_i, window_of(y == 1 ? a_i : 0, y == 1 ? n : a_i), f,
                         sm, st);
        if (threadIdx.x == 0) K.hv_state[slot] = 0;
        flush_stats(K, task, dict, m, st);
    }
}

struct SelArgs {
    KbDev D;
    KbState K;
    const float* state;  // [n_envs][nv] new state
    int32_t big_par;     // the list that orders this launch; the other one is written for the next step (-1: none)
    int32_t gemm;        // shared mode: shared_fgemm_kernel has scored this state for the large dictionaries
};

// per-learner part of KBRL_Control.select_action (kbrl_control.py:44-63): the smallest candidate the classifier
// accepts.  The reference scans c = 0, 1, 2, ... and stops at the first +1 (about 15 candidates in); so does this:
// 64 candidates per pass of the landmarks, in order, until a pass contains the answer.
// MODE 3: the scores come from shared_fgemm_kernel's partial sums (Fp: this replica's row, `part` doubles between parts)
template <int MODE>
__device__ __forceinline__ int select_scan(const KbDev& D, const KbState& K, const uint64_t* sh, int m, int d, int task, int env,
                                           int s, const Lds& sm, uint64_t* n_scored, const double* Fp = nullptr, size_t part = 0) {
    const int n = D.n_prbs, lane = threadIdx.x & 63;
    int found = -1;
    for (int g = 0; 64 * g <= n && found < 0; ++g) {
        double f[1];
        if (MODE == 3) {
            int cc = 64 * g + lane;
            cc = cc < n ? cc : n;
            f[0] = ((Fp[cc] + Fp[part + cc]) + (Fp[2 * part + cc] + Fp[3 * part + cc])) +
                   ((Fp[4 * part + cc] + Fp[5 * part + cc]) + (Fp[6 * part + cc] + Fp[7 * part + cc]));
        } else if (g == 0) {
            score<1, MODE == 3 ? 2 : MODE>(D, K, sh, m, d, sm, 64 * g, 1, f);
        } else {
            score<1, MODE == 0 ? 1 : (MODE == 3 ? 2 : MODE)>(D, K, sh, m, d, sm, 64 * g, 1, f);
        }
        const int c = 64 * g + lane;
        const int c1 = 64 * g + 63 < n ? 64 * g + 63 : n;
        *n_scored += (uint64_t)(c1 - 64 * g + 1);
        unsigned long long cand = __ballot(c <= n && f[0] >= 0.0);  // positive, or a tie to be drawn
        // walk the (rare) exact ties in order; each consumes one draw (kernel.py:26-27)
        while (cand) {
            const int l = __builtin_ctzll(cand);
            const double fl = readlane_f64(f[0], l);
            if (fl > 0.0) { found = 64 * g + l; break; }
            int dr = 0;
            if (threadIdx.x == 0) dr = tie_draw(K, task, env, s);
            dr = __builtin_amdgcn_readfirstlane(dr);
            if (dr == 1) { found = 64 * g + l; break; }
            cand &= cand - 1;
        }
    }
    return found;
}

__global__ __launch_bounds__(64, KB_OCC) void select_kernel(SelArgs A) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    __shared__ Lds sm;
    const int T = D.n_envs * D.S;
    const int task = learner_of_block(K, T, A.big_par);
    if (task < 0) return;
    const int env = task / D.S, s = task - env * D.S;
    const int d = D.dims[s] + 1, n = D.n_prbs;
    const int dict = dict_of(D, task);
    const uint64_t* sh = shells_of(D, K, dict);
    const int m = K.m[dict];
    if (A.big_par >= 0 && threadIdx.x == 0) {  // the next step's list (the other of the two)
        const int pw = 1 - A.big_par;
        int listed = 0;
        if (m >= KB_BIG_M) {
            int32_t* L = K.big + (size_t)pw * (1 + KB_BIG_MAX);
            const int slot = atomicAdd(&L[0], 1);
            if (slot < KB_BIG_MAX) {
                L[1 + slot] = task;
                listed = 1;
            }
        }
        K.isbig[(size_t)pw * T + task] = listed;
    }
    load_gtab(K, sm);
    stage_state(D, A.state, env, s, d, sm);
    __syncthreads();
    int found = -1;
    uint64_t n_scored = 0;
    if (m > 0) {
        if (D.shared) {
            bool from_gemm = A.gemm && gemm_applies(D, K, dict, m);
            if (from_gemm) {
                const double* E = K.workE + (size_t)s * KB_GEMM_KS * D.n_envs + env;
                double emax = 0.0;
                for (int kp = 0; kp < KB_GEMM_KS; ++kp) emax = E[(size_t)kp * D.n_envs] > emax ? E[(size_t)kp * D.n_envs] : emax;
                from_gemm = emax >= 1e-250;
            }
            if (from_gemm)
                found = select_scan<3>(D, K, sh, m, d, task, env, s, sm, &n_scored,
                                       K.workF + (size_t)s * KB_GEMM_KS * D.n_envs * 256 + (size_t)env * 256, (size_t)D.n_envs * 256);
            else
                found = select_scan<2>(D, K, sh, m, d, task, env, s, sm, &n_scored);
        } else {
            found = select_scan<0>(D, K, sh, m, d, task, env, s, sm, &n_scored);
        }
    }
    if (threadIdx.x == 0) {
        const uint64_t n_pred = found >= 0 ? (uint64_t)found + 1 : (uint64_t)n + 1;
        const int offset = K.security[env * D.S + s];
        int act, margin = 0;
        if (found >= 0) {
            int a = n < found + offset ? n : found + offset;
            margin = a - found;
            act = a;
        } else {
            act = n;
        }
        K.action[env * D.S + s] = act;
        K.margins[env * D.S + s] = margin;
        uint64_t* st = K.stats + (size_t)task * 4;
        st[0] += n_pred;
        st[3] += n_scored * (uint64_t)m;
    }
}

// ---- KBRL_Control.select_action for one agent per replica (kbrl_control.py:44-63), round 4: the scores of EVERY candidate
// as a dense product on the matrix cores, sixteen learners at a time.
//   select_bin_kernel   a wave per learner walks its landmarks once (the exp, the D0 / E rows of the state) and sums
//                       coeff_j E_j per grid index: W[learner][a] (bin_pass)
//   select_gemm_kernel  F = T W^T for sixteen learners per workgroup: T[c][a] = G[|a - c|] (candidates x grid indices) is the
//                       same Toeplitz matrix for every learner of the handle -- its 16 x 4 operand tiles are read straight
//                       out of the G table --, W^T (grid indices x 16 learners) comes through LDS.  A wave owns up to four
//                       16-candidate tiles (four independent accumulators, one B operand read for four v_mfma_f64_16x16x4);
//                       per output this is the chain of fused multiply-adds over a = 0, 1, ... that chain_scores spells
//                       out.  Then every learner's row of F gets the exact exponentials of the landmarks that could not be
//                       binned, is left in K.F (update_control of this state starts from it) and is scanned for the first
//                       accepted candidate, in order, exact ties drawing (kernel.py:26-27).
// Per learner: one pass over the landmarks plus 208 x 204 multiply-adds on the matrix pipe, whatever m is.  Round 3 walked
// every landmark for every group of 64 candidates, here and again in update_control.
#ifndef KB_BIN_OCC
#define KB_BIN_OCC 4
#endif
// The next step's list of large learners (the other of the two) and who is on it, from the dictionary sizes: a workgroup counts
// its large learners and reserves their places with ONE atomic.  Until round 5 every wave of select_bin_kernel took its place
// itself -- some 4,000 returning atomics on one address per launch at step 3000 of config 3, served one after the other by
// one L2 channel: THE duration of that kernel (0.2 ms where its memory traffic takes 0.1; tools/ubench/scatter_read.hip).
__global__ __launch_bounds__(256) void big_list_kernel(KbDev D, KbState K, int big_par) {
    __shared__ int wcount[4], base;
    const int T = D.n_envs * D.S, pw = 1 - big_par;
    const int task = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool big = task < T && K.m[dict_of(D, task)] >= KB_BIG_M;
    const unsigned long long mask = __ballot(big);
    if (lane == 0) wcount[wv] = __builtin_popcountll(mask);
    __syncthreads();
    int32_t* L = K.big + (size_t)pw * (1 + KB_BIG_MAX);
    if (threadIdx.x == 0) {
        const int tot = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        base = tot ? atomicAdd(&L[0], tot) : 0;
    }
    __syncthreads();
    int pos = base + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
    for (int q = 0; q < wv; ++q) pos += wcount[q];
    const int listed = big && pos < KB_BIG_MAX ? 1 : 0;
    if (listed) L[1 + pos] = task;
    if (task < T) K.isbig[(size_t)pw * T + task] = listed;
}

// The learners of the large-learner list, a workgroup of KB_BINBIG_WAVES waves each: wave w bins segments w, w + waves, ... of
// the dictionary (KB_BIN_SEG chunks each) into its own LDS array, and the segments' sums are added up in increasing order after
// every round -- the order bin_pass keeps on one wave, so both give the same bits.  One wave walking a dictionary of 1,645
// landmarks chunk after chunk was select_bin_kernel's whole duration at step 3000 of config 3 (26 chunks x 7.6 us: round 5,
// tools/stamps_probe.sh); its mean learner has 3.5 chunks.
#ifndef KB_BINBIG_WAVES
#define KB_BINBIG_WAVES 4
#endif
#define KB_BINBIG_GRID 1024
struct BinBigLds {
    double W[256];
    double Ws[KB_BINBIG_WAVES][256];
    double dls[KB_BINBIG_WAVES][KB_DLIST * 3];
    double x[KB_DMAX];
    int res[KB_BINBIG_WAVES];
};
__device__ __forceinline__ void select_bin_big_body(const SelArgs& A, BinBigLds& L_) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    double (&W)[256] = L_.W;
    double (&Ws)[KB_BINBIG_WAVES][256] = L_.Ws;
    double (&dls)[KB_BINBIG_WAVES][KB_DLIST * 3] = L_.dls;
    double (&x)[KB_DMAX] = L_.x;
    int (&res)[KB_BINBIG_WAVES] = L_.res;
    const int T = D.n_envs * D.S;
    // KB_BINBIG_GRID workgroups walk the list (an empty list costs a thousand workgroups that leave at once, not four thousand)
    for (int slot = (int)blockIdx.x; slot < KB_BIG_MAX; slot += KB_BINBIG_GRID) {
    const int task = learner_of_slot(K, T, A.big_par, slot);  // (slots below KB_BIG_MAX: the list)
    if (task < 0) return;  // (the list is dense from its start: nothing further on either)
    __syncthreads();       // (the previous learner's LDS is done with)
    const int env = task / D.S, s = task - env * D.S;
    const int d = D.dims[s] + 1;
    const int dict = dict_of(D, task);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t* sh = shells_of(D, K, dict);
    const uint64_t shv = shell_vector(D, sh);
    const int m = K.m[dict];
    if (m < 2) {
        if (threadIdx.x == 0) K.fdirect[task] = 0;
        continue;
    }
    if (threadIdx.x < 256) W[threadIdx.x] = 0.0;
    if (threadIdx.x < d - 1) x[threadIdx.x] = (double)A.state[(size_t)env * D.nv + D.off[s] + threadIdx.x];
    __syncthreads();
    const int nch = (m + 63) >> 6, nseg = (nch + KB_BIN_SEG - 1) / KB_BIN_SEG;
    const bool f32 = KB_F32_ROWS && d - 1 == 10 && K.f32bad[dict] == 0;
    double* dlist = K.dlist + (size_t)task * (KB_DLIST * 3);
    int flags = 0, ndir = 0;
    for (int s0 = 0; s0 < nseg; s0 += KB_BINBIG_WAVES) {
        const int sg = s0 + wv;
        int r = 0;
        if (sg < nseg) {
#pragma unroll
            for (int k = 0; k < 4; ++k) Ws[wv][lane + 64 * k] = 0.0;
            const int b0 = sg * KB_BIN_SEG, b1 = b0 + KB_BIN_SEG < nch ? b0 + KB_BIN_SEG : nch;
            r = bin_chunks<0, KB_BIN_DEEP>(D, K, sh, shv, m, d, x, b0, b1, Ws[wv], dls[wv], 0, f32);
        }
        if (lane == 0) res[wv] = r;
        __syncthreads();
        const int here = nseg - s0 < KB_BINBIG_WAVES ? nseg - s0 : KB_BINBIG_WAVES;
        if (threadIdx.x < 256) {  // the segments of this round, in order
            double acc = W[threadIdx.x];
            for (int q = 0; q < here; ++q) acc += Ws[q][threadIdx.x];
            W[threadIdx.x] = acc;
        }
        // the direct-evaluation lists of the round's segments, in order, behind what earlier rounds listed (all threads walk the
        // same counts; the first KB_DLIST entries overall are kept, as bin_pass keeps them)
        for (int q = 0; q < here; ++q) {
            const int rq = res[q], nq = rq >> 8;
            flags |= rq & 3;
            const int keep = nq < KB_DLIST ? nq : KB_DLIST;
            for (int e = threadIdx.x; e < 3 * keep; e += blockDim.x)
                if (ndir + e / 3 < KB_DLIST) dlist[3 * ndir + e] = dls[q][e];
            ndir += nq;
        }
        __syncthreads();
    }
    if (threadIdx.x < 256) K.Wg[(size_t)task * 256 + threadIdx.x] = W[threadIdx.x];
    if (threadIdx.x == 0) K.fdirect[task] = flags | (ndir << 8);
    }
}
__global__ __launch_bounds__(64 * KB_BINBIG_WAVES, KB_BIN_OCC) void select_bin_big_kernel(SelArgs A) {
    __shared__ BinBigLds L_;
    select_bin_big_body(A, L_);
}

// every other learner (and, on a handle without the list, every learner): a wave each.  slot0: KB_BIG_MAX when the list's
// places are select_bin_big_kernel's, else 0
struct BinLds {
    double W[256];
    double Wseg[256];
    double x[KB_DMAX];
};
// one wave's barrier with its own LDS traffic: __syncthreads where the wave is the whole workgroup (SOLO), a wave barrier with LDS
// fences where other waves of the workgroup have left or do other work (select_bin_all_kernel)
template <bool SOLO>
__device__ __forceinline__ void bin_wave_sync() {
    if (SOLO) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
template <bool SOLO>
__device__ __forceinline__ void select_bin_body(const SelArgs& A, int slot, BinLds& L_) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    double (&W)[256] = L_.W;
    double (&Wseg)[256] = L_.Wseg;
    double (&x)[KB_DMAX] = L_.x;
    const int T = D.n_envs * D.S;
#ifdef KB_BIN_STAMPS  // experiment build: where a wave's time goes (every 16th wave adds its s_memtime differences to hv_work[4..7])
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
    const int task = learner_of_slot(K, T, A.big_par, slot);
    if (task < 0) return;
    const int env = task / D.S, s = task - env * D.S;
    const int d = D.dims[s] + 1;
    const int dict = dict_of(D, task);
    const int lane = threadIdx.x & 63;
    const uint64_t* sh = shells_of(D, K, dict);
    const uint64_t shv = shell_vector(D, sh);  // (requested together with m: one round trip, not two)
    const int m = K.m[dict];
#ifdef KB_BIN_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    if (m < 2) {  // (nothing to bin: select_gemm_kernel scores the single landmark in float32, kernel.py:16)
        if (lane == 0) K.fdirect[task] = 0;
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) W[lane + 64 * k] = 0.0;
    if (lane < d - 1) x[lane] = (double)A.state[(size_t)env * D.nv + D.off[s] + lane];
    bin_wave_sync<SOLO>();
    const int direct = bin_pass<0, KB_BIN_DEEP>(D, K, sh, m, d, x, W, Wseg, K.dlist + (size_t)task * (KB_DLIST * 3), shv,
                                                KB_F32_ROWS && d - 1 == 10 && K.f32bad[dict] == 0);
#ifdef KB_BIN_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && (blockIdx.x & 15) == 0) {
        atomicAdd(&K.hv_work[4], st1 - st0);                        // entry -> task, m, shell offsets known
        atomicAdd(&K.hv_work[5], st2 - st1);                        // the landmarks (all chunks, stores drained)
        atomicAdd(&K.hv_work[6], (unsigned long long)((m + 63) >> 6));  // chunks
        atomicAdd(&K.hv_work[7], 1ull);                             // waves recorded
    }
#endif
    bin_wave_sync<SOLO>();
    double* Wg = K.Wg + (size_t)task * 256;
#pragma unroll
    for (int k = 0; k < 4; ++k) Wg[lane + 64 * k] = W[lane + 64 * k];
    if (lane == 0) K.fdirect[task] = direct;
}
__global__ __launch_bounds__(64, KB_BIN_OCC) void select_bin_kernel(SelArgs A, int slot0) {
    __shared__ BinLds L_;
    select_bin_body<true>(A, slot0 + (int)blockIdx.x, L_);
}
// Both in ONE launch (round 6): the first KB_BINBIG_GRID workgroups walk the large-learner list four waves a learner, every
// other workgroup takes four of the remaining learners, a wave each.  As two launches on one stream the two kernels ran one after
// the other (0.085 + 0.095 ms at step 3000 of config 3), each waiting three quarters of its cycles for memory; side by side they
// share the chip.  Same routines, same sums, same bits.
__global__ __launch_bounds__(64 * KB_BINBIG_WAVES, KB_BIN_OCC) void select_bin_all_kernel(SelArgs A) {
    __shared__ union {
        BinBigLds big;
        BinLds one[KB_BINBIG_WAVES];
    } L_;
    if ((int)blockIdx.x < KB_BINBIG_GRID) {
        select_bin_big_body(A, L_.big);
    } else {
        const int wv = threadIdx.x >> 6;
        select_bin_body<false>(A, KB_BIG_MAX + ((int)blockIdx.x - KB_BINBIG_GRID) * KB_BINBIG_WAVES + wv, L_.one[wv]);
    }
}

#define KB_WT_LD 17  // doubles between the rows of W^T in LDS (16 learners + 1: the transposing stores spread over the banks)
struct GemmLds {
    double G2[512];
    union {
        double Wt[256 * KB_WT_LD];          // W^T[a][learner]
        double Fs[KB_SEL_WAVES][256];       // F[learner][candidate], once every wave is done with W^T
    };
    int task[KB_SEL_WAVES], m[KB_SEL_WAVES];
};

__global__ __launch_bounds__(256) void select_gemm_kernel(SelArgs A) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    __shared__ GemmLds sm;
    const int T = D.n_envs * D.S, n = D.n_prbs;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < KB_SEL_WAVES) {
        const int t = learner_of_slot(K, T, A.big_par, (int)blockIdx.x * KB_SEL_WAVES + (int)threadIdx.x);
        sm.task[threadIdx.x] = t;
        sm.m[threadIdx.x] = t >= 0 ? K.m[dict_of(D, t)] : 0;
    }
    for (int k = threadIdx.x; k < 512; k += blockDim.x) sm.G2[k] = K.gtab[k < 256 ? 256 - k : k - 256];
    __syncthreads();
    // ---- W^T of the block's sixteen learners (wave w brings learners 4 w .. 4 w + 3; a learner without a W is a zero column)
    {
        double v[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int l = 4 * wv + q;
            const bool on = sm.task[l] >= 0 && sm.m[l] >= 2;
            const double* Wg = K.Wg + (size_t)(on ? sm.task[l] : 0) * 256;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[q][k] = on ? Wg[lane + 64 * k] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) sm.Wt[(lane + 64 * k) * KB_WT_LD + 4 * wv + q] = v[q][k];
    }
    __syncthreads();
    // ---- F = T W^T: wave w owns the candidate tiles w, w + 4, w + 8, w + 12 (16 candidates each) of all sixteen learners
    const int nt = n / 16 + 1, KA = (n + 4) & ~3;
    const int li = lane & 15, kq = lane >> 4;
    kb_f64x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (kb_f64x4){0.0, 0.0, 0.0, 0.0};
    {
        const double* Wb = sm.Wt + kq * KB_WT_LD + li;      // B operand: W^T[a0 + kq][learner li]
        const double* Ga = sm.G2 + 256 + kq - (16 * wv + li);  // A operand of tile t: T[16 (w + 4 t) + li][a0 + kq] = G2[256 + a - c]
        // (all four tiles unconditionally -- a tile past the last candidate costs its MFMAs and is never stored; a wave-uniform
        // "if (tile < nt)" around each MFMA made the compiler park the accumulators in VGPRs and move them through the same
        // eight AGPRs around every instruction: 340 cycles per MFMA instead of 64)
        for (int a0 = 0; a0 < KA; a0 += 4) {
            const double b = Wb[a0 * KB_WT_LD];
            double ta[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) ta[t] = Ga[a0 - 64 * t];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[t], b, acc[t], 0, 0, 0);
        }
    }
    __syncthreads();  // (every wave is done with W^T: F takes its place)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (wv + 4 * t < nt) {
            // the lane holds F[candidate 16 (w + 4 t) + kq + 4 v][learner li]
#pragma unroll
            for (int v = 0; v < 4; ++v) sm.Fs[li][16 * (wv + 4 * t) + kq + 4 * v] = acc[t][v];
        }
    }
    __syncthreads();
    // ---- per learner: the row of F, the first accepted candidate (kbrl_control.py:54-61); wave w takes learners w, w + 4, ...
    for (int l = wv; l < KB_SEL_WAVES; l += 4) {
        const int task = sm.task[l];
        if (task < 0) continue;
        const int m = sm.m[l];
        const int env = task / D.S, s = task - env * D.S;
        const int d = D.dims[s] + 1;
        const int dict = dict_of(D, task);
        const uint64_t* sh = shells_of(D, K, dict);
        const int offset = K.security[env * D.S + s];  // (needed at the very end: in flight meanwhile)
        double f[4];
        if (m >= 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) f[g] = 64 * g <= n ? sm.Fs[l][64 * g + lane] : 0.0;
            const int direct = K.fdirect[task];
            if (direct) add_direct_terms<4>(D, K, sh, m, d, 0, n / 64 + 1, direct, K.dlist + (size_t)task * (KB_DLIST * 3), f);
        } else if (m == 1) {
            double x[KB_DMAX];
#pragma unroll
            for (int q = 0; q < KB_DMAX - 1; ++q) x[q] = q < d - 1 ? (double)A.state[(size_t)env * D.nv + D.off[s] + q] : 0.0;
            score_single<4, 0>(D, K, sh, d, x, 0, f);
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) f[g] = 0.0;
        }
        {
            double* F = K.F + (size_t)task * 256;
#pragma unroll
            for (int g = 0; g < 4; ++g) F[64 * g + lane] = f[g];
            if (lane < d - 1) K.fstate[(size_t)task * 16 + lane] = A.state[(size_t)env * D.nv + D.off[s] + lane];
            if (lane == 0) K.fver[task] = K.ver[dict];
        }
        int found = -1;
        uint64_t n_scored = 0;
        if (m > 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (64 * g <= n && found < 0) {
                    const int c = 64 * g + lane;
                    const int c1 = 64 * g + 63 < n ? 64 * g + 63 : n;
                    n_scored += (uint64_t)(c1 - 64 * g + 1);
                    unsigned long long cand = __ballot(c <= n && f[g] >= 0.0);  // positive, or a tie to be drawn
                    while (cand) {  // walk the (rare) exact ties in order; each consumes one draw (kernel.py:26-27)
                        const int ll = __builtin_ctzll(cand);
                        const double fl = readlane_f64(f[g], ll);
                        if (fl > 0.0) { found = 64 * g + ll; break; }
                        int dr = 0;
                        if (lane == 0) dr = tie_draw(K, task, env, s);
                        dr = __builtin_amdgcn_readfirstlane(dr);
                        if (dr == 1) { found = 64 * g + ll; break; }
                        cand &= cand - 1;
                    }
                }
            }
        }
        if (lane == 0) {
            const uint64_t n_pred = found >= 0 ? (uint64_t)found + 1 : (uint64_t)n + 1;
            int act, margin = 0;
            if (found >= 0) {
                int a = n < found + offset ? n : found + offset;
                margin = a - found;
                act = a;
            } else {
                act = n;
            }
            K.action[env * D.S + s] = act;
            K.margins[env * D.S + s] = margin;
            unsigned long long* st = (unsigned long long*)(K.stats + (size_t)task * 4);
            atomicAdd(&st[0], (unsigned long long)n_pred);  // (no return value: nothing waits for it)
            atomicAdd(&st[3], (unsigned long long)(n_scored * (uint64_t)m));
        }
    }
}

// cross-learner part of select_action + adjust_action (kbrl_control.py:65-78)
__global__ void adjust_kernel(KbDev D, KbState K, int32_t* action_out, int big_par) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env == 0 && big_par >= 0) {  // the list select_kernel just used is the one the next select writes: empty it (a count past
                                     // KB_BIG_MAX in the other one only means that some learners kept their place)
        K.big[(size_t)big_par * (1 + KB_BIG_MAX)] = 0;
        int32_t* Lw = K.big + (size_t)(1 - big_par) * (1 + KB_BIG_MAX);
        if (Lw[0] > KB_BIG_MAX) Lw[0] = KB_BIG_MAX;
    }
    if (env >= D.n_envs) return;
    long assigned = 0;
    for (int s = 0; s < D.S; ++s) assigned += K.action[env * D.S + s];
    int adjusted = 0;
    if (assigned > D.n_prbs) {
        adjusted = 1;
        for (int s = 0; s < D.S; ++s) {
            int a = K.action[env * D.S + s];
            double p = (double)a / (double)assigned;
            int na = (int)(int16_t)__builtin_floor((double)D.n_prbs * p);
            K.margins[env * D.S + s] = (int)(int16_t)(K.margins[env * D.S + s] - (a - na));
            K.action[env * D.S + s] = na;
        }
    }
    K.adjusted[env] = adjusted;
    if (action_out)
        for (int s = 0; s < D.S; ++s) action_out[env * D.S + s] = K.action[env * D.S + s];
}

// ---- per-step histories of KBRL_Control.run (kbrl_control.py:119-124,135-141), recorded on the device by the resident
// loop: one column per step, one row per replica
struct HistArgs {
    KbDev D;
    KbState K;
    const double* reward;      // env: reward of the step just executed
    const int32_t* labels;     // env: [n_envs][S]
    const int32_t* violations; // env: [n_envs][S]
    const int32_t* hits;       // [n_envs][S] of this update_control
    double* h_reward;          // [n_envs][steps]
    int16_t* h_resources;      // [n_envs][steps]   action.sum() of the NEW action
    int16_t* h_hits;           // [n_envs][S][steps]
    int16_t* h_adjusted;       // [n_envs][steps]
    int16_t* h_sla;            // [n_envs][steps]   SLA_labels.sum()
    int16_t* h_violation;      // [n_envs][steps]   total_violations
    int32_t* cursor;           // [1] next column
    int32_t steps;
};

__global__ void history_kernel(HistArgs A) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = A.cursor[0];
    if (env < A.D.n_envs && i < A.steps) {
        const int S = A.D.S;
        int res = 0, sla = 0, viol = 0;
        for (int s = 0; s < S; ++s) {
            res += A.K.action[env * S + s];
            sla += A.labels[env * S + s];
            viol += A.violations[env * S + s];
            A.h_hits[((size_t)env * S + s) * A.steps + i] = (int16_t)A.hits[env * S + s];
        }
        const size_t o = (size_t)env * A.steps + i;
        A.h_reward[o] = A.reward[env];
        A.h_resources[o] = (int16_t)res;
        A.h_adjusted[o] = (int16_t)A.K.adjusted[env];
        A.h_sla[o] = (int16_t)sla;
        A.h_violation[o] = (int16_t)viol;
    }
}
__global__ void history_advance_kernel(int32_t* cursor) { cursor[0] += 1; }

// ---- shared-dictionary mode (build-defined extension): one dictionary per slice index,
// learned from every replica on every GPU.  A step is a few rounds of
//   scan    each replica finds its first mistake (in the reference's augmentation order) against the frozen
//           shared dictionary                                                  [shared_scan_kernel]
//   collect the first B proposers per slice, in replica order                  [shared_collect_kernel]
//   (host)  all-gather of the proposal lists over RCCL, merge by global replica id, keep the first B
//   apply   every rank applies the same merged list, in order, through Projectron.predict/update, so all
//           ranks hold bitwise-identical dictionaries                           [shared_apply_kernel]
//   commit  replicas whose proposal was taken move their cursor past it         [shared_commit_kernel]
// With a single replica this is exactly the reference's sequential loop (kbrl_control.py:103-112).

#define KB_PROP_W (2 + KB_DMAX)  // doubles per proposal: global replica id, c | (y << 16) packed as double, x...

struct ScanArgs {
    KbDev D;
    KbState K;
    const float* state;
    const int32_t* action;
    const int32_t* labels;
    int32_t* hits;
    int32_t* cursor;  // [T] next candidate to examine; < 0: range exhausted
    int32_t* cstar;   // [T] candidate proposed in this round, -1 none
    int32_t round;
};

// ---- a large shared dictionary scores ALL replicas of the rank at once.  With E[r][j] = exp(-gamma |l_j[:d-1] - state_r|^2)
// (the RBF Gram block of replicas x landmarks) and Q[j][c] = coeff_j G[|a_j - c|] (landmarks x candidates),
//     F[r][c] = sum_j E[r][j] Q[j][c]
// is a dense contraction: v_mfma_f64_16x16x4, a wave per (16 replicas, part of the landmarks, 8 candidate tiles); the A
// operand E is produced in place (each lane the distance and the exp of its own (replica, landmark) pair), B comes from
// the tiles shared_q_kernel lays out.  shared_scan_kernel then only adds the KB_GEMM_KS partial sums of its window.  A
// replica whose E_j are ALL below 1e-250 (an outlier state, see score_pass) keeps the streaming pass and its direct
// exponentials; so does a dictionary with off-grid landmarks.

__global__ __launch_bounds__(256) void shared_q_kernel(KbDev D, KbState K) {
    const int s = blockIdx.x;
    const int m = K.m[s];
    if (!gemm_applies(D, K, s, m)) return;
    __shared__ double G[KB_GTAB];
    for (int k = threadIdx.x; k < KB_GTAB; k += blockDim.x) G[k] = K.gtab[k];
    __syncthreads();
    const int capr = kb_capr(D.cap), mp = (m + 3) & ~3;
    const uint64_t* sh = shells_of(D, K, s);
    double* Q = K.workq + (size_t)s * 16 * capr * 16;
    const int nt = D.n_prbs / 16 + 1;  // tiles that cover candidates 0..n_prbs
    for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < nt * mp * 16; e += gridDim.y * blockDim.x) {
        const int nn = e & 15, j = (e >> 4) % mp, ct = (e >> 4) / mp;
        double v = 0.0;
        if (j < m) {
            const double* P = vec_page(K, sh, j >> 6);
            const int a = ((const int32_t*)(P + KB_ROW_IDX * KB_CH))[j & 63];
            int o = a - (16 * ct + nn);
            o = o < 0 ? -o : o;
            v = P[KB_ROW_CO * KB_CH + (j & 63)] * G[o < 255 ? o : 255];
        }
        Q[((size_t)ct * capr + j) * 16 + nn] = v;
    }
}

__global__ __launch_bounds__(256) void shared_fgemm_kernel(KbDev D, KbState K, const float* state) {
    const int s = blockIdx.x;
    const int m = K.m[s];
    if (!gemm_applies(D, K, s, m)) return;
    const int rt = blockIdx.y, ch = blockIdx.z / (KB_GEMM_KS / 4), kp = (blockIdx.z % (KB_GEMM_KS / 4)) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, kq = lane >> 4, li = lane & 15;
    const int d = D.dims[s] + 1, N = D.n_envs, capr = kb_capr(D.cap);
    const int nt = D.n_prbs / 16 + 1, ct0 = 8 * ch;
    const int ntc = nt - ct0 < 8 ? nt - ct0 : 8;
    if (ntc <= 0) return;
    const uint64_t* sh = shells_of(D, K, s);
    const double* Q = K.workq + (size_t)s * 16 * capr * 16;
    const int r = 16 * rt + li;
    double sx[KB_DMAX - 1];
#pragma unroll
    for (int q = 0; q < KB_DMAX - 1; ++q) sx[q] = (q < d - 1 && r < N) ? (double)state[(size_t)r * D.nv + D.off[s] + q] : 0.0;
    kb_f64x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (kb_f64x4){0.0, 0.0, 0.0, 0.0};
    double emax = 0.0;
    const int nk = (m + 3) >> 2, per = (nk + KB_GEMM_KS - 1) / KB_GEMM_KS;
    const int k_lo = kp * per, k_hi = (kp + 1) * per < nk ? (kp + 1) * per : nk;
    // two slabs of four landmarks per trip: all their loads (coordinates, Q tiles) are issued before the first use
    for (int ks = k_lo; ks < k_hi; ks += 2) {
        double E2[2], b2[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = 4 * (ks + h) + kq;
            const bool valid = ks + h < k_hi && j < m;
            const double* P = vec_page(K, sh, (valid ? j : 0) >> 6);
            const int l = (valid ? j : 0) & 63;
            double cv[KB_DMAX - 1];
#pragma unroll
            for (int q = 0; q < KB_DMAX - 1; ++q) cv[q] = q < d - 1 ? P[q * KB_CH + l] : 0.0;
#pragma unroll
            for (int t = 0; t < 8; ++t) b2[h][t] = (t < ntc && valid) ? Q[((size_t)(ct0 + t) * capr + j) * 16 + li] : 0.0;
            double d0 = 0.0;
#pragma unroll
            for (int q = 0; q < KB_DMAX - 1; ++q) {
                if (q < d - 1) {
                    const double u = cv[q] - sx[q];
                    d0 += u * u;
                }
            }
            E2[h] = valid ? rs_exp_nonpos(-D.gamma * d0) : 0.0;
            emax = E2[h] > emax ? E2[h] : emax;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t < ntc) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(E2[h], b2[h][t], acc[t], 0, 0, 0);
    }
    // the lane holds F[replica 16 rt + kq + 4 v][candidate 16 (ct0 + t) + li]
    double* F = K.workF + ((size_t)s * KB_GEMM_KS + kp) * N * 256;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (t < ntc) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int rr = 16 * rt + kq + 4 * v;
                if (rr < N) F[(size_t)rr * 256 + 16 * (ct0 + t) + li] = acc[t][v];
            }
        }
    }
    if (ch == 0) {  // the largest E of replica r over this part of the landmarks (lanes li, li + 16, li + 32, li + 48)
        double e2 = __shfl_xor(emax, 16);
        emax = e2 > emax ? e2 : emax;
        e2 = __shfl_xor(emax, 32);
        emax = e2 > emax ? e2 : emax;
        if (lane < 16 && r < N) K.workE[((size_t)s * KB_GEMM_KS + kp) * N + r] = emax;
    }
}

#ifndef KB_SCAN_OCC
#define KB_SCAN_OCC 3
#endif
__global__ __launch_bounds__(64, KB_SCAN_OCC) void shared_scan_kernel(ScanArgs A) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    __shared__ Lds sm;
    const int task = blockIdx.x, env = task / D.S, s = task - env * D.S;
    const int d = D.dims[s] + 1, n = D.n_prbs;
    const int dict = dict_of(D, task);
    const uint64_t* sh = shells_of(D, K, dict);
    const int m = K.m[dict];
    const int a_i = A.action[env * D.S + s];
    const int y = A.labels[env * D.S + s];
    uint64_t n_pred = 0, n_eval = 0;
    int c_from = A.round == 0 ? (y == 1 ? a_i : 0) : A.cursor[task];
    const int c_to = y == 1 ? n : a_i;
    if (A.round > 0 && !(c_from >= 0 && c_from <= c_to)) {  // nothing left to examine
        if (threadIdx.x == 0) {
            A.cstar[task] = -1;
            A.cursor[task] = -1;
        }
        return;
    }
    load_gtab(K, sm);
    stage_state(D, A.state, env, s, d, sm);
    __syncthreads();
    const Win w = window_of(y == 1 ? a_i : 0, c_to);
    double f[4];
    bool from_gemm = gemm_applies(D, K, dict, m);
    if (from_gemm) {
        const double* E = K.workE + (size_t)s * KB_GEMM_KS * D.n_envs + env;
        double emax = 0.0;
        for (int kp = 0; kp < KB_GEMM_KS; ++kp) emax = E[(size_t)kp * D.n_envs] > emax ? E[(size_t)kp * D.n_envs] : emax;
        from_gemm = emax >= 1e-250;
    }
    if (from_gemm) {
        const double* F = K.workF + (size_t)s * KB_GEMM_KS * D.n_envs * 256 + (size_t)env * 256;
        const size_t part = (size_t)D.n_envs * 256;
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int c = w.base + 64 * g + lane;
            c = c < n ? c : n;
            f[g] = g < w.ng ? ((F[c] + F[part + c]) + (F[2 * part + c] + F[3 * part + c])) +
                                  ((F[4 * part + c] + F[5 * part + c]) + (F[6 * part + c] + F[7 * part + c]))
                            : 0.0;
        }
    } else {
        score<4, 2>(D, K, sh, m, d, sm, w.base, w.ng, f);
    }
    if (A.round == 0) {
        // y_pred, accuracy table, security factor: kbrl_control.py:88-101 (as update_control_kernel)
        n_pred += 1;
        n_eval += (uint64_t)m;
        control_bookkeeping(D, K, task, env, s, m, f_of(f, w, a_i), y, A.hits, sm);
    }
    int cstar = -1;
    if (c_from >= 0 && c_from <= c_to) {
        n_eval += (uint64_t)(c_to - c_from + 1) * (uint64_t)m;
        int zeros;
        cstar = first_mistake(f, w, y, c_from, c_to, &zeros);
        const int last = cstar >= 0 ? cstar : c_to;
        n_pred += (uint64_t)(last - c_from + 1);
        if (m > 0 && zeros > 0 && threadIdx.x == 0) K.tie_ctr[task] += (uint32_t)zeros;  // Q11
    }
    if (threadIdx.x == 0) {
        A.cstar[task] = cstar;
        A.cursor[task] = cstar >= 0 ? cstar : -1;  // stays on the proposal until it is committed
        uint64_t* st = K.stats + (size_t)task * 4;
        st[0] += n_pred;
        st[3] += n_eval;
    }
}

// first `budget` proposers of each slice in replica order -> props[s][i][KB_PROP_W], counts[s] = all proposers
// Rank of every local proposer of slice s (replicas with cstar >= 0) in replica order, by all threads of the block:
// thread t owns a contiguous run of replicas, counts its proposers, the block scans the counts.  Calls
// emit(env, rank) for every proposer and returns the total.  (Launch with KB_RANK_THREADS threads.)
#define KB_RANK_THREADS 1024
template <class F>
__device__ __forceinline__ int ranked_proposers(const KbDev& D, const int32_t* cstar, int s, F emit) {
    __shared__ int sc[KB_RANK_THREADS];
    const int T = (int)blockDim.x, t = (int)threadIdx.x;
    const int per = (D.n_envs + T - 1) / T;
    const int e0 = t * per, e1 = e0 + per < D.n_envs ? e0 + per : D.n_envs;
    int c = 0;
    for (int env = e0; env < e1; ++env) c += cstar[env * D.S + s] >= 0 ? 1 : 0;
    sc[t] = c;
    __syncthreads();
    int incl = c;
    for (int d = 1; d < T; d <<= 1) {  // Hillis-Steele inclusive scan
        const int o = t >= d ? sc[t - d] : 0;
        __syncthreads();
        incl += o;
        sc[t] = incl;
        __syncthreads();
    }
    const int total = sc[T - 1];
    int pos = incl - c;
    for (int env = e0; env < e1; ++env)
        if (cstar[env * D.S + s] >= 0) emit(env, pos++);
    return total;
}

__device__ __forceinline__ void write_proposal(const KbDev& D, const float* state, const int32_t* labels, int s, int env,
                                               int c, double* p) {
    const int d = D.dims[s] + 1;
    p[0] = (double)(D.first_env + env);
    p[1] = (double)(c * 4 + (labels[env * D.S + s] == 1 ? 1 : 0));
    for (int q = 0; q < d - 1; ++q) p[2 + q] = (double)state[(size_t)env * D.nv + D.off[s] + q];
}

__global__ __launch_bounds__(KB_RANK_THREADS) void shared_collect_kernel(KbDev D, const float* state, const int32_t* labels,
                                                                         const int32_t* cstar, int budget, double* props,
                                                                         int32_t* counts) {
    const int s = blockIdx.x;
    const int total = ranked_proposers(D, cstar, s, [&](int env, int rank) {
        if (rank < budget)
            write_proposal(D, state, labels, s, env, cstar[env * D.S + s], props + ((size_t)s * budget + rank) * KB_PROP_W);
    });
    if (threadIdx.x == 0) counts[s] = total;
}

// ---- device-resident exchange (kb_shared_step): the proposal block a rank contributes to the all-gather is
//   block = [S doubles: proposers per slice] [S][budget][KB_PROP_W] proposals
__global__ __launch_bounds__(KB_RANK_THREADS) void shared_collect_block_kernel(KbDev D, const float* state,
                                                                               const int32_t* labels, const int32_t* cstar,
                                                                               int budget, double* block) {
    const int s = blockIdx.x;
    double* props = block + D.S;
    const int total = ranked_proposers(D, cstar, s, [&](int env, int rank) {
        if (rank < budget)
            write_proposal(D, state, labels, s, env, cstar[env * D.S + s], props + ((size_t)s * budget + rank) * KB_PROP_W);
    });
    if (threadIdx.x == 0) block[s] = (double)total;
}

// Merge the gathered blocks of all W ranks for slice s = blockIdx.x: ascending global replica id, the first `budget`
// (the rule of kbrl_dev.merge_proposals, evaluated where the data is).  taken[s] = how many of rank `me`'s proposers
// made it; total[0] += proposers of all ranks (0 ends the learning step); total[1] = 1 when a rank's block carries the
// failure mark (a negative count: kb_shared_step's abort path).
__global__ __launch_bounds__(256) void shared_merge_kernel(KbDev D, const double* gathered, int W, int me, int budget,
                                                           size_t blk_doubles, double* mprops, int32_t* mcounts,
                                                           int32_t* taken, int32_t* total) {
    const int s = blockIdx.x;
    __shared__ int n_of[64];   // candidates rank w contributes
    __shared__ int off_of[65];
    __shared__ int s_taken, s_all;
    if (threadIdx.x == 0) {
        int o = 0, all = 0, failed = 0;
        for (int w = 0; w < W; ++w) {
            int c = (int)gathered[(size_t)w * blk_doubles + s];
            if (c < 0) {  // a rank that could not complete its round says so in the place of its count: every rank sees it here
                failed = 1;
                c = 0;
            }
            all += c;
            n_of[w] = c < budget ? c : budget;
            off_of[w] = o;
            o += n_of[w];
        }
        off_of[W] = o;
        s_taken = 0;
        s_all = all;
        if (failed) atomicOr(&total[1], 1);
    }
    __syncthreads();
    const int n = off_of[W];
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        int w = 0;
        while (e >= off_of[w + 1]) ++w;
        const int i = e - off_of[w];
        const double* p = gathered + (size_t)w * blk_doubles + D.S + ((size_t)s * budget + i) * KB_PROP_W;
        const double id = p[0];
        int rank = 0;
        for (int w2 = 0; w2 < W; ++w2)
            for (int i2 = 0; i2 < n_of[w2]; ++i2) {
                const double id2 = gathered[(size_t)w2 * blk_doubles + D.S + ((size_t)s * budget + i2) * KB_PROP_W];
                rank += (id2 < id || (id2 == id && (w2 < w || (w2 == w && i2 < i)))) ? 1 : 0;
            }
        if (rank < budget) {
            double* q = mprops + ((size_t)s * budget + rank) * KB_PROP_W;
            for (int k = 0; k < KB_PROP_W; ++k) q[k] = p[k];
            if (w == me) atomicAdd(&s_taken, 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mcounts[s] = n < budget ? n : budget;
        taken[s] = s_taken;
        atomicAdd(total, s_all);
    }
}

// the first n_accept[s] local proposers of slice s (replica order) had their sample applied: move on
__global__ __launch_bounds__(KB_RANK_THREADS) void shared_commit_kernel(KbDev D, const int32_t* cstar, const int32_t* n_accept,
                                                                        int32_t* cursor) {
    const int s = blockIdx.x;
    const int left = n_accept[s];
    ranked_proposers(D, cstar, s, [&](int env, int rank) {
        if (rank < left) cursor[env * D.S + s] = cstar[env * D.S + s] + 1;
    });
}

// ---- A dictionary at capacity only projects: landmarks and Kinv are fixed for a whole proposal list, so everything
// that does not involve the coefficients is computed for ALL proposals at once by kernels as wide as the chip
// (shared_cols_kernel, shared_matvec_kernel: one workgroup per slice would leave 250 CUs idle), and only f = k . coeff
// and the coefficient update run one proposal after the other (shared_apply_kernel, one wave, no block barriers).
// Work area per slice: KF [budget][capr] kernel columns, DS [budget][capr] d* = Kinv k_f.
__device__ __forceinline__ bool batch_applies(const KbDev& D, int m) { return m >= D.cap && m >= 2 && !D.serial_apply; }

#define KB_COLS_BLOCKS 64
#define KB_MATVEC_BLOCKS 256

// kernel columns of all proposals of the full dictionaries (kernel_column_full's arithmetic)
__global__ __launch_bounds__(256) void shared_cols_kernel(KbDev D, KbState K, const double* props, const int32_t* counts,
                                                          int budget) {
    const int s = blockIdx.x;
    const int m = K.m[s];
    if (!batch_applies(D, m)) return;
    const int np = counts[s] < budget ? counts[s] : budget;
    const int d = D.dims[s] + 1, capr = kb_capr(D.cap);
    const uint64_t* sh = shells_of(D, K, s);
    const double* pr = props + (size_t)s * budget * KB_PROP_W;
    double* KF = K.workb + (size_t)s * 2 * budget * capr;
    for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < np * m; e += gridDim.y * blockDim.x) {
        const int p = e / m, j = e - p * m;
        const double* x = pr + (size_t)p * KB_PROP_W + 2;
        const double* P = vec_page(K, sh, j >> 6);
        const int l = j & 63;
        double d0 = 0.0;
        for (int q = 0; q < d - 1; ++q) {
            const double t = P[q * KB_CH + l] - x[q];
            d0 += t * t;
        }
        const int c = ((int)pr[(size_t)p * KB_PROP_W + 1]) >> 2;
        const double dl = P[(d - 1) * KB_CH + l] - (double)c / (double)D.n_prbs;
        KF[(size_t)p * capr + j] = rs_exp(-D.gamma * (d0 + dl * dl));
    }
}

// d* = Kinv k_f for all proposals of the full dictionaries, each output summed exactly as matvec_colsum does (column
// walk; eight partial sums over the row classes j mod 8, fused multiply-adds in increasing j; the same tree): a lane owns
// output i and walks the proposals four at a time
__global__ __launch_bounds__(256) void shared_matvec_kernel(KbDev D, KbState K, const int32_t* counts, int budget, int mfma) {
    const int s = blockIdx.x;
    const int m = K.m[s];
    if (!batch_applies(D, m) || (mfma && !(m & 63))) return;  // (mfma: shared_matvec_mfma_kernel takes the multiples of 64)
    const int np = counts[s] < budget ? counts[s] : budget;
    const int capr = kb_capr(D.cap);
    const uint64_t* sh = shells_of(D, K, s);
    const int lane = threadIdx.x & 63;
    const int nb = (m + 63) >> 6;
    const double* KF = K.workb + (size_t)s * 2 * budget * capr;
    double* DS = K.workb + (size_t)s * 2 * budget * capr + (size_t)budget * capr;
    const int nwork = nb * ((np + 3) >> 2);  // (column block, group of four proposals) per wave
    const int wave = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.y * (blockDim.x >> 6);
    for (int wk = wave; wk < nwork; wk += nw) {
        const int bi = wk % nb, p0 = (wk / nb) * 4;
        double a[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[u][r] = 0.0;
        for (int bj = 0; bj < nb; ++bj) {
            const int rows = m - 64 * bj < 64 ? m - 64 * bj : 64;
            const double* tp = kinv_tile(K, sh, bj, bi) + lane;
            double kfv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) kfv[r] = KF[(size_t)(p0 + r < np ? p0 + r : p0) * capr + 64 * bj + lane];
            for (int r0 = 0; r0 < rows; r0 += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {  // row r0 + u is of class u (64 bj is a multiple of eight)
                    const int rr = r0 + u;
                    if (rr < rows) {
                        const double kv = tp[rr * 64];
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[u][r] = __builtin_fma(kv, readlane_f64(kfv[r], rr), a[u][r]);
                    }
                }
            }
        }
        const int i = 64 * bi + lane;
        if (i < m) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (p0 + r < np)
                    DS[(size_t)(p0 + r) * capr + i] = ((a[0][r] + a[1][r]) + (a[2][r] + a[3][r])) + ((a[4][r] + a[5][r]) + (a[6][r] + a[7][r]));
        }
    }
}

// The same d* = Kinv k_f for all proposals of a full dictionary as ONE dense product on the matrix cores, DS = KF Kinv
// (n_p x m x m), and still every output bit for bit the column walk's sum.  v_mfma_f64_16x16x4 accumulates its four
// products as a chain of fused multiply-adds in k order onto the accumulator (tools/experiments/mfma_order.hip: 51,200 of
// 51,200 outputs equal fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c)))) bit for bit), which is how matvec_colsum forms
// ONE of its eight partial sums: class c = the rows j = c (mod 8), in increasing j.  So a wave keeps eight accumulator
// tiles, one per row class, feeds tile c the rows c + 8 (4 h + k) of every 64-row block (k = the instruction's
// contraction index, h = 0, 1: sixteen instructions per block of rows) and adds the eight tiles in the column walk's
// tree at the end.  A wave owns 16 proposals x 16 outputs; the four waves of a workgroup take four proposal tiles of the
// same output tile (they read the same rows of Kinv together).  Dictionaries whose size is a multiple of 64 only (no
// padded rows: a padding term fma(0, 0, acc) would turn an accumulator of -0.0 into +0.0).
__global__ __launch_bounds__(256) void shared_matvec_mfma_kernel(KbDev D, KbState K, const int32_t* counts, int budget) {
    const int s = blockIdx.x;
    const int m = K.m[s];
    if (!batch_applies(D, m) || (m & 63)) return;
    const int np = counts[s] < budget ? counts[s] : budget;
    const int capr = kb_capr(D.cap);
    const uint64_t* sh = shells_of(D, K, s);
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    const int nb = m >> 6;
    const double* KF = K.workb + (size_t)s * 2 * budget * capr;
    double* DS = K.workb + (size_t)s * 2 * budget * capr + (size_t)budget * capr;
    const int npt = (np + 15) >> 4, nct = m >> 4;
    const int nwork = npt * nct;  // (output tile, proposal tile), proposal tile fastest: a workgroup shares its rows of Kinv
    const int wave = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.y * (blockDim.x >> 6);
    for (int wk = wave; wk < nwork; wk += nw) {
        const int ct = wk / npt, pt = wk - ct * npt;
        const int p = 16 * pt + li;                 // A operand: proposal p, row class c, contraction slot kq
        const int bi = ct >> 2, ci = 16 * (ct & 3) + li;  // B operand: output 64 bi + ci
        const double* ar = KF + (size_t)(p < np ? p : 0) * capr + 8 * kq;
        kb_f64x4 acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = (kb_f64x4){0.0, 0.0, 0.0, 0.0};
        for (int bj = 0; bj < nb; ++bj) {
            const double* tp = kinv_tile(K, sh, bj, bi) + (size_t)(8 * kq) * 64 + ci;
            double a[2][8], b[2][8];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    // row 64 bj + c + 8 (4 h + kq) of the block
                    a[h][c] = p < np ? ar[64 * bj + 32 * h + c] : 0.0;
                    b[h][c] = tp[(size_t)(32 * h + c) * 64];
                }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[h][c], b[h][c], acc[c], 0, 0, 0);
        }
        // register r of the lane: proposal 16 pt + kq + 4 r, output 64 bi + ci
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int po = 16 * pt + kq + 4 * r;
            if (po < np)
                DS[(size_t)po * capr + 64 * bi + ci] =
                    ((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) + ((acc[4][r] + acc[5][r]) + (acc[6][r] + acc[7][r]));
        }
    }
}

// A full dictionary only projects: proposal p is still a mistake iff  f_p = k_p . coeff <= 0  for the coefficients as the
// EARLIER projections of the list left them,
//     coeff = coeff_0 + sum_{q < p, applied} y_q d*_q    =>    f_p = k_p . coeff_0 + sum_{q < p, applied} y_q (k_p . d*_q),
// so all the order touches is a matrix of scalars: the proposals' Gram block A = KF DS^T (n_p x n_p x m).
// shared_gram_kernel forms it as a dense contraction on the matrix cores -- v_mfma_f64_16x16x4, one 16 x 16 tile per wave,
// the lower block triangle only, as wide as the chip -- together with f_p^0 = k_p . coeff_0 (a wave per proposal);
// shared_apply_kernel then walks the list with one wave (a dot over the earlier proposals per step, rows of A prefetched)
// and gives the coefficients the applied d*_q in list order, element by element exactly as the one-by-one path adds them
// (same bits).
__global__ __launch_bounds__(256) void shared_gram_kernel(KbDev D, KbState K, const int32_t* counts, int budget) {
    const int s = blockIdx.x;
    const int m = K.m[s];
    if (!batch_applies(D, m)) return;
    const int np = counts[s] < budget ? counts[s] : budget;
    const int capr = kb_capr(D.cap);
    const int lane = threadIdx.x & 63, kq = lane >> 4, li = lane & 15;
    const double* KF = K.workb + (size_t)s * 2 * budget * capr;
    const double* DS = KF + (size_t)budget * capr;
    double* A = K.workg + (size_t)s * budget * budget;
    double* F0 = K.workf + (size_t)s * budget;
    const int nt = (np + 15) >> 4, ntile = nt * (nt + 1) / 2;
    const int wave = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.y * (blockDim.x >> 6);
    for (int wk = wave; wk < ntile + np; wk += nw) {
        if (wk >= ntile) {  // f_p^0 (the one-by-one path's own dot)
            const int p = wk - ntile;
            const uint64_t* sh = shells_of(D, K, s);
            double part[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                part[v] = 0.0;
                for (int j = 64 * v + lane; j < m; j += 256) part[v] += KF[(size_t)p * capr + j] * vec_page(K, sh, j >> 6)[KB_ROW_CO * KB_CH + lane];
            }
            for (int dd = 32; dd >= 1; dd >>= 1) {
#pragma unroll
                for (int v = 0; v < 4; ++v) part[v] += __shfl_xor(part[v], dd);
            }
            double t = 0.0;
#pragma unroll
            for (int v = 0; v < 4; ++v) t += __shfl(part[v], 0);
            if (lane == 0) F0[p] = t;
            continue;
        }
        // tile (ti, tj), tj <= ti, of A[p][q] = k_p . d*_q.  MFMA step u of a 16-landmark slab contracts landmarks
        // k0 + 4 kq + u (any pairing of k works as long as both operands use it): four consecutive doubles per lane and slab
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= wk) ++ti;
        const int tj = wk - ti * (ti + 1) / 2;
        const int pa = 16 * ti + li, qb = 16 * tj + li;
        const double* ar = KF + (size_t)(pa < np ? pa : 0) * capr;
        const double* br = DS + (size_t)(qb < np ? qb : 0) * capr;
        kb_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < m; k0 += 32) {
            double a[8], b[8];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + 16 * h + 4 * kq + u;
                    a[4 * h + u] = (pa < np && k < m) ? ar[k] : 0.0;
                    b[4 * h + u] = (qb < np && k < m) ? br[k] : 0.0;
                }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
        // the lane holds A[16 ti + kq + 4 r][16 tj + li]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = 16 * ti + kq + 4 * r, q = 16 * tj + li;
            if (p < np && q < np) A[(size_t)q * budget + p] = acc[r];  // stored by COLUMN: the walk reads column q when q is applied
        }
    }
}

// the ordered part, by the slice's own workgroup: co = the coefficient column, wl = scratch (4 x budget doubles), in LDS
__device__ uint64_t apply_full_batch(const KbDev& D, const KbState& K, int s, int m, const double* pr, int np, int budget,
                                     double* co, double* wl) {
    const int capr = kb_capr(D.cap);
    const uint64_t* sh = shells_of(D, K, s);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* DS = K.workb + (size_t)s * 2 * budget * capr + (size_t)budget * capr;
    const double* A = K.workg + (size_t)s * budget * budget;
    const double* F0 = K.workf + (size_t)s * budget;
    double* wq = wl;               // [budget] y_q of the applied proposals, 0 for the others
    double* lst = wl + budget;     // [budget] the applied proposals, compacted in list order
    double* cnt = wl + 2 * budget; // [1]
    for (int j = threadIdx.x; j < m; j += blockDim.x) co[j] = *vec_at(K, sh, KB_ROW_CO, j);
    for (int q = threadIdx.x; q < budget; q += blockDim.x) wq[q] = 0.0;
    __syncthreads();
    uint64_t n_mist = 0;
    if (wave == 0) {
        bool sat = false;
        int napp = 0;
        const int nq = (np + 63) >> 6;  // lanes hold proposals lane, lane + 64, ...
        // lane l keeps g_p = sum_{applied q < p} y_q A[p][q] for its proposals p = l, l + 64, ...: nothing is summed across
        // lanes; when proposal q is applied every lane adds its piece of A's column q (prefetched two steps ahead)
        double g[4] = {0.0, 0.0, 0.0, 0.0}, f0v[4];
        int yv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = lane + 64 * k;
            yv[k] = q < np ? ((((int)pr[(size_t)q * KB_PROP_W + 1]) & 1) ? 1 : -1) : 1;
            f0v[k] = q < np ? F0[q] : 0.0;
        }
        double nxt[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 4; ++k) nxt[h][k] = (h < np && k < nq && lane + 64 * k < np) ? A[(size_t)h * budget + lane + 64 * k] : 0.0;
        for (int p = 0; p < np; ++p) {
            double col[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                col[k] = nxt[p & 1][k];
                nxt[p & 1][k] = (p + 2 < np && k < nq && lane + 64 * k < np) ? A[(size_t)(p + 2) * budget + lane + 64 * k] : 0.0;
            }
            const int pk = p >> 6, pl = p & 63;
            const int y = __builtin_amdgcn_readlane(pk == 0 ? yv[0] : (pk == 1 ? yv[1] : (pk == 2 ? yv[2] : yv[3])), pl);
            const double f = readlane_f64(pk == 0 ? f0v[0] + g[0] : (pk == 1 ? f0v[1] + g[1] : (pk == 2 ? f0v[2] + g[2] : f0v[3] + g[3])), pl);
            if (f * (double)y <= 0.0) {
                n_mist += 1;
                // delta = 1 - k_p . d*_p is the Gram block's diagonal: a full dictionary met a sample it would have grown for
                const double dg = readlane_f64(pk == 0 ? col[0] : (pk == 1 ? col[1] : (pk == 2 ? col[2] : col[3])), pl);
                const double delta = 1.0 - dg;
                sat = sat || (delta > 0.0 ? delta : 0.0) > D.eta;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (lane + 64 * k > p) g[k] = __builtin_fma((double)y, col[k], g[k]);
                if (lane == 0) {
                    wq[p] = (double)y;
                    lst[napp] = (double)p;
                }
                napp += 1;
            }
        }
        if (lane == 0) cnt[0] = (double)napp;
        if (sat && lane == 0) atomicOr(&K.err[0], 8);
    }
    __syncthreads();
    // ---- coeff_j takes the applied d*_q in list order (the one-by-one path's own expression), four loads in flight
    const int napp = (int)cnt[0];
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        double c = co[j];
        for (int a0 = 0; a0 < napp; a0 += 4) {
            double v[4], yy[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = (int)lst[a0 + u < napp ? a0 + u : a0];
                v[u] = DS[(size_t)q * capr + j];
                yy[u] = wq[q];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (a0 + u < napp) c = c + yy[u] * v[u];
        }
        *vec_at(K, sh, KB_ROW_CO, j) = c;
    }
    return n_mist;
}

// apply a merged proposal list to the dictionary of slice s = blockIdx.x, in order.  Dynamic LDS: kb_apply_lds_doubles().
__global__ __launch_bounds__(1024) void shared_apply_kernel(KbDev D, KbState K, const double* props, const int32_t* counts,
                                                         int budget, uint64_t* gstats) {
    extern __shared__ double kb_dyn_lds[];
    __shared__ Lds sm;
    const int s = blockIdx.x;
    const int d = D.dims[s] + 1;
    const uint64_t* sh = shells_of(D, K, s);
    int m = K.m[s];
    const int np = counts[s] < budget ? counts[s] : budget;
    uint64_t n_mist = 0, n_grow = 0;
    const unsigned long long t_in = wall_clock64();  // (developer aid: per-slice time and applied samples, gstats[8..])
    if (np > 0 && batch_applies(D, m)) {
        // kernel columns and d* of the whole list are in the work area (shared_cols_kernel, shared_matvec_kernel)
        n_mist = apply_full_batch(D, K, s, m, props + (size_t)s * budget * KB_PROP_W, np, budget, kb_dyn_lds,
                                  kb_dyn_lds + kb_capr(D.cap));
        if (threadIdx.x == 0) {
            atomicAdd((unsigned long long*)&gstats[1], (unsigned long long)n_mist);
            atomicAdd((unsigned long long*)&gstats[8 + s], wall_clock64() - t_in);
            atomicAdd((unsigned long long*)&gstats[16 + s], (unsigned long long)n_mist);
            atomicAdd((unsigned long long*)&gstats[24 + s], (unsigned long long)np);
        }
        return;
    }
    // A dictionary that can still grow: the proposals are predicted against it TOGETHER (a wave per proposal, each sum in
    // the shape the one-by-one predict uses: kernel values in kernel_column_full's arithmetic, wave_dot256's partial
    // sums), the first one in list order that is still a mistake is applied (Projectron.update), and what follows it is
    // predicted again.  The list costs one parallel pass per APPLIED sample instead of one predict per proposal -- most
    // of a round's proposals stop being mistakes once the first few are learned -- and every number is the one the
    // one-by-one walk produces.
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63;
    double* fp = kb_dyn_lds;  // [budget] predictions of the remaining proposals
    int from = 0;
    while (from < np) {
        // The first remaining proposal, in order, that is still a mistake.  The remaining proposals are predicted in
        // chunks that double (one proposal per wave first): while a list is being learned its next mistake is usually among
        // the first few, and a pass over all of the rest for every applied sample was most of a long list's time.  The
        // predictions past the one that is applied are discarded either way (the dictionary changes under them).
        int i = 0x7fffffff;
        for (int lo = from, span = nw; lo < np && i == 0x7fffffff; lo += span, span *= 2) {
            const int hi = lo + span < np ? lo + span : np;
            __syncthreads();
            for (int ii = lo + wave; ii < hi; ii += nw) {
                const int i = ii;
                const double* p = props + ((size_t)s * budget + i) * KB_PROP_W;
                const double t = (double)(((int)p[1]) >> 2) / (double)D.n_prbs;
                double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    for (int j = 64 * v + lane; j < m; j += 256) {
                        const double* P = vec_page(K, sh, j >> 6);
                        double d0 = 0.0;
                        for (int q = 0; q < d - 1; ++q) {
                            const double u = P[q * KB_CH + lane] - p[2 + q];
                            d0 += u * u;
                        }
                        const double dl = P[(d - 1) * KB_CH + lane] - t;
                        double k = rs_exp(-D.gamma * (d0 + dl * dl));
                        if (m == 1) k = (double)(float)k;
                        part[v] += k * P[KB_ROW_CO * KB_CH + lane];
                    }
                }
                for (int dd = 32; dd >= 1; dd >>= 1) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) part[v] += __shfl_xor(part[v], dd);
                }
                double f = 0.0;
#pragma unroll
                for (int v = 0; v < 4; ++v) f += __shfl(part[v], 0);
                if (m == 1) {  // float32 while a single landmark is held (kernel.py:16, projectron.py:9)
                    const double* P = vec_page(K, sh, 0);
                    double d0 = 0.0;
                    for (int q = 0; q < d - 1; ++q) {
                        const double u = P[q * KB_CH] - p[2 + q];
                        d0 += u * u;
                    }
                    const double dl = P[(d - 1) * KB_CH] - t;
                    f = (double)(float)((float)rs_exp(-D.gamma * (d0 + dl * dl)) * (float)P[KB_ROW_CO * KB_CH]);
                }
                if (m == 0) f = 0.0;
                if (lane == 0) fp[i] = f;
                    }
            __syncthreads();
            if (threadIdx.x == 0) sm.ired[5] = 0x7fffffff;
            __syncthreads();
            for (int q = lo + (int)threadIdx.x; q < hi; q += blockDim.x) {
                const int y = (((int)props[((size_t)s * budget + q) * KB_PROP_W + 1]) & 1) ? 1 : -1;
                if (fp[q] * (double)y <= 0.0) atomicMin(&sm.ired[5], q);
            }
            __syncthreads();
            i = sm.ired[5];
            __syncthreads();
        }
        if (i == 0x7fffffff) break;
        const double* p = props + ((size_t)s * budget + i) * KB_PROP_W;
        const int packed = (int)p[1];
        const int c = packed >> 2, y = (packed & 1) ? 1 : -1;
        if ((int)threadIdx.x < d - 1) sm.x[threadIdx.x] = p[2 + threadIdx.x];
        __syncthreads();
        const double t = (double)c / (double)D.n_prbs;
        kernel_column_full(D, K, sh, m, d, sm.x, t);
        int branch;
        double delta;
        const int m_new = apply_update(D, K, s, 0, m, d, sm.x, t, c, y, sm, &branch, &delta);
        n_mist += 1;
        if (branch == 2 && m_new > m) n_grow += 1;
        m = m_new;
        from = i + 1;
    }
    if (threadIdx.x == 0) {
        K.m[s] = m;
        if (np > 0) K.kf_owner[s] = -1;
        atomicAdd((unsigned long long*)&gstats[1], (unsigned long long)n_mist);
        atomicAdd((unsigned long long*)&gstats[2], (unsigned long long)n_grow);
        atomicAdd((unsigned long long*)&gstats[8 + s], wall_clock64() - t_in);
        atomicAdd((unsigned long long*)&gstats[16 + s], (unsigned long long)n_mist);
        atomicAdd((unsigned long long*)&gstats[24 + s], (unsigned long long)np);
    }
}

// ---- single-call entry points behind Projectron.predict / update (drop-in API, N=1 plumbing)

struct OneArgs {
    KbDev D;
    KbState K;
    int task;
    int y;
    double x[KB_DMAX];
    double* out;  // [4]: y_pred, f, branch, delta
};

__global__ __launch_bounds__(256) void predict_one_kernel(OneArgs A) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    __shared__ Lds sm;
    const int task = A.task, env = task / D.S, s = task - env * D.S;
    const int dt = dict_of(D, task);
    const uint64_t* sh = shells_of(D, K, dt);
    const int d = D.dims[s] + 1;
    const int m = K.m[dt];
    double p = 0.0;
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        double* P = vec_page(K, sh, j >> 6);
        const int l = j & 63;
        double dist = 0.0;
        for (int q = 0; q < d; ++q) {
            double t = P[q * KB_CH + l] - A.x[q];
            dist += t * t;
        }
        double k = rs_exp(-D.gamma * dist);
        if (m == 1) k = (double)(float)k;
        P[KB_ROW_KF * KB_CH + l] = k;
        p += k * P[KB_ROW_CO * KB_CH + l];
    }
    double f = block_sum(p, sm);
    if (m == 1) f = (double)(float)((float)*vec_at(K, sh, KB_ROW_KF, 0) * (float)*vec_at(K, sh, KB_ROW_CO, 0));
    if (threadIdx.x == 0) {
        int y = 0;
        if (m > 0) {
            y = f > 0.0 ? 1 : (f < 0.0 ? -1 : 0);
            if (y == 0) y = tie_draw(K, task, env, s);
        } else {
            f = 0.0;
        }
        K.f_last[task] = f;
        K.m_last[task] = m;
        K.kf_owner[dt] = task;
        A.out[0] = (double)y;
        A.out[1] = f;
        K.stats[(size_t)task * 4 + 0] += 1;
    }
}

__global__ __launch_bounds__(256) void update_one_kernel(OneArgs A) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    __shared__ Lds sm;
    const int task = A.task, env = task / D.S, s = task - env * D.S;
    const int dt = dict_of(D, task);
    const int d = D.dims[s] + 1;
    int m = K.m[dt];
    // Q12: Projectron.update uses the (f, K_f) cached by the predict that immediately preceded it.  Learners bound
    // into a KBRL_Control share their dictionary with update_control / select_action, which may have grown it (or
    // reused its K_f row) in between; the reference would then multiply arrays of different lengths (numpy raises).
    if (K.m_last[task] != m || (m > 0 && K.kf_owner[dt] != task)) {
        if (threadIdx.x == 0) { A.out[2] = -1.0; A.out[3] = 0.0; }
        return;
    }
    const double f = K.f_last[task];
    if (!(f * (double)A.y <= 0.0)) {
        if (threadIdx.x == 0) { A.out[2] = 0.0; A.out[3] = 0.0; }
        return;
    }
    for (int q = threadIdx.x; q < d - 1; q += blockDim.x) sm.x[q] = A.x[q];
    __syncthreads();
    int branch;
    double delta;
    const int m_new = apply_update(D, K, dt, env, m, d, sm.x, A.x[d - 1], grid_index(A.x[d - 1], D.n_prbs), A.y, sm, &branch, &delta);
    if (threadIdx.x == 0) {
        K.m[dt] = m_new;
        K.kf_owner[dt] = -1;  // update_control reuses the row
        A.out[2] = (double)branch;
        A.out[3] = delta;
        K.stats[(size_t)task * 4 + 1] += 1;
        if (branch == 2 && m_new > m) K.stats[(size_t)task * 4 + 2] += 1;
    }
}

// dense copies of one dictionary for the host (kb_get_learner, kb_get_kernel_row)
__global__ __launch_bounds__(256) void gather_learner_kernel(KbDev D, KbState K, int dict, int d, int m, double* landmarks,
                                                             double* coeff, double* kinv, double* kf_row) {
    const uint64_t* sh = shells_of(D, K, dict);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    if (landmarks)
        for (int e = tid; e < m * d; e += nt) {
            const int j = e / d, q = e - j * d;
            landmarks[e] = *vec_at(K, sh, q, j);
        }
    if (coeff)
        for (int j = tid; j < m; j += nt) coeff[j] = *vec_at(K, sh, KB_ROW_CO, j);
    if (kf_row)
        for (int j = tid; j < m; j += nt) kf_row[j] = *vec_at(K, sh, KB_ROW_KF, j);
    if (kinv)
        for (size_t e = tid; e < (size_t)m * m; e += nt) {
            const int i = (int)(e / m), j = (int)(e - (size_t)i * m);
            // (triangle storage: the upper block triangle is the lower one transposed)
            kinv[e] = (D.tri && (j >> 6) > (i >> 6)) ? kinv_tile_lo(K, sh, j >> 6, i >> 6)[(j & 63) * 64 + (i & 63)]
                                                     : kinv_tile(K, sh, i >> 6, j >> 6)[(i & 63) * 64 + (j & 63)];
        }
}

// end of the update phase: the queues are emptied for the next step; how many large learners were queued goes to a word of
// host memory the launcher reads WITHOUT synchronising (it decides whether the next steps enqueue the repair rounds at all)
// (seen: the work queued for the large learners' repairs in this step, in tiles of Kinv per repair pass: what the host goes
// by when it decides whether the next steps enqueue the chip-wide rounds -- a handful of dictionaries of 300 landmarks
// are repaired by a workgroup each in less time than nine idle launches take, ONE of 1,500 is not)
__global__ void heavy_reset_kernel(KbState K, volatile int32_t* seen) {
    if (seen) *seen = K.heavy[3];
    if (seen) seen[1] = K.big[0] > K.big[1 + KB_BIG_MAX] ? K.big[0] : K.big[1 + KB_BIG_MAX];  // listed (large) learners
    K.heavy[0] = K.heavy[1] = K.heavy[2] = K.heavy[3] = 0;
}

__global__ void kb_gtab_kernel(KbDev D, KbState K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < KB_GTAB) {
        const double t = (double)k / (double)D.n_prbs;
        K.gtab[k] = rs_exp_nonpos(-D.gamma * (t * t));
    }
}

__global__ void kb_reset_kernel(KbDev D, KbState K, const int32_t* init_action, const int32_t* init_sec,
                                const uint64_t* seeds, int n_dict) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int T = D.n_envs * D.S;
    if (i < T) {  // per-learner state; the dictionaries (K.m: one per learner, or one per slice when shared) are
                  // cleared by kb_reset with their own count
        K.f_last[i] = 0.0;
        K.m_last[i] = 0;
        K.tie_ctr[i] = 0;
        K.action[i] = init_action[i];
        K.security[i] = init_sec[i];
        K.margins[i] = 0;
        for (int k = 0; k < 4; ++k) K.stats[(size_t)i * 4 + k] = 0;
        for (int c = 0; c < D.n_prbs; ++c) K.acc[(size_t)i * D.n_prbs + c] = (D.lo + D.hi) / 2;
    }
    if (i < n_dict) K.kf_owner[i] = -1;
    if (i < D.n_envs) {
        K.seeds[i] = seeds[i];
        K.adjusted[i] = 0;
        K.err[i] = 0;
    }
    if (i == 0) K.pool_top[0] = 64;  // offset 0 means "no shell"
}

}  // namespace kb

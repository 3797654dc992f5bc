// kb_kbrl.hip -- the KBRL agent's kernel machinery on gfx950, batched over env replicas.
//
// What it computes (reference call tree):
//   GaussianKernel.k / predict       algorithms/kernel.py:8-28
//   Projectron.predict / update      algorithms/projectron.py:32-60
//   KBRL_Control.select_action / adjust_action / update_control   kbrl_control.py:41-114
//
// One workgroup (4 waves) owns one learner = (replica, slice).  Both hot loops of the agent --
// the augmentation loop of update_control and the linear scan of select_action -- evaluate the
// classifier on candidates x_c = (state, c / n_prbs) that differ only in the last coordinate.
// For landmark l_j:  ||l_j - x_c||^2 = D0_j + (lam_j - t_c)^2 = [D0_j + lam_j^2, -2 lam_j, 1] . [1, t_c, t_c^2]
// so the [landmarks x candidates] distance block is a K=4 dense contraction: exactly one
// v_mfma_f64_16x16x4_f64 per 16x16 tile.  exp() of the tile and the contraction with the
// coefficient vector follow on the VALU (4 exps per lane per tile: this, not the MFMA, bounds
// the kernel; DESIGN.md gives the flop model).  The reference's sequentially dependent loop
// (predict, update, predict, ...) is reproduced exactly in order: score the whole remaining
// range, find the first mistake, apply that one Projectron update, rescore what follows (H4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rs_philox.h"

namespace kb {

#define KB_DMAX 16      // max len(x)
#define KB_CAND_MAX 272 // n_prbs + 1 rounded up to 16 (n_prbs <= 256)

typedef double kb_f64x4 __attribute__((ext_vector_type(4)));

struct KbDev {
    int32_t n_envs, S, n_prbs, cap, nv;
    int32_t dims[8];  // state variables per learner (len(x) = dims+1)
    int32_t off[8];   // first state variable of learner s
    double alfa, lo, hi, gamma, eta;
    int32_t shared;   // 1: one dictionary per slice shared by all replicas (build-defined extension)
    int32_t first_env; // global id of local replica 0 (shared mode proposals carry global ids)
    int32_t serial_apply; // shared mode: apply a full dictionary's proposals one by one as well (KBRL_SERIAL_APPLY, tests)
};

// dictionary a learner (task = env * S + s) reads and writes
__device__ __forceinline__ int dict_of(const KbDev& D, int task) { return D.shared ? task % D.S : task; }

struct KbState {
    int32_t* m;        // [T] landmarks per learner (T = n_envs * S)
    double* L;         // [T][KB_DMAX][cap] landmarks, coordinate-major
    double* coeff;     // [T][cap]
    double* Kinv;      // [T][cap][cap]
    double* kf;        // [T][cap]  K_f cached by the last predict (projectron.py:34)
    double* f_last;    // [T]
    int32_t* m_last;   // [T] dictionary size the cached (f_last, kf) was computed against (Q12 guard)
    uint32_t* tie_ctr; // [T] Philox counter of the tie-break stream (kernel.py:26-27)
    uint64_t* seeds;   // [n_envs]
    int32_t* action;   // [n_envs][S]
    int32_t* security; // [n_envs][S]
    int32_t* margins;  // [n_envs][S]
    int32_t* adjusted; // [n_envs]
    double* acc;       // [n_envs][S][n_prbs]
    int32_t* err;      // [n_envs]
    uint64_t* stats;   // [T][4]: predicts, mistakes, grows, kernel evaluations (candidates x landmarks)
    double* work;      // [blocks][2][cap rounded up to 16] per-block columns of an update in progress (kf, d*)
    double* workb;     // shared mode: [S][2][budget_cap][cap rounded up to 16] kernel columns and d* of a proposal list
};

__device__ __forceinline__ int tie_draw(const KbState& K, int task, int env, int s) {
    uint64_t seed = K.seeds[env];
    rs_stream st = {(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)s, 0xFFFFFFFFu, K.tie_ctr[task]};
    int v = rs_stream_pm1(&st);
    K.tie_ctr[task] = st.ctr;
    return v;
}

// Working set of one learner.  Three columns live in LDS (carved out of dynamic shared memory by capacity: 27 KB at
// capacity 1024, so five blocks stay resident per CU; the MFMA operands D0 + lam^2 and -2 lam are formed from them on
// the fly).  The two columns of an update in progress (kernel column, d* = Kinv k_f) are touched by the one learner
// in ten that updates in a step: they sit in a per-block row of global memory (KbState.work) behind the same
// pointers.
struct Lds {
    double* lam;  // lam_j (last coordinate of the landmark)
    double* d0;   // D0_j
    double* co;   // coeff_j
    double* kf;   // kernel column of the candidate being updated   (global, per block)
    double* ds;   // d* = Kinv k_f                                   (global, per block)
    double* f;    // [KB_CAND_MAX]
    double* x;    // [KB_DMAX]
    double* red;  // [16]
    int* ired;    // [8]
    int capr;     // column length (capacity rounded up to 16)
};

__host__ __device__ inline int kb_capr(int cap) { return (cap + 15) & ~15; }
__host__ __device__ inline size_t kb_lds_bytes(int cap) {
    return ((size_t)3 * kb_capr(cap) + KB_CAND_MAX + KB_DMAX + 16) * sizeof(double) + 8 * sizeof(int);
}

__device__ __forceinline__ Lds carve_lds(int cap, double* work) {
    extern __shared__ double kb_dyn_lds[];
    Lds sm;
    const int c = kb_capr(cap);
    double* p = kb_dyn_lds;
    sm.lam = p; p += c;
    sm.d0 = p; p += c;
    sm.co = p; p += c;
    sm.f = p; p += KB_CAND_MAX;
    sm.x = p; p += KB_DMAX;
    sm.red = p; p += 16;
    sm.ired = (int*)p;
    sm.kf = work + (size_t)blockIdx.x * 2 * c;
    sm.ds = sm.kf + c;
    sm.capr = c;
    return sm;
}

__device__ __forceinline__ double block_sum(double v, Lds& sm) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm.red[w];
    return t;
}

// D0_j, lam_j and the MFMA operands for the current state (first d-1 coordinates in sm.x)
__device__ void prepare_operands(const KbDev& D, const KbState& K, int dict, int m, int d, Lds& sm) {
    const int cap = D.cap;
    const double* L = K.L + (size_t)dict * KB_DMAX * cap;
    // entries beyond the dictionary must read as zero coefficients up to the end of the last 16-landmark tile
    // (apply_update clears a new tile when the dictionary grows into it)
    int lim = (m + 16) & ~15;
    lim = lim < sm.capr ? lim : sm.capr;
    for (int j = threadIdx.x; j < lim; j += blockDim.x) {
        double d0 = 0.0, lam = 0.0, co = 0.0;
        if (j < m) {
            for (int q = 0; q < d - 1; ++q) {
                double t = L[(size_t)q * cap + j] - sm.x[q];
                d0 += t * t;
            }
            lam = L[(size_t)(d - 1) * cap + j];
            co = K.coeff[(size_t)dict * cap + j];
        }
        sm.d0[j] = d0;
        sm.lam[j] = lam;
        sm.co[j] = co;
    }
    __syncthreads();
}

// f(c) for c in [c_lo, c_hi] -> sm.f[c]; one wave per 16-candidate strip, MFMA distance tiles
__device__ void score_range(const KbDev& D, int m, int c_lo, int c_hi, Lds& sm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const double inv_n = 1.0;  // t_c = c / n_prbs computed below with a true division
    (void)inv_n;
    if (m == 0) {
        for (int c = c_lo + (int)threadIdx.x; c <= c_hi; c += blockDim.x) sm.f[c] = 0.0;
        __syncthreads();
        return;
    }
    if (m == 1) {
        // single landmark: numpy keeps k and coeff in float32 (kernel.py:16, projectron.py:9)
        for (int c = c_lo + (int)threadIdx.x; c <= c_hi; c += blockDim.x) {
            double t = (double)c / (double)D.n_prbs;
            double dl = sm.lam[0] - t;
            double k = rs_exp(-D.gamma * (sm.d0[0] + dl * dl));
            sm.f[c] = (double)(float)((float)k * (float)sm.co[0]);
        }
        __syncthreads();
        return;
    }
    const int n_strips = (c_hi - c_lo) / 16 + 1;
    const int mt = (m + 15) / 16;
    const int kq = lane >> 4, li = lane & 15;
    for (int strip = wave; strip < n_strips; strip += nw) {
        const int c = c_lo + strip * 16 + li;
        const double t = (double)c / (double)D.n_prbs;
        // B[k][cand]: (1, t, t^2, 0)
        const double bval = kq == 0 ? 1.0 : (kq == 1 ? t : (kq == 2 ? t * t : 0.0));
        double part = 0.0;
        for (int jt = 0; jt < mt; ++jt) {
            const int j = jt * 16 + li;
            // A[landmark][k]: (D0 + lam^2, -2 lam, 1, 0)
            const double lam_j = sm.lam[j];
            const double aval = kq == 0 ? sm.d0[j] + lam_j * lam_j : (kq == 1 ? -2.0 * lam_j : (kq == 2 ? 1.0 : 0.0));
            kb_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aval, bval, acc, 0, 0, 0);
            // lane holds dist[landmark = kq + 4 r][cand = li]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jj = jt * 16 + kq + 4 * r;
                double dist = acc[r] > 0.0 ? acc[r] : 0.0;
                part += rs_exp_nonpos(-D.gamma * dist) * sm.co[jj];  // dist >= 0: branch-free, the 4 chains interleave
            }
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (kq == 0 && c <= c_hi) sm.f[c] = part;
    }
    __syncthreads();
}

// kernel column of candidate c against the dictionary -> sm.kf (exact (lam - t)^2 form)
__device__ void kernel_column(const KbDev& D, int m, int c, Lds& sm) {
    const double t = (double)c / (double)D.n_prbs;
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        double dl = sm.lam[j] - t;
        double k = rs_exp(-D.gamma * (sm.d0[j] + dl * dl));
        sm.kf[j] = m == 1 ? (double)(float)k : k;
    }
    __syncthreads();
}

// Projectron.update (projectron.py:39-60) for x = (sm.x[0..d-2], t_last), given its kernel column in sm.kf.
// Returns the new m.  branch: 1 = projection onto the dictionary, 2 = dictionary grew.
// R: rows of Kinv per pass of the mat-vec (their loads stay in flight together; the 1024-thread shared apply uses 8,
// the per-replica kernels 1 -- more would cost them registers they need for residency)
template <int R = 1>
__device__ int apply_update(const KbDev& D, const KbState& K, int dict, int err_env, int m, int d, double t_last, int y,
                            Lds& sm, int* branch, double* delta_out, bool* saturated = nullptr) {
    const int cap = D.cap;
    double* Kinv = K.Kinv + (size_t)dict * cap * cap;
    double* coeffg = K.coeff + (size_t)dict * cap;
    double dot;
    if (m <= 1) {
        float kinv = m == 0 ? 0.0f : 1.0f;
        float ds = kinv * (float)(m == 0 ? 0.0 : sm.kf[0]);
        if (threadIdx.x == 0) sm.ds[0] = (double)ds;
        dot = (double)(float)(ds * (float)(m == 0 ? 0.0 : sm.kf[0]));
        __syncthreads();
    } else {
        // d* = Kinv k_f: one row per wave-strided lane group (rows are contiguous -> coalesced)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
        // (eight rows per pass: their loads are independent and stay in flight together; each row's sum is formed
        // exactly as before -- lane-strided partial sums in increasing j, then the xor butterfly)
        for (int i0 = wave; i0 < m; i0 += nw * R) {
            double a[R];
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = 0.0;
            for (int j = lane; j < m; j += 64) {
                const double kfj = sm.kf[j];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = i0 + r * nw;
                    const double kv = Kinv[(size_t)(i < m ? i : i0) * cap + j];
                    if (i < m) a[r] += kv * kfj;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double v = a[r];
                for (int dd = 32; dd >= 1; dd >>= 1) v += __shfl_xor(v, dd);
                const int i = i0 + r * nw;
                if (lane == 0 && i < m) sm.ds[i] = v;
            }
        }
        __syncthreads();
        // 256 strided partial sums whatever the block size (the shared-dictionary kernel runs 1024 threads and must
        // produce the bits of the 256-thread per-replica kernels); waves beyond the fourth add exact zeros
        double p = 0.0;
        if (threadIdx.x < 256)
            for (int j = threadIdx.x; j < m; j += 256) p += sm.ds[j] * sm.kf[j];
        dot = block_sum(p, sm);
    }
    double delta = 1.0 - dot;  // Kii = k(x, x) = 1
    delta = delta > 0.0 ? delta : 0.0;
    *delta_out = delta;
    // A dictionary that has reached its capacity projects every further sample onto its span (the fixed-budget
    // reading of Projectron) instead of growing as the reference's unbounded SVvariable would: learning goes on,
    // nothing is dropped, and the replica is flagged (err bit 8, "saturated": reported by kb_get_sizes == capacity
    // and by the host classes as a warning, not as an error).
    const bool full = m >= cap;
    if (delta > D.eta && full && threadIdx.x == 0) atomicOr(&K.err[err_env], 8);
    if (saturated) *saturated = delta > D.eta && full;
    if (delta <= D.eta || full) {
        *branch = 1;
        for (int j = threadIdx.x; j < m; j += blockDim.x) {
            double nc = sm.co[j] + (double)y * sm.ds[j];
            if (m == 1) nc = (double)(float)nc;
            sm.co[j] = nc;
            coeffg[j] = nc;
        }
        __syncthreads();
        return m;
    }
    *branch = 2;
    // SVvariable.extend / insert; Kinv <- [[Kinv,0],[0,0]] + outer([d*,-1],[d*,-1]) / delta
    double* L = K.L + (size_t)dict * KB_DMAX * cap;
    const double t = t_last;
    if ((m & 15) == 15 && threadIdx.x < 16 && m + 1 + (int)threadIdx.x < sm.capr) {
        // the landmark after this one opens a new 16-landmark tile: make it read as empty
        const int j = m + 1 + threadIdx.x;
        sm.d0[j] = 0.0; sm.lam[j] = 0.0; sm.co[j] = 0.0;
    }
    if (threadIdx.x == 0) {
        for (int q = 0; q < d - 1; ++q) L[(size_t)q * cap + m] = sm.x[q];
        L[(size_t)(d - 1) * cap + m] = t;
        coeffg[m] = (double)y;
        sm.co[m] = (double)y;
        sm.d0[m] = 0.0;  // the new landmark shares the current state
        sm.lam[m] = t;
        sm.ds[m] = -1.0;
    }
    __syncthreads();
    if (m == 0) {
        if (threadIdx.x == 0) Kinv[0] = 1.0;
    } else {
        const int m1 = m + 1;
        for (int e = threadIdx.x; e < m1 * m1; e += blockDim.x) {
            const int i = e / m1, j = e - i * m1;
            double old = (i < m && j < m) ? Kinv[(size_t)i * cap + j] : 0.0;
            Kinv[(size_t)i * cap + j] = old + (sm.ds[i] * sm.ds[j]) / delta;
        }
    }
    __syncthreads();
    return m + 1;
}

struct CtlArgs {
    KbDev D;
    KbState K;
    const float* state;     // [n_envs][nv] state the action was taken in
    const int32_t* action;  // [n_envs][S]
    const int32_t* labels;  // [n_envs][S]
    int32_t* hits;          // [n_envs][S]
};

// KBRL_Control.update_control for one learner (kbrl_control.py:83-112)
__global__ __launch_bounds__(256, 5) void update_control_kernel(CtlArgs A) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    Lds sm = carve_lds(D.cap, K.work);
    const int task = blockIdx.x, env = task / D.S, s = task - env * D.S;
    const int d = D.dims[s] + 1, n = D.n_prbs;
    const int dict = dict_of(D, task);
    int m = K.m[dict];
    if (threadIdx.x < d - 1) sm.x[threadIdx.x] = (double)A.state[(size_t)env * D.nv + D.off[s] + threadIdx.x];
    __syncthreads();
    prepare_operands(D, K, dict, m, d, sm);
    const int a_i = A.action[env * D.S + s];
    const int y = A.labels[env * D.S + s];
    uint64_t n_pred = 0, n_mist = 0, n_grow = 0, n_eval = 0;

    // ---- y_pred = predict((state, a_i / n)) (kbrl_control.py:88-89)
    score_range(D, m, a_i, a_i, sm);
    n_pred += 1;
    n_eval += (uint64_t)m;
    int y_pred = 0;
    {
        double f0 = sm.f[a_i];
        if (m > 0) {
            y_pred = f0 > 0.0 ? 1 : (f0 < 0.0 ? -1 : 0);
            if (y_pred == 0) {
                if (threadIdx.x == 0) sm.ired[0] = tie_draw(K, task, env, s);
                __syncthreads();
                y_pred = sm.ired[0];
                __syncthreads();
            }
        }
    }
    const int hit = y == y_pred;
    // ---- accuracy table and security factor (kbrl_control.py:92-99)
    {
        int margin = K.margins[env * D.S + s];
        margin = margin > 0 ? margin : 0;
        double* acc = K.acc + ((size_t)env * D.S + s) * n;
        if (y_pred == 1) {
            for (int c = threadIdx.x; c < n; c += blockDim.x) {
                if (!hit) {
                    if (c < margin + 1) acc[c] = (1 - D.alfa) * acc[c];
                } else {
                    if (c >= margin) acc[c] = (1 - D.alfa) * acc[c] + D.alfa;
                }
            }
        }
        __syncthreads();
        if (!K.adjusted[env]) {
            if (threadIdx.x == 0) sm.ired[1] = 0x7fffffff;
            __syncthreads();
            int first = 0x7fffffff;
            for (int c = threadIdx.x; c < n; c += blockDim.x)
                if (acc[c] > D.lo) { first = c; break; }
            if (first != 0x7fffffff) atomicMin(&sm.ired[1], first);
            __syncthreads();
            if (threadIdx.x == 0) K.security[env * D.S + s] = sm.ired[1] == 0x7fffffff ? 0 : sm.ired[1];
        }
        if (threadIdx.x == 0) A.hits[env * D.S + s] = hit;
    }
    // ---- sample augmentation (kbrl_control.py:102-112), in the reference's order
    int c_from = y == 1 ? a_i : 0;
    const int c_to = y == 1 ? n : a_i;
    bool rescore = true;
    while (c_from <= c_to) {
        if (rescore) {
            score_range(D, m, c_from, c_to, sm);
            n_eval += (uint64_t)(c_to - c_from + 1) * (uint64_t)m;
        }
        // first candidate (in order) with f * y <= 0
        if (threadIdx.x == 0) sm.ired[2] = 0x7fffffff;
        __syncthreads();
        int firstc = 0x7fffffff;
        for (int c = c_from + (int)threadIdx.x; c <= c_to; c += blockDim.x)
            if (sm.f[c] * (double)y <= 0.0) { firstc = c; break; }
        if (firstc != 0x7fffffff) atomicMin(&sm.ired[2], firstc);
        __syncthreads();
        const int cstar = sm.ired[2];
        __syncthreads();
        const int last = cstar == 0x7fffffff ? c_to : cstar;
        n_pred += (uint64_t)(last - c_from + 1);
        // the predictions made on the way each consume a tie-break draw when f == 0 (Q11)
        if (m > 0) {
            int zeros = 0;
            for (int c = c_from + (int)threadIdx.x; c <= last; c += blockDim.x) zeros += sm.f[c] == 0.0 ? 1 : 0;
            int tz = (int)block_sum((double)zeros, sm);
            if (threadIdx.x == 0 && tz > 0) K.tie_ctr[task] += (uint32_t)tz;
        }
        if (cstar == 0x7fffffff) break;
        n_mist += 1;
        kernel_column(D, m, cstar, sm);
        int branch;
        double delta;
        bool saturated;
        const int m_new = apply_update(D, K, dict, env, m, d, (double)cstar / (double)n, y, sm, &branch, &delta, &saturated);
        // A FULL dictionary that met a sample it would have added cannot represent this region: the remaining
        // candidates would meet the same wall one O(m^2) projection at a time, so the augmentation of this learner
        // stops for this step (build-defined; the reference's dictionary is unbounded; the oracle does the same)
        if (saturated) break;
        c_from = cstar + 1;
        rescore = true;
        if (branch == 2 && m_new > m) {
            n_grow += 1;
            // The dictionary grew by the landmark (state, c*/n) with coefficient y and nothing else changed, so
            // the scores of the remaining candidates move by one kernel value each (same state: only the last
            // coordinate differs) -- instead of a full rescoring.  The float32 regime of a
            // single-landmark dictionary (kernel.py:16) keeps the full pass.
            if (m >= 2) {
                const double ts = (double)cstar / (double)n;
                for (int c = c_from + (int)threadIdx.x; c <= c_to; c += blockDim.x) {
                    const double dl = ts - (double)c / (double)n;
                    sm.f[c] += (double)y * rs_exp(-D.gamma * (dl * dl));
                }
                n_eval += (uint64_t)(c_to - c_from + 1 > 0 ? c_to - c_from + 1 : 0);
                __syncthreads();
                rescore = false;
            }
        }
        m = m_new;
    }
    if (threadIdx.x == 0) {
        K.m[dict] = m;
        uint64_t* st = K.stats + (size_t)task * 4;
        st[0] += n_pred;
        st[1] += n_mist;
        st[2] += n_grow;
        st[3] += n_eval;
    }
}

struct SelArgs {
    KbDev D;
    KbState K;
    const float* state;  // [n_envs][nv] new state
};

// per-learner part of KBRL_Control.select_action (kbrl_control.py:44-63): the smallest candidate the classifier
// accepts.  The reference scans c = 0, 1, 2, ... and stops at the first +1 (about 15 candidates in); so does this:
// batches of one 16-candidate strip per wave, in order, until a batch contains the answer.
__global__ __launch_bounds__(256) void select_kernel(SelArgs A) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    Lds sm = carve_lds(D.cap, K.work);
    const int task = blockIdx.x, env = task / D.S, s = task - env * D.S;
    const int d = D.dims[s] + 1, n = D.n_prbs;
    const int dict = dict_of(D, task);
    const int m = K.m[dict];
    if (threadIdx.x < d - 1) sm.x[threadIdx.x] = (double)A.state[(size_t)env * D.nv + D.off[s] + threadIdx.x];
    __syncthreads();
    int found = -1;
    uint64_t n_scored = 0;
    if (m > 0) {
        prepare_operands(D, K, dict, m, d, sm);
        const int batch = 16 * (int)(blockDim.x >> 6);
        for (int c0 = 0; c0 <= n && found < 0; c0 += batch) {
            const int c1 = c0 + batch - 1 < n ? c0 + batch - 1 : n;
            score_range(D, m, c0, c1, sm);
            n_scored += (uint64_t)(c1 - c0 + 1);
            if (threadIdx.x == 0) sm.ired[0] = 0x7fffffff;
            __syncthreads();
            const int c = c0 + (int)threadIdx.x;
            if (c <= c1 && sm.f[c] >= 0.0) atomicMin(&sm.ired[0], c);  // positive, or a tie to be drawn
            __syncthreads();
            if (threadIdx.x == 0) {
                // walk the (rare) exact ties of this batch in order; each consumes one draw (kernel.py:26-27)
                int cc = sm.ired[0], hit = -1;
                while (cc <= c1) {
                    if (sm.f[cc] > 0.0) { hit = cc; break; }
                    if (sm.f[cc] == 0.0 && tie_draw(K, task, env, s) == 1) { hit = cc; break; }
                    int nx = 0x7fffffff;
                    for (int q = cc + 1; q <= c1; ++q)
                        if (sm.f[q] >= 0.0) { nx = q; break; }
                    cc = nx;
                }
                sm.ired[1] = hit;
            }
            __syncthreads();
            found = sm.ired[1];
            __syncthreads();
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

// cross-learner part of select_action + adjust_action (kbrl_control.py:65-78)
__global__ void adjust_kernel(KbDev D, KbState K, int32_t* action_out) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
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

__global__ __launch_bounds__(256) void shared_scan_kernel(ScanArgs A) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    Lds sm = carve_lds(D.cap, K.work);
    const int task = blockIdx.x, env = task / D.S, s = task - env * D.S;
    const int d = D.dims[s] + 1, n = D.n_prbs;
    const int dict = dict_of(D, task);
    const int m = K.m[dict];
    if (threadIdx.x < d - 1) sm.x[threadIdx.x] = (double)A.state[(size_t)env * D.nv + D.off[s] + threadIdx.x];
    __syncthreads();
    prepare_operands(D, K, dict, m, d, sm);
    const int a_i = A.action[env * D.S + s];
    const int y = A.labels[env * D.S + s];
    uint64_t n_pred = 0, n_eval = 0;
    int c_from;
    if (A.round == 0) {
        // y_pred, accuracy table, security factor: kbrl_control.py:88-101 (as update_control_kernel)
        score_range(D, m, a_i, a_i, sm);
        n_pred += 1;
        n_eval += (uint64_t)m;
        int y_pred = 0;
        double f0 = sm.f[a_i];
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
        if (y_pred == 1) {
            for (int c = threadIdx.x; c < n; c += blockDim.x) {
                if (!hit) {
                    if (c < margin + 1) acc[c] = (1 - D.alfa) * acc[c];
                } else {
                    if (c >= margin) acc[c] = (1 - D.alfa) * acc[c] + D.alfa;
                }
            }
        }
        __syncthreads();
        if (!K.adjusted[env]) {
            if (threadIdx.x == 0) sm.ired[1] = 0x7fffffff;
            __syncthreads();
            int first = 0x7fffffff;
            for (int c = threadIdx.x; c < n; c += blockDim.x)
                if (acc[c] > D.lo) { first = c; break; }
            if (first != 0x7fffffff) atomicMin(&sm.ired[1], first);
            __syncthreads();
            if (threadIdx.x == 0) K.security[env * D.S + s] = sm.ired[1] == 0x7fffffff ? 0 : sm.ired[1];
        }
        if (threadIdx.x == 0) A.hits[env * D.S + s] = hit;
        c_from = y == 1 ? a_i : 0;
    } else {
        c_from = A.cursor[task];
    }
    const int c_to = y == 1 ? n : a_i;
    int cstar = -1;
    if (c_from >= 0 && c_from <= c_to) {
        score_range(D, m, c_from, c_to, sm);
        n_eval += (uint64_t)(c_to - c_from + 1) * (uint64_t)m;
        if (threadIdx.x == 0) sm.ired[2] = 0x7fffffff;
        __syncthreads();
        int firstc = 0x7fffffff;
        for (int c = c_from + (int)threadIdx.x; c <= c_to; c += blockDim.x)
            if (sm.f[c] * (double)y <= 0.0) { firstc = c; break; }
        if (firstc != 0x7fffffff) atomicMin(&sm.ired[2], firstc);
        __syncthreads();
        if (sm.ired[2] != 0x7fffffff) cstar = sm.ired[2];
        const int last = cstar >= 0 ? cstar : c_to;
        n_pred += (uint64_t)(last - c_from + 1);
        if (m > 0) {  // predictions with f == 0 consume a tie-break draw each (Q11)
            int zeros = 0;
            for (int c = c_from + (int)threadIdx.x; c <= last; c += blockDim.x) zeros += sm.f[c] == 0.0 ? 1 : 0;
            int tz = (int)block_sum((double)zeros, sm);
            if (threadIdx.x == 0 && tz > 0) K.tie_ctr[task] += (uint32_t)tz;
        }
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
// made it; total[0] += proposers of all ranks (0 ends the learning step).
__global__ __launch_bounds__(256) void shared_merge_kernel(KbDev D, const double* gathered, int W, int me, int budget,
                                                           size_t blk_doubles, double* mprops, int32_t* mcounts,
                                                           int32_t* taken, int32_t* total) {
    const int s = blockIdx.x;
    __shared__ int n_of[64];   // candidates rank w contributes
    __shared__ int off_of[65];
    __shared__ int s_taken, s_all;
    if (threadIdx.x == 0) {
        int o = 0, all = 0;
        for (int w = 0; w < W; ++w) {
            const int c = (int)gathered[(size_t)w * blk_doubles + s];
            all += c;
            n_of[w] = c < budget ? c : budget;
            off_of[w] = o;
            o += n_of[w];
        }
        off_of[W] = o;
        s_taken = 0;
        s_all = all;
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

// sum of the 256 strided partial products Σ_{j = t, t + 256, ...} a[j] b[j] (t < 256) exactly as block_sum forms it in the
// 256-thread kernels -- per 64-lane wave the xor butterfly (lane 0's value), then the four wave totals in order -- but
// by ONE wave, without block barriers.  All lanes return the value.
__device__ __forceinline__ double wave_dot256(const double* a, const double* b, int m) {
    const int lane = threadIdx.x & 63;
    double part[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        part[v] = 0.0;
        for (int j = 64 * v + lane; j < m; j += 256) part[v] += a[j] * b[j];
    }
    for (int dd = 32; dd >= 1; dd >>= 1) {  // four independent butterflies, stepped together
#pragma unroll
        for (int v = 0; v < 4; ++v) part[v] += __shfl_xor(part[v], dd);
    }
    double t = 0.0;
#pragma unroll
    for (int v = 0; v < 4; ++v) t += __shfl(part[v], 0);
    return t;  // (the 1024-thread kernels add exact zeros for their waves 4..15)
}

// ---- A dictionary at capacity only projects: landmarks and Kinv are fixed for a whole proposal list, so everything
// that does not involve the coefficients is computed for ALL proposals at once by kernels as wide as the chip
// (shared_cols_kernel, shared_matvec_kernel: one workgroup per slice would leave 250 CUs idle and walk 256 dependent L2
// round trips per wave), and only f = k . coeff and the coefficient update run one proposal after the other
// (shared_apply_kernel, one wave, no block barriers).  Work area per slice: KF [budget][capr] kernel columns,
// DS [budget][capr] d* = Kinv k_f.
__device__ __forceinline__ bool batch_applies(const KbDev& D, int m) { return m >= D.cap && m >= 2 && !D.serial_apply; }

#define KB_COLS_BLOCKS 16
#define KB_MATVEC_BLOCKS 32

// kernel columns of all proposals of the full dictionaries (prepare_operands' D0 and kernel_column's arithmetic)
__global__ __launch_bounds__(256) void shared_cols_kernel(KbDev D, KbState K, const double* props, const int32_t* counts,
                                                          int budget) {
    const int s = blockIdx.x;
    const int m = K.m[s];
    if (!batch_applies(D, m)) return;
    const int np = counts[s] < budget ? counts[s] : budget;
    const int d = D.dims[s] + 1, cap = D.cap, capr = kb_capr(cap);
    const double* pr = props + (size_t)s * budget * KB_PROP_W;
    const double* L = K.L + (size_t)s * KB_DMAX * cap;
    double* KF = K.workb + (size_t)s * 2 * budget * capr;
    for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < np * m; e += gridDim.y * blockDim.x) {
        const int p = e / m, j = e - p * m;
        const double* x = pr + (size_t)p * KB_PROP_W + 2;
        double d0 = 0.0;
        for (int q = 0; q < d - 1; ++q) {
            const double t = L[(size_t)q * cap + j] - x[q];
            d0 += t * t;
        }
        const int c = ((int)pr[(size_t)p * KB_PROP_W + 1]) >> 2;
        const double dl = L[(size_t)(d - 1) * cap + j] - (double)c / (double)D.n_prbs;
        KF[(size_t)p * capr + j] = rs_exp(-D.gamma * (d0 + dl * dl));
    }
}

// d* = Kinv k_f for all proposals of the full dictionaries: a wave takes rows of Kinv and walks the proposals four at a
// time (each row sum as in apply_update: lane-strided partial sums in increasing j, then the xor butterfly, lane 0)
__global__ __launch_bounds__(256) void shared_matvec_kernel(KbDev D, KbState K, const int32_t* counts, int budget) {
    const int s = blockIdx.x;
    const int m = K.m[s];
    if (!batch_applies(D, m)) return;
    const int np = counts[s] < budget ? counts[s] : budget;
    const int cap = D.cap, capr = kb_capr(cap);
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.y * (blockDim.x >> 6);
    const double* Kinv = K.Kinv + (size_t)s * cap * cap;
    const double* KF = K.workb + (size_t)s * 2 * budget * capr;
    double* DS = K.workb + (size_t)s * 2 * budget * capr + (size_t)budget * capr;
    for (int i = wave; i < m; i += nw) {
        const double* row = Kinv + (size_t)i * cap;
        for (int p0 = 0; p0 < np; p0 += 4) {
            double a[4] = {0.0, 0.0, 0.0, 0.0};
            for (int j = lane; j < m; j += 64) {
                const double kv = row[j];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double kf = KF[(size_t)(p0 + r < np ? p0 + r : p0) * capr + j];
                    a[r] += kv * kf;
                }
            }
            for (int dd = 32; dd >= 1; dd >>= 1) {  // the four butterflies step together
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] += __shfl_xor(a[r], dd);
            }
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (p0 + r < np) DS[(size_t)(p0 + r) * capr + i] = a[r];
            }
        }
    }
}

// the ordered part, by the slice's own workgroup (shared_apply_kernel): delta per proposal, then predict + projection in
// list order.  Returns the number of mistakes.
__device__ uint64_t apply_full_batch(const KbDev& D, const KbState& K, int s, int m, const double* pr, int np, int budget,
                                     Lds& sm) {
    const int cap = D.cap, capr = sm.capr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    double* coeffg = K.coeff + (size_t)s * cap;
    const double* KF = K.workb + (size_t)s * 2 * budget * capr;
    const double* DS = KF + (size_t)budget * capr;
    for (int j = threadIdx.x; j < m; j += blockDim.x) sm.co[j] = coeffg[j];
    // ---- delta = 1 - k_f . d* per proposal (only the "saturated" flag depends on it here)
    for (int p = wave; p < np; p += nw) {
        const double dot = wave_dot256(DS + (size_t)p * capr, KF + (size_t)p * capr, m);
        double delta = 1.0 - dot;
        delta = delta > 0.0 ? delta : 0.0;
        if (lane == 0) sm.f[p] = delta > D.eta ? 1.0 : 0.0;
    }
    __syncthreads();
    // ---- the proposals in order: predict against the evolving coefficients, project if still a mistake
    uint64_t n_mist = 0;
    if (wave == 0) {
        bool sat = false;
        for (int p = 0; p < np; ++p) {
            const int y = (((int)pr[(size_t)p * KB_PROP_W + 1]) & 1) ? 1 : -1;
            const double f = wave_dot256(KF + (size_t)p * capr, sm.co, m);
            if (f * (double)y <= 0.0) {
                n_mist += 1;
                sat = sat || sm.f[p] != 0.0;
                const double* ds = DS + (size_t)p * capr;
                for (int j = lane; j < m; j += 64) sm.co[j] = sm.co[j] + (double)y * ds[j];
            }
        }
        if (sat && lane == 0) atomicOr(&K.err[0], 8);  // a full dictionary met a sample it would have grown for
    }
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += blockDim.x) coeffg[j] = sm.co[j];
    return n_mist;
}

// apply a merged proposal list to the dictionary of slice s = blockIdx.x, in order
__global__ __launch_bounds__(1024) void shared_apply_kernel(KbDev D, KbState K, const double* props, const int32_t* counts,
                                                         int budget, uint64_t* gstats) {
    Lds sm = carve_lds(D.cap, K.work);
    const int s = blockIdx.x;
    const int d = D.dims[s] + 1;
    int m = K.m[s];
    const int np = counts[s] < budget ? counts[s] : budget;
    uint64_t n_mist = 0, n_grow = 0;
    if (np > 0 && batch_applies(D, m)) {
        // kernel columns and d* of the whole list are in the work area (shared_cols_kernel, shared_matvec_kernel)
        n_mist = apply_full_batch(D, K, s, m, props + (size_t)s * budget * KB_PROP_W, np, budget, sm);
        if (threadIdx.x == 0) atomicAdd((unsigned long long*)&gstats[1], (unsigned long long)n_mist);
        return;
    }
    for (int i = 0; i < np; ++i) {
        const double* p = props + ((size_t)s * budget + i) * KB_PROP_W;
        const int packed = (int)p[1];
        const int c = packed >> 2, y = (packed & 1) ? 1 : -1;
        __syncthreads();
        if (threadIdx.x < d - 1) sm.x[threadIdx.x] = p[2 + threadIdx.x];
        __syncthreads();
        prepare_operands(D, K, s, m, d, sm);
        // Projectron.predict on (state, c/n): f = k . coeff (float32 while a single landmark is held)
        kernel_column(D, m, c, sm);
        double part = 0.0;
        if (threadIdx.x < 256)  // same 256 strided partial sums as the 256-thread kernels
            for (int j = threadIdx.x; j < m; j += 256) part += sm.kf[j] * sm.co[j];
        double f = block_sum(part, sm);
        if (m == 1) f = (double)(float)((float)sm.kf[0] * (float)sm.co[0]);
        if (m == 0) f = 0.0;
        if (f * (double)y <= 0.0) {  // still a mistake against the evolving dictionary
            int branch;
            double delta;
            const int m_new = apply_update<8>(D, K, s, 0, m, d, (double)c / (double)D.n_prbs, y, sm, &branch, &delta);
            n_mist += 1;
            if (branch == 2 && m_new > m) n_grow += 1;
            m = m_new;
        }
    }
    if (threadIdx.x == 0) {
        K.m[s] = m;
        atomicAdd((unsigned long long*)&gstats[1], (unsigned long long)n_mist);
        atomicAdd((unsigned long long*)&gstats[2], (unsigned long long)n_grow);
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
    Lds sm = carve_lds(D.cap, K.work);
    const int task = A.task, env = task / D.S, s = task - env * D.S;
    const int dt = dict_of(D, task);
    const int d = D.dims[s] + 1, cap = D.cap;
    const int m = K.m[dt];
    double* kfg = K.kf + (size_t)task * cap;
    double p = 0.0;
    const double* L = K.L + (size_t)dt * KB_DMAX * cap;
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        double dist = 0.0;
        for (int q = 0; q < d; ++q) {
            double t = L[(size_t)q * cap + j] - A.x[q];
            dist += t * t;
        }
        double k = rs_exp(-D.gamma * dist);
        if (m == 1) k = (double)(float)k;
        kfg[j] = k;
        p += k * K.coeff[(size_t)dt * cap + j];
    }
    double f = block_sum(p, sm);
    if (m == 1) f = (double)(float)((float)kfg[0] * (float)K.coeff[(size_t)dt * cap]);
    if (threadIdx.x == 0) {
        int y = 0;
        if (m > 0) {
            y = f > 0.0 ? 1 : (f < 0.0 ? -1 : 0);
            if (y == 0) y = tie_draw(K, task, env, s);
        } else {
            f = 0.0;
            kfg[0] = 0.0;
        }
        K.f_last[task] = f;
        K.m_last[task] = m;
        A.out[0] = (double)y;
        A.out[1] = f;
        K.stats[(size_t)task * 4 + 0] += 1;
    }
}

__global__ __launch_bounds__(256) void update_one_kernel(OneArgs A) {
    const KbDev& D = A.D;
    const KbState& K = A.K;
    Lds sm = carve_lds(D.cap, K.work);
    const int task = A.task, env = task / D.S, s = task - env * D.S;
    const int dt = dict_of(D, task);
    const int d = D.dims[s] + 1, cap = D.cap;
    int m = K.m[dt];
    // Q12: Projectron.update uses the (f, K_f) cached by the predict that immediately preceded it.  Learners bound
    // into a KBRL_Control share their dictionary with update_control / select_action, which may have grown it in
    // between; the reference would then multiply arrays of different lengths (numpy raises).  Report it.
    if (K.m_last[task] != m) {
        if (threadIdx.x == 0) { A.out[2] = -1.0; A.out[3] = 0.0; }
        return;
    }
    const double f = K.f_last[task];
    if (!(f * (double)A.y <= 0.0)) {
        if (threadIdx.x == 0) { A.out[2] = 0.0; A.out[3] = 0.0; }
        return;
    }
    // stage x, the coefficients and the cached kernel row for apply_update
    for (int q = threadIdx.x; q < d - 1; q += blockDim.x) sm.x[q] = A.x[q];
    for (int j = threadIdx.x; j < sm.capr; j += blockDim.x) {
        sm.co[j] = j < m ? K.coeff[(size_t)dt * cap + j] : 0.0;
        sm.kf[j] = j < (m > 0 ? m : 1) ? K.kf[(size_t)task * cap + j] : 0.0;
    }
    __syncthreads();
    int branch;
    double delta;
    const int m_new = apply_update(D, K, dt, env, m, d, A.x[d - 1], A.y, sm, &branch, &delta);
    if (threadIdx.x == 0) {
        K.m[dt] = m_new;
        A.out[2] = (double)branch;
        A.out[3] = delta;
        K.stats[(size_t)task * 4 + 1] += 1;
        if (branch == 2 && m_new > m) K.stats[(size_t)task * 4 + 2] += 1;
    }
}

__global__ void kb_reset_kernel(KbDev D, KbState K, const int32_t* init_action, const int32_t* init_sec,
                                const uint64_t* seeds) {
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
    if (i < D.n_envs) {
        K.seeds[i] = seeds[i];
        K.adjusted[i] = 0;
        K.err[i] = 0;
    }
}

}  // namespace kb

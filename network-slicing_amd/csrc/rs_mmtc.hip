// rs_mmtc.hip -- one observation period of every mMTC slice of every replica, on gfx950.
//
// What it computes:
//   SliceL1mMTC.slot        slice_l1.py:87-125  (NB-IoT style FIFO: one carrier per PRB, each
//                                                backlogged device needs `repetitions` slots)
//   SliceRANmMTC.slot/...   slice_ran.py:91-148 (1000 periodic devices, SLA on mean delay)
//
// Mapping: one wavefront per (replica, mMTC slice).  The FIFO lives in LDS for the whole step;
// device d of the slice is owned by lane d % 64, which keeps its next arrival time in a
// register, so a slot without arrivals costs one compare + ballot.  The per-slot means the
// reference takes with numpy (delays.mean(), repetitions.mean()) are integer sums, kept as
// running totals and updated on arrival / transmission / completion.  RNG is used only at
// reset (slice_ran.py:97-101), from the slice's Philox stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rs_device.h"
#include "../../include/rs_philox.h"
#include "../../include/ranslice.h"

struct rs_handle;

namespace rs {

#define MTC_DEV_PER_LANE 16
#define MTC_DEV_MAX (64 * MTC_DEV_PER_LANE)
#define MTC_CAP_MAX 2048

struct MtcState {
    int32_t* n_users;   // [task]
    int64_t* s_start;   // [task] sum of t_start over the backlog
    int64_t* s_rep;     // [task] sum of remaining repetitions over the backlog
    int32_t* dev_next;  // [task][MTC_DEV_MAX] absolute slot of the next message
    int32_t* dev_period;
    int32_t* dev_rep;
    int32_t* q_rep;     // [task][cap]
    int32_t* q_start;   // [task][cap]
    size_t n_tasks;
    int cap;
};

__global__ void mtc_reset_kernel(const RsDev* D, MtcState M, const uint64_t* seeds) {
    const int task = blockIdx.x;
    const int rep = task / D->n_mmtc;
    const int sl = D->n_embb + (task - rep * D->n_mmtc);
    const uint64_t seed = seeds[rep];
    for (int i = threadIdx.x; i < MTC_DEV_MAX; i += blockDim.x) {
        size_t o = (size_t)task * MTC_DEV_MAX + i;
        if (i < D->mtc_n_dev) {
            // SliceRANmMTC.reset (slice_ran.py:97-101): three choices per device, in device order
            rs_stream st = {(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)sl, 0u, (uint32_t)(3 * i)};
            int r = D->mtc_rep_set[rs_stream_integers(&st, D->mtc_n_rep)];
            int p = D->mtc_period_set[rs_stream_integers(&st, D->mtc_n_period)];
            int t = 1 + (int)rs_stream_integers(&st, p);
            M.dev_rep[o] = r;
            M.dev_period[o] = p;
            M.dev_next[o] = t;
        } else {
            M.dev_rep[o] = 0;
            M.dev_period[o] = 0;
            M.dev_next[o] = RS_NEVER;
        }
    }
    if (threadIdx.x == 0) {
        M.n_users[task] = 0;
        M.s_start[task] = 0;
        M.s_rep[task] = 0;
    }
}

struct MtcArgs {
    const RsDev* D;
    MtcState M;
    const int32_t* actions;
    const int64_t* run;  // device-side run state: [0] = slots elapsed since reset
    float* obs;
    int32_t* labels;
    int32_t* violations;
    double* info;
    int32_t* err;
};

// one wave per task, 4 tasks per 256-thread block; dynamic LDS = 4 * cap * 8 bytes
__global__ __launch_bounds__(256) void mtc_step_kernel(MtcArgs A) {
    extern __shared__ int32_t lds[];
    const RsDev* __restrict__ D = A.D;
    const MtcState& M = A.M;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int cap = M.cap;
    int32_t* q_rep = lds + (size_t)w * 2 * cap;
    int32_t* q_start = q_rep + cap;
    const int task = blockIdx.x * 4 + w;
    if (task >= (int)M.n_tasks) return;  // whole wave exits; no block-level barrier is used
    const int rep = task / D->n_mmtc;
    const int ms = task - rep * D->n_mmtc;
    const int sl = D->n_embb + ms;
    const int n_prbs = A.actions[rep * D->n_slices + sl];

    int n_users = M.n_users[task];
    int64_t s_start = M.s_start[task], s_rep = M.s_rep[task];
    for (int i = lane; i < n_users; i += 64) {
        q_rep[i] = M.q_rep[(size_t)task * cap + i];
        q_start[i] = M.q_start[(size_t)task * cap + i];
    }
    int dnext[MTC_DEV_PER_LANE];
    int min_next = RS_NEVER;
#pragma unroll
    for (int k = 0; k < MTC_DEV_PER_LANE; ++k) {
        dnext[k] = M.dev_next[(size_t)task * MTC_DEV_MAX + k * 64 + lane];
        min_next = dnext[k] < min_next ? dnext[k] : min_next;
    }
    double i_delay = 0.0, i_rep = 0.0, i_dev = 0.0;
    int err = 0;

    const int slots = D->slots;
    const int clock0 = (int)A.run[0];
    for (int t = 0; t < slots; ++t) {
        const int now = clock0 + t + 1;  // SliceL1mMTC.time (slice_l1.py:88)
        // ---- arrivals in device-index order (slice_ran.py:106-121, slice_l1.py:76-82)
        if (__builtin_amdgcn_ballot_w64(min_next == now) != 0ull) {
#pragma unroll
            for (int k = 0; k < MTC_DEV_PER_LANE; ++k) {
                const bool fire = dnext[k] == now;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(fire);
                if (mk != 0ull) {
                    const int before = __popcll(mk & ((1ull << lane) - 1ull));
                    const int cnt = __popcll(mk);
                    if (fire) {
                        const size_t o = (size_t)task * MTC_DEV_MAX + k * 64 + lane;
                        const int pos = n_users + before;
                        const int r = M.dev_rep[o];
                        if (pos < cap) {
                            q_rep[pos] = r;
                            q_start[pos] = now;
                        } else {
                            err = 1;
                        }
                        dnext[k] = now + M.dev_period[o];
                    }
                    // running sums: every lane needs the totals -> reduce the arrivals' reps
                    int rsum = fire ? M.dev_rep[(size_t)task * MTC_DEV_MAX + k * 64 + lane] : 0;
                    for (int d = 32; d >= 1; d >>= 1) rsum += __shfl_xor(rsum, d);
                    int take = n_users + cnt <= cap ? cnt : cap - n_users;
                    if (take < cnt) {
                        // capacity overflow: results are undefined from here on and flagged
                        err = 1;
                        take = take > 0 ? take : 0;
                    }
                    s_rep += rsum;
                    s_start += (int64_t)cnt * now;
                    n_users += take;
                }
            }
            min_next = RS_NEVER;
#pragma unroll
            for (int k = 0; k < MTC_DEV_PER_LANE; ++k) min_next = dnext[k] < min_next ? dnext[k] : min_next;
        }
        // ---- transmissions: the first n_tx backlogged devices use one carrier each
        const int n_tx = n_prbs < n_users ? n_prbs : n_users;
        bool any_done = false;
        for (int base = 0; base < n_tx; base += 64) {
            const int i = base + lane;
            bool done = false;
            if (i < n_tx) {
                int r = q_rep[i] - 1;
                q_rep[i] = r;
                done = r <= 0;
            }
            any_done = any_done || (__builtin_amdgcn_ballot_w64(done) != 0ull);
        }
        s_rep -= n_tx;
        // ---- drop finished devices, order preserved (slice_l1.py:102-107)
        if (any_done) {
            int wpos = 0;
            int64_t removed_start = 0;
            for (int base = 0; base < n_users; base += 64) {
                const int i = base + lane;
                int r = 0, st = 0;
                if (i < n_users) {
                    r = q_rep[i];
                    st = q_start[i];
                }
                const bool keep = i < n_users && r > 0;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(keep);
                const int pos = wpos + __popcll(mk & ((1ull << lane) - 1ull));
                int rs_ = (i < n_users && !keep) ? st : 0;
                for (int d = 32; d >= 1; d >>= 1) rs_ += __shfl_xor(rs_, d);
                removed_start += rs_;
                if (keep) {
                    q_rep[pos] = r;
                    q_start[pos] = st;
                }
                wpos += __popcll(mk);
            }
            s_start -= removed_start;
            n_users = wpos;
        }
        // ---- per-slot summary (slice_l1.py:109-125) -> SliceRANmMTC.update_info (slice_ran.py:139-142)
        double delay = 0.0, avg_rep = 0.0;
        if (n_users > 0) {
            int64_t sd = (int64_t)n_users * now - s_start;  // delays are never negative here
            delay = (double)sd / (double)n_users;
            avg_rep = RS_RINT((double)s_rep / (double)n_users);
        }
        i_delay += delay;
        i_rep += avg_rep;
        i_dev += (double)n_users;
    }

    // ---- write back
    for (int i = lane; i < n_users; i += 64) {
        M.q_rep[(size_t)task * cap + i] = q_rep[i];
        M.q_start[(size_t)task * cap + i] = q_start[i];
    }
#pragma unroll
    for (int k = 0; k < MTC_DEV_PER_LANE; ++k) M.dev_next[(size_t)task * MTC_DEV_MAX + k * 64 + lane] = dnext[k];
    const bool any_err = __builtin_amdgcn_ballot_w64(err != 0) != 0ull;
    if (lane == 0) {
        M.n_users[task] = n_users;
        M.s_start[task] = s_start;
        M.s_rep[task] = s_rep;
        // get_state order: devices, avg_rep, delay (scenario_creator.py:92; slice_ran.py:133-137)
        float* o = A.obs + (size_t)rep * D->n_vars + D->n_embb * RS_N_EMBB_VARS + ms * RS_N_MMTC_VARS;
        o[0] = (float)(i_dev / D->norm_mmtc[0]);
        o[1] = (float)(i_rep / D->norm_mmtc[1]);
        o[2] = (float)(i_delay / D->norm_mmtc[2]);
        double* inf = A.info + ((size_t)rep * D->n_slices + sl) * 10;
        inf[0] = i_delay;
        inf[1] = i_rep;
        inf[2] = i_dev;
        for (int k = 3; k < 10; ++k) inf[k] = 0.0;
        // SliceRANmMTC.compute_reward (slice_ran.py:145-148)
        const bool ok = i_delay / slots < D->sla_mtc_delay;
        A.violations[rep * D->n_slices + sl] = ok ? 0 : 1;
        A.labels[rep * D->n_slices + sl] = ok ? 1 : -1;
        if (any_err) atomicOr(&A.err[rep], 4);  // bit 2: mMTC queue capacity
    }
}

}  // namespace rs

// host helpers used by rs_api.hip (defined there after rs_handle is complete)
static int mtc_alloc(rs_handle* h, rs::MtcState* m, size_t n_tasks, const RsDev& d);
static int mtc_reset(rs_handle* h, rs::MtcState* m);
static int mtc_step(rs_handle* h, rs::MtcState* m, hipStream_t stream);

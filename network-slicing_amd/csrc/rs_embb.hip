// rs_embb.hip -- one observation period (slots_per_step slots) of every eMBB slice of every
// replica, on gfx950.
//
// What it computes (reference call tree):
//   NodeB.step                 node_b.py:59-91      (set_prbs + the slot loop for eMBB slices)
//   SliceL1eMBB.slot           slice_l1.py:193-228
//   SliceRANeMBB.slot/...      slice_ran.py:195-325 (arrivals, CAC, departures, update_info, SLA)
//   UE.*                       slice_ran.py:20-58
//   ProportionalFair.allocate  schedulers.py:21-76
//   SINRSelectiveFading / MCSCodeset / macro_cell   channel_models.py
//   CbrSource / VbrSource      traffic_generators.py
//
// Mapping to the machine: a G-lane group (G = 16 or 32; 64/G tasks per wavefront) owns one (replica, slice)
// task for the whole step, lane u = UE u; hot per-UE state sits in VGPRs for all 50 slots, cold state (timers,
// stream counters, accumulators) in LDS, HBM is touched once in and once out.  A task that needs more UE lanes
// than the instance has leaves its state untouched, raises a redo flag and is replayed by the G = 32 instance of
// the same kernel -- identical arithmetic, so results do not depend on G.
//
// Two kinds of work alternate inside a step:
//   * CHANNEL ESTIMATES (get_snr + estimate_snr).  A UE's fading trajectory depends on nothing the scheduler
//     does (the walker's redraws are addressed by time, include/rs_philox.h), so round(mean(column)) of every
//     (UE, slot) of the next CH slots is evaluated AHEAD of the slot loop: every lane of the wave takes one
//     (UE, slot) item and sums its column alone, in numpy's pairwise order (8 strided accumulators held by the
//     lane), the items of the wave's 64/G tasks being dealt out to all 64 lanes.  Results go to an int16 table in
//     LDS.  This is the only part with real HBM/L2 traffic, and here it runs with every lane busy and every load
//     independent of the scheduling chain.
//   * THE SLOT LOOP (traffic, PF, response, reception, SLA sums): serial from slot to slot through queue and
//     throughput average.  PF is closed-form when the slot is under-loaded (order of service cannot matter, see
//     below), otherwise the leader of the metric runs the reference loop alone until it loses the argmax.  The
//     reception model (MI average -> effective SNR -> probability) is evaluated only for UEs that actually sent
//     bits: for the others the Bernoulli draw is consumed (the stream counter advances) but cannot change state.
// Compiled with -ffp-contract=off: all f64 arithmetic is IEEE and in the oracle's order.

#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rs_device.h"

#if defined(__HIP_DEVICE_COMPILE__)
// out-of-line exp/log for device code (see include/rs_detmath.h, RS_EXP_CALL)
__device__ double rs_exp_ool(double x);
__device__ double rs_log_ool(double x);
#define RS_EXP_CALL rs_exp_ool
#define RS_LOG_CALL rs_log_ool
#endif
#include "../../include/rs_philox.h"
#include "../../include/ranslice.h"

typedef double rs_d2 __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __noinline__ double rs_exp_ool(double x) { return rs_exp(x); }
__device__ __noinline__ double rs_log_ool(double x) { return rs_log(x); }
// rs_exp of two independent arguments in one call: the two dependent chains (reduction, a 13-step Horner form, the
// scaling) interleave instruction by instruction, so a wave that is waiting on its own arithmetic -- the heaviest waves of a
// launch, which the others make way for -- gets through two exponentials in little more than the time of one.  Same
// values as rs_exp, special cases by selection instead of branches.
__device__ __forceinline__ double rs_exp_sel(double x, double core) {
    double v = x < -745.2 ? 0.0 : core;
    v = x > 709.782712893384 ? rs_inf() : v;
    return x != x ? x : v;
}
__device__ __noinline__ rs_d2 rs_exp2_ool(rs_d2 x) {
    const double a = (x.x >= -745.2 && x.x <= 709.782712893384) ? x.x : 0.0;
    const double b = (x.y >= -745.2 && x.y <= 709.782712893384) ? x.y : 0.0;
    const double ca = rs_exp_core(a), cb = rs_exp_core(b);
    rs_d2 o;
    o.x = rs_exp_sel(x.x, ca);
    o.y = rs_exp_sel(x.y, cb);
    return o;
}
// MCSCodeset's logistic curve (rs_sigmoid) of two arguments at once: same values
__device__ __forceinline__ rs_d2 rs_sigmoid2(double x1, double x2, double x0, double k) {
    rs_d2 t;
    t.x = (-k) * (x1 - x0);
    t.y = (-k) * (x2 - x0);
    const rs_d2 e = rs_exp2_ool(t);
    rs_d2 o;
    o.x = 1.0 / (1.0 + e.x);
    o.y = 1.0 / (1.0 + e.y);
    return o;
}
#else
__device__ rs_d2 rs_sigmoid2(double x1, double x2, double x0, double k);  // (host pass: device code only)
#endif

namespace rs {

// Optional per-section cycle accounting (build with -DRS_SECTION_PROFILE; tools/section_profile.py).
#ifndef RS_PRIO_A
#define RS_PRIO_A 10u
#define RS_PRIO_B 4u
#define RS_PRIO_C 2u
#endif
#ifndef RS_OCC
#define RS_OCC 5
#endif
#ifndef RS_LOADS
#define RS_LOADS 4  // column sums: fading samples a lane has in flight (4 or 8)
#endif
#ifndef RS_OCC_OTHER
#define RS_OCC_OTHER 3  // waves per SIMD the tracing / 8- and 32-lane instances are compiled for (5: profiles/HISTORY.md)
#endif
#ifndef RS_DYN_PRIO
#define RS_DYN_PRIO 1
#endif
#ifndef RS_BLOCK_MIN
#define RS_BLOCK_MIN 4     // PF: block rounds while a contender's share of the free RB pairs is at least this ...
#endif
#ifndef RS_BLOCK_PAIRS
#define RS_BLOCK_PAIRS 8   // ... in slices of at least this many RB pairs (profiles/HISTORY.md: 24 until the shares of a block
#endif                     //     round lost their run-time branch; with agents in the loop 8 is 4 % faster late in learning)
#ifndef RS_HINT_PAIRS
#define RS_HINT_PAIRS 24   // the BLOCK instance is picked for allocations with slices of this many pairs (rs_api.hip: auto_hint,
#endif                     //     rs_step): on the random script's 12-25-pair slices block rounds lose to the trip loop
#ifndef RS_PACE_3B
#define RS_PACE_3B 46ull  // the same three thresholds in the BLOCK instances (agents' allocations: a few wide tasks carry the launch, and
#define RS_PACE_2B 44ull  // a wave that is merely a little behind should not yet take issue slots from them): > 1.44 / 1.375 / 1.31 x the
#define RS_PACE_1B 42ull  // reference pace (profiles/HISTORY.md: step kernel 1.30 -> 1.25 ms early, 1.545 -> 1.50 late in learning)
#endif
#ifndef RS_PACE_3
#define RS_PACE_3 40ull  // > 1.25 x the reference pace: priority 3 (the plateau of a sweep, profiles/HISTORY.md)
#define RS_PACE_2 35ull  // > 1.09: 2
#define RS_PACE_1 28ull  // > 0.875: 1
#endif
#ifndef RS_LPU
#define RS_LPU 4
#endif

#ifdef RS_SECTION_PROFILE
#define SEC_DECL unsigned long long sec_t0 = __builtin_amdgcn_s_memtime(), sec_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SEC_MARK(i)                                              \
    {                                                            \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();    \
        sec_acc[i] += t_ - sec_t0;                               \
        sec_t0 = t_;                                             \
    }
#define SEC_FLUSH(buf)                                                                                  \
    if ((threadIdx.x & 63u) == 0u) {                                                                    \
        unsigned long long tot_ = 0;                                                                    \
        for (int i_ = 0; i_ < 13; ++i_) tot_ += sec_acc[i_];                                            \
        for (int i_ = 0; i_ < 16; ++i_)                                                                 \
            if (i_ != 14) atomicAdd((unsigned long long*)&(buf)[i_], sec_acc[i_]);                      \
        /* slowest wave of any launch so far; its own section split goes to a second bank (racy, profiling only) */ \
        if (tot_ > atomicMax((unsigned long long*)&(buf)[14], tot_))                                    \
            for (int i_ = 0; i_ < 16; ++i_) (buf)[16 + 4 * (size_t)n_tasks + i_] = sec_acc[i_];        \
    }                                                                                                   \
    if ((threadIdx.x & 63u) != 0u && sec_acc[13]) atomicAdd((unsigned long long*)&(buf)[13], sec_acc[13]); \
    if (gl == 0 && valid) { /* per task: cycles of its wave, UEs and RBs at the start, contested PF trips */ \
        unsigned long long tot_ = 0;                                                                    \
        for (int i_ = 0; i_ < 13; ++i_) tot_ += sec_acc[i_];                                            \
        (buf)[16 + task * 4 + 0] = tot_;                                                                \
        /* upper halves: where the wave ran (HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID) */ \
        (buf)[16 + task * 4 + 1] = (unsigned long long)S.t_n_ue[task] |                                 \
                                   ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);        \
        (buf)[16 + task * 4 + 2] = (unsigned long long)A.actions[rep * n_slices + sl] |                 \
                                   ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32);        \
        (buf)[16 + task * 4 + 3] = (unsigned long long)(stat >> 18) | ((sec_t0 - tot_) << 24); /* start time */ \
    }
#else
#define SEC_DECL
#define SEC_MARK(i)
#define SEC_FLUSH(buf)
#endif

__device__ __forceinline__ int bperm(int v, int src_lane) {
    return __builtin_amdgcn_ds_bpermute(src_lane << 2, v);
}
__device__ __forceinline__ unsigned bperm(unsigned v, int src_lane) {
    return (unsigned)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}
__device__ __forceinline__ double bperm(double v, int src_lane) {
    uint64_t u = rs_d2u(v);
    int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)u);
    int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)(u >> 32));
    return rs_u2d(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }

// ---- cross-lane primitives on DPP (VALU latency) instead of ds_bpermute (LDS latency).
// DPP controls: quad_perm [1,0,3,2] = 0xB1 (lane^1), [2,3,0,1] = 0x4E (lane^2), row_half_mirror = 0x141
// (i <-> 7-i inside 8 lanes), row_mirror = 0x140 (i <-> 15-i inside a 16-lane row), row_shr:n = 0x110+n,
// row_shl:n = 0x100+n, row_bcast:15 = 0x142.  v_permlane16_swap (gfx950) exchanges the two rows of a
// 32-lane group.  A group of G lanes never straddles a row boundary for G <= 16.
#define DPP_XOR1 0xB1
#define DPP_XOR2 0x4E
#define DPP_HMIRROR 0x141
#define DPP_MIRROR 0x140
#define DPP_BCAST15 0x142

// The permutations used through dpp_i / dpp_d read only lanes of the caller's own subgroup, which share its
// EXEC state, so every destination lane has a valid source and no `old` operand has to be materialised
// (row_shl:1 in the pairwise remainder may pull an undefined value into a subgroup's last lane; it never
// reaches lane 0 within the LPU - 1 shifts that are consumed).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    uint64_t u = rs_d2u(v);
    int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)u, CTRL, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(u >> 32), CTRL, 0xF, 0xF, false);
    return rs_u2d(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
// maximum of two finite doubles: one v_max_f64 (the generic fmax also quiets its operands first).  The result
// usually feeds a DPP move next, and the compiler cannot see a VALU write inside the asm, so the two wait
// states that hazard needs are spelled out here.
__device__ __forceinline__ double max_finite(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ballot restricted to this lane's G-lane group (bit u = lane u of the group)
template <int G>
__device__ __forceinline__ unsigned group_ballot(bool c, int gbase) {
    const unsigned long long b = __builtin_amdgcn_ballot_w64(c) >> gbase;
    return G == 32 ? (unsigned)b : ((unsigned)b & ((1u << G) - 1u));
}

// sum over the G lanes of a group; every lane gets the total.  Used only for exactly representable
// integers (int, or integer-valued doubles), so the association order is irrelevant.
template <int G>
__device__ __forceinline__ int group_sum(int v) {
    v += dpp_i<DPP_XOR1>(v);
    v += dpp_i<DPP_XOR2>(v);
    v += dpp_i<DPP_HMIRROR>(v);
    if (G >= 16) v += dpp_i<DPP_MIRROR>(v);
    if (G == 32) {
        auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        v = (int)r[0] + (int)r[1];
    }
    return v;
}
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    v += dpp_d<DPP_XOR1>(v);
    v += dpp_d<DPP_XOR2>(v);
    v += dpp_d<DPP_HMIRROR>(v);
    if (G >= 16) v += dpp_d<DPP_MIRROR>(v);
    if (G == 32) {
        uint64_t u = rs_d2u(v);
        auto lo = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);
        auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
        v = rs_u2d(((uint64_t)hi[0] << 32) | lo[0]) + rs_u2d(((uint64_t)hi[1] << 32) | lo[1]);
    }
    return v;
}

template <int G>
__device__ __forceinline__ double group_max(double v) {
    double o;
    // operands are finite (metrics >= 0 or the -1 / -2 sentinels)
    o = dpp_d<DPP_XOR1>(v); v = max_finite(o, v);
    o = dpp_d<DPP_XOR2>(v); v = max_finite(o, v);
    o = dpp_d<DPP_HMIRROR>(v); v = max_finite(o, v);
    if (G >= 16) { o = dpp_d<DPP_MIRROR>(v); v = max_finite(o, v); }
    if (G == 32) {
        uint64_t u = rs_d2u(v);
        auto lo = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);
        auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
        double a = rs_u2d(((uint64_t)hi[0] << 32) | lo[0]);
        double b = rs_u2d(((uint64_t)hi[1] << 32) | lo[1]);
        v = max_finite(a, b);
    }
    return v;
}

// maximum of finite floats over the G lanes of a group (DPP moves fold into the v_max_f32)
template <int G>
__device__ __forceinline__ float group_max_f32(float v) {
    v = __builtin_fmaxf(v, __builtin_bit_cast(float, dpp_i<DPP_XOR1>(__builtin_bit_cast(int, v))));
    v = __builtin_fmaxf(v, __builtin_bit_cast(float, dpp_i<DPP_XOR2>(__builtin_bit_cast(int, v))));
    v = __builtin_fmaxf(v, __builtin_bit_cast(float, dpp_i<DPP_HMIRROR>(__builtin_bit_cast(int, v))));
    if (G >= 16) v = __builtin_fmaxf(v, __builtin_bit_cast(float, dpp_i<DPP_MIRROR>(__builtin_bit_cast(int, v))));
    if (G == 32) {
        auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
        v = __builtin_fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
    }
    return v;
}

// exclusive prefix sum over the G lanes of a group: Hillis-Steele with row_shr (sources outside the
// group masked by lane index), plus row_bcast:15 to carry row 0's total into row 1 when G = 32
template <int G>
__device__ __forceinline__ int group_excl_scan(int v, int gl) {
    int inc = v, t;
    t = __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true); inc += gl >= 1 ? t : 0;
    t = __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true); inc += gl >= 2 ? t : 0;
    t = __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true); inc += gl >= 4 ? t : 0;
    if (G >= 16) { t = __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, true); inc += (gl & 15) >= 8 ? t : 0; }
    if (G == 32) inc += __builtin_amdgcn_update_dpp(0, inc, DPP_BCAST15, 0xA, 0xF, false);
    return inc - v;
}

// index of the k-th (0-based) set bit of m, or 0 if there is none
__device__ __forceinline__ int kth_set_bit(unsigned m, int k) {
    for (int z = 0; z < k; ++z) m &= m - 1u;
    return m ? __ffs((int)m) - 1 : 0;
}

// numpy's pairwise sum (np.sum / np.mean of a contiguous f64 vector, n <= 256) splits a vector longer than 128 in
// halves rounded down to a multiple of 8, recursively, and sums a block of <= 128 elements with 8 strided
// accumulators R_0..R_7, the tree ((R0+R1)+(R2+R3))+((R4+R5)+(R6+R7)) and a sequential remainder.
// Two evaluators of the same order follow: one lane alone (channel estimates), and the LPU = 4 lanes of a subgroup
// together (MI averages of the response).
struct PwSplit {
    int n0, n1, n2;  // block lengths (n1, n2 may be 0): result = B(0, n0) + (B(n0, n1) + B(n0 + n1, n2))
    __device__ __forceinline__ explicit PwSplit(int n) {
        n0 = n;
        n1 = 0;
        n2 = 0;
        if (n > 128) {  // pw(n) = pw(h) + pw(n - h), h = n/2 - (n/2) % 8  (<= 128 for n <= 256)
            int h = n >> 1;
            h -= h & 7;
            n0 = h;
            n1 = n - h;
            if (n1 > 128) {  // 129..135: split once more
                int h2 = n1 >> 1;
                h2 -= h2 & 7;
                n2 = n1 - h2;
                n1 = h2;
            }
        }
    }
};

// one block by ONE lane: ld(i) = element i; `on` masks lanes without work (loops are wave-uniform)
template <class F>
__device__ __forceinline__ double lane_block(int off, int n, bool on, F ld) {
    const int lim = n >= 8 ? n - (n & 7) : 0;
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = 0.0;
    if (on && n >= 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = ld(off + j);
    }
    for (int i = 8; wave_any(on && i < lim); i += 8) {
        if (on && i < lim) {
#pragma unroll
            for (int h = 0; h < 8; h += RS_LOADS) {  // RS_LOADS loads in flight at a time (register budget)
                double v[RS_LOADS];
#pragma unroll
                for (int j = 0; j < RS_LOADS; ++j) v[j] = ld(off + i + h + j);
#pragma unroll
                for (int j = 0; j < RS_LOADS; ++j) r[h + j] += v[j];
            }
        }
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));  // n < 8: 0.0, as numpy starts
    const int rem = n - lim;
    if (wave_any(on && rem > 0)) {
        // the (up to seven) remainder elements are added one after the other, but fetched together: one memory round
        // trip instead of one per element
        double v[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = ld(off + lim + (j < rem ? j : 0));
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (on && j < rem) res += v[j];
    }
    return res;
}

template <class F>
__device__ __forceinline__ double lane_pairwise(int n, bool on, F ld) {
    const PwSplit s(n);
    double a = lane_block(0, s.n0, on, ld);
    if (wave_any(on && s.n1 > 0)) {
        double b = lane_block(s.n0, s.n1, on && s.n1 > 0, ld);
        if (wave_any(on && s.n2 > 0)) {
            const double c = lane_block(s.n0 + s.n1, s.n2, on && s.n2 > 0, ld);
            b = s.n2 > 0 ? b + c : b;
        }
        a = s.n1 > 0 ? a + b : a;
    }
    return a;
}

// one block by the 8 lanes of a TEAM: lane j owns R_j.  Every lane of the team must call it with the same (off, n).
// The elements are expensive (the response evaluates a sigmoid of a table value per element), so every lane walks its own
// accumulator serially and the team meets only for the tree and the remainder.  f2(i1, p1, i2, p2) returns elements i1
// and i2 together (p: the lane needs that one; the other half of the pair is garbage otherwise): a lane's elements are
// evaluated two at a time -- the remainder element with the first of the accumulator, then the accumulator's in pairs --
// and added one after the other in the same order as ever.  Loops are wave-uniform (`on` masks idle teams).  The result
// is valid in lane 0 of the team.
template <class F2>
__device__ __forceinline__ double team_block(int off, int n, int j, bool on, F2 f2) {
    double res = 0.0;
    const int lim = n >= 8 ? n - (n & 7) : 0;
    const int rem = n - lim;
    const bool hv = on && j < rem, h0 = on && n >= 8;
    // the sequential remainder's operands are independent of the tree: evaluated with the accumulator's first element
    double v = 0.0, r = 0.0;
    if (wave_any(hv || h0)) {
        const rs_d2 s = f2(off + lim + j, hv, off + j, h0);
        v = hv ? s.x : 0.0;
        r = h0 ? s.y : 0.0;
    }
    if (wave_any(on && n >= 8)) {
        for (int i = 8; wave_any(on && i < lim); i += 16) {
            const bool p1 = on && i < lim, p2 = on && i + 8 < lim;
            const rs_d2 s = f2(off + i + j, p1, off + i + 8 + j, p2);
            if (p1) r += s.x;
            if (p2) r += s.y;
        }
        // ((R0+R1)+(R2+R3))+((R4+R5)+(R6+R7)): each pairing is commutative, so both partners agree
        r += dpp_d<DPP_XOR1>(r);
        r += dpp_d<DPP_XOR2>(r);
        r += dpp_d<DPP_HMIRROR>(r);
        res = n >= 8 ? r : 0.0;
    }
    if (wave_any(on && rem > 0)) {
#pragma unroll
        for (int s_ = 0; s_ < 7; ++s_) {
            if (s_ < rem) res += v;    // lane 0: element s_ of the remainder
            v = dpp_d<0x101>(v);       // row_shl:1: lane i <- lane i+1 (lane 7 of a team may pull its neighbour's
                                       // value: it never reaches lane 0 within the 7 shifts that are consumed)
        }
    }
    return res;
}

template <class F2>
__device__ __forceinline__ double team_pairwise(int n, int j, bool on, F2 f2) {
    const PwSplit s(n);
    double a = team_block(0, s.n0, j, on, f2);
    if (wave_any(on && s.n1 > 0)) {
        double b = team_block(s.n0, s.n1, j, on && s.n1 > 0, f2);
        if (wave_any(on && s.n2 > 0)) {
            const double c = team_block(s.n0 + s.n1, s.n2, j, on && s.n2 > 0, f2);
            b = s.n2 > 0 ? b + c : b;
        }
        a = s.n1 > 0 ? a + b : a;
    }
    return a;
}

// macro_cell (channel_models.py:84-97) on the UE's own stream.  The stream travels by value and the advanced draw
// counter comes back beside the result: a pointer argument would put the caller's stream on the stack (scratch).
// (a vector type, so that it is returned in registers: x = nominal SINR, y = the stream's draw counter afterwards)
typedef double MacroCell __attribute__((ext_vector_type(2)));
__device__ __noinline__ MacroCell macro_cell_draw(const RsDev* D, rs_stream st_in) {
    rs_stream stv = st_in;
    rs_stream* st = &stv;
    double x, y;
    for (;;) {
        x = rs_stream_uniform(st);
        y = rs_stream_uniform(st);
        // generate_xy / find_y_value (channel_models.py:44-76), evaluated literally
        double m, b;
        m = (0.0 - 0.5) / (0.25 - 0.0); b = -m * 0.0 + 0.5;   bool c1 = y > m * x + b;
        m = (0.5 - 0.0) / (1.0 - 0.75); b = -m * 0.75 + 0.0;  bool c2 = y > m * x + b;
        m = (1.0 - 0.5) / (0.25 - 0.0); b = -m * 0.0 + 0.5;   bool c3 = y < m * x + b;
        m = (0.5 - 1.0) / (1.0 - 0.75); b = -m * 0.75 + 1.0;  bool c4 = y < m * x + b;
        if (c1 && c2 && c3 && c4) break;
    }
    // rs_stream_normal(st, 0.0, 10.0) spelled out with the in-line logarithm: this function must stay a leaf (a nested
    // call would make it save its return address through a stack slot, i.e. scratch)
    double LogF;
    {
        double v1, v2, r2;
        do {
            v1 = 2.0 * rs_stream_uniform(st) - 1.0;
            v2 = 2.0 * rs_stream_uniform(st) - 1.0;
            r2 = v1 * v1 + v2 * v2;
        } while (r2 >= 1.0 || r2 == 0.0);
        const double z = v1 * RS_SQRT((-2.0 * rs_log(r2)) / r2);
        LogF = 0.0 + 10.0 * z;
    }
    double x_t = x - 0.5 / 2;
    double distance = RS_SQRT(x_t * x_t + y * y);
    double cos_theta = x_t / distance;
    double theta = rs_acos(cos_theta);
    theta = theta * RS_RAD2DEG - 60;
    double R = distance * 2 > 0.1 ? distance * 2 : 0.1;
    double t65 = theta / 65;
    double att = 12 * (t65 * t65);
    double G = 15 + (-1 * (att < 20 ? att : 20));
    double lr = rs_log10(R);
    double L = D->prop_A + D->prop_B * lr;
    double gamma = 2.6;
    double FSPL = 20 * rs_log10(2.0) + 92.45 + gamma * 10 * lr;
    L = L > FSPL ? L : FSPL;
    double loss = L + LogF - G;
    double Rx_pw = 30 - (loss > 70 ? loss : 70);
    MacroCell out;
    out.x = Rx_pw - (-110) - 9;
    out.y = (double)stv.ctr;
    return out;
}

__device__ __forceinline__ int rint_slots(double seconds_or_slots, double slot_length) {
    // np.rint(x / slot_length) as a non-negative slot count, saturated far below RS_NEVER
    double v = RS_RINT(seconds_or_slots / slot_length);
    return v < 1.0e9 ? (int)v : 1000000000;
}

struct StepArgs {
    const RsDev* D;
    const RsState* S;         // device copy of the state pointers (kept out of the kernarg SGPRs)
    const double* fad;        // [trace][time][P]
    const float* fad32;       // the same samples in float32 (decisions by guard band; rs_api.hip: upload_fading)
    const double* fps;        // [trace][time][P + 1] prefix sums of a column's samples
    const uint8_t* fad_valid; // [trace][time]
    const int32_t* actions;   // [n_envs][n_slices]
    const int64_t* run;       // device-side run state: [0] slots elapsed since reset before this step (rs_api.hip)
    float* obs;               // [n_envs][n_vars]
    int32_t* labels;          // [n_envs][n_slices]
    int32_t* violations;      // [n_envs][n_slices]
    double* info;             // [n_envs][n_slices][10]
    uint64_t* counters;       // [n_tasks][4]
    rs_alloc_rec* trace;      // [n_tasks][slots][RS_GROUP] or null
    uint64_t* sections;       // [16] cycle sums per code section (RS_SECTION_PROFILE builds)
    int32_t* redo;            // [n_tasks] set by a G < 32 launch for tasks it could not hold; consumed by the G = 32 replay
    int32_t replay;           // 1: process only tasks whose redo flag is set
    unsigned long long* pace; // [0] sum, [1] count of the wave paces (cycles per slot) of the launch in flight,
                              // [2] mean pace of the previous launch: the reference of the dynamic issue priority
    const int32_t* order;     // [n_tasks] launch order of the tasks (rs_order.hip) or null = task index order
    int32_t spread;           // 1: ONE task per wave (its first group; the other groups idle).  For batches of at most one wave
                              // per SIMD of the chip: groups of a wave run their loops in lockstep, so a wave costs the maximum
                              // over its tasks loop by loop, and a batch this small gains nothing from packing them.
    int32_t order_off, order_cnt;  // this launch takes order[order_off .. order_off + order_cnt) (order_cnt < 0: every task) --
                              // the split step: the head of the cost ranking on the 16-lane instance, the rest on the 8-lane one
};

// select among three wave-uniform values by a per-lane index 0..2
template <class T>
__device__ __forceinline__ T sel3(int i, T a, T b, T c) {
    return i == 0 ? a : (i == 1 ? b : c);
}

// SINRSelectiveFading.get_snr's index walk (channel_models.py:171-191): one step, redraw on leaving [0, T)
// (time-addressed, include/rs_philox.h), skipping columns that contain NaN (Q10)
__device__ __forceinline__ void walker_advance(int& findex, int& fstep, int Tn, bool has_nan, const uint8_t* valid_col,
                                               uint32_t key0, uint32_t key1, uint32_t sl, uint32_t serial, uint32_t now) {
    unsigned attempt = 0u;
    for (;;) {
        findex += fstep;
        if (findex >= Tn || findex < 0) rs_walker_redraw(key0, key1, sl, serial, now, attempt++, Tn, &findex, &fstep);
        if (!has_nan || valid_col[findex]) break;
    }
}

// Response sums of the spans flagged `mine` (at most RS_WIDE_SPAN RBs each, or wider than RS_WIDE_MAX): R1 + R2 fused --
// np.mean's pairwise sum of the mutual information over a UE's RBs (channel_models.py:303-307).  The spans of the
// whole wave are dealt out to TEAMS of 8 lanes, lane j of a team owning numpy's accumulator R_j: it evaluates the
// sigmoid of its elements one after the other and adds them in numpy's order, the team meets for the tree and the
// remainder.  No per-RB values are stored anywhere.  Returns the lane's own sum (or `keep` if it has no span).
// mi: the logistic curves' (x0, k) per modulation in LDS, mi[md] and mi[4 + md] (six doubles held in registers across the
// response were what pushed five loop-long values out to scratch in every slot)
__device__ __forceinline__ double team_response(const double* mi, const double* fad, const double* nom_wave, bool mine,
                                                int rbs, int span_col, int mod, double keep) {
    const int lane = (int)(threadIdx.x & 63u);
    double sum_rx = keep;
    const unsigned long long wmask = __builtin_amdgcn_ballot_w64(mine);
    const int n_sp = __popcll(wmask);
    const int my_sp = __popcll(wmask & ((1ull << lane) - 1ull));   // my span's index in the wave
    const int team = lane >> 3, j = lane & 7;
    unsigned long long rest = wmask;
    for (int round = 0; round * 8 < n_sp; ++round) {
        // owners of this round's 8 spans (uniform), then mine by team
        int owner = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int o = rest ? __builtin_ctzll(rest) : 0;
            rest &= rest - 1ull;
            owner = team == k ? o : owner;
        }
        const bool on = round * 8 + team < n_sp;
        const int c0 = bperm(span_col, owner);
        const int n = bperm(rbs, owner);
        const int md = bperm(mod, owner);
        const double nom = nom_wave[owner];
        const double x0 = mi[md], kk = mi[4 + md];
        const double* __restrict__ sp = fad + (on ? c0 : 0);
        // a UE holding a single RB skips the MI average (channel_models.py:305): keep x itself
        const bool single = n == 1;
        const double sv = team_pairwise(n, j, on, [&](int i1, bool p1, int i2, bool p2) {
            const double x1 = (p1 ? sp[i1] : 0.0) + nom, x2 = (p2 ? sp[i2] : 0.0) + nom;  // (both fetches in flight)
            rs_d2 o;
            o.x = x1;
            o.y = x2;
            return single ? o : rs_sigmoid2(x1, x2, x0, kk);
        });
        const double got = bperm(sv, (my_sp & 7) << 3);
        if (mine && (my_sp >> 3) == round) sum_rx = got;
    }
    return sum_rx;
}

// Response sums of spans wider than RS_WIDE_SPAN RBs (an agent's allocation: ~180 RBs in one slice).  In a team such a
// span would keep 8 lanes busy with 12-23 sigmoids each in a row; here the WHOLE wave evaluates it, one RB per lane
// and pass, into the wave's LDS buffer `wmi`, and one team adds the buffer up in numpy's order.  Out of line on
// purpose: random-action batches never come here, and in line its registers cost the hot loop spills.
#ifndef RS_WIDE_SPAN
#define RS_WIDE_SPAN 64
#endif
#define RS_WIDE_MAX 192   // wider spans (193..256 RBs) take the team path
__device__ __noinline__ double wide_response(const double* mi, const double* fad, double* wmi, const double* nom_wave,
                                             bool mine, int rbs, int span_col, int mod) {
    const int lane = (int)(threadIdx.x & 63u);
    double out = 0.0;
    unsigned long long wm = __builtin_amdgcn_ballot_w64(mine);
    while (wm != 0ull) {
        const int ol = __builtin_ctzll(wm);  // the span's owner lane (uniform)
        wm &= wm - 1ull;
        const int n = __builtin_amdgcn_readlane(rbs, ol);
        const int c0 = __builtin_amdgcn_readlane(span_col, ol);
        const int md = __builtin_amdgcn_readlane(mod, ol);
        const double nom = nom_wave[ol];
        const double x0 = mi[md], kk = mi[4 + md];
        for (int k0 = 0; k0 < n; k0 += 128) {  // two RBs per lane and pass
            const int k1 = k0 + lane, k2 = k0 + 64 + lane;
            const double x1 = (k1 < n ? fad[c0 + k1] : 0.0) + nom, x2 = (k2 < n ? fad[c0 + k2] : 0.0) + nom;
            const rs_d2 sg = rs_sigmoid2(x1, x2, x0, kk);
            if (k1 < n) wmi[k1] = sg.x;
            if (k2 < n) wmi[k2] = sg.y;
        }
        __builtin_amdgcn_wave_barrier();
        const double sv = team_pairwise(n, lane & 7, lane < 8, [&](int i1, bool p1, int i2, bool p2) {
            rs_d2 o;
            o.x = p1 ? wmi[i1] : 0.0;
            o.y = p2 ? wmi[i2] : 0.0;
            return o;
        });
        const double got = bperm(sv, 0);
        if (lane == ol) out = got;
        __builtin_amdgcn_wave_barrier();
    }
    return out;
}

// ---- The reception TEST without the reception PROBABILITY (non-tracing instances).
// All the simulator does with MCSCodeset.response (channel_models.py:297-313) is one comparison, `u < p_rx`
// (slice_l1.py:219-224), and p_rx is a strictly increasing function of the MI sum S = sum_i sigmoid(snr_i):
//     u < p_rx  <=>  S > S*(u),   S* = n sigmoid_mod(s*),  s* = ref(mcs) + (B - ln((1-u)/u)) / A        (A, k > 0).
// So the kernel forms S~ ~ S and S*~ ~ S* in float32 on the transcendental unit (v_exp_f32 / v_log_f32 / v_rcp_f32:
// ~9 instructions per RB where the exact sigmoid is ~45, and the sum may take any order) and decides by them whenever
// they lie further apart than a guard band that covers both approximation errors several times over (RsDev.rx_band,
// derived in rs_api.hip: rx_fast_setup); a UE inside the band, or whose draw is within 1e-4 of 0 or 1, is evaluated
// exactly as before.  The outcome is the exact path's outcome in every case, so results stay bit-identical to the oracle;
// only ~2e-4 of the evaluations take the exact path (counted: rs_get_rx_stats).
//
// sigmoid of snr - x0 = (v + hi) + lo (v: the float32 sample; hi + lo: nominal SINR - x0 split into two floats, so that no
// rounding is relative to anything but the argument itself) with slope constant c1 = -k log2(e), loc = lo c1: absolute error below
// 4e-7 whatever the magnitudes (the argument's relative error of 2e-7 enters through t sigma'(t) <= 0.23; v_exp_f32 overflows
// to +inf and underflows to 0, which v_rcp_f32 turns into the limits 0 and 1)
__device__ __forceinline__ float fast_sigmoid(float v, float hi, float c1, float loc) {
    const float t2 = __builtin_fmaf(v + hi, c1, loc);
    const float e = __builtin_amdgcn_exp2f(t2);
    return __builtin_amdgcn_rcpf(1.0f + e);
}
typedef float rs_f4u __attribute__((ext_vector_type(4), aligned(4)));
// sum of the sigmoids of the (up to four) elements k0 .. k0 + 3 < n held in q: (s0 + s1) + (s2 + s3) in float32
__device__ __forceinline__ float fast_quad(rs_f4u q, int k0, int n, float hi, float c1, float loc) {
    const float s0 = fast_sigmoid(q.x, hi, c1, loc), s1 = fast_sigmoid(q.y, hi, c1, loc);
    const float s2 = fast_sigmoid(q.z, hi, c1, loc), s3 = fast_sigmoid(q.w, hi, c1, loc);
    return ((k0 < n ? s0 : 0.0f) + (k0 + 1 < n ? s1 : 0.0f)) + ((k0 + 2 < n ? s2 : 0.0f) + (k0 + 3 < n ? s3 : 0.0f));
}

// S~ of the spans flagged `mine` (2..64 RBs), dealt to 8-lane teams like team_response: lane j of a team takes elements
// 4 j .. 4 j + 3 and 32 + 4 j .. 35 + 4 j -- two 16-byte loads, both in flight; any order of summation will do
#define RS_FAST_WIDE 64
template <class F>
__device__ __forceinline__ double fast_team_sums(const double* mi, const float* c1s, const float* fad32, const double* nom_wave,
                                                 bool mine, int rbs, int span_col, int mod, F meanwhile) {
    const int lane = (int)(threadIdx.x & 63u);
    double out = 0.0;
    const unsigned long long wmask = __builtin_amdgcn_ballot_w64(mine);
    const int n_sp = __popcll(wmask);
    const int my_sp = __popcll(wmask & ((1ull << lane) - 1ull));
    const int team = lane >> 3, j = lane & 7;
    unsigned long long rest = wmask;
    for (int round = 0; round * 8 < n_sp; ++round) {
        int owner = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int o = rest ? __builtin_ctzll(rest) : 0;
            rest &= rest - 1ull;
            owner = team == k ? o : owner;
        }
        const bool on = round * 8 + team < n_sp;
        const int c0 = bperm(span_col, owner);
        const int n_o = bperm(rbs, owner);
        const int n = on ? n_o : 0;
        const int md = bperm(mod, owner);
        const float* __restrict__ sp = fad32 + (on ? c0 : 0) + 4 * j;
        rs_f4u qa = {0.0f, 0.0f, 0.0f, 0.0f}, qb = {0.0f, 0.0f, 0.0f, 0.0f};
        if (4 * j < n) qa = *(const rs_f4u*)sp;
        if (32 + 4 * j < n) qb = *(const rs_f4u*)(sp + 32);
        const double nomx = nom_wave[owner] - mi[md];
        const float c1 = c1s[md];
        if (round == 0) meanwhile();  // the caller's arithmetic that needs none of this, under the loads' latency
        const float hi = (float)nomx, loc = (float)(nomx - (double)hi) * c1;
        double acc = (double)fast_quad(qa, 4 * j, n, hi, c1, loc);
        if (wave_any(n > 32)) acc += (double)fast_quad(qb, 32 + 4 * j, n, hi, c1, loc);
        acc += dpp_d<DPP_XOR1>(acc);
        acc += dpp_d<DPP_XOR2>(acc);
        acc += dpp_d<DPP_HMIRROR>(acc);
        const double got = bperm(acc, (my_sp & 7) << 3);
        if (mine && (my_sp >> 3) == round) out = got;
    }
    return out;
}

// S~ of spans wider than RS_FAST_WIDE RBs (an agent's allocation): the whole wave, lane l elements 4 l .. 4 l + 3 of the span in
// one 16-byte load (n <= 256); the next span's load is issued before this one's arithmetic
__device__ __forceinline__ double fast_wide_sums(const double* mi, const float* c1s, const float* fad32, const double* nom_wave,
                                                 bool mine, int rbs, int span_col, int mod) {
    const int lane = (int)(threadIdx.x & 63u);
    double out = 0.0;
    unsigned long long wm = __builtin_amdgcn_ballot_w64(mine);
    rs_f4u q = {0.0f, 0.0f, 0.0f, 0.0f};
    {
        const int ol = __builtin_ctzll(wm);
        if (4 * lane < __builtin_amdgcn_readlane(rbs, ol)) q = *(const rs_f4u*)(fad32 + __builtin_amdgcn_readlane(span_col, ol) + 4 * lane);
    }
    while (wm != 0ull) {
        const int ol = __builtin_ctzll(wm);  // the span's owner lane (uniform)
        wm &= wm - 1ull;
        const int n = __builtin_amdgcn_readlane(rbs, ol);
        const int md = __builtin_amdgcn_readlane(mod, ol);
        rs_f4u qn = {0.0f, 0.0f, 0.0f, 0.0f};
        if (wm != 0ull) {
            const int on_ = __builtin_ctzll(wm);
            if (4 * lane < __builtin_amdgcn_readlane(rbs, on_)) qn = *(const rs_f4u*)(fad32 + __builtin_amdgcn_readlane(span_col, on_) + 4 * lane);
        }
        const double nomx = nom_wave[ol] - mi[md];
        const float c1 = c1s[md];
        const float hi = (float)nomx, loc = (float)(nomx - (double)hi) * c1;
        double acc = (double)fast_quad(q, 4 * lane, n, hi, c1, loc);
        acc += dpp_d<DPP_XOR1>(acc);
        acc += dpp_d<DPP_XOR2>(acc);
        acc += dpp_d<DPP_HMIRROR>(acc);
        acc += dpp_d<DPP_MIRROR>(acc);
        acc += bperm(acc, lane ^ 16);
        acc += bperm(acc, lane ^ 32);
        if (lane == ol) out = acc;
        q = qn;
    }
    return out;
}

// 5 waves per SIMD for the production instance (the whole 4096-replica batch is then co-resident).  The tracing
// instances are test tooling: they keep the register budget of 3 waves per SIMD.
// BLOCK: the contested PF allocation may hand out RB pairs in block rounds (wide slices); without it the instance
// carries only the trip loop (fewer live registers: the whole point at 5 waves per SIMD).
// FDIV: pf_b * bits / slot_length by the verified reciprocal form (rs_create checks every reachable `bits`; RsDev.pf_div_fast)
// -- a template parameter since round 4: as a run-time flag it put a branch into every one of the four shares a block-round
// iteration forms and kept their (independent) chains from being interleaved.
template <int G, bool TRACE, bool BLOCK, bool FDIV>
__global__ __launch_bounds__(256, (G == 16 && !TRACE) ? RS_OCC : RS_OCC_OTHER) void embb_step_kernel(StepArgs A) {
    static_assert(G == 8 || G == 16 || G == 32, "lanes per task");
    constexpr int TPB = 256 / G;                     // tasks per block
    constexpr int TPW = 64 / G;                      // tasks per wave
    constexpr int CH = 12;                           // slots per chunk of channel estimates (table row = 24 B)
    // Cold per-UE state lives in LDS (one slot per thread, conflict-free), so that the hot loop keeps few
    // enough VGPRs for 5 resident waves per SIMD.  Timers are absolute slot numbers; `evt_at` (a VGPR)
    // is the earliest of them, so these arrays are touched only in slots where something happens.
    __shared__ unsigned short L_burst[RS_BURSTS][256];  // VBR burst end times (rs_burst_code), 0 = free
    __shared__ int L_hold[256];              // departure time
    __shared__ int L_uvbr[256];              // next burst arrival of the UE's VBR source
    __shared__ unsigned L_serial[256], L_ctr[256];
    __shared__ int L_acc_traf[256], L_acc_bits[256], L_acc_prbs[256];
    __shared__ double L_nom[256];            // nominal SINR
    __shared__ short T_esnr[256][CH];        // round(mean(snr)) of the UE in the CH slots of the current chunk
    __shared__ int L_lut[RS_LUT_MAX];        // e_snr -> modulation << 24 | mcs << 16 | rate (mcs_rate_vs_error)
    __shared__ double L_ref[32];             // MCS reference SNR (estimate_rx_prob)
    __shared__ double L_mi[8];               // logistic MI curves: x0 of the three modulations, pad, k of the three, pad
    __shared__ float L_c1[4];                // -k log2(e) of the three modulations in float32 (the reception test by guard band)
    __shared__ int L_task[TPB][4];           // per task: cbr_at, vbr_at, slice draw counter, next UE serial
    __shared__ double W_mi[4][RS_WIDE_MAX];  // per wave: MI values of one WIDE span (response of 65..192 RBs); sized so
                                             // that a block stays within 25 LDS granules of 1280 B: 5 blocks per CU
    const RsDev* __restrict__ D = A.D;
    const int tid = (int)threadIdx.x;
    if (tid < RS_LUT_MAX) L_lut[tid] = tid < D->lut_n ? ((D->mcs_mod[D->lut_mcs[tid]] << 24) | (D->lut_mcs[tid] << 16) | D->lut_rate[tid]) : 0;
    if (tid >= 64 && tid < 96) L_ref[tid - 64] = D->mcs_ref[tid - 64];
    if (tid >= 96 && tid < 104) L_mi[tid - 96] = (tid & 3) == 3 ? 0.0 : (tid < 100 ? D->mi_x0[tid - 96] : D->mi_k[tid - 100]);
    if (tid >= 104 && tid < 108) L_c1[tid - 104] = tid < 107 ? D->rx_c1[tid - 104] : 0.0f;
    __syncthreads();
    const RsState& S = *A.S;
    const int clock0 = (int)A.run[0];
    const int lane = (int)(threadIdx.x & 63u);
    const int gl = lane & (G - 1);       // UE index owned by this lane
    const int gbase = lane & ~(G - 1);   // first lane of my group inside the wave
    const int tb = tid - gl;             // first thread of my group in the block
    const int wb = tid & ~63;            // first thread of my wave in the block
    const int tq = tid / G;              // my group's index in the block
    const int n_tasks = D->n_envs * D->n_embb;
    int task = A.spread ? (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6) : (int)blockIdx.x * TPB + (int)(threadIdx.x / G);
    const int n_mine = A.order_cnt >= 0 ? A.order_cnt : n_tasks;  // tasks of this launch
    const bool in_range = task < n_mine && (!A.spread || gbase == 0);
    if (!in_range) task = n_mine - 1;
    if (A.order) task = A.order[A.order_off + task];
    const bool selected = in_range && (!A.replay || A.redo[task] != 0);
    if (!wave_any(selected)) return;  // replay launch: nothing flagged in this wave
    const int rep = task / D->n_embb;
    const int sl = task - rep * D->n_embb;
    const int n_slices = D->n_slices;
    const int P = D->P;
    const double slot_len = D->slot_length;
    const double pf_a = D->pf_a, pf_b = D->pf_b;
    const double slot_rc = D->slot_rc;
    // (the plain 16-lane instance keeps the run-time flag: its trip loop has one share per iteration, and it measured 1.5 %
    // slower with the constant -- a different schedule at the same 96 registers)
    const bool pf_div_fast = BLOCK ? FDIV : (D->pf_div_fast != 0);
    const int gran = D->gran;
    const bool has_nan = D->has_nan != 0;
    const bool rx_fast = D->rx_band > 0.0;  // the reception test by guard band is available for this configuration
    const double est_band = D->est_band;
    const bool est_fast = !TRACE && est_band > 0.0 && A.fps != nullptr;
    const int co0 = D->col_off[0], co1 = D->col_off[1], co2 = D->col_off[2];
    const int T0 = D->T[0], T1 = D->T[1], T2 = D->T[2];
    const int fo0 = (int)D->fad_off[0], fo1 = (int)D->fad_off[1], fo2 = (int)D->fad_off[2];  // < 2^31 (rs_load_fading)
    const int vo0 = (int)D->valid_off[0], vo1 = (int)D->valid_off[1], vo2 = (int)D->valid_off[2];

    // set_prbs (node_b.py:71-74): contiguous ranges in slice order
    int prb_lo = 0;
    for (int q = 0; q < sl; ++q) prb_lo += A.actions[rep * n_slices + q];
    int n_prb = A.actions[rep * n_slices + sl];

    // ---- load persistent state
    int n_ue = selected ? S.t_n_ue[task] : 0;
    // a G < 32 instance gives up a task that has more UEs than lanes: nothing is written back and the
    // G = 32 replay redoes the whole step for it
    bool aborted = G < 32 && selected && n_ue > G;
    bool valid = selected && !aborted;
    if (!valid) { n_prb = 0; n_ue = 0; }
    int slice_evt;  // earlier of the slice's next CBR / VBR arrival checks (the pair itself sits in L_task)
    {
        const int cbr_at = S.t_cbr_at[task], vbr_at = S.t_vbr_at[task];
        if (gl == 0) {
            L_task[tq][0] = cbr_at;
            L_task[tq][1] = vbr_at;
            L_task[tq][2] = (int)S.t_ctr[task];
            L_task[tq][3] = (int)S.t_serial[task];
        }
        slice_evt = valid ? (cbr_at < vbr_at ? cbr_at : vbr_at) : RS_NEVER;
    }
    const uint64_t seed = S.seeds[rep];
    const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
    int err = 0;

    const int ui = task * RS_GROUP + gl;  // HBM layout keeps 32 UE slots per task whatever G is
    bool active = gl < n_ue;
    double queue = 0.0, th = 0.0;
    int e_snr = 0, findex = 0, ue_bits = 0, ue_prbs = 0, flags = 0;
    int n_act = 0;          // VBR bursts still running when the slot begins
    int evt_at = RS_NEVER;  // earliest of: departure, next burst arrival, next burst end
    {
        int hold_at = RS_NEVER, uvbr_at = RS_NEVER;
        unsigned uctr = 0u, userial = 0u;
        double nominal = 0.0;
        if (active) {
            queue = S.u_queue[ui];
            th = S.u_th[ui];
            nominal = S.u_nominal[ui];
            hold_at = S.u_hold_at[ui];
            e_snr = S.u_e_snr[ui];
            findex = S.u_findex[ui];
            ue_bits = S.u_bits[ui];
            ue_prbs = S.u_prbs[ui];
            uvbr_at = S.u_vbr_at[ui];
            uctr = S.u_ctr[ui];
            userial = S.u_serial[ui];
            flags = S.u_flags[ui];
        }
        evt_at = hold_at < uvbr_at ? hold_at : uvbr_at;
#pragma unroll
        for (int k = 0; k < RS_BURSTS; ++k) {
            const unsigned e = active ? S.u_burst[(task * RS_BURSTS + k) * RS_GROUP + gl] : 0u;
            L_burst[k][tid] = (unsigned short)e;
            if (e != 0u) {  // occupied entries are always still running (they are freed in the slot they end)
                n_act += 1;
                const int endt = clock0 + rs_burst_rel(e, clock0);
                evt_at = endt < evt_at ? endt : evt_at;
            }
        }
        n_act += (flags >> 8) & 0xff;  // bursts that never end (Q5) are only counted
        L_hold[tid] = hold_at;
        L_uvbr[tid] = uvbr_at;
        L_serial[tid] = userial;
        L_ctr[tid] = uctr;
        L_nom[tid] = nominal;
        L_acc_traf[tid] = 0;
        L_acc_bits[tid] = 0;
        L_acc_prbs[tid] = 0;
    }

    // per-step accumulators: lane k (<10) holds info[k] of this slice (slice_ran.py:270-273)
    double infok = 0.0;
    double infok_hi = 0.0;  // G = 8 only: info[8], info[9] in lanes 0, 1
    // per-UE running sums (traffic, th(bits), prb) live in L_acc_*; flush() folds them into info[] by class
    // per-step statistics of the task in one register: UE-slots (bits 0-11, <= 50 x 32), scheduled slots (bits
    // 12-17, <= 50), PF trips (bits 18-31, <= 50 x 128); the fading-sample and RB-pair counters follow from them
    unsigned stat = 0u;
    unsigned n_rx_tests = 0u;  // reception tests of this wave in this step (wave-uniform; rs_get_rx_stats)
    {
        // The launch ends when its slowest wave ends, and all waves of the batch are co-resident, so waves
        // whose tasks were expensive in the previous step (persistent backlog -> long contested PF loops)
        // get issue priority over their lighter neighbours on the SIMD for the whole step.
        if (A.order) {
            // launch order = cost rank (rs_order.hip): the heaviest tenth of the waves, the next fifth, ...
            const unsigned w = blockIdx.x * 4u + (threadIdx.x >> 6), nw = gridDim.x * 4u;
            if (w * RS_PRIO_A < nw) __builtin_amdgcn_s_setprio(3);
            else if (w * RS_PRIO_B < nw) __builtin_amdgcn_s_setprio(2);
            else if (w * RS_PRIO_C < nw) __builtin_amdgcn_s_setprio(1);
        } else {
            int cost = valid ? S.t_cost[task] : 0;
#pragma unroll
            for (int d_ = G; d_ < 64; d_ <<= 1) {
                const int o = bperm(cost, lane ^ d_);
                cost = o > cost ? o : cost;
            }
            const int c0_ = __builtin_amdgcn_readfirstlane(cost);
            if (c0_ > 1200) __builtin_amdgcn_s_setprio(3);
            else if (c0_ > 600) __builtin_amdgcn_s_setprio(2);
            else if (c0_ > 300) __builtin_amdgcn_s_setprio(1);
        }
    }

    auto flush = [&]() {
        // SliceRANeMBB.update_info's three integer-valued sums (slice_ran.py:282-285,296-299):
        // exact in f64 whatever the order, so they are accumulated per UE and folded here.
        const bool is_vbr = (flags & 1) != 0;
        const int acc_traffic = L_acc_traf[tid], acc_bits = L_acc_bits[tid], acc_prbs = L_acc_prbs[tid];
        int t_c = group_sum<G>((active && !is_vbr) ? acc_traffic : 0);
        int t_v = group_sum<G>((active && is_vbr) ? acc_traffic : 0);
        int b_c = group_sum<G>((active && !is_vbr) ? acc_bits : 0);
        int b_v = group_sum<G>((active && is_vbr) ? acc_bits : 0);
        int p_c = group_sum<G>((active && !is_vbr) ? acc_prbs : 0);
        int p_v = group_sum<G>((active && is_vbr) ? acc_prbs : 0);
        double add = 0.0;
        add = gl == 0 ? (double)t_c : add;
        add = gl == 1 ? (double)b_c : add;
        add = gl == 2 ? (double)p_c : add;
        add = gl == 5 ? (double)t_v : add;
        add = gl == 6 ? (double)b_v : add;
        add = gl == 7 ? (double)p_v : add;
        infok += add;
        L_acc_traf[tid] = 0;
        L_acc_bits[tid] = 0;
        L_acc_prbs[tid] = 0;
    };

    // pf_b * bits / slot_length of the throughput averages (schedulers.py:54, slice_ran.py:55)
    auto pf_share = [&](int b) -> double {
        const double xb = pf_b * (double)b;
        if (pf_div_fast) {  // == xb / slot_len for every `bits` a slot can reach (checked by rs_create)
            const double q0 = xb * slot_rc;
            return __builtin_fma(__builtin_fma(-q0, slot_len, xb), slot_rc, q0);
        }
        return xb / slot_len;
    };

    const int slots = D->slots;
    const int n_pairs_full = n_prb / gran;  // RB pairs of full size in this slice
    int t0 = 0, CL = 0;                     // current chunk of channel estimates: slots [t0, t0 + CL)
    SEC_DECL
    // Dynamic issue priority.  All waves of the batch are co-resident (five per SIMD) and the launch ends with its
    // slowest wave, whose cost last step's statistics predict poorly.  So every wave paces itself against the mean
    // pace of the previous launch: behind schedule -> it issues ahead of its SIMD neighbours, ahead -> it yields.
    // (Timing only; no result depends on it.)
    unsigned long long pace_ref = 0ull;
    const unsigned long long pace_t0 = __builtin_amdgcn_s_memtime();
#ifdef RS_WAVE_LOG
    if ((tid & 63) == 0) ((unsigned long long*)A.sections)[16 + 16 * (size_t)(blockIdx.x * 4u + (threadIdx.x >> 6)) + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (RS_DYN_PRIO && A.pace) {
        pace_ref = __builtin_nontemporal_load(&A.pace[2]);
        if (blockIdx.x == 0 && tid == 0) {  // nobody adds to [0], [1] before the end of its 50 slots
            const unsigned long long s_ = A.pace[0], c_ = A.pace[1];
            if (c_ != 0ull) {
                A.pace[2] = (unsigned long long)((double)s_ / (double)c_);
                A.pace[0] = 0ull;
                A.pace[1] = 0ull;
            }
        }
        pace_ref = __builtin_amdgcn_readfirstlane((unsigned)pace_ref);  // < 2^32 cycles per slot
    }
    for (int t = 0; t < slots; ++t) {
        // (the slot's own copy of the thread index, opaque to the optimiser: the addresses of this thread's entries in the
        // dozen LDS arrays are then formed where they are used -- one base register and the array's offset in the
        // instruction -- instead of being hoisted out of the loop, one register each, and spilled: the hottest spill slots
        // of the kernel held nothing but such addresses)
        int lt = tid;
        asm volatile("" : "+v"(lt));
        if (RS_DYN_PRIO && pace_ref != 0ull && t >= 2) {
            const unsigned long long el = (__builtin_amdgcn_s_memtime() - pace_t0) * 32ull;
            const unsigned long long due = (unsigned long long)t * pace_ref;
            if (el > due * (BLOCK ? RS_PACE_3B : RS_PACE_3)) __builtin_amdgcn_s_setprio(3);       // in 32nds of the reference pace
            else if (el > due * (BLOCK ? RS_PACE_2B : RS_PACE_2)) __builtin_amdgcn_s_setprio(2);
            else if (el > due * (BLOCK ? RS_PACE_1B : RS_PACE_1)) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        const int now = clock0 + t + 1;
        const int slot_counter = t + 1;
        const bool chunk_start = t == t0 + CL;  // wave-uniform
        if (chunk_start) {
            t0 = t;
            CL = slots - t < CH ? slots - t : CH;
        }
        const int tt0 = t - t0;
        const int n_ue_before = n_ue;
        int n_new = 0;

        // ================= SliceRANeMBB.slot: arrivals (slice_ran.py:205-249)
        if (wave_any(slice_evt == now)) {
            const bool fire = slice_evt == now;
            int cbr_at = L_task[tq][0], vbr_at = L_task[tq][1];
            uint32_t sl_ctr = (uint32_t)L_task[tq][2], next_serial = (uint32_t)L_task[tq][3];
            const bool cbr_fire = fire && cbr_at == now;
            const bool vbr_fire = fire && vbr_at == now;
            int n_pend = 0;
            int pend_type0 = 0, pend_type1 = 0;
            if (wave_any(cbr_fire)) flush();  // cbr_cac reads this step's running sums
            double i1 = bperm(infok, gbase + 1), i2 = bperm(infok, gbase + 2);
            if (cbr_fire) {
                rs_stream st = {key0, key1, (uint32_t)sl, 0u, sl_ctr};
                double ia = rs_stream_exponential(&st, D->cbr_ia_scale);
                sl_ctr = st.ctr;
                cbr_at = now + 1 + rint_slots(ia, slot_len);
                // cbr_cac (slice_ran.py:195-203)
                int cslots = slot_counter > 1 ? slot_counter : 1;
                double time = cslots * slot_len;
                double c_prb = i2 / cslots;
                double c_th = i1 / time;
                if (!(c_prb >= D->sla[1] || c_th >= D->sla[0])) {
                    pend_type0 = 0;
                    n_pend = 1;
                }
            }
            if (vbr_fire) {
                rs_stream st = {key0, key1, (uint32_t)sl, 0u, sl_ctr};
                double ia = rs_stream_exponential(&st, D->vbr_ia_scale);
                sl_ctr = st.ctr;
                vbr_at = now + 1 + rint_slots(ia, slot_len);
                if (n_pend == 0) pend_type0 = 1; else pend_type1 = 1;
                n_pend += 1;
            }
            if (n_ue + n_pend > G) {
                if (G == 32) {
                    err |= 1;  // RS_EOVERFLOW: UE capacity
                    n_pend = G - n_ue;
                } else {
                    // does not fit this instance: drop the task here, the G = 32 replay redoes the step
                    aborted = true;
                    valid = false;
                    active = false;
                    n_pend = 0;
                    n_ue = 0;
                    n_prb = 0;
                    cbr_at = RS_NEVER;
                    vbr_at = RS_NEVER;
                    evt_at = RS_NEVER;
                    queue = 0.0;
                }
            }
            const bool is_new = gl >= n_ue && gl < n_ue + n_pend;
            if (is_new) {
                const int k = gl - n_ue;
                const int type = k == 0 ? pend_type0 : pend_type1;
                const unsigned userial = next_serial + (uint32_t)k;
                rs_stream st = {key0, key1, (uint32_t)sl, userial, 0u};
                queue = 0.0; th = 0.0; e_snr = 0; ue_bits = 0; ue_prbs = 0;
                L_acc_traf[lt] = 0; L_acc_bits[lt] = 0; L_acc_prbs[lt] = 0;
#pragma unroll
                for (int q = 0; q < RS_BURSTS; ++q) L_burst[q][lt] = 0;
                n_act = 0;
                int uvbr_at = RS_NEVER;
                if (type == 1) {  // VbrSource.__init__ (traffic_generators.py:62-68)
                    int v = (int)RS_RINT(rs_stream_exponential(&st, D->vbr_inter));
                    uvbr_at = v >= 1 ? now + v - 1 : RS_NEVER;  // Q5: 0 never fires
                }
                double hold = rs_stream_exponential(&st, type == 0 ? D->cbr_hold_scale : D->vbr_hold_scale);
                int hv = rint_slots(hold, slot_len);
                const int hold_at = hv >= 1 ? now + hv - 1 : RS_NEVER;    // Q5
                int ftype = 0, fstep = 1;
                double nominal = 0.0;
                findex = 0;
                if (hold_at != now) {  // Q13: a one-slot holding time never joins the slice
                    // SINRSelectiveFading.insert_user (channel_models.py:163-169)
                    ftype = (int)rs_stream_integers(&st, RS_N_TRACES);
                    findex = (int)rs_stream_integers(&st, sel3(ftype, T0, T1, T2));
                    fstep = rs_stream_pm1(&st);
                    const MacroCell mc = macro_cell_draw(D, st);
                    nominal = mc.x;
                    st.ctr = (uint32_t)mc.y;
                }
                L_hold[lt] = hold_at;
                L_uvbr[lt] = uvbr_at;
                L_serial[lt] = userial;
                L_ctr[lt] = st.ctr;
                L_nom[lt] = nominal;
                evt_at = hold_at < uvbr_at ? hold_at : uvbr_at;
                flags = type | (ftype << 1) | ((fstep > 0 ? 1 : 0) << 3);
                active = true;
            }
            if (fire) {
                n_ue += n_pend;
                n_new = n_pend;
                next_serial += (uint32_t)n_pend;
                if (gl == 0) {
                    L_task[tq][0] = cbr_at;
                    L_task[tq][1] = vbr_at;
                    L_task[tq][2] = (int)sl_ctr;
                    L_task[tq][3] = (int)next_serial;
                }
                slice_evt = valid ? (cbr_at < vbr_at ? cbr_at : vbr_at) : RS_NEVER;
            }
        }

        SEC_MARK(9)
        // ================= channel estimates of the chunk's slots (get_snr + estimate_snr: channel_models.py:171-191,
        // slice_ran.py:43-45), at the start of a chunk for every UE of the wave's tasks, and for a UE that joins
        // mid-chunk from its arrival slot on.  One (UE, slot) item per lane and round.
        if (chunk_start || wave_any(n_new > 0)) {
            const int span = CL - tt0;                                // slots still ahead in the chunk (wave-uniform)
            const int row0 = chunk_start ? 0 : n_ue_before;           // first UE row to estimate (group-uniform)
            const int cnt = chunk_start ? n_ue : n_new;
            const int items = (valid && n_prb > 0) ? cnt * span : 0;  // group-uniform
            int pre[TPW + 1];
            pre[0] = 0;
#pragma unroll
            for (int g = 0; g < TPW; ++g) pre[g + 1] = pre[g] + __builtin_amdgcn_readlane(items, g * G);
            const int total = pre[TPW];
            const int Mdiv = (1048576 + span - 1) / span;            // loc / span == (loc * Mdiv) >> 20 for loc < 2^11
            for (int base = 0; base < total; base += 64) {
                const int idx = base + lane;
                const bool have = idx < total;
                int g = 0, pg = 0;
#pragma unroll
                for (int k = 1; k < TPW; ++k)
                    if (idx >= pre[k]) { g = k; pg = pre[k]; }
                const int loc = have ? idx - pg : 0;
                const int r = (loc * Mdiv) >> 20;
                const int tt = tt0 + (loc - r * span);
                const int gl0 = g * G;                               // first lane of the item's group in the wave
                const int row = bperm(row0, gl0) + r;
                const int src = gl0 + (have ? row : 0);              // lane that owns the item's UE
                const int f0 = bperm(findex, src), fl = bperm(flags, src);
                const int np = bperm(n_prb, gl0), plo = bperm(prb_lo, gl0);
                const uint32_t k0 = bperm(key0, gl0), k1 = bperm(key1, gl0);
                const int isl = bperm(sl, gl0);
                const double nom = L_nom[wb + src];
                const int dep = L_hold[wb + src];
                const int adv = tt - tt0 + 1;                        // walker steps from the owner's current state
                // a UE that departs in slot `dep` is extracted before its get_snr of that slot
                const bool on = have && clock0 + t0 + tt + 1 < dep;
                const int ftype = (fl >> 1) & 3;
                int fs = (fl & 8) ? 1 : -1;
                const int Tn = sel3(ftype, T0, T1, T2);
                int f = f0 + fs * adv;
                const bool straight = !has_nan && f >= 0 && f < Tn;
                if (wave_any(on && !straight)) {
                    if (on && !straight) {
                        const uint32_t ser = L_serial[wb + src];
                        const uint8_t* vcol = A.fad_valid + sel3(ftype, vo0, vo1, vo2);
                        f = f0;
                        for (int a = 0; a < adv; ++a)
                            walker_advance(f, fs, Tn, has_nan, vcol, k0, k1, (uint32_t)isl, ser, (uint32_t)(now + a));
                    }
                }
                // round(np.mean(snr)) (half-to-even, Q7) from the column's prefix sums -- two loads whatever the slice's width --
                // unless the mean lies within est_band of a half-integer (rs_api.hip: upload_fading); then, as in the tracing
                // instances' every estimate, numpy's pairwise sum of the samples themselves decides
                bool todo = on;
                if (est_fast) {
                    const double* __restrict__ ps = A.fps + (on ? (sel3(ftype, co0, co1, co2) + f) * (P + 1) + plo : 0);
                    const double pa = ps[0], pb = ps[on ? np : 0];
                    const double mean = (pb - pa) / (double)np + nom;
                    const double rr = RS_RINT(mean);
                    const bool sure = __builtin_fabs(mean - rr) < 0.5 - est_band && __builtin_fabs(mean) < 3.0e4;
                    if (on && sure) T_esnr[wb + src][tt] = (short)(int)rr;
                    todo = on && !sure;
                }
                if (wave_any(todo)) {
                    const double* __restrict__ colp = A.fad + (todo ? sel3(ftype, fo0, fo1, fo2) + f * P + plo : 0);
                    const double sum = lane_pairwise(np, todo, [&](int i) { return colp[i] + nom; });
                    if (todo) T_esnr[wb + src][tt] = (short)(int)RS_RINT(sum / (double)np);  // round(np.mean(...)): half-to-even (Q7)
                }
            }
        }

        SEC_MARK(7)
        // ================= per-UE timer events: departures, VBR burst ends and arrivals
        int n_cur = n_act;  // bursts that emit in THIS slot (an arrival of this slot starts emitting next slot)
        if (wave_any(active && evt_at == now)) {
            // ---- departures (slice_ran.py:251-261) + extract_users (slice_l1.py:187-191)
            const bool depart = active && evt_at == now && L_hold[lt] == now;
            if (wave_any(depart)) {
                flush();
                const unsigned keep = group_ballot<G>(active && !depart, gbase);
                const int n_keep = __popc(keep);
                const int su = kth_set_bit(keep, gl);
                const int src = gbase + su;
                queue = bperm(queue, src);
                th = bperm(th, src);
                e_snr = bperm(e_snr, src);
                findex = bperm(findex, src);
                ue_bits = bperm(ue_bits, src);
                ue_prbs = bperm(ue_prbs, src);
                flags = bperm(flags, src);
                n_act = bperm(n_act, src);
                evt_at = bperm(evt_at, src);
                // LDS-resident fields: every lane reads its source slot, then all write (in-order LDS)
                const int st_ = tb + su;
                const int m_hold = L_hold[st_], m_uvbr = L_uvbr[st_];
                const unsigned m_ser = L_serial[st_], m_ctr = L_ctr[st_];
                const double m_nom = L_nom[st_];
                unsigned short m_b[RS_BURSTS];
#pragma unroll
                for (int k = 0; k < RS_BURSTS; ++k) m_b[k] = L_burst[k][st_];
                int m_e[CH / 2];
#pragma unroll
                for (int k = 0; k < CH / 2; ++k) m_e[k] = ((const int*)T_esnr[st_])[k];
                __builtin_amdgcn_wave_barrier();
                n_ue = n_keep;
                active = gl < n_ue;
                L_hold[lt] = active ? m_hold : RS_NEVER;
                L_uvbr[lt] = active ? m_uvbr : RS_NEVER;
                L_serial[lt] = active ? m_ser : 0u;
                L_ctr[lt] = m_ctr;
                L_nom[lt] = m_nom;
#pragma unroll
                for (int k = 0; k < RS_BURSTS; ++k) L_burst[k][lt] = m_b[k];
#pragma unroll
                for (int k = 0; k < CH / 2; ++k) ((int*)T_esnr[lt])[k] = m_e[k];
                if (!active) { evt_at = RS_NEVER; n_act = 0; }
                n_cur = n_act;
            }
            // ---- VbrSource.step events (traffic_generators.py:70-99) on absolute end times
            if (active && evt_at == now) {
                int cnt = (flags >> 8) & 0xff, nxt = RS_NEVER;  // the never-ending bursts (Q5) always emit
                unsigned freek = RS_BURSTS;                      // a free entry for a burst that may start now
#pragma unroll
                for (int k = 0; k < RS_BURSTS; ++k) {
                    const unsigned e = L_burst[k][lt];
                    const int rel = e != 0u ? rs_burst_rel(e, now) : 0;
                    if (e != 0u && rel <= 0) L_burst[k][lt] = 0;  // ends exactly now: dropped without emitting
                    if (rel > 0) {
                        cnt += 1;
                        nxt = now + rel < nxt ? now + rel : nxt;
                    } else if (freek == RS_BURSTS) {
                        freek = (unsigned)k;
                    }
                }
                n_cur = cnt;
                int uvbr_at = L_uvbr[lt];
                if (uvbr_at == now) {
                    rs_stream st = {key0, key1, (uint32_t)sl, L_serial[lt], L_ctr[lt]};
                    const int d = (int)RS_RINT(rs_stream_exponential(&st, D->vbr_b_size));
                    const int v = (int)RS_RINT(rs_stream_exponential(&st, D->vbr_inter));
                    L_ctr[lt] = st.ctr;
                    if (d < 1) {  // Q5: a duration that rounds to 0 never counts down to 0: the burst emits for ever
                        if (((flags >> 8) & 0xff) == 0xff) err |= 2;
                        else flags += 1 << 8;
                        cnt += 1;
                    } else if (freek == RS_BURSTS || d >= RS_BURST_MAX_LEN) {
                        err |= 2;  // RS_EOVERFLOW: more than RS_BURSTS bursts running, or one longer than the 15-bit clock can hold
                    } else {
#pragma unroll
                        for (int k = 0; k < RS_BURSTS; ++k)
                            if ((unsigned)k == freek) L_burst[k][lt] = (unsigned short)rs_burst_code(now + d);
                        cnt += 1;
                        nxt = now + d < nxt ? now + d : nxt;
                    }
                    uvbr_at = v >= 1 ? now + v : RS_NEVER;
                    L_uvbr[lt] = uvbr_at;
                }
                n_act = cnt;
                const int h = L_hold[lt];
                int e = h < uvbr_at ? h : uvbr_at;
                evt_at = e < nxt ? e : nxt;
            }
        }

        const bool is_vbr = (flags & 1) != 0;
        SEC_MARK(0)

        // ================= UE.traffic_step (slice_ran.py:47-49)
        if (active) {
            const double new_bits = is_vbr ? (double)n_cur * D->vbr_p_size : D->cbr_bits;
            queue += new_bits;
            atomicAdd(&L_acc_traf[lt], (int)new_bits);
        }
        const bool any_queue = group_ballot<G>(active && queue > 0.0, gbase) != 0u;

        SEC_MARK(1)
        // ================= channel: this slot's walker step and the estimate tabulated for it (Q3: with no PRBs the
        // walker does not move and e_snr is stale)
        int col = 0;  // element offset of this UE's fading column
        const int ftype = (flags >> 1) & 3;
        if (n_prb > 0 && active) {
            int fstep = (flags & 8) ? 1 : -1;
            walker_advance(findex, fstep, sel3(ftype, T0, T1, T2), has_nan, A.fad_valid + sel3(ftype, vo0, vo1, vo2), key0,
                           key1, (uint32_t)sl, L_serial[lt], (uint32_t)now);
            flags = (flags & ~8) | ((fstep > 0 ? 1 : 0) << 3);
            col = sel3(ftype, fo0, fo1, fo2) + findex * P;
            e_snr = T_esnr[lt][tt0];
        }
        stat += (unsigned)n_ue;

        SEC_MARK(2)
        // ================= scheduling (slice_l1.py:215-224)
        const bool sched = valid && any_queue && n_prb > 0;
        double p_rx = 0.0;
        if (wave_any(sched)) {
            // ---- ProportionalFair.allocate (schedulers.py:21-76)
            int li = e_snr - D->lut_lo;
            li = li < 0 ? 0 : (li >= D->lut_n ? D->lut_n - 1 : li);
            const int lut = L_lut[li];
            const int mcs = (lut >> 16) & 0xff;
            const int mod = lut >> 24;
            const int rate = lut & 0xffff;
            const double rate_d = (double)rate;
            int q = active ? (int)(queue < 1073741824.0 ? queue : 1073741824.0) : 0;
            double thl = th > 1.0 ? th : 1.0;
            int rbs = 0, bits = 0;
            int r = 0;
            {
                // Under-loaded slot (the common case): while any UE has data an RB pair goes to a UE with data (a
                // positive metric beats the zeros), and a UE that drains stops competing, so UE u is granted exactly
                // k_u = ceil(q_u / (gran * rate_u)) pairs WHATEVER the order in which the argmax serves them,
                // provided all of those grants land on full pairs: sum(k_u) <= floor(n_prb / gran).  Then
                // rbs_u = k_u * gran, every queue is sent whole, and the idle remainder goes to UE 0 (Q4).  The
                // local throughput averages that decide the order are discarded by the reference
                // (schedulers.py:64-76), so nothing else is needed.
                const int per_it = gran * rate;
                const int k_u = (active && q > 0) ? (int)((double)(q + per_it - 1) / (double)per_it) : 0;
                const int need = group_sum<G>(k_u);
                if (sched && need <= n_pairs_full) {
                    rbs = k_u * gran;
                    bits = k_u > 0 ? q : 0;
                    if (gl == 0) rbs += n_prb - need * gran;
                    r = n_prb;
                } else {
                    // over-loaded, but a single UE holds all the data: it wins every pair (k_u > floor(n_prb / gran), so
                    // it is not drained before the last, possibly single-RB, pair)
                    const unsigned nz = group_ballot<G>(sched && k_u > 0, gbase);
                    if (sched && (nz & (nz - 1u)) == 0u) {
                        if (k_u > 0) {
                            const int cap_bits = n_prb * rate;
                            rbs = n_prb;
                            bits = q < cap_bits ? q : cap_bits;
                        }
                        r = n_prb;
                    }
                }
            }
            SEC_MARK(8)
            if (wave_any(sched && r < n_prb)) {
                // Contested slot.  The reference hands out one RB pair per iteration to the first UE of maximal
                // metric m_u = rate_u / th_u and updates only that UE's local average (schedulers.py:44-62).  A
                // UE's sequence of head metrics h_u0, h_u1, ... (after 0, 1, ... pairs) therefore depends on
                // nobody else, and the greedy loop is a merge of those sequences: the pair a UE takes with j
                // pairs already in hand comes in the order of its key k_uj = min(h_u0..h_uj) (descending, equal
                // keys by UE index).  Pick any target pair g* = (u*, j*) with key L: the pairs handed out up to
                // and including g* are u*'s first j* + 1 and, for every other UE, those with k_uj > L, or = L
                // and u < u* -- a prefix of its sequence, which it can count ALONE.  (At the moment u* shows the
                // head that sets L it is the first maximum, so every other head is below (L, u*) and stays there
                // while u* runs on to j*; and a UE that has ever shown a head below (L, u*) before that moment
                // would not have been served past it.)  Two kinds of rounds use this:
                //  - a TRIP: the target is the runner-up's current head, so only the leader has pairs above it
                //    and runs the reference loop alone until it stops being the argmax;
                //  - a BLOCK round (BLOCK instances, slices of >= RS_BLOCK_PAIRS pairs while B = full pairs left /
                //    contenders >= RS_BLOCK_MIN): every contender steps B - 1 pairs ahead on a copy of its state,
                //    e_u = its B-th key, L = max e_u at u* (first maximum); then every contender takes its pairs
                //    above (L, u*) -- at most B each, so the round cannot overdraw the slice -- and u* exactly B.
                //    All UEs of all tasks of the wave step at once: between B and contenders x B pairs for
                //    2 B - 1 steps, where trips pay a round of reductions for every two or three pairs.
                double m = active ? ((q > 0 ? rate_d : 0.0) / thl) : -1.0;
                bool need_full = true;
                double mx = 0.0;
                int idx = 0;
                int rf = r == 0 ? n_pairs_full : 0;  // full pairs still free (the closed forms above hand out all or nothing)
                bool blk = BLOCK && n_pairs_full >= RS_BLOCK_PAIRS;  // this task may still take block rounds (one way)
                for (;;) {
                    const bool more = sched && r < n_prb;
                    if (!wave_any(more)) break;
                    if (more) stat += 1u << 18;
    #ifdef RS_SECTION_PROFILE
                    sec_acc[15] += 1;  // PF rounds (not cycles)
    #endif
                    bool bmode = false;
                    if (BLOCK && wave_any(more && blk)) {
                        const int per_it = gran * rate;
                        const int uc = __popc(group_ballot<G>(more && m > 0.0, gbase));
                        // an under-estimate of rf / uc serves as well as the quotient (float reciprocal, rounded down)
                        const int B = (int)((float)rf * __builtin_amdgcn_rcpf((float)(uc > 1 ? uc : 1)) * 0.999f);
                        blk = blk && uc >= 2 && B >= RS_BLOCK_MIN;
                        bmode = more && blk;
                        const bool cont = bmode && m > 0.0;
                        // Both passes step FOUR pairs per iteration: th_j = pf_a th_(j-1) + share(bits_j), and the
                        // bits after j more pairs are known in advance (bits + j per_it until the queue runs dry), so
                        // the four shares are independent chains and only the four mul-adds depend on each other.
                        // Same operations on the same operands as the one-pair-at-a-time loop, same doubles.
                        const int kl = cont ? (int)((double)(q + per_it - 1) / (double)per_it) : 0;  // pairs that empty my queue (exact, see above)
                        // ---- pass 1: my key after B - 1 more pairs (largest th seen <=> smallest metric)
                        double e = m;
                        {
                            const bool run1 = cont && kl > B - 1;  // still holding data after B - 1 more pairs (else: key 0)
                            double t1 = thl, tmax1 = thl;
                            int b1 = bits;
                            for (int j = 1; wave_any(run1 && j < B); j += 4) {
                                const double s1 = pf_share(b1 + per_it), s2 = pf_share(b1 + 2 * per_it);
                                const double s3 = pf_share(b1 + 3 * per_it), s4 = pf_share(b1 + 4 * per_it);
                                const double a1 = pf_a * t1 + s1;
                                const double a2 = pf_a * a1 + s2;
                                const double a3 = pf_a * a2 + s3;
                                const double a4 = pf_a * a3 + s4;
                                const double x1 = max_finite(tmax1, a1), x2 = max_finite(x1, a2);
                                const double x3 = max_finite(x2, a3), x4 = max_finite(x3, a4);
                                const int left = run1 ? B - j : 0;  // steps still to do, this one included
                                t1 = left >= 4 ? a4 : (left == 3 ? a3 : (left == 2 ? a2 : (left == 1 ? a1 : t1)));
                                tmax1 = left >= 4 ? x4 : (left == 3 ? x3 : (left == 2 ? x2 : (left == 1 ? x1 : tmax1)));
                                b1 += 4 * per_it;
                            }
                            if (cont) e = run1 ? rate_d / tmax1 : 0.0;
                        }
                        SEC_MARK(11)
                        const double Lk = group_max<G>(e);
                        const int ustar = __ffs((int)group_ballot<G>(e == Lk, gbase)) - 1;
                        // ---- pass 2: take my pairs above (L, u*)
                        int cnt = 0;
                        {
                            const bool target = cont && gl == ustar;  // takes exactly B (fewer if it drains: L = 0 then)
                            const bool low = gl < ustar;
                            // fl(rate / tmax) against L without the IEEE divide: fl(L (1 +- 2^-40) tmax) is within
                            // 2^-51 of the product, so rate above the upper one puts the exact quotient above the
                            // double after L, rate below the lower one under the double before it, and rounding is
                            // monotone.  Anything closer (ties between equal UEs included) takes the divide (in the loop below).
                            const double L_hi = Lk * 0x1.0000000001p+0, L_lo = Lk * 0x1.fffffffffep-1;
                            double tmax = thl;
                            int kl2 = kl;  // pairs until my queue is empty
                            bool open_ = cont && (target || m > Lk || (m == Lk && low));
                            for (;;) {
                                if (!wave_any(open_)) break;
                                const int lim = B - cnt;  // pairs I may still take in this round
                                const int c1 = per_it < q ? per_it : q, c2 = 2 * per_it < q ? 2 * per_it : q;
                                const int c3 = 3 * per_it < q ? 3 * per_it : q, c4 = 4 * per_it < q ? 4 * per_it : q;
                                const double s1 = pf_share(bits + c1), s2 = pf_share(bits + c2);
                                const double s3 = pf_share(bits + c3), s4 = pf_share(bits + c4);
                                const double a1 = pf_a * thl + s1;
                                const double a2 = pf_a * a1 + s2;
                                const double a3 = pf_a * a2 + s3;
                                const double a4 = pf_a * a3 + s4;
                                const double x1 = max_finite(tmax, a1), x2 = max_finite(x1, a2);
                                const double x3 = max_finite(x2, a3), x4 = max_finite(x3, a4);
                                // ab_i: my key after i pairs of this iteration is above (L, u*).  All four without a branch (two
                                // multiplies and two compares each: the guard-band test of `above`); only a lane whose key falls
                                // INSIDE the band of some pair takes the IEEE divides, all four together, in one wave-uniform branch
                                // -- as four short-circuit lambdas this was ~60 instructions and two divergent branches per pair.
                                const bool h1 = rate_d > L_hi * x1, h2 = rate_d > L_hi * x2, h3 = rate_d > L_hi * x3, h4 = rate_d > L_hi * x4;
                                bool ab1 = h1, ab2 = h2, ab3 = h3, ab4 = h4;
                                {
                                    const bool u1 = !h1 && !(rate_d < L_lo * x1), u2 = !h2 && !(rate_d < L_lo * x2);
                                    const bool u3 = !h3 && !(rate_d < L_lo * x3), u4 = !h4 && !(rate_d < L_lo * x4);
                                    if (wave_any(open_ && !target && (u1 || u2 || u3 || u4))) {
                                        const double k1 = rate_d / x1, k2 = rate_d / x2, k3 = rate_d / x3, k4 = rate_d / x4;
                                        ab1 = u1 ? (k1 > Lk || (k1 == Lk && low)) : h1;
                                        ab2 = u2 ? (k2 > Lk || (k2 == Lk && low)) : h2;
                                        ab3 = u3 ? (k3 > Lk || (k3 == Lk && low)) : h3;
                                        ab4 = u4 ? (k4 > Lk || (k4 == Lk && low)) : h4;
                                    }
                                }
                                // p_i: with i pairs of this iteration in hand, the next one is mine too
                                const bool p1 = open_ & (kl2 > 1) & (1 < lim) & (target | ab1);
                                const bool p2 = p1 & (kl2 > 2) & (2 < lim) & (target | ab2);
                                const bool p3 = p2 & (kl2 > 3) & (3 < lim) & (target | ab3);
                                const bool p4 = p3 & (kl2 > 4) & (4 < lim) & (target | ab4);
                                if (open_) {
                                    const int n = 1 + (p1 ? 1 : 0) + (p2 ? 1 : 0) + (p3 ? 1 : 0);
                                    const int cn = p3 ? c4 : (p2 ? c3 : (p1 ? c2 : c1));
                                    q -= cn;
                                    bits += cn;
                                    rbs += n * gran;
                                    cnt += n;
                                    kl2 -= n;
                                    thl = p3 ? a4 : (p2 ? a3 : (p1 ? a2 : a1));  // (after the pair that empties the queue: unused)
                                    tmax = p3 ? x4 : (p2 ? x3 : (p1 ? x2 : x1));
                                }
                                open_ = p4;
                            }
                        }
                        SEC_MARK(12)
                        const int T = group_sum<G>(cnt);
                        if (cnt > 0) m = q > 0 ? rate_d / thl : 0.0;
                        if (bmode) {
                            r += T * gran;
                            rf -= T;
                            need_full = true;
                        }
                    }
                    const bool tm = more && !bmode;  // tasks on a trip this round
                    if (!BLOCK || wave_any(tm)) {  // (without BLOCK, tm == more: some lane is on a trip)
                        if (wave_any(tm && need_full)) {
                            const double fm = group_max<G>(m);
                            const unsigned eq = group_ballot<G>(m == fm, gbase);
                            if (need_full) {
                                mx = fm;
                                idx = __ffs((int)eq) - 1;  // np.argmax: first maximum
                            }
                        }
                        const double m_rest = gl == idx ? -2.0 : m;
                        const double m2 = group_max<G>(m_rest);
                        const unsigned eq2 = group_ballot<G>(m_rest == m2, gbase);
                        const int idx2 = __ffs((int)eq2) - 1;
                        int take = 0;
                        SEC_MARK(11)
                        if (tm) {
                            if (mx == 0.0) {
                                // every queue is empty: argmax of an all-zero metric is UE 0 for all the
                                // remaining RB pairs (Q4); its local th is discarded afterwards
                                if (gl == 0) rbs += n_prb - r;
                            } else if (gl == idx) {
                                // Only the leader's metric changes while it keeps winning, so the leader's lane
                                // runs the reference loop alone until it stops being the argmax.
                                if (m2 <= 0.0) {
                                    // nobody else has data: it wins every RB pair until drained -> closed form
                                    const int R = n_prb - r;
                                    const int per_it = gran * rate;
                                    // ceil divisions through one IEEE f64 divide each (exact: operands < 2^31 and the
                                    // quotient of two integers is never within rounding distance of the next integer);
                                    // the 32-bit integer divide costs ~40 VALU instructions on this ISA
                                    const int k_full = (int)((double)(q + per_it - 1) / (double)per_it);  // RB pairs to drain
                                    const int K = (int)((double)(R + gran - 1) / (double)gran);          // RB pairs left
                                    const bool all = k_full >= K;
                                    const int cap_bits = R * rate;
                                    const int tx = all ? (q < cap_bits ? q : cap_bits) : q;
                                    take = all ? K * gran : k_full * gran;
                                    rbs += all ? R : take;
                                    q -= tx;
                                    bits += tx;
                                    m = 0.0;  // drained, or no RBs left
                                } else {
                                    int rr = r;
                                    bool keep;
                                    do {
    #ifdef RS_SECTION_PROFILE
                                        sec_acc[13] += 1;  // leader-run iterations (all lanes are summed)
    #endif
                                        const int prbs = n_prb - rr < gran ? n_prb - rr : gran;
                                        rbs += prbs;
                                        const int tx = prbs * rate < q ? prbs * rate : q;
                                        q -= tx;
                                        bits += tx;
                                        rr += gran;
                                        if (q > 0) {  // a drained UE's metric is 0 whatever its (discarded) local th
                                            thl = pf_a * thl + pf_share(bits);
                                            // fl(rate / thl) against the runner-up WITHOUT the IEEE divide (a dozen dependent
                                            // instructions) on the run's critical path: t = fl(m2 * thl) is within 2^-53 of
                                            // m2 * thl, so rate > t (1 + 2^-40) puts the exact quotient above the double
                                            // after m2, rate < t (1 - 2^-40) below the one before it; rounding is monotone,
                                            // so the rounded quotient compares the same way.  Anything closer (ties between
                                            // equal UEs included) takes the divide.
                                            const double tm2 = m2 * thl;
                                            if (rate_d > tm2 * 0x1.0000000001p+0) {
                                                keep = true;
                                            } else if (rate_d < tm2 * 0x1.fffffffffep-1) {
                                                keep = false;
                                            } else {
                                                const double mm = rate_d / thl;
                                                keep = mm > m2 || (mm == m2 && gl < idx2);
                                            }
                                        } else {
                                            keep = false;  // m2 > 0 here
                                        }
                                    } while (keep && rr < n_prb);
                                    m = q > 0 ? rate_d / thl : 0.0;
                                    take = rr - r;
                                }
                            }
                        }
                        SEC_MARK(12)
                        const int tk = bperm(take, gbase + idx);
                        if (tm) {
                            r = mx == 0.0 ? n_prb : r + tk;
                            if (mx != 0.0 && m2 > 0.0) {
                                // the run ended below the runner-up (or the RBs ran out): next leader is known
                                need_full = false;
                                mx = m2;
                                idx = idx2;
                            } else {
                                need_full = true;
                            }
                        }
                    }
                }
            }

            SEC_MARK(3)
            // RBs are laid out contiguously in UE order (schedulers.py:66-76)
            const int prb_i = group_excl_scan<G>(rbs, gl);
            // ---- MCSCodeset.response (channel_models.py:297-313), for the UEs whose reception outcome can change
            // anything: those that sent bits.  (A UE holding RBs without data still consumes its Bernoulli draw
            // below; the allocation trace wants every probability, so the tracing instances evaluate them all.)
            const bool needed = sched && active && rbs > 0 && (TRACE || bits > 0);
            const int span_col = col + prb_lo + prb_i;  // first element of my span in the fading table
            // ---- the reception test by guard band (see fast_sigmoid above): decided in float32 for all but ~2e-4 of the UEs
            bool exact = needed;   // UEs whose probability is evaluated exactly
            bool rx_ok = false;    // outcome of the others
            if (!TRACE) n_rx_tests += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(needed));  // (scalar: statistics only)
            if (!TRACE && rx_fast) {
                const bool single = rbs == 1;  // no MI average (channel_models.py:305): the RB's SINR against s* itself
                const bool multi = needed && !single;
                const bool fw = multi && rbs > RS_FAST_WIDE;
                const double x0 = L_mi[mod], ref = L_ref[mcs];
                double S1 = 0.0;
                if (needed && single) S1 = A.fad[span_col];
                bool central = false;
                double St = 0.0;
                // the draw and the threshold S*(u): independent of the sums, so they run while the samples are on their way
                auto threshold = [&]() {
                    double u = 0.5;
                    if (needed) {
                        rs_stream st = {key0, key1, (uint32_t)sl, L_serial[lt], L_ctr[lt]};
                        u = rs_stream_uniform(&st);  // the draw the reception step consumes below
                    }
                    central = u >= 1.0e-4 && u <= 1.0 - 1.0e-4;
                    const float lf = 0.6931471805599453f * __builtin_amdgcn_logf((float)(1.0 - u) * __builtin_amdgcn_rcpf((float)u));
                    const float dq = (D->rx_B - lf) * D->rx_invA;  // s* - ref(mcs), dB
                    const float ystar = fast_sigmoid((float)((ref - x0) + (double)dq), 0.0f, L_c1[mod], 0.0f);
                    St = single ? (double)dq : (double)rbs * (double)ystar;
                };
                double S = 0.0;
                bool thr_done = false;
                if (wave_any(multi && !fw)) {
                    S = fast_team_sums(L_mi, L_c1, A.fad32, &L_nom[wb], multi && !fw, rbs, span_col, mod, threshold);
                    thr_done = true;
                }
                if (wave_any(fw)) {
                    const double s2 = fast_wide_sums(L_mi, L_c1, A.fad32, &L_nom[wb], fw, rbs, span_col, mod);
                    S = fw ? s2 : S;
                }
                if (!thr_done) threshold();
                if (needed && single) S = (S1 + L_nom[lt]) - ref;
                const double band = single ? D->rx_band1 : (double)rbs * D->rx_band;
                const double dd = S - St;
                const bool sure = central && (dd > band || dd < -band);  // (NaN anywhere: not sure)
                rx_ok = dd > 0.0;
                exact = needed && !sure;
            }
            SEC_MARK(10)
            if (wave_any(exact)) {
                // R1 + R2 fused: np.mean's pairwise sum of the mutual information over a UE's RBs.  The spans to evaluate
                // (all tasks of the wave) are dealt out to TEAMS of 8 lanes, lane j of a team owning numpy's accumulator
                // R_j: it evaluates the sigmoid of its elements one after the other and adds them in numpy's order, the
                // team meets for the tree and the remainder.  No per-RB values are stored anywhere.
                double sum_rx = 0.0;
                constexpr int WIDE = RS_WIDE_SPAN;
                const bool wide_sp = exact && rbs > WIDE && rbs <= RS_WIDE_MAX;
                if (wave_any(wide_sp)) sum_rx = wide_response(L_mi, A.fad, W_mi[tid >> 6], &L_nom[wb], wide_sp, rbs, span_col, mod);
                sum_rx = team_response(L_mi, A.fad, &L_nom[wb], exact && !wide_sp, rbs, span_col, mod, sum_rx);
                // R3: effective SNR and reception probability, every evaluated UE in its own lane
                if (exact) {
                    const double x0 = L_mi[mod], kk = L_mi[4 + mod];
                    double s_eff = sum_rx;  // rbs == 1: the RB's SINR itself (0 + x, numpy's n < 8 path)
                    if (rbs > 1) s_eff = rs_inv_sigmoid(sum_rx / (double)rbs, x0, kk);
                    const double x = D->mcsA * (s_eff - L_ref[mcs]) - D->mcsB;
                    p_rx = rs_sigmoid(x, 0.0, 1.0);
                    if (!TRACE) atomicAdd((unsigned long long*)&A.counters[(size_t)task * 4 + 1], 1ull);
                }
            }
            SEC_MARK(4)
            // ---- reception + UE.transmission_step (slice_l1.py:219-224, slice_ran.py:51-55)
            if (sched && active) {
                bool received = false;
                if (rbs > 0) {  // the draw is consumed whether or not anything rides on it
                    const unsigned c = L_ctr[lt];
                    if (exact) {
                        rs_stream st = {key0, key1, (uint32_t)sl, L_serial[lt], c};
                        received = rs_stream_uniform(&st) < p_rx;
                    } else if (needed) {
                        received = rx_ok;
                    }
                    L_ctr[lt] = c + 1u;
                }
                if (!received) bits = 0;
                double nq = queue - (double)bits;
                queue = nq > 0.0 ? nq : 0.0;
                th = pf_a * th + pf_share(bits);
                ue_bits = bits;
                ue_prbs = rbs;
            }
            if (sched) stat += 1u << 12;
        }

        SEC_MARK(5)
        // ================= SliceRANeMBB.update_info (slice_ran.py:278-305); Q2: stale bits/prbs count
        if (active) {
            atomicAdd(&L_acc_bits[lt], ue_bits);
            atomicAdd(&L_acc_prbs[lt], ue_prbs);
        }
        {
            const unsigned m_c = group_ballot<G>(active && !is_vbr, gbase);
            const unsigned m_v = group_ballot<G>(active && is_vbr, gbase);
            int n_c = __popc(m_c), n_v = __popc(m_v);
            n_c = n_c > 1 ? n_c : 1;
            n_v = n_v > 1 ? n_v : 1;
            double q_c = 0.0, q_v = 0.0;
            if (wave_any(active && queue != 0.0)) {
                q_c = group_sum<G>((active && !is_vbr) ? queue : 0.0);
                q_v = group_sum<G>((active && is_vbr) ? queue : 0.0);
            }
            // both e_snr sums in one integer reduction: |sum| < 2^15 per class
            int packed = active ? (is_vbr ? e_snr * 65536 : e_snr) : 0;
            packed = group_sum<G>(packed);
            int s_c = (int)(int16_t)(packed & 0xffff);
            int s_v = (packed - s_c) >> 16;
            // lane 3: cbr_queue, 4: cbr_snr, 8: vbr_queue, 9: vbr_snr -- one IEEE divide per lane
            if (G >= 16) {
                double num = 0.0;
                num = gl == 3 ? q_c : num;
                num = gl == 4 ? (double)s_c : num;
                num = gl == 8 ? q_v : num;
                num = gl == 9 ? (double)s_v : num;
                const double den = (double)(gl < 5 ? n_c : n_v);
                infok += num / den;
            } else {
                // lanes 3, 4 take the CBR quotients, lanes 0, 1 (second register) the VBR ones
                double num = 0.0;
                num = gl == 3 ? q_c : num;
                num = gl == 4 ? (double)s_c : num;
                num = gl == 0 ? q_v : num;
                num = gl == 1 ? (double)s_v : num;
                const double den = (double)(gl >= 3 ? n_c : n_v);
                const double quo = num / den;
                if (gl == 3 || gl == 4) infok += quo;
                if (gl == 0 || gl == 1) infok_hi += quo;
            }
        }

        SEC_MARK(6)
        if (TRACE) {
            if (valid) {
                rs_alloc_rec rec;
                rec.serial = active ? (int32_t)L_serial[lt] : 0;
                rec.type = active ? (flags & 1) : 0;
                rec.e_snr = active ? e_snr : 0;
                rec.prbs = active ? ue_prbs : 0;
                rec.bits = active ? (int64_t)ue_bits : 0;
                rec.queue = active ? queue : 0.0;
                rec.th = active ? th : 0.0;
                rec.p = (active && sched) ? p_rx : 0.0;
                rs_alloc_rec* row = A.trace + ((size_t)task * slots + t) * RS_GROUP;
                row[gl] = rec;
                if (G < 32) {  // the trace always has 32 entries per slot: clear the ones this instance has no lane for
                    rs_alloc_rec zero = {};
                    for (int e2 = gl + G; e2 < RS_GROUP; e2 += G) row[e2] = zero;
                }
            }
        }
    }
    flush();
    SEC_FLUSH(A.sections)
    const bool wave_worked = wave_any(valid);
    // reception tests of the wave: upper half of counter [1] of its first task (the lower half counts the UEs evaluated exactly)
    if (!TRACE && (tid & 63) == 0 && n_rx_tests != 0u) atomicAdd((unsigned long long*)&A.counters[(size_t)task * 4 + 1], (unsigned long long)n_rx_tests << 32);
    if (RS_DYN_PRIO && A.pace && (tid & 63) == 0 && wave_worked) {
        atomicAdd(&A.pace[0], (__builtin_amdgcn_s_memtime() - pace_t0) / (unsigned long long)slots);
        atomicAdd(&A.pace[1], 1ull);
#ifdef RS_WAVE_LOG  // developer build (tools/wave_log.py): when and where every wave of the production kernel ran
        {
            unsigned long long* wl_ = (unsigned long long*)A.sections + 16 + 16 * (size_t)(blockIdx.x * 4u + (threadIdx.x >> 6));
            wl_[0] = __builtin_amdgcn_s_memtime() - pace_t0;
            wl_[2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);  // HW_ID, XCC_ID
            wl_[3] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
        }
#endif
#ifdef RS_PACE_XCC  // developer build (profiles/HISTORY.md): wave paces per XCD through the section-profile buffer
        const unsigned xcc_ = __builtin_amdgcn_s_getreg(63508) & 7u;
        atomicAdd((unsigned long long*)&A.sections[2 * xcc_], (__builtin_amdgcn_s_memtime() - pace_t0) / (unsigned long long)slots);
        atomicAdd((unsigned long long*)&A.sections[2 * xcc_ + 1], 1ull);
#endif
    }

    // ---- outputs: get_state (slice_ran.py:321-325), compute_reward (slice_ran.py:307-319)
    // (the state pointers are fetched again here, through a pointer the compiler cannot see through, instead of
    // being kept alive -- twenty of them -- across the whole step)
    const RsState* sp_end = A.S;
    asm volatile("" : "+s"(sp_end));
    const RsState& SE = *sp_end;
    const double i1 = bperm(infok, gbase + 1), i2 = bperm(infok, gbase + 2), i3 = bperm(infok, gbase + 3);
    const double i6 = bperm(infok, gbase + 6), i7 = bperm(infok, gbase + 7);
    // info[8], info[9] live in lanes 8, 9: a G = 8 group keeps them in lanes 0, 1 of a second register
    const double i8 = G >= 16 ? bperm(infok, gbase + 8) : bperm(infok_hi, gbase + 0);
    if (selected && gl == 0) A.redo[task] = aborted ? 1 : 0;
    if (valid) {
        if (gl < RS_N_EMBB_VARS && gl < G) {
            A.obs[(size_t)rep * D->n_vars + sl * RS_N_EMBB_VARS + gl] = (float)(infok / D->norm[gl]);
            A.info[((size_t)rep * n_slices + sl) * 10 + gl] = infok;
        }
        if (G == 8 && gl < 2) {
            A.obs[(size_t)rep * D->n_vars + sl * RS_N_EMBB_VARS + 8 + gl] = (float)(infok_hi / D->norm[8 + gl]);
            A.info[((size_t)rep * n_slices + sl) * 10 + 8 + gl] = infok_hi;
        }
        if (gl == 0) {
            const double obs_time = slots * slot_len;
            bool cbr_ok = (i1 / obs_time > D->sla[0]) || (i2 / slots > D->sla[1]) || (i3 / slots < D->sla[2]);
            bool vbr_ok = (i6 / obs_time > D->sla[3]) || (i7 / slots > D->sla[4]) || (i8 / slots < D->sla[5]);
            int viol = !(cbr_ok && vbr_ok);
            A.violations[rep * n_slices + sl] = viol;
            A.labels[rep * n_slices + sl] = viol == 0 ? 1 : -1;
            SE.t_n_ue[task] = n_ue;
            SE.t_cbr_at[task] = L_task[tq][0];
            SE.t_vbr_at[task] = L_task[tq][1];
            SE.t_ctr[task] = (uint32_t)L_task[tq][2];
            SE.t_serial[task] = (uint32_t)L_task[tq][3];
            SE.t_cost[task] = (int)(stat >> 18);
#ifdef RS_WAVE_LOG  // [4 + group]: UEs at the end | RBs << 8 | PF rounds of the step << 16 | UE-slots << 32
            ((unsigned long long*)A.sections)[16 + 16 * (size_t)(blockIdx.x * 4u + (threadIdx.x >> 6)) + 4 + (gbase / G)] =
                (unsigned long long)n_ue | ((unsigned long long)n_prb << 8) | ((unsigned long long)(stat >> 18) << 16) | ((unsigned long long)(stat & 0xfffu) << 32);
#endif
            uint64_t* c = A.counters + (size_t)task * 4;
            const unsigned cnt_ue = stat & 0xfffu, n_sched = (stat >> 12) & 0x3fu;
            c[0] += cnt_ue * (unsigned)n_prb;  // every UE reads n_prb fading samples per slot (n_prb is fixed for the step)
            c[2] += n_sched * (unsigned)((n_prb + gran - 1) / gran);  // RB pairs per scheduled slot
            c[3] += cnt_ue;
        }
        if (err != 0) atomicOr(&SE.err[rep], err);  // bit 0: UE capacity, bit 1: VBR burst capacity
        if (active) {
            SE.u_queue[ui] = queue;
            SE.u_th[ui] = th;
            SE.u_nominal[ui] = L_nom[tid];
            SE.u_hold_at[ui] = L_hold[tid];
            SE.u_e_snr[ui] = e_snr;
            SE.u_findex[ui] = findex;
            SE.u_bits[ui] = ue_bits;
            SE.u_prbs[ui] = ue_prbs;
            SE.u_vbr_at[ui] = L_uvbr[tid];
            SE.u_ctr[ui] = L_ctr[tid];
            SE.u_serial[ui] = L_serial[tid];
            SE.u_flags[ui] = flags;
#pragma unroll
            for (int k = 0; k < RS_BURSTS; ++k) SE.u_burst[(task * RS_BURSTS + k) * RS_GROUP + gl] = L_burst[k][tid];
        }
    }
}

}  // namespace rs

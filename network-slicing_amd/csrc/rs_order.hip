// Launch order of the eMBB step tasks.
//
// All waves of a 4096-replica launch are co-resident (5 per SIMD), so the launch ends when its slowest wave
// ends, and a wave's 64/G tasks run in lockstep: every loop runs for the largest trip count among them.
// Measured (tools/section_profile.py): the slowest waves are the ones where two or three expensive tasks met,
// because their peaks fall into different slots and add up.  Tasks are therefore ranked by predicted cost
// (fading samples per slot + the contested PF trips of their previous step) and every wave gets ONE task from
// the heavy end of the ranking and 64/G - 1 from the light end, heaviest waves first: the dispatcher fills
// the SIMDs round by round (tools/ubench/placement.hip: every SIMD receives one wave of blocks
// [256k, 256k + 256)), so each SIMD also gets one wave of each cost stratum, and for batches larger than the
// chip the order is longest-processing-time-first.  Every second round is dealt backwards (a serpentine, end of round 4): in plain
// cost order one SIMD held the heaviest wave of every round and another the lightest of every round (tools/wave_log.py: the five
// waves of a SIMD are w, w + 1024 +- 3, w + 2048 +- 3, ...; their last end fell from 0.86 to 0.74 of the launch across w with agents
// in the loop, 0.85 to 0.82 with the serpentine).  The order only decides which lanes simulate which
// (replica, slice); results do not depend on it.  RANSLICE_ORDER=0 turns it off, 1-3 sort without pairing
// (measured slower: homogeneous heavy waves are the slowest of all).
//
// Counting sort in two launches: keys + per-bin ranks (atomics), then prefix sums + scatter.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rs_device.h"

namespace rs {

#define RS_ORDER_BINS 2048

__device__ __forceinline__ int order_key(int mode, int n_ue, int n_prb, int cost) {
    int key;
    mode = mode > 3 ? mode - 3 : mode;
    if (mode == 4) key = ((n_ue * n_prb) >> 2) + 2 * cost;  // PF trips dominate (a trip costs what ~30 fading samples do)
    else if (mode == 5) key = cost;
    else if (mode == 2) key = n_ue * 64 + (n_prb >> 2);  // UE count first, then width
    else if (mode == 3) key = n_ue * n_prb + cost;  // work + last step's contested PF trips
    else key = n_ue * n_prb;                         // fading samples per slot
    key = key < 0 ? 0 : key;
    return key < RS_ORDER_BINS ? key : RS_ORDER_BINS - 1;
}

// hist: this step's bin counters (zero on entry); slot[task] = bin | rank-in-bin << 11.  Ranks are taken inside the
// block first (LDS atomics), then one global atomic per (block, occupied bin) reserves the block's range.
__global__ __launch_bounds__(256) void order_key_kernel(const RsDev* __restrict__ D, const RsState* __restrict__ Sp,
                                                        const int32_t* __restrict__ actions, int mode, int* hist,
                                                        uint64_t* slot, int4 kw, int split_ue) {
    __shared__ int cnt[RS_ORDER_BINS];  // tasks of this block per bin, then the block's base rank in the bin
    for (int k = threadIdx.x; k < RS_ORDER_BINS; k += 256) cnt[k] = 0;
    __syncthreads();
    const int n_tasks = D->n_envs * D->n_embb;
    const int task = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    int bin = 0, local = 0;
    if (task < n_tasks) {
        const int rep = task / D->n_embb, sl = task - rep * D->n_embb;
        const int n_prb = actions[rep * D->n_slices + sl];
        int key;
        if (kw.w > 0 || kw.x != 16 || kw.y != 16 || kw.z != 0) {
            // developer knob RANSLICE_KEY_W = "a,b,c,d" (sixteenths): a x fading samples per slot + b x last step's PF rounds +
            // c x UEs + d x (slots the backlog needs at this allocation, up to 50) x (UEs + 1)
            const int n_ue = Sp->t_n_ue[task];
            int need_rb = 0;
            if (kw.w > 0) {
                for (int u = 0; u < n_ue && u < RS_GROUP; ++u) {
                    const double q = Sp->u_queue[(size_t)task * RS_GROUP + u];
                    int li = Sp->u_e_snr[(size_t)task * RS_GROUP + u] - D->lut_lo;
                    li = li < 0 ? 0 : (li >= D->lut_n ? D->lut_n - 1 : li);
                    const int rate = D->lut_rate[li] > 0 ? D->lut_rate[li] : 1;
                    const double rb = q / (double)rate;
                    need_rb += rb < 100000.0 ? (int)rb : 100000;
                }
            }
            int drain = n_prb > 0 ? need_rb / n_prb : 0;
            drain = drain < 50 ? drain : 50;
            key = (kw.x * (n_ue * n_prb) + kw.y * Sp->t_cost[task] + kw.z * n_ue + kw.w * drain * (n_ue + 1)) >> 4;
            key = key < 0 ? 0 : (key < RS_ORDER_BINS ? key : RS_ORDER_BINS - 1);
        } else {
            key = order_key(mode, Sp->t_n_ue[task], n_prb, Sp->t_cost[task]);
        }
        // split step (rs_api.hip): tasks with split_ue UEs or more lead the ranking whatever their cost -- they do not fit the
        // 8-lane instance -- and the cost key orders each class at half its resolution
        if (split_ue > 0) key = (Sp->t_n_ue[task] >= split_ue ? RS_ORDER_BINS / 2 : 0) + ((key >> 1) < RS_ORDER_BINS / 2 ? (key >> 1) : RS_ORDER_BINS / 2 - 1);
        bin = RS_ORDER_BINS - 1 - key;  // heaviest first
        local = atomicAdd(&cnt[bin], 1);
    }
    __syncthreads();
    if (task < n_tasks && local == 0) cnt[bin] = atomicAdd(&hist[bin], cnt[bin]);  // first of its bin in the block
    __syncthreads();
    if (task < n_tasks) slot[task] = (uint64_t)bin | ((uint64_t)(cnt[bin] + local) << 11);
}

// order[start(bin) + rank] = task; also clears the other parity's counters for the next step
__global__ __launch_bounds__(256) void order_scatter_kernel(const RsDev* __restrict__ D, const int* __restrict__ hist,
                                                            int* hist_next, const uint64_t* __restrict__ slot,
                                                            int32_t* order, int pair, int tpw, int snake, int snake_mask, int rot_mask,
                                                            int seg_lo, int seg_hi) {
    __shared__ int start[RS_ORDER_BINS];
    __shared__ int part[256];
    constexpr int PER = RS_ORDER_BINS / 256;
    int loc[PER];
    int s = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        loc[k] = s;
        s += hist[threadIdx.x * PER + k];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    // exclusive scan of the 256 partial sums (Hillis-Steele)
    int incl = s;
    for (int d = 1; d < 256; d <<= 1) {
        const int o = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        incl += o;
        part[threadIdx.x] = incl;
        __syncthreads();
    }
    const int excl = incl - s;
#pragma unroll
    for (int k = 0; k < PER; ++k) start[threadIdx.x * PER + k] = excl + loc[k];
    __syncthreads();
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < RS_ORDER_BINS; k += 256) hist_next[k] = 0;
    const int n_all = D->n_envs * D->n_embb;
    const int task = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (task >= n_all) return;
    const uint64_t v = slot[task];
    int p = start[(int)(v & (RS_ORDER_BINS - 1))] + (int)(v >> 11);  // rank, heaviest first
    // The split step (rs_api.hip) deals the ranking out to several launches: ranks [seg_lo, seg_hi) are the packed 16-lane
    // launch's and get the wave composition below among themselves; the ranks before (one task per wave) and after (the
    // 8-lane instance, eight like tasks per wave) keep their place.  Without a split the segment is the whole ranking.
    if (p < seg_lo || p >= seg_hi) {
        order[p] = task;
        return;
    }
    const int n_tasks = seg_hi - seg_lo;
    p -= seg_lo;
    if (pair > 0 && tpw > 1 && n_tasks % tpw == 0) {
        // The heaviest `pair`/256 of the waves get ONE task from the heavy end of the ranking, filled up with
        // tpw - 1 from the light end: the heavy task's trip counts then set the wave's pace alone instead of adding
        // up with other heavy tasks' peaks (those waves end the launch).  The waves after them take the middle of
        // the ranking in order: tasks of similar cost run their loops in lockstep with the least idle trips.
        const int W = n_tasks / tpw;
        const int Hw = (int)(((long long)W * pair) >> 8);
        if (p < Hw) p = tpw * p;
        else if (p >= n_tasks - (tpw - 1) * Hw) {
            const int q = n_tasks - 1 - p;
            p = tpw * (q / (tpw - 1)) + 1 + q % (tpw - 1);
        } else {
            p = tpw * Hw + (p - Hw);
        }
    }
    if (snake > 0 && tpw > 0) {
        // The dispatcher deals the blocks out round by round: waves w, w + R, w + 2 R, ... (R = SIMDs of the chip) share a
        // SIMD (tools/wave_log.py: the blocks of an XCD go round its CUs with stride 32).  In plain cost order SIMD 0 then
        // holds the heaviest wave of EVERY round and SIMD R - 1 the lightest of every round -- a whole wave's cost range
        // between them.  Every second complete round is therefore dealt backwards (a serpentine): w and 2 R - 1 - w share
        // a SIMD, and the sums even out.
        int Wv = p / tpw;
        const int in = p - Wv * tpw, k = Wv / snake;
        if (((snake_mask >> (k & 31)) & 1) != 0 && (k + 1) * snake <= n_tasks / tpw) Wv = k * snake + (snake - 1 - (Wv - k * snake));
        // (developer knob: rounds rotated by half a round instead -- evens out a U-shaped cost profile where reversal evens out a slope)
        if (((rot_mask >> (k & 31)) & 1) != 0 && (k + 1) * snake <= n_tasks / tpw) Wv = k * snake + (Wv - k * snake + snake / 2) % snake;
        p = Wv * tpw + in;
    }
    order[seg_lo + p] = task;
}

}  // namespace rs

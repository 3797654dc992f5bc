// rs_mux.hip -- create_env(..., L1_level=False) (scenario_creator.py:168-177): the slices are multiplexed in the L1.
//
//   * every eMBB RAN slice of a replica sits under ONE SliceL1eMBB (slice_l1.py:126-228 with several slices_ran): one
//     UE list in arrival order across the slices, one PRB range, one ProportionalFair.allocate over all of them;
//     arrivals / admission control / departures / update_info / SLA stay per RAN slice (slice_ran.py:150-325);
//   * every mMTC RAN slice feeds ONE SliceL1mMTC FIFO (slice_l1.py:15-125), each entry remembering its slice.
// The action has one entry per L1 slice; labels / violations likewise (violations = RAN slices in breach,
// slice_l1.py:160-171); the observation and the info rows keep one block per RAN slice.
//
// Mapping: one wavefront per replica and L1 slice; lane u = UE u of the shared list (capacity 64).  This is the
// reference's rarely used second mode (none of its experiment scripts passes L1_level=False), so the kernels are
// the straightforward ones -- the per-RB-pair PF loop with wave-wide reductions, channel estimates inside the slot
// loop -- built from the same arithmetic pieces as rs_embb.hip (walker, pairwise sums, response sums, streams), which
// is what makes them bit-identical to the oracle.  No order kernels, no replay: 64 lanes are the capacity.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rs_embb.hip"
#include "rs_mmtc.hip"

namespace rs {

#define RS_MUX_UE 64
#define RS_MUX_RAN 8

__device__ __forceinline__ int wave_sum(int v) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {  // exactly representable integers only (order irrelevant)
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    for (int d = 32; d >= 1; d >>= 1) {
        const double o = __shfl_xor(v, d);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int wave_excl_scan(int v, int lane) {
    int inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    return inc - v;
}
__device__ __forceinline__ int kth_set_bit64(unsigned long long m, int k) {
    for (int z = 0; z < k; ++z) m &= m - 1ull;
    return m ? __builtin_ctzll(m) : 0;
}

// UE slot u of replica rep in the per-UE state arrays: the replica's n_embb tasks own 32 slots each
__device__ __forceinline__ size_t mux_ui(int rep, int n_embb, int u) { return ((size_t)rep * n_embb + (u >> 5)) * RS_GROUP + (u & 31); }
__device__ __forceinline__ size_t mux_bi(int rep, int n_embb, int u, int k) {
    return (((size_t)rep * n_embb + (u >> 5)) * RS_BURSTS + k) * RS_GROUP + (u & 31);
}

template <bool TRACE>
__global__ __launch_bounds__(64) void embb_mux_step_kernel(StepArgs A) {
    __shared__ unsigned short L_burst[RS_BURSTS][64];  // VBR burst end times (rs_burst_code), 0 = free
    __shared__ int L_hold[64], L_uvbr[64];
    __shared__ unsigned L_serial[64], L_ctr[64];
    __shared__ int L_acc_traf[64], L_acc_bits[64], L_acc_prbs[64];
    __shared__ double L_nom[64];
    __shared__ double L_info[RS_MUX_RAN][10];   // info accumulators of every RAN slice (slice_ran.py:270-273)
    __shared__ int L_slice[RS_MUX_RAN][4];      // per RAN slice: cbr_at, vbr_at, slice draw counter, next UE serial
    __shared__ double W_mi[RS_MAX_PRBS];
    __shared__ int L_lut[RS_LUT_MAX];
    __shared__ double L_ref[32];
    __shared__ double L_mi[8];  // logistic MI curves (x0 x 3, pad, k x 3, pad): team_response / wide_response
    const RsDev* __restrict__ D = A.D;
    const RsState& S = *A.S;
    const int lane = (int)threadIdx.x;
    const int rep = (int)blockIdx.x;
    const int M = D->n_embb;
    const int cap = M * RS_GROUP < RS_MUX_UE ? M * RS_GROUP : RS_MUX_UE;
    L_lut[lane] = lane < D->lut_n ? ((D->mcs_mod[D->lut_mcs[lane]] << 24) | (D->lut_mcs[lane] << 16) | D->lut_rate[lane]) : 0;
    if (lane < 32) L_ref[lane] = D->mcs_ref[lane];
    if (lane >= 32 && lane < 40) L_mi[lane - 32] = (lane & 3) == 3 ? 0.0 : (lane < 36 ? D->mi_x0[lane - 32] : D->mi_k[lane - 36]);
    for (int i = lane; i < M * 10; i += 64) L_info[i / 10][i % 10] = 0.0;
    if (lane < M) {
        const int task = rep * M + lane;
        L_slice[lane][0] = S.t_cbr_at[task];
        L_slice[lane][1] = S.t_vbr_at[task];
        L_slice[lane][2] = (int)S.t_ctr[task];
        L_slice[lane][3] = (int)S.t_serial[task];
    }
    __builtin_amdgcn_wave_barrier();
    const int clock0 = (int)A.run[0];
    const int n_act_entries = (M > 0 ? 1 : 0) + (D->n_mmtc > 0 ? 1 : 0);
    const int n_prb = A.actions[rep * n_act_entries + 0];
    const int prb_lo = 0;
    const int P = D->P;
    const double slot_len = D->slot_length;
    const double pf_a = D->pf_a, pf_b = D->pf_b;
    const int gran = D->gran;
    const bool has_nan = D->has_nan != 0;
    const int T0 = D->T[0], T1 = D->T[1], T2 = D->T[2];
    const int fo0 = (int)D->fad_off[0], fo1 = (int)D->fad_off[1], fo2 = (int)D->fad_off[2];
    const int vo0 = (int)D->valid_off[0], vo1 = (int)D->valid_off[1], vo2 = (int)D->valid_off[2];
    const uint64_t seed = S.seeds[rep];
    const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
    int err = 0;

    int n_ue = S.t_n_ue[rep * M];
    bool active = lane < n_ue;
    double queue = 0.0, th = 0.0;
    int e_snr = 0, findex = 0, ue_bits = 0, ue_prbs = 0, flags = 0;  // flags: bit0 type, bits1-2 trace, bit3 step sign, bits4-6 RAN slice
    int n_act = 0, evt_at = RS_NEVER;
    {
        int hold_at = RS_NEVER, uvbr_at = RS_NEVER;
        unsigned uctr = 0u, userial = 0u;
        double nominal = 0.0;
        const size_t ui = mux_ui(rep, M, lane);
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
            const unsigned e = active ? S.u_burst[mux_bi(rep, M, lane, k)] : 0u;
            L_burst[k][lane] = (unsigned short)e;
            if (e != 0u) {  // occupied entries are always still running (they are freed in the slot they end)
                n_act += 1;
                const int endt = clock0 + rs_burst_rel(e, clock0);
                evt_at = endt < evt_at ? endt : evt_at;
            }
        }
        n_act += (flags >> 8) & 0xff;  // bursts that never end (Q5) are only counted
        L_hold[lane] = hold_at;
        L_uvbr[lane] = uvbr_at;
        L_serial[lane] = userial;
        L_ctr[lane] = uctr;
        L_nom[lane] = nominal;
        L_acc_traf[lane] = 0;
        L_acc_bits[lane] = 0;
        L_acc_prbs[lane] = 0;
    }
    unsigned stat_ue = 0u, stat_sched = 0u;

    // SliceRANeMBB.update_info's integer sums (slice_ran.py:282-285,296-299), folded per RAN slice and class
    auto flush = [&]() {
        const int ran = (flags >> 4) & 7;
        const bool is_vbr = (flags & 1) != 0;
        for (int m = 0; m < M; ++m) {
            for (int cls = 0; cls < 2; ++cls) {
                const bool mine = active && ran == m && (is_vbr ? 1 : 0) == cls;
                const int t_ = wave_sum(mine ? L_acc_traf[lane] : 0);
                const int b_ = wave_sum(mine ? L_acc_bits[lane] : 0);
                const int p_ = wave_sum(mine ? L_acc_prbs[lane] : 0);
                if (lane == 0) {
                    L_info[m][cls * 5 + 0] += (double)t_;
                    L_info[m][cls * 5 + 1] += (double)b_;
                    L_info[m][cls * 5 + 2] += (double)p_;
                }
            }
        }
        L_acc_traf[lane] = 0;
        L_acc_bits[lane] = 0;
        L_acc_prbs[lane] = 0;
        __builtin_amdgcn_wave_barrier();
    };
    auto pf_share = [&](int b) -> double { return (pf_b * (double)b) / slot_len; };

    const int slots = D->slots;
    const int n_pairs_full = n_prb / gran;
    for (int t = 0; t < slots; ++t) {
        const int now = clock0 + t + 1;
        const int slot_counter = t + 1;
        // ================= RAN slice by RAN slice: SliceRANeMBB.slot, extract_users, add_users (slice_l1.py:195-198)
        for (int m = 0; m < M; ++m) {
            int cbr_at = L_slice[m][0], vbr_at = L_slice[m][1];
            const bool cbr_fire = cbr_at == now, vbr_fire = vbr_at == now;
            const bool depart = active && ((flags >> 4) & 7) == m && L_hold[lane] == now;
            const bool any_dep = wave_any(depart);
            if (!(cbr_fire || vbr_fire || any_dep)) continue;
            uint32_t sl_ctr = (uint32_t)L_slice[m][2], next_serial = (uint32_t)L_slice[m][3];
            int n_pend = 0, pend_type0 = 0, pend_type1 = 0;
            if (cbr_fire || any_dep) flush();  // cbr_cac reads this step's running sums
            if (cbr_fire) {
                rs_stream st = {key0, key1, (uint32_t)m, 0u, sl_ctr};
                const double ia = rs_stream_exponential(&st, D->cbr_ia_scale);
                sl_ctr = st.ctr;
                cbr_at = now + 1 + rint_slots(ia, slot_len);
                // cbr_cac (slice_ran.py:195-203)
                const int cslots = slot_counter > 1 ? slot_counter : 1;
                const double time = cslots * slot_len;
                const double c_prb = L_info[m][2] / cslots;
                const double c_th = L_info[m][1] / time;
                if (!(c_prb >= D->sla[1] || c_th >= D->sla[0])) {
                    pend_type0 = 0;
                    n_pend = 1;
                }
            }
            if (vbr_fire) {
                rs_stream st = {key0, key1, (uint32_t)m, 0u, sl_ctr};
                const double ia = rs_stream_exponential(&st, D->vbr_ia_scale);
                sl_ctr = st.ctr;
                vbr_at = now + 1 + rint_slots(ia, slot_len);
                if (n_pend == 0) pend_type0 = 1; else pend_type1 = 1;
                n_pend += 1;
            }
            // ---- departures of this RAN slice (slice_ran.py:251-261) + extract_users (slice_l1.py:187-191)
            if (any_dep) {
                const unsigned long long keep = __builtin_amdgcn_ballot_w64(active && !depart);
                const int n_keep = __popcll(keep);
                const int src = kth_set_bit64(keep, lane);
                queue = bperm(queue, src);
                th = bperm(th, src);
                e_snr = bperm(e_snr, src);
                findex = bperm(findex, src);
                ue_bits = bperm(ue_bits, src);
                ue_prbs = bperm(ue_prbs, src);
                flags = bperm(flags, src);
                n_act = bperm(n_act, src);
                evt_at = bperm(evt_at, src);
                const int m_hold = L_hold[src], m_uvbr = L_uvbr[src];
                const unsigned m_ser = L_serial[src], m_ctr = L_ctr[src];
                const double m_nom = L_nom[src];
                unsigned short m_b[RS_BURSTS];
#pragma unroll
                for (int k = 0; k < RS_BURSTS; ++k) m_b[k] = L_burst[k][src];
                __builtin_amdgcn_wave_barrier();
                n_ue = n_keep;
                active = lane < n_ue;
                L_hold[lane] = active ? m_hold : RS_NEVER;
                L_uvbr[lane] = active ? m_uvbr : RS_NEVER;
                L_serial[lane] = active ? m_ser : 0u;
                L_ctr[lane] = m_ctr;
                L_nom[lane] = m_nom;
#pragma unroll
                for (int k = 0; k < RS_BURSTS; ++k) L_burst[k][lane] = m_b[k];
                if (!active) { evt_at = RS_NEVER; n_act = 0; }
                __builtin_amdgcn_wave_barrier();
            }
            // ---- add_users: the new UEs join the end of the shared list, insert_user draws on their own streams
            if (n_ue + n_pend > cap) {
                err |= 1;  // RS_EOVERFLOW: UE capacity of the multiplexed L1 slice
                n_pend = cap - n_ue;
            }
            const bool is_new = lane >= n_ue && lane < n_ue + n_pend;
            if (is_new) {
                const int k = lane - n_ue;
                const int type = k == 0 ? pend_type0 : pend_type1;
                const unsigned userial = next_serial + (uint32_t)k;
                rs_stream st = {key0, key1, (uint32_t)m, userial, 0u};
                queue = 0.0; th = 0.0; e_snr = 0; ue_bits = 0; ue_prbs = 0; findex = 0;
                L_acc_traf[lane] = 0; L_acc_bits[lane] = 0; L_acc_prbs[lane] = 0;
#pragma unroll
                for (int q = 0; q < RS_BURSTS; ++q) L_burst[q][lane] = 0;
                n_act = 0;
                int uvbr_at = RS_NEVER;
                if (type == 1) {  // VbrSource.__init__ (traffic_generators.py:62-68)
                    const int v = (int)RS_RINT(rs_stream_exponential(&st, D->vbr_inter));
                    uvbr_at = v >= 1 ? now + v - 1 : RS_NEVER;  // Q5
                }
                const double hold = rs_stream_exponential(&st, type == 0 ? D->cbr_hold_scale : D->vbr_hold_scale);
                const int hv = rint_slots(hold, slot_len);
                const int hold_at = hv >= 1 ? now + hv - 1 : RS_NEVER;  // Q5
                int ftype = 0, fstep = 1;
                double nominal = 0.0;
                if (hold_at != now) {  // Q13
                    ftype = (int)rs_stream_integers(&st, RS_N_TRACES);
                    findex = (int)rs_stream_integers(&st, sel3(ftype, T0, T1, T2));
                    fstep = rs_stream_pm1(&st);
                    const MacroCell mc = macro_cell_draw(D, st);
                    nominal = mc.x;
                    st.ctr = (uint32_t)mc.y;
                }
                L_hold[lane] = hold_at;
                L_uvbr[lane] = uvbr_at;
                L_serial[lane] = userial;
                L_ctr[lane] = st.ctr;
                L_nom[lane] = nominal;
                evt_at = hold_at < uvbr_at ? hold_at : uvbr_at;
                flags = type | (ftype << 1) | ((fstep > 0 ? 1 : 0) << 3) | (m << 4);
                active = true;
            }
            n_ue += n_pend;
            next_serial += (uint32_t)n_pend;
            // Q13: a new UE whose holding time is one slot leaves in its arrival slot, before it is served
            {
                const bool gone = active && is_new && L_hold[lane] == now;
                if (wave_any(gone)) {
                    const unsigned long long keep = __builtin_amdgcn_ballot_w64(active && !gone);
                    // such a UE sits at the end of the list: lanes below it keep their places unless two arrived
                    const int src = kth_set_bit64(keep, lane);
                    queue = bperm(queue, src); th = bperm(th, src); e_snr = bperm(e_snr, src); findex = bperm(findex, src);
                    ue_bits = bperm(ue_bits, src); ue_prbs = bperm(ue_prbs, src); flags = bperm(flags, src);
                    n_act = bperm(n_act, src); evt_at = bperm(evt_at, src);
                    const int m_hold = L_hold[src], m_uvbr = L_uvbr[src];
                    const unsigned m_ser = L_serial[src], m_ctr = L_ctr[src];
                    const double m_nom = L_nom[src];
                    unsigned short m_b[RS_BURSTS];
#pragma unroll
                    for (int k = 0; k < RS_BURSTS; ++k) m_b[k] = L_burst[k][src];
                    __builtin_amdgcn_wave_barrier();
                    n_ue = __popcll(keep);
                    active = lane < n_ue;
                    L_hold[lane] = active ? m_hold : RS_NEVER;
                    L_uvbr[lane] = active ? m_uvbr : RS_NEVER;
                    L_serial[lane] = active ? m_ser : 0u;
                    L_ctr[lane] = m_ctr;
                    L_nom[lane] = m_nom;
#pragma unroll
                    for (int k = 0; k < RS_BURSTS; ++k) L_burst[k][lane] = m_b[k];
                    if (!active) { evt_at = RS_NEVER; n_act = 0; }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (lane == 0) {
                L_slice[m][0] = cbr_at;
                L_slice[m][1] = vbr_at;
                L_slice[m][2] = (int)sl_ctr;
                L_slice[m][3] = (int)next_serial;
            }
            __builtin_amdgcn_wave_barrier();
        }

        // ================= VbrSource.step events (traffic_generators.py:70-99) on absolute end times
        int n_cur = n_act;
        if (active && evt_at == now) {
            int cnt = (flags >> 8) & 0xff, nxt = RS_NEVER;  // the never-ending bursts (Q5) always emit
            unsigned freek = RS_BURSTS;                      // a free entry for a burst that may start now
#pragma unroll
            for (int k = 0; k < RS_BURSTS; ++k) {
                const unsigned e = L_burst[k][lane];
                const int rel = e != 0u ? rs_burst_rel(e, now) : 0;
                if (e != 0u && rel <= 0) L_burst[k][lane] = 0;  // ends exactly now: dropped without emitting
                if (rel > 0) {
                    cnt += 1;
                    nxt = now + rel < nxt ? now + rel : nxt;
                } else if (freek == RS_BURSTS) {
                    freek = (unsigned)k;
                }
            }
            n_cur = cnt;
            int uvbr_at = L_uvbr[lane];
            if (uvbr_at == now) {
                rs_stream st = {key0, key1, (uint32_t)((flags >> 4) & 7), L_serial[lane], L_ctr[lane]};
                const int d = (int)RS_RINT(rs_stream_exponential(&st, D->vbr_b_size));
                const int v = (int)RS_RINT(rs_stream_exponential(&st, D->vbr_inter));
                L_ctr[lane] = st.ctr;
                if (d < 1) {  // Q5: a duration that rounds to 0 never counts down to 0: the burst emits for ever
                    if (((flags >> 8) & 0xff) == 0xff) err |= 2;
                    else flags += 1 << 8;
                    cnt += 1;
                } else if (freek == RS_BURSTS || d >= RS_BURST_MAX_LEN) {
                    err |= 2;  // RS_EOVERFLOW: more than RS_BURSTS bursts running, or one longer than the 15-bit clock can hold
                } else {
#pragma unroll
                    for (int k = 0; k < RS_BURSTS; ++k)
                        if ((unsigned)k == freek) L_burst[k][lane] = (unsigned short)rs_burst_code(now + d);
                    cnt += 1;
                    nxt = now + d < nxt ? now + d : nxt;
                }
                uvbr_at = v >= 1 ? now + v : RS_NEVER;
                L_uvbr[lane] = uvbr_at;
            }
            n_act = cnt;
            const int h = L_hold[lane];
            const int e = h < uvbr_at ? h : uvbr_at;
            evt_at = e < nxt ? e : nxt;
        }
        const bool is_vbr = (flags & 1) != 0;
        const int ran = (flags >> 4) & 7;

        // ================= UE.traffic_step (slice_ran.py:47-49)
        if (active) {
            const double new_bits = is_vbr ? (double)n_cur * D->vbr_p_size : D->cbr_bits;
            queue += new_bits;
            L_acc_traf[lane] += (int)new_bits;
        }
        const bool any_queue = wave_any(active && queue > 0.0);

        // ================= channel: get_snr + estimate_snr (channel_models.py:171-191, slice_ran.py:43-45)
        int col = 0;
        const int ftype = (flags >> 1) & 3;
        if (n_prb > 0) {
            const bool on = active;
            if (on) {
                int fstep = (flags & 8) ? 1 : -1;
                walker_advance(findex, fstep, sel3(ftype, T0, T1, T2), has_nan, A.fad_valid + sel3(ftype, vo0, vo1, vo2), key0,
                               key1, (uint32_t)ran, L_serial[lane], (uint32_t)now);
                flags = (flags & ~8) | ((fstep > 0 ? 1 : 0) << 3);
                col = sel3(ftype, fo0, fo1, fo2) + findex * P;
            }
            const double nom = L_nom[lane];
            const double* __restrict__ colp = A.fad + (on ? col + prb_lo : 0);
            const double sum = lane_pairwise(n_prb, on, [&](int i) { return colp[i] + nom; });
            if (on) e_snr = (int)RS_RINT(sum / (double)n_prb);
        }
        stat_ue += (unsigned)n_ue;

        // ================= scheduling (slice_l1.py:215-224): one PF over every UE of the L1 slice
        const bool sched = any_queue && n_prb > 0;
        double p_rx = 0.0;
        if (sched) {
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
            const int per_it = gran * rate;
            // under-loaded slot: closed form (see rs_embb.hip)
            const int k_u = (active && q > 0) ? (int)((double)(q + per_it - 1) / (double)per_it) : 0;
            const int need = wave_sum(k_u);
            if (need <= n_pairs_full) {
                rbs = k_u * gran;
                bits = k_u > 0 ? q : 0;
                if (lane == 0) rbs += n_prb - need * gran;
            } else {
                // the reference loop, one RB pair per trip (schedulers.py:47-63)
                double m = active ? ((q > 0 ? rate_d : 0.0) / thl) : -1.0;
                for (int r = 0; r < n_prb; r += gran) {
                    const unsigned long long nz = __builtin_amdgcn_ballot_w64(active && q > 0);
                    if (nz == 0ull) {  // Q4: an all-zero metric sends every remaining pair to UE 0
                        if (lane == 0) rbs += n_prb - r;
                        break;
                    }
                    const double mx = wave_max(m);
                    const unsigned long long eq = __builtin_amdgcn_ballot_w64(m == mx);
                    const int idx = __builtin_ctzll(eq);  // np.argmax: first maximum
                    if (lane == idx) {
                        const int prbs = n_prb - r < gran ? n_prb - r : gran;
                        rbs += prbs;
                        const int tx = prbs * rate < q ? prbs * rate : q;
                        q -= tx;
                        bits += tx;
                        thl = pf_a * thl + pf_share(bits);
                        m = q > 0 ? rate_d / thl : 0.0;
                    }
                }
            }
            const int prb_i = wave_excl_scan(rbs, lane);
            // ---- MCSCodeset.response (channel_models.py:297-313)
            const bool needed = active && rbs > 0 && (TRACE || bits > 0);
            const int span_col = col + prb_lo + prb_i;
            double sum_rx = 0.0;
            const bool wide_sp = needed && rbs > RS_WIDE_SPAN;
            if (wave_any(wide_sp)) sum_rx = wide_response(L_mi, A.fad, W_mi, L_nom, wide_sp, rbs, span_col, mod);
            sum_rx = team_response(L_mi, A.fad, L_nom, needed && !wide_sp, rbs, span_col, mod, sum_rx);
            if (needed) {
                const double x0 = D->mi_x0[mod], kk = D->mi_k[mod];
                double s_eff = sum_rx;
                if (rbs > 1) s_eff = rs_inv_sigmoid(sum_rx / (double)rbs, x0, kk);
                const double x = D->mcsA * (s_eff - L_ref[mcs]) - D->mcsB;
                p_rx = rs_sigmoid(x, 0.0, 1.0);
            }
            // ---- reception + UE.transmission_step (slice_l1.py:219-224, slice_ran.py:51-55)
            if (active) {
                bool received = false;
                if (rbs > 0) {
                    const unsigned c = L_ctr[lane];
                    if (needed) {
                        rs_stream st = {key0, key1, (uint32_t)ran, L_serial[lane], c};
                        received = rs_stream_uniform(&st) < p_rx;
                    }
                    L_ctr[lane] = c + 1u;
                }
                if (!received) bits = 0;
                const double nq = queue - (double)bits;
                queue = nq > 0.0 ? nq : 0.0;
                th = pf_a * th + pf_share(bits);
                ue_bits = bits;
                ue_prbs = rbs;
            }
            stat_sched += 1u;
        }

        // ================= SliceRANeMBB.update_info of every RAN slice (slice_ran.py:278-305); Q2: stale bits/prbs count
        if (active) {
            L_acc_bits[lane] += ue_bits;
            L_acc_prbs[lane] += ue_prbs;
        }
        for (int m = 0; m < M; ++m) {
            const bool mine = active && ran == m;
            const unsigned long long m_c = __builtin_amdgcn_ballot_w64(mine && !is_vbr);
            const unsigned long long m_v = __builtin_amdgcn_ballot_w64(mine && is_vbr);
            int n_c = __popcll(m_c), n_v = __popcll(m_v);
            n_c = n_c > 1 ? n_c : 1;
            n_v = n_v > 1 ? n_v : 1;
            const double q_c = wave_sum((mine && !is_vbr) ? queue : 0.0);
            const double q_v = wave_sum((mine && is_vbr) ? queue : 0.0);
            const int s_c = wave_sum((mine && !is_vbr) ? e_snr : 0);
            const int s_v = wave_sum((mine && is_vbr) ? e_snr : 0);
            if (lane == 0) {
                L_info[m][3] += q_c / (double)n_c;
                L_info[m][4] += (double)s_c / (double)n_c;
                L_info[m][8] += q_v / (double)n_v;
                L_info[m][9] += (double)s_v / (double)n_v;
            }
        }
        __builtin_amdgcn_wave_barrier();

        if (TRACE) {
            rs_alloc_rec rec;
            rec.serial = active ? (int32_t)L_serial[lane] : 0;
            rec.type = active ? ((flags & 1) | (ran << 8)) : 0;
            rec.e_snr = active ? e_snr : 0;
            rec.prbs = active ? ue_prbs : 0;
            rec.bits = active ? (int64_t)ue_bits : 0;
            rec.queue = active ? queue : 0.0;
            rec.th = active ? th : 0.0;
            rec.p = (active && sched) ? p_rx : 0.0;
            A.trace[((size_t)rep * slots + t) * RS_MUX_UE + lane] = rec;
        }
    }
    flush();

    // ---- outputs: get_state of every RAN slice (slice_ran.py:321-325), compute_reward (slice_ran.py:307-319,
    // slice_l1.py:160-171: the L1 slice reports how many of its RAN slices are in breach)
    for (int i = lane; i < M * 10; i += 64) {
        const int m = i / 10, k = i % 10;
        A.obs[(size_t)rep * D->n_vars + m * RS_N_EMBB_VARS + k] = (float)(L_info[m][k] / D->norm[k]);
        A.info[((size_t)rep * D->n_slices + m) * 10 + k] = L_info[m][k];
    }
    if (lane == 0) {
        const double obs_time = slots * slot_len;
        int viol = 0;
        for (int m = 0; m < M; ++m) {
            const double* I = L_info[m];
            const bool cbr_ok = (I[1] / obs_time > D->sla[0]) || (I[2] / slots > D->sla[1]) || (I[3] / slots < D->sla[2]);
            const bool vbr_ok = (I[6] / obs_time > D->sla[3]) || (I[7] / slots > D->sla[4]) || (I[8] / slots < D->sla[5]);
            viol += !(cbr_ok && vbr_ok);
        }
        A.violations[rep * n_act_entries + 0] = viol;
        A.labels[rep * n_act_entries + 0] = viol == 0 ? 1 : -1;
        S.t_n_ue[rep * M] = n_ue;
        S.t_cost[rep * M] = 0;
        uint64_t* c = A.counters + (size_t)(rep * M) * 4;
        c[0] += (uint64_t)stat_ue * (unsigned)n_prb;
        c[2] += (uint64_t)stat_sched * (unsigned)((n_prb + gran - 1) / gran);
        c[3] += stat_ue;
    }
    if (lane < M) {
        const int task = rep * M + lane;
        S.t_cbr_at[task] = L_slice[lane][0];
        S.t_vbr_at[task] = L_slice[lane][1];
        S.t_ctr[task] = (uint32_t)L_slice[lane][2];
        S.t_serial[task] = (uint32_t)L_slice[lane][3];
    }
    if (err != 0) atomicOr(&S.err[rep], err);
    if (active) {
        const size_t ui = mux_ui(rep, M, lane);
        S.u_queue[ui] = queue;
        S.u_th[ui] = th;
        S.u_nominal[ui] = L_nom[lane];
        S.u_hold_at[ui] = L_hold[lane];
        S.u_e_snr[ui] = e_snr;
        S.u_findex[ui] = findex;
        S.u_bits[ui] = ue_bits;
        S.u_prbs[ui] = ue_prbs;
        S.u_vbr_at[ui] = L_uvbr[lane];
        S.u_ctr[ui] = L_ctr[lane];
        S.u_serial[ui] = L_serial[lane];
        S.u_flags[ui] = flags;
#pragma unroll
        for (int k = 0; k < RS_BURSTS; ++k) S.u_burst[mux_bi(rep, M, lane, k)] = L_burst[k][lane];
    }
}

// ---- the multiplexed SliceL1mMTC: one wave per replica, one FIFO for all mMTC RAN slices.  A FIFO entry packs the
// remaining repetitions (low 24 bits) and its RAN slice (bits 24..); arrivals join RAN slice by RAN slice, device order
// inside a slice (slice_l1.py:90-93); the per-slice means are running integer sums as in mtc_step_kernel.
// dynamic LDS = 2 * cap_total * 4 bytes (FIFO) + n_mmtc * MTC_DEV_MAX * 4 bytes (next arrival of every device)
__global__ __launch_bounds__(64) void mtc_mux_step_kernel(MtcArgs A) {
    extern __shared__ int32_t lds[];
    const RsDev* __restrict__ D = A.D;
    const MtcState& M = A.M;
    const int lane = (int)threadIdx.x;
    const int rep = (int)blockIdx.x;
    const int NS = D->n_mmtc;
    const int cap1 = M.cap, cap = cap1 * NS;
    int32_t* q_rep = lds;            // packed: repetitions | slice << 24
    int32_t* q_start = lds + cap;
    int32_t* dnext = lds + 2 * cap;  // [NS][MTC_DEV_MAX]
    const int n_act_entries = (D->n_embb > 0 ? 1 : 0) + 1;
    const int n_prbs = A.actions[rep * n_act_entries + (D->n_embb > 0 ? 1 : 0)];
    const size_t task0 = (size_t)rep * NS;

    int n_users = 0;
    int n_of[RS_MUX_RAN];
    int64_t s_start[RS_MUX_RAN], s_rep[RS_MUX_RAN];
    int min_next[RS_MUX_RAN];
    double i_delay[RS_MUX_RAN], i_rep[RS_MUX_RAN], i_dev[RS_MUX_RAN];
#pragma unroll
    for (int s = 0; s < RS_MUX_RAN; ++s) {
        n_of[s] = 0; s_start[s] = 0; s_rep[s] = 0; min_next[s] = RS_NEVER;
        i_delay[s] = 0.0; i_rep[s] = 0.0; i_dev[s] = 0.0;
        if (s < NS) {
            n_of[s] = M.n_users[task0 + s];
            s_start[s] = M.s_start[task0 + s];
            s_rep[s] = M.s_rep[task0 + s];
            n_users += n_of[s];
            for (int i = lane; i < MTC_DEV_MAX; i += 64) {
                const int v = M.dev_next[(task0 + s) * MTC_DEV_MAX + i];
                dnext[s * MTC_DEV_MAX + i] = v;
                min_next[s] = v < min_next[s] ? v : min_next[s];
            }
        }
    }
    for (int i = lane; i < n_users; i += 64) {  // the FIFO occupies the replica's rows of the per-task queues
        q_rep[i] = M.q_rep[task0 * cap1 + i];
        q_start[i] = M.q_start[task0 * cap1 + i];
    }
    __builtin_amdgcn_wave_barrier();
    int err = 0;
    const int slots = D->slots;
    const int clock0 = (int)A.run[0];
    for (int t = 0; t < slots; ++t) {
        const int now = clock0 + t + 1;
        // ---- arrivals: RAN slice by RAN slice, device-index order (slice_l1.py:90-93, slice_ran.py:103-121)
#pragma unroll
        for (int s = 0; s < RS_MUX_RAN; ++s) {
            if (s >= NS) continue;
            if (__builtin_amdgcn_ballot_w64(min_next[s] == now) == 0ull) continue;
            int mn = RS_NEVER;
            for (int k = 0; k < MTC_DEV_PER_LANE; ++k) {
                const int dv = k * 64 + lane;
                int nx = dnext[s * MTC_DEV_MAX + dv];
                const bool fire = nx == now;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(fire);
                if (mk != 0ull) {
                    const int before = __popcll(mk & ((1ull << lane) - 1ull));
                    const int cnt = __popcll(mk);
                    int rsum = 0;
                    if (fire) {
                        const size_t o = (task0 + s) * MTC_DEV_MAX + dv;
                        const int pos = n_users + before;
                        const int r = M.dev_rep[o];
                        rsum = r;
                        if (pos < cap) {
                            q_rep[pos] = r | (s << 24);
                            q_start[pos] = now;
                        } else {
                            err = 1;
                        }
                        nx = now + M.dev_period[o];
                        dnext[s * MTC_DEV_MAX + dv] = nx;
                    }
                    rsum = wave_sum(rsum);
                    int take = n_users + cnt <= cap ? cnt : cap - n_users;
                    if (take < cnt) {
                        err = 1;
                        take = take > 0 ? take : 0;
                    }
                    s_rep[s] += rsum;
                    s_start[s] += (int64_t)cnt * now;
                    n_of[s] += take;
                    n_users += take;
                }
                mn = nx < mn ? nx : mn;
            }
            min_next[s] = mn;
            __builtin_amdgcn_wave_barrier();
        }
        // ---- transmissions: the first n_tx entries of the FIFO use one carrier each, whatever their slice
        const int n_tx = n_prbs < n_users ? n_prbs : n_users;
        bool any_done = false;
        for (int base = 0; base < n_tx; base += 64) {
            const int i = base + lane;
            bool done = false;
            int sl_i = -1;
            if (i < n_tx) {
                const int v = q_rep[i] - 1;
                q_rep[i] = v;
                done = (v & 0xffffff) == 0;
                sl_i = v >> 24;
            }
            any_done = any_done || (__builtin_amdgcn_ballot_w64(done) != 0ull);
#pragma unroll
            for (int s = 0; s < RS_MUX_RAN; ++s)
                if (s < NS) s_rep[s] -= __popcll(__builtin_amdgcn_ballot_w64(sl_i == s));
        }
        // ---- drop finished entries, order preserved (slice_l1.py:102-107)
        if (any_done) {
            int wpos = 0;
            __builtin_amdgcn_wave_barrier();
            for (int base = 0; base < n_users; base += 64) {
                const int i = base + lane;
                int v = 0, st = 0;
                if (i < n_users) {
                    v = q_rep[i];
                    st = q_start[i];
                }
                const bool keep = i < n_users && (v & 0xffffff) > 0;
                const bool drop = i < n_users && !keep;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(keep);
                const int pos = wpos + __popcll(mk & ((1ull << lane) - 1ull));
#pragma unroll
                for (int s = 0; s < RS_MUX_RAN; ++s) {
                    if (s >= NS) continue;
                    const bool ds = drop && (v >> 24) == s;
                    const unsigned long long dm = __builtin_amdgcn_ballot_w64(ds);
                    if (dm != 0ull) {
                        int rs_ = ds ? st : 0;
                        rs_ = wave_sum(rs_);
                        s_start[s] -= rs_;
                        n_of[s] -= __popcll(dm);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (keep) {
                    q_rep[pos] = v;
                    q_start[pos] = st;
                }
                __builtin_amdgcn_wave_barrier();
                wpos += __popcll(mk);
            }
            n_users = wpos;
        }
        // ---- per-slot summary of every RAN slice over its own entries (slice_l1.py:112-125)
#pragma unroll
        for (int s = 0; s < RS_MUX_RAN; ++s) {
            if (s >= NS) continue;
            double delay = 0.0, avg_rep = 0.0;
            if (n_of[s] > 0) {
                const int64_t sd = (int64_t)n_of[s] * now - s_start[s];
                delay = (double)sd / (double)n_of[s];
                avg_rep = RS_RINT((double)s_rep[s] / (double)n_of[s]);
            }
            i_delay[s] += delay;
            i_rep[s] += avg_rep;
            i_dev[s] += (double)n_of[s];
        }
    }
    // ---- write back
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n_users; i += 64) {
        M.q_rep[task0 * cap1 + i] = q_rep[i];
        M.q_start[task0 * cap1 + i] = q_start[i];
    }
    const bool any_err = __builtin_amdgcn_ballot_w64(err != 0) != 0ull;
    int viol = 0;
#pragma unroll
    for (int s = 0; s < RS_MUX_RAN; ++s) {
        if (s >= NS) continue;
        for (int i = lane; i < MTC_DEV_MAX; i += 64) M.dev_next[(task0 + s) * MTC_DEV_MAX + i] = dnext[s * MTC_DEV_MAX + i];
        const bool ok = i_delay[s] / slots < D->sla_mtc_delay;
        viol += ok ? 0 : 1;
        if (lane == 0) {
            M.n_users[task0 + s] = n_of[s];
            M.s_start[task0 + s] = s_start[s];
            M.s_rep[task0 + s] = s_rep[s];
            float* o = A.obs + (size_t)rep * D->n_vars + D->n_embb * RS_N_EMBB_VARS + s * RS_N_MMTC_VARS;
            o[0] = (float)(i_dev[s] / D->norm_mmtc[0]);
            o[1] = (float)(i_rep[s] / D->norm_mmtc[1]);
            o[2] = (float)(i_delay[s] / D->norm_mmtc[2]);
            double* inf = A.info + ((size_t)rep * D->n_slices + D->n_embb + s) * 10;
            inf[0] = i_delay[s];
            inf[1] = i_rep[s];
            inf[2] = i_dev[s];
            for (int k = 3; k < 10; ++k) inf[k] = 0.0;
        }
    }
    if (lane == 0) {
        const int a = D->n_embb > 0 ? 1 : 0;
        A.violations[rep * n_act_entries + a] = viol;
        A.labels[rep * n_act_entries + a] = viol == 0 ? 1 : -1;
        if (any_err) atomicOr(&A.err[rep], 4);
    }
}

}  // namespace rs

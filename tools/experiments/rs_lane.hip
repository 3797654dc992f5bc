// NOT PART OF THE PRODUCT: the lane-per-task form of the eMBB step, kept as the record of an experiment (round 2: exact, 4-12x
// slower than the group kernels at every batch size).  It was compiled into libranslice.so behind RANSLICE_LANE=1
// until round 3; to build it again, include it from csrc/rs_api.hip after rs_embb.hip and restore the launch hook of
// commit 5d583f7.
// eMBB step, LANE-PER-TASK form (VERDICT r1 #3 asked for it for batches of >= 8192 replicas per GPU).
//
// embb_step_kernel (rs_embb.hip) gives a (replica, slice) task 16 lanes, of which three or four hold a UE: its
// instruction stream is issued for 64 lanes and used by a fifth of them.  Here ONE LANE owns a task for the whole step
// and walks its UEs one after the other, the way the reference's Python does: 64 tasks per wavefront, no cross-lane
// traffic at all, every loop runs for the longest of the 64 tasks; an eighth of the instructions per task.
//
// Same state fields, same Philox streams, same arithmetic in the same order as the group kernel, and bit-exact
// against the oracle at full size (tests/test_gpu_parity.py::test_lane_engine_matches_oracle; the whole of
// tests/test_gpu_fullsize.py passes with RANSLICE_LANE=1).  Only the HBM layout of the per-UE arrays differs --
// lane-major, [task / 64][UE][task % 64], one 256/512-byte segment per field and wave -- which is why a handle keeps
// the engine it was created with.
//
// MEASURED (tools/lane_check.sh, MI355X, scenario_0, stationary population): 14.1 ms per step at 4096 replicas
// (group kernel 1.13), 15.5 at 8192 (2.08), 17.2 at 16384 (3.83), 47.6 at 65536 (13.8).  The instruction count is
// not what bounds it: a lane's UE records sit in HBM/L2 behind loads that depend on each other (~1 us each, some
// 200 per slot), every lane walks its own fading column (64 cache lines per load instruction, nine batches of four per
// UE and slot), and at 235 VGPRs two waves per SIMD cannot hide any of that.  Moving the records into LDS (~84 B per
// UE, eight UEs per lane: three waves per CU) would cut the chain to an estimated 1.5 ms per round of waves -- a
// third better than the group kernel at >= 8192 replicas at best, and worse below.  So nothing selects this engine; it
// stays as the exact, measured answer to "would one lane per task pay on this chip?" (DESIGN.md section 8).
#pragma once
#include "rs_embb.hip"

namespace rs {

struct LaneWork {
    int32_t* evt;    // [U] earliest of: departure, next burst arrival, next burst end (RS_NEVER if none)
    int32_t* nact;   // [U] VBR bursts still running when the slot begins
    int32_t* q;      // [U] PF: queue in bits (clamped to 2^30)
    int32_t* rate;   // [U] PF: bits per RB of the UE's MCS | mcs << 16 | modulation << 24
    double* thl;     // [U] PF: local throughput average
    double* m;       // [U] PF: metric rate * [q > 0] / thl
};

struct LaneArgs {
    StepArgs a;
    LaneWork w;
};

__global__ __launch_bounds__(256, 2) void embb_lane_step_kernel(LaneArgs LA) {
    const StepArgs& A = LA.a;
    const LaneWork& W = LA.w;
    __shared__ int L_lut[RS_LUT_MAX];
    __shared__ double L_ref[32];
    const RsDev* __restrict__ D = A.D;
    const int tid = (int)threadIdx.x;
    if (tid < RS_LUT_MAX) L_lut[tid] = tid < D->lut_n ? ((D->mcs_mod[D->lut_mcs[tid]] << 24) | (D->lut_mcs[tid] << 16) | D->lut_rate[tid]) : 0;
    if (tid >= 64 && tid < 96) L_ref[tid - 64] = D->mcs_ref[tid - 64];
    __syncthreads();
    const RsState& S = *A.S;
    const int n_tasks = D->n_envs * D->n_embb;
    int task = (int)blockIdx.x * 256 + tid;
    const bool valid = task < n_tasks;
    if (!wave_any(valid)) return;
    const int lane = tid & 63;
    const size_t wv = (size_t)(task >> 6);
    if (!valid) task = n_tasks - 1;
    const int rep = task / D->n_embb;
    const int sl = task - rep * D->n_embb;
    const int n_slices = D->n_slices;
    const int P = D->P;
    const int slots = D->slots;
    const double slot_len = D->slot_length;
    const double pf_a = D->pf_a, pf_b = D->pf_b, slot_rc = D->slot_rc;
    const bool pf_div_fast = D->pf_div_fast != 0;
    const int gran = D->gran;
    const bool has_nan = D->has_nan != 0;
    const int clock0 = (int)A.run[0];
    auto pf_share = [&](int b) -> double {  // pf_b * bits / slot_length, as in the group kernel
        const double xb = pf_b * (double)b;
        if (pf_div_fast) {
            const double q0 = xb * slot_rc;
            return __builtin_fma(__builtin_fma(-q0, slot_len, xb), slot_rc, q0);
        }
        return xb / slot_len;
    };
    // lane-major index of UE u of my task, and of its burst entry k
    auto UI = [&](int u) -> size_t { return ((wv * RS_GROUP + (size_t)u) << 6) + (size_t)lane; };
    auto BI = [&](int u, int k) -> size_t { return ((wv * RS_BURSTS * RS_GROUP + (size_t)(k * RS_GROUP + u)) << 6) + (size_t)lane; };

    int prb_lo = 0;
    for (int q = 0; q < sl; ++q) prb_lo += A.actions[rep * n_slices + q];
    const int n_prb = valid ? A.actions[rep * n_slices + sl] : 0;
    const int n_pairs_full = n_prb / gran;
    const uint64_t seed = S.seeds[rep];
    const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);

    int n_ue = valid ? S.t_n_ue[task] : 0;
    int cbr_at = valid ? S.t_cbr_at[task] : RS_NEVER, vbr_at = valid ? S.t_vbr_at[task] : RS_NEVER;
    uint32_t sl_ctr = S.t_ctr[task], next_serial = S.t_serial[task];
    int err = 0;
    // earliest departure among my UEs
    int next_dep = RS_NEVER;
    for (int u = 0; wave_any(u < n_ue); ++u) {
        const int h = S.u_hold_at[UI(u)];
        if (u < n_ue) next_dep = h < next_dep ? h : next_dep;
    }
    // SliceRANeMBB.info (slice_ran.py:270-273): ten running sums
    double info[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) info[k] = 0.0;
    unsigned cnt_ue = 0, n_sched = 0, pf_rounds = 0;

    for (int t = 0; t < slots; ++t) {
        const int now = clock0 + t + 1;
        const int slot_counter = t + 1;
        // ================= SliceRANeMBB.slot: arrivals (slice_ran.py:205-249)
        const bool cbr_fire = valid && cbr_at == now, vbr_fire = valid && vbr_at == now;
        if (wave_any(cbr_fire || vbr_fire)) {
            int n_pend = 0, type0 = 0, type1 = 0;
            if (cbr_fire) {
                rs_stream st = {key0, key1, (uint32_t)sl, 0u, sl_ctr};
                const double ia = rs_stream_exponential(&st, D->cbr_ia_scale);
                sl_ctr = st.ctr;
                cbr_at = now + 1 + rint_slots(ia, slot_len);
                // cbr_cac (slice_ran.py:195-203) on this step's running sums
                const int cslots = slot_counter > 1 ? slot_counter : 1;
                const double time = cslots * slot_len;
                const double c_prb = info[2] / cslots;
                const double c_th = info[1] / time;
                if (!(c_prb >= D->sla[1] || c_th >= D->sla[0])) {
                    type0 = 0;
                    n_pend = 1;
                }
            }
            if (vbr_fire) {
                rs_stream st = {key0, key1, (uint32_t)sl, 0u, sl_ctr};
                const double ia = rs_stream_exponential(&st, D->vbr_ia_scale);
                sl_ctr = st.ctr;
                vbr_at = now + 1 + rint_slots(ia, slot_len);
                if (n_pend == 0) type0 = 1; else type1 = 1;
                n_pend += 1;
            }
            if (n_ue + n_pend > RS_GROUP) {
                err |= 1;  // RS_EOVERFLOW: UE capacity
                n_pend = RS_GROUP - n_ue;
            }
            for (int k = 0; wave_any(k < n_pend); ++k) {
                if (k < n_pend) {
                    const int u = n_ue + k;
                    const int type = k == 0 ? type0 : type1;
                    const unsigned userial = next_serial + (uint32_t)k;
                    rs_stream st = {key0, key1, (uint32_t)sl, userial, 0u};
                    int uvbr_at = RS_NEVER;
                    if (type == 1) {  // VbrSource.__init__ (traffic_generators.py:62-68)
                        const int v = (int)RS_RINT(rs_stream_exponential(&st, D->vbr_inter));
                        uvbr_at = v >= 1 ? now + v - 1 : RS_NEVER;  // Q5: 0 never fires
                    }
                    const double hold = rs_stream_exponential(&st, type == 0 ? D->cbr_hold_scale : D->vbr_hold_scale);
                    const int hv = rint_slots(hold, slot_len);
                    const int hold_at = hv >= 1 ? now + hv - 1 : RS_NEVER;  // Q5
                    int ftype = 0, fstep = 1, findex = 0;
                    double nominal = 0.0;
                    if (hold_at != now) {  // Q13: a one-slot holding time never joins the slice
                        // SINRSelectiveFading.insert_user (channel_models.py:163-169)
                        ftype = (int)rs_stream_integers(&st, RS_N_TRACES);
                        findex = (int)rs_stream_integers(&st, D->T[ftype]);
                        fstep = rs_stream_pm1(&st);
                        const MacroCell mc = macro_cell_draw(D, st);
                        nominal = mc.x;
                        st.ctr = (uint32_t)mc.y;
                    }
                    const size_t i = UI(u);
                    S.u_queue[i] = 0.0;
                    S.u_th[i] = 0.0;
                    S.u_nominal[i] = nominal;
                    S.u_hold_at[i] = hold_at;
                    S.u_e_snr[i] = 0;
                    S.u_findex[i] = findex;
                    S.u_bits[i] = 0;
                    S.u_prbs[i] = 0;
                    S.u_vbr_at[i] = uvbr_at;
                    S.u_ctr[i] = st.ctr;
                    S.u_serial[i] = userial;
                    S.u_flags[i] = type | (ftype << 1) | ((fstep > 0 ? 1 : 0) << 3);
                    for (int b = 0; b < RS_BURSTS; ++b) S.u_burst[BI(u, b)] = 0;
                    W.evt[i] = hold_at < uvbr_at ? hold_at : uvbr_at;
                    W.nact[i] = 0;
                    next_dep = hold_at < next_dep ? hold_at : next_dep;
                }
            }
            n_ue += n_pend;
            next_serial += (uint32_t)n_pend;
        }

        // ================= departures (slice_ran.py:251-261) + extract_users (slice_l1.py:187-191): the list keeps its order
        if (wave_any(next_dep == now)) {
            if (next_dep == now) {
                int w = 0, nd = RS_NEVER;
                for (int u = 0; u < n_ue; ++u) {
                    const size_t i = UI(u);
                    const int h = S.u_hold_at[i];
                    if (h == now) continue;
                    if (w != u) {
                        const size_t o = UI(w);
                        S.u_queue[o] = S.u_queue[i];
                        S.u_th[o] = S.u_th[i];
                        S.u_nominal[o] = S.u_nominal[i];
                        S.u_hold_at[o] = h;
                        S.u_e_snr[o] = S.u_e_snr[i];
                        S.u_findex[o] = S.u_findex[i];
                        S.u_bits[o] = S.u_bits[i];
                        S.u_prbs[o] = S.u_prbs[i];
                        S.u_vbr_at[o] = S.u_vbr_at[i];
                        S.u_ctr[o] = S.u_ctr[i];
                        S.u_serial[o] = S.u_serial[i];
                        S.u_flags[o] = S.u_flags[i];
                        for (int b = 0; b < RS_BURSTS; ++b) S.u_burst[BI(w, b)] = S.u_burst[BI(u, b)];
                        W.evt[o] = W.evt[i];
                        W.nact[o] = W.nact[i];
                    }
                    nd = h < nd ? h : nd;
                    w += 1;
                }
                n_ue = w;
                next_dep = nd;
            }
        }

        // ================= pass 1 over my UEs: VBR burst events, traffic_step, walker + channel estimate, PF set-up
        const bool has_prb = valid && n_prb > 0;
        int need = 0, n_cont = 0;       // RB pairs that would drain every queue; UEs with data
        bool any_queue = false;
        for (int u = 0; wave_any(u < n_ue); ++u) {
            const bool on = u < n_ue;
            const size_t i = UI(u);
            int flags = S.u_flags[i];
            double queue = S.u_queue[i];
            const bool is_vbr = (flags & 1) != 0;
            int n_cur = 0;
            if (wave_any(on && is_vbr)) n_cur = W.nact[i];
            // ---- VbrSource.step events (traffic_generators.py:70-99) on absolute end times
            const int evt = W.evt[i];
            if (wave_any(on && evt == now)) {
                if (on && evt == now) {
                    int cnt = (flags >> 8) & 0xff, nxt = RS_NEVER;  // the never-ending bursts (Q5) always emit
                    int freek = RS_BURSTS;                           // a free entry for a burst that may start now
                    for (int k = 0; k < RS_BURSTS; ++k) {
                        const unsigned e = S.u_burst[BI(u, k)];
                        const int rel = e != 0u ? rs_burst_rel(e, now) : 0;
                        if (e != 0u && rel <= 0) S.u_burst[BI(u, k)] = 0;  // ends exactly now: dropped without emitting
                        if (rel > 0) {
                            cnt += 1;
                            nxt = now + rel < nxt ? now + rel : nxt;
                        } else if (freek == RS_BURSTS) {
                            freek = k;
                        }
                    }
                    n_cur = cnt;
                    int uvbr_at = S.u_vbr_at[i];
                    if (uvbr_at == now) {
                        rs_stream st = {key0, key1, (uint32_t)sl, S.u_serial[i], S.u_ctr[i]};
                        const int d = (int)RS_RINT(rs_stream_exponential(&st, D->vbr_b_size));
                        const int v = (int)RS_RINT(rs_stream_exponential(&st, D->vbr_inter));
                        S.u_ctr[i] = st.ctr;
                        if (d < 1) {  // Q5: a duration that rounds to 0 never counts down to 0: the burst emits for ever
                            if (((flags >> 8) & 0xff) == 0xff) err |= 2;
                            else flags += 1 << 8;
                            cnt += 1;
                        } else if (freek == RS_BURSTS || d >= RS_BURST_MAX_LEN) {
                            err |= 2;  // RS_EOVERFLOW: more than RS_BURSTS bursts running, or one too long for its code
                        } else {
                            S.u_burst[BI(u, freek)] = (uint16_t)rs_burst_code(now + d);
                            cnt += 1;
                            nxt = now + d < nxt ? now + d : nxt;
                        }
                        uvbr_at = v >= 1 ? now + v : RS_NEVER;
                        S.u_vbr_at[i] = uvbr_at;
                    }
                    W.nact[i] = cnt;
                    const int h = S.u_hold_at[i];
                    const int e2 = h < uvbr_at ? h : uvbr_at;
                    W.evt[i] = e2 < nxt ? e2 : nxt;
                }
            }
            // ---- UE.traffic_step (slice_ran.py:47-49)
            const double new_bits = is_vbr ? (double)n_cur * D->vbr_p_size : D->cbr_bits;
            if (on) {
                queue += new_bits;
                info[is_vbr ? 5 : 0] += (double)(int)new_bits;
            }
            // ---- channel: this slot's walker step and estimate (Q3: with no PRBs the walker does not move, e_snr is stale)
            int e_snr = S.u_e_snr[i];
            if (wave_any(on && has_prb)) {
                const int ftype = (flags >> 1) & 3;
                int findex = S.u_findex[i];
                int fstep = (flags & 8) ? 1 : -1;
                const bool go = on && has_prb;
                if (go) {
                    walker_advance(findex, fstep, D->T[ftype], has_nan, A.fad_valid + (size_t)D->valid_off[ftype], key0, key1,
                                   (uint32_t)sl, S.u_serial[i], (uint32_t)now);
                    flags = (flags & ~8) | ((fstep > 0 ? 1 : 0) << 3);
                }
                const double nom = S.u_nominal[i];
                const double* __restrict__ colp = A.fad + (go ? (size_t)D->fad_off[ftype] + (size_t)findex * P + prb_lo : 0);
                const double sum = lane_pairwise(n_prb, go, [&](int j) { return colp[j] + nom; });
                if (go) {
                    e_snr = (int)RS_RINT(sum / (double)n_prb);  // round(np.mean(...)): half-to-even (Q7)
                    S.u_findex[i] = findex;
                    S.u_e_snr[i] = e_snr;
                }
            }
            if (on) {
                S.u_flags[i] = flags;
                S.u_queue[i] = queue;
                // ---- ProportionalFair.allocate set-up (schedulers.py:30-43)
                int li = e_snr - D->lut_lo;
                li = li < 0 ? 0 : (li >= D->lut_n ? D->lut_n - 1 : li);
                const int lut = L_lut[li];
                const int rate = lut & 0xffff;
                const int q = (int)(queue < 1073741824.0 ? queue : 1073741824.0);
                const double th = S.u_th[i];
                const double thl = th > 1.0 ? th : 1.0;
                const int per_it = gran * rate;
                const int k_u = q > 0 ? (int)((double)(q + per_it - 1) / (double)per_it) : 0;
                W.q[i] = q;
                W.rate[i] = lut;
                W.thl[i] = thl;
                need += k_u;
                n_cont += q > 0 ? 1 : 0;
                any_queue = any_queue || queue > 0.0;
            }
        }
        cnt_ue += (unsigned)n_ue;

        // ================= scheduling (slice_l1.py:215-224)
        const bool sched = valid && any_queue && n_prb > 0;
        if (sched) n_sched += 1;
        if (wave_any(sched)) {
            // ---- closed forms (see rs_embb.hip): under-loaded slot; one UE holds all the data; contested otherwise
            const bool under = sched && need <= n_pairs_full;
            const bool single = sched && !under && n_cont == 1;
            const bool contested = sched && !under && !single;
            for (int u = 0; wave_any((under || single) && u < n_ue); ++u) {
                if ((under || single) && u < n_ue) {
                    const size_t i = UI(u);
                    const int q = W.q[i], rate = W.rate[i] & 0xffff;
                    int rbs = 0, bits = 0;
                    if (under) {
                        const int per_it = gran * rate;
                        const int k_u = q > 0 ? (int)((double)(q + per_it - 1) / (double)per_it) : 0;
                        rbs = k_u * gran;
                        bits = k_u > 0 ? q : 0;
                        if (u == 0) rbs += n_prb - need * gran;  // the idle remainder goes to UE 0 (Q4)
                    } else if (q > 0) {
                        const int cap_bits = n_prb * rate;
                        rbs = n_prb;
                        bits = q < cap_bits ? q : cap_bits;
                    }
                    S.u_prbs[i] = rbs;
                    S.u_bits[i] = bits;
                }
            }
            if (wave_any(contested)) {
                // the reference loop (schedulers.py:44-62), one RB pair per iteration to the first UE of maximal metric
                for (int u = 0; wave_any(contested && u < n_ue); ++u) {
                    if (contested && u < n_ue) {
                        const size_t i = UI(u);
                        const int q = W.q[i];
                        W.m[i] = (q > 0 ? (double)(W.rate[i] & 0xffff) : 0.0) / W.thl[i];
                        S.u_prbs[i] = 0;
                        S.u_bits[i] = 0;
                    }
                }
                int r = contested ? 0 : n_prb;
                while (wave_any(r < n_prb)) {
                    const bool go = r < n_prb;
                    // np.argmax: first maximum
                    double best = -1.0;
                    int idx = 0;
                    for (int u = 0; wave_any(go && u < n_ue); ++u) {
                        const double mu = W.m[UI(u)];
                        if (go && u < n_ue && mu > best) {
                            best = mu;
                            idx = u;
                        }
                    }
                    if (go) {
                        pf_rounds += 1;
                        if (best <= 0.0) {
                            // every queue is empty: argmax of an all-zero metric is UE 0 for all the remaining pairs (Q4)
                            S.u_prbs[UI(0)] += n_prb - r;
                            r = n_prb;
                        } else {
                            const size_t i = UI(idx);
                            const int rate = W.rate[i] & 0xffff;
                            int q = W.q[i], bits = S.u_bits[i], rbs = S.u_prbs[i];
                            double thl = W.thl[i];
                            const double rate_d = (double)rate;
                            // the leader keeps the pairs while it stays the (first) maximum: second-best metric and index
                            double m2 = -1.0;
                            int idx2 = 0;
                            for (int u = 0; u < n_ue; ++u) {
                                const double mu = W.m[UI(u)];
                                if (u != idx && mu > m2) {
                                    m2 = mu;
                                    idx2 = u;
                                }
                            }
                            double mnew;
                            for (;;) {
                                const int prbs = n_prb - r < gran ? n_prb - r : gran;
                                rbs += prbs;
                                const int tx = prbs * rate < q ? prbs * rate : q;
                                q -= tx;
                                bits += tx;
                                r += gran;
                                if (q > 0) {
                                    thl = pf_a * thl + pf_share(bits);
                                    mnew = rate_d / thl;
                                } else {
                                    mnew = 0.0;
                                }
                                if (r >= n_prb) break;
                                if (!(mnew > m2 || (mnew == m2 && idx < idx2))) break;
                            }
                            if (r > n_prb) r = n_prb;
                            W.q[i] = q;
                            W.thl[i] = thl;
                            W.m[i] = mnew;
                            S.u_bits[i] = bits;
                            S.u_prbs[i] = rbs;
                        }
                    }
                }
            }

            // ================= pass 2: MCSCodeset.response (channel_models.py:297-313), reception, UE.transmission_step
            int prb_i = 0;
            for (int u = 0; wave_any(sched && u < n_ue); ++u) {
                const bool on = sched && u < n_ue;
                const size_t i = UI(u);
                const int rbs = on ? S.u_prbs[i] : 0;
                int bits = S.u_bits[i];
                const int lut = W.rate[i];
                const int mcs = (lut >> 16) & 0xff, mod = lut >> 24;
                const bool needed = on && rbs > 0 && bits > 0;
                double p_rx = 0.0;
                if (wave_any(needed)) {
                    const int flags = S.u_flags[i];
                    const int ftype = (flags >> 1) & 3;
                    const double nom = S.u_nominal[i];
                    const double x0 = sel3(mod, D->mi_x0[0], D->mi_x0[1], D->mi_x0[2]);
                    const double kk = sel3(mod, D->mi_k[0], D->mi_k[1], D->mi_k[2]);
                    const double* __restrict__ sp =
                        A.fad + (needed ? (size_t)D->fad_off[ftype] + (size_t)S.u_findex[i] * P + prb_lo + prb_i : 0);
                    const bool one = rbs == 1;
                    const double sum_rx = lane_pairwise(rbs, needed, [&](int j) {
                        const double x = sp[j] + nom;
                        return one ? x : rs_sigmoid(x, x0, kk);
                    });
                    if (needed) {
                        double s_eff = sum_rx;  // rbs == 1: the RB's SINR itself (0 + x, numpy's n < 8 path)
                        if (rbs > 1) s_eff = rs_inv_sigmoid(sum_rx / (double)rbs, x0, kk);
                        const double x = D->mcsA * (s_eff - L_ref[mcs]) - D->mcsB;
                        p_rx = rs_sigmoid(x, 0.0, 1.0);
                    }
                }
                if (on) {
                    bool received = false;
                    if (rbs > 0) {  // the draw is consumed whether or not anything rides on it
                        const unsigned c = S.u_ctr[i];
                        if (needed) {
                            rs_stream st = {key0, key1, (uint32_t)sl, S.u_serial[i], c};
                            received = rs_stream_uniform(&st) < p_rx;
                        }
                        S.u_ctr[i] = c + 1u;
                    }
                    if (!received) bits = 0;
                    const double queue = S.u_queue[i];
                    const double nq = queue - (double)bits;
                    S.u_queue[i] = nq > 0.0 ? nq : 0.0;
                    S.u_th[i] = pf_a * S.u_th[i] + pf_share(bits);
                    S.u_bits[i] = bits;
                    prb_i += rbs;
                }
            }
        }

        // ================= SliceRANeMBB.update_info (slice_ran.py:278-305); Q2: stale bits/prbs count
        {
            int n_c = 0, n_v = 0, s_c = 0, s_v = 0;
            double q_c = 0.0, q_v = 0.0;
            for (int u = 0; wave_any(u < n_ue); ++u) {
                if (u < n_ue) {
                    const size_t i = UI(u);
                    const bool is_vbr = (S.u_flags[i] & 1) != 0;
                    const double queue = S.u_queue[i];
                    const int e_snr = S.u_e_snr[i], ub = S.u_bits[i], up = S.u_prbs[i];
                    if (is_vbr) {
                        n_v += 1;
                        q_v += queue;
                        s_v += e_snr;
                        info[6] += (double)ub;
                        info[7] += (double)up;
                    } else {
                        n_c += 1;
                        q_c += queue;
                        s_c += e_snr;
                        info[1] += (double)ub;
                        info[2] += (double)up;
                    }
                }
            }
            n_c = n_c > 1 ? n_c : 1;
            n_v = n_v > 1 ? n_v : 1;
            if (valid) {
                info[3] += q_c / (double)n_c;
                info[4] += (double)s_c / (double)n_c;
                info[8] += q_v / (double)n_v;
                info[9] += (double)s_v / (double)n_v;
            }
        }
    }

    // ---- outputs: get_state (slice_ran.py:321-325), compute_reward (slice_ran.py:307-319)
    if (valid) {
#pragma unroll
        for (int k = 0; k < RS_N_EMBB_VARS; ++k) {
            A.obs[(size_t)rep * D->n_vars + sl * RS_N_EMBB_VARS + k] = (float)(info[k] / D->norm[k]);
            A.info[((size_t)rep * n_slices + sl) * 10 + k] = info[k];
        }
        const double obs_time = slots * slot_len;
        const bool cbr_ok = (info[1] / obs_time > D->sla[0]) || (info[2] / slots > D->sla[1]) || (info[3] / slots < D->sla[2]);
        const bool vbr_ok = (info[6] / obs_time > D->sla[3]) || (info[7] / slots > D->sla[4]) || (info[8] / slots < D->sla[5]);
        const int viol = !(cbr_ok && vbr_ok);
        A.violations[rep * n_slices + sl] = viol;
        A.labels[rep * n_slices + sl] = viol == 0 ? 1 : -1;
        S.t_n_ue[task] = n_ue;
        S.t_cbr_at[task] = cbr_at;
        S.t_vbr_at[task] = vbr_at;
        S.t_ctr[task] = sl_ctr;
        S.t_serial[task] = next_serial;
        S.t_cost[task] = (int)pf_rounds;
        uint64_t* c = A.counters + (size_t)task * 4;
        c[0] += cnt_ue * (unsigned)n_prb;
        c[2] += n_sched * (unsigned)((n_prb + gran - 1) / gran);
        c[3] += cnt_ue;
        if (err != 0) atomicOr(&S.err[rep], err);
    }
}

}  // namespace rs

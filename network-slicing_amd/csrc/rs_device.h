// rs_device.h -- device-side data layout of the batched RAN-slice simulator (gfx950).
//
// One "task" is one eMBB slice of one env replica.  A task is advanced by a group of 16 or 32
// lanes: lane u of the group owns UE u of the slice (UE list order = arrival order,
// reference slice_l1.py:183-191), so every per-UE quantity lives in a register and the
// per-slot reductions the reference does with Python loops (argmax of the PF metric,
// RB prefix sums, per-class sums) are cross-lane operations.  Persistent state is kept in HBM
// as structure-of-arrays with the UE index fastest, so a group's load of one field is one
// contiguous 128/256-byte segment.
#pragma once
#include <stdint.h>

#define RS_GROUP 32          // lanes per task == UE capacity per slice
#define RS_BURSTS 16         // VBR bursts that can run at once per UE (8 overflowed once per ~1e11 UE-slots: a bench run)
#define RS_BURST_MAX_LEN 16000  // longest burst the 15-bit end-time code holds (P(Exp(500) >= 16000) = 1e-14)
#define RS_LUT_MAX 64
#define RS_NEVER 0x7fffffff  // absolute slot time that never arrives (reference quirk Q5)
#define RS_MAX_PRBS 256

// End slot of a running VBR burst in 16 bits: bit 15 = occupied, bits 0-14 = absolute end slot modulo 2^15.  A burst is
// freed in the very slot it ends, so an occupied entry is always within RS_BURST_MAX_LEN slots of its end and the
// signed 15-bit difference to the clock is its remaining life.
#define rs_burst_code(end_abs) (0x8000u | ((unsigned)(end_abs) & 0x7fffu))
#define rs_burst_rel(code, now) ((int)(((((unsigned)(code)) - (unsigned)(now)) & 0x7fffu) ^ 0x4000u) - 0x4000)

// Immutable parameters, one copy in HBM, read through scalar loads.
struct RsDev {
    int32_t n_envs, n_prbs, n_embb, n_mmtc, n_slices, slots, n_vars;  // n_slices = RAN slices (info rows)
    int32_t n_act;          // action / label entries per replica: n_slices, or one per L1 slice when multiplexed
    int32_t mux;            // rs_config.l1_multiplex
    int32_t P;              // row length of the fading tables (PRBs after row extension)
    int32_t T[3];           // time samples per trace
    int32_t has_nan;        // any trace column flagged invalid
    int64_t fad_off[3];     // element offset of trace f inside the table buffer
    int64_t valid_off[3];   // byte offset of trace f inside the column-valid buffer
    double slot_length;
    double cbr_bits;        // CbrSource packet size = bit_rate * 1e-3 (traffic_generators.py:56-59)
    double cbr_ia_scale, cbr_hold_scale;   // 1/lambda, t_mean (slice_ran.py:208,220)
    double vbr_ia_scale, vbr_hold_scale;   // slice_ran.py:243,238
    double vbr_p_size, vbr_b_size, vbr_inter;  // traffic_generators.py:62-66
    double sla[6], norm[10];
    double prop_A, prop_B;
    double mcsA, mcsB;      // MCSCodeset.compute_factors (channel_models.py:272-279)
    double pf_a, pf_b;      // 1 - 1/window, 1/window (schedulers.py:16-17, slice_ran.py:30-31)
    double slot_rc;         // RN(1 / slot_length)
    int32_t pf_div_fast;    // 1: (pf_b * bits) / slot_length may be formed as q = x * slot_rc refined by two fmas;
                            // rs_create checked every integer `bits` a slot can reach against the IEEE divide
    int32_t gran;           // PF granularity
    int32_t lut_lo, lut_n;  // e_snr -> (mcs, rate) lookup, clamped outside [lut_lo, lut_lo+lut_n)
    int32_t lut_mcs[RS_LUT_MAX], lut_rate[RS_LUT_MAX];
    double mcs_ref[32];
    int32_t mcs_mod[32];    // modulation of each MCS (0 qpsk, 1 16qam, 2 64qam)
    double mi_x0[3], mi_k[3];  // mutual-information sigmoid per modulation (channel_models.py:268-270)
    // the reception test by guard band (rs_embb.hip: fast_sigmoid; set up by rx_fast_setup in rs_api.hip)
    double rx_band;         // guard band of the MI-sum comparison, per RB of the span; 0 = every UE takes the exact path
    double est_band;        // guard band of round(mean SINR) formed from the prefix sums; 0 = every estimate by the pairwise sum
    int32_t col_off[3];     // columns of the traces before trace f (fad_off / P)
    int32_t pad_rx;
    double rx_band1;        // guard band (dB) of the single-RB comparison
    float rx_c1[3];         // -k log2(e) per modulation
    float rx_invA, rx_B;    // 1 / mcsA, mcsB
    double penalty;
    // mMTC
    int32_t mtc_n_dev, mtc_cap, mtc_n_rep, mtc_n_period;
    int32_t mtc_rep_set[8], mtc_period_set[8];
    double sla_mtc_delay, norm_mmtc[3];
};

// Persistent per-task / per-UE state (device pointers).
struct RsState {
    // per task [n_envs * n_embb]
    int32_t* t_n_ue;
    int32_t* t_cbr_at;      // absolute slot at which the next CBR arrival check fires
    int32_t* t_vbr_at;
    uint32_t* t_ctr;        // slice-level Philox draw counter
    uint32_t* t_serial;     // next UE serial
    int32_t* t_cost;        // PF loop trips of the task's previous step (scheduling hint only, never affects results)
    // per UE [task][RS_GROUP]
    double* u_queue;
    double* u_th;
    double* u_nominal;
    int32_t* u_hold_at;     // absolute slot at which the UE departs
    int32_t* u_e_snr;
    int32_t* u_findex;
    int32_t* u_bits;        // last slot's UE.bits (kept for Q2)
    int32_t* u_prbs;
    int32_t* u_vbr_at;      // absolute slot of the source's next burst arrival
    uint32_t* u_ctr;
    uint32_t* u_serial;
    int32_t* u_flags;       // bit0 type (0 CBR, 1 VBR), bits1-2 fading trace, bit3 step sign (+1 if set), bits4-6 RAN slice
                            // (multiplexed L1 only), bits8-15 VBR bursts that never end (Q5)
    uint16_t* u_burst;      // [task][RS_BURSTS][RS_GROUP] end slots as rs_burst_code, 0 = free
    // per replica
    uint64_t* seeds;
    int32_t* err;           // [n_envs] sticky error flags (RS_EOVERFLOW ...)
};

/* rs_oracle.c -- CPU restatement of the reference RanSlice.step path (see rs_oracle.h).
 * TEST INFRASTRUCTURE: not part of the product.  Build with -ffp-contract=off.
 *
 * Every function names the reference lines it follows.  Quirks Q1..Q12 of SURVEY.md §8a are
 * reproduced on purpose; they are marked where they occur.
 */
#include "rs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rs_philox.h"

#define RSO_MAX_UE 256
#define RSO_MAX_BURSTS 128

enum { CBR = 0, VBR = 1 };
/* info slots, eMBB (slice_ran.py:271-272) */
enum { I_CBR_TRAFFIC = 0, I_CBR_TH, I_CBR_PRB, I_CBR_QUEUE, I_CBR_SNR,
       I_VBR_TRAFFIC, I_VBR_TH, I_VBR_PRB, I_VBR_QUEUE, I_VBR_SNR };

typedef struct {
    int type;
    int ran;      /* RAN slice the UE belongs to (UE.slice_ran_id); matters when slices are multiplexed in one L1 */
    uint32_t serial;
    rs_stream st;
    double hold;  /* remaining_time[id] (slice_ran.py:222,240) */
    double queue; /* UE.queue */
    double th;    /* UE.th */
    int64_t e_snr;
    int ftype;    /* SINRSelectiveFading.users[id] (channel_models.py:169) */
    int64_t findex;
    int fstep;
    double nominal;
    double new_bits;
    int64_t bits;
    int64_t prbs;
    double p;
    /* VbrSource (traffic_generators.py:61-99) */
    double vbr_next;
    int n_bursts;
    double burst[RSO_MAX_BURSTS];
} rso_ue;

typedef struct {
    int n_prbs, prb_lo;
    int n_ue;
    rso_ue* ue;
    double cbr_next, vbr_next;
    int slot_counter;
    double info[10];
    rs_stream st;
    uint32_t next_serial;
} rso_embb;

typedef struct {
    int n_prbs;
    int64_t time;
    int n_dev;
    int64_t* period;
    int64_t* t_to_arrival;
    int64_t* dev_rep;
    int n_users, cap;
    int64_t* q_rep;
    int64_t* q_start;
    double info[3]; /* delay, avg_rep, devices */
    rs_stream st;
} rso_mmtc;

struct rs_oracle {
    rs_config cfg;
    int n_slices, n_vars, max_ue, max_bursts, max_queue;
    rso_embb* embb;
    rso_mmtc* mmtc;
    /* L1_level=False (scenario_creator.py:168-177): every eMBB RAN slice under ONE SliceL1eMBB (one UE list, one PF
     * scheduler, one PRB range), every mMTC RAN slice in ONE SliceL1mMTC (one FIFO) */
    int mux;
    rso_embb mux_embb; /* the UE list and the PRB range of the multiplexed L1 slice (its RAN-level fields are unused) */
    rso_mmtc mux_mmtc; /* time, n_prbs and the FIFO of the multiplexed mMTC L1 slice */
    int64_t* mq_ran;   /* RAN slice of every FIFO entry (SliceL1mMTC.slice_ran_ids) */
    /* fading tables, device layout [trace][time][prb] */
    double* fad[RS_N_TRACES];
    uint8_t* fad_valid[RS_N_TRACES];
    int fad_T[RS_N_TRACES];
    int fad_P; /* rows of the (extended) tables */
    /* randomness */
    int use_tape;
    const uint8_t* tape_kind;
    const double* tape_val;
    int64_t tape_n, tape_pos;
    uint64_t seed;
    int64_t slots_done; /* slots since reset (completed steps) */
    int64_t now;        /* absolute number of the slot being simulated (1-based) */
    double mcsA, mcsB;
    uint64_t counters[4];
    int err;
    char errmsg[256];
};

static void fail(rs_oracle* o, int code, const char* msg) {
    if (!o->err) {
        o->err = code;
        snprintf(o->errmsg, sizeof o->errmsg, "%s", msg);
    }
}

/* ------------------------------------------------------------------ randomness */

static double tape_pop(rs_oracle* o, int kind) {
    if (o->tape_pos >= o->tape_n) {
        fail(o, RS_ESTATE, "tape exhausted");
        return 0.5;
    }
    if (o->tape_kind[o->tape_pos] != kind) {
        char b[128];
        snprintf(b, sizeof b, "tape kind mismatch at %lld: want %d have %d", (long long)o->tape_pos, kind,
                 (int)o->tape_kind[o->tape_pos]);
        fail(o, RS_ESTATE, b);
    }
    return o->tape_val[o->tape_pos++];
}

static double dr_random(rs_oracle* o, rs_stream* s) {
    return o->use_tape ? tape_pop(o, RSO_K_RANDOM) : rs_stream_uniform(s);
}
static double dr_exponential(rs_oracle* o, rs_stream* s, double scale, int kind) {
    return o->use_tape ? tape_pop(o, kind) : rs_stream_exponential(s, scale);
}
static int64_t dr_integers(rs_oracle* o, rs_stream* s, int64_t n) {
    return o->use_tape ? (int64_t)tape_pop(o, RSO_K_INT) : rs_stream_integers(s, n);
}
static int dr_pm1(rs_oracle* o, rs_stream* s) {
    return o->use_tape ? (int)tape_pop(o, RSO_K_CHOICE) : rs_stream_pm1(s);
}
static int64_t dr_choice_set(rs_oracle* o, rs_stream* s, const int32_t* set, int k) {
    return o->use_tape ? (int64_t)tape_pop(o, RSO_K_CHOICE) : (int64_t)set[rs_stream_integers(s, k)];
}
static int64_t dr_choice_arange(rs_oracle* o, rs_stream* s, int64_t n) {
    return o->use_tape ? (int64_t)tape_pop(o, RSO_K_CHOICE) : rs_stream_integers(s, n);
}
static double dr_normal(rs_oracle* o, rs_stream* s, double loc, double scale) {
    return o->use_tape ? tape_pop(o, RSO_K_NORMAL) : rs_stream_normal(s, loc, scale);
}

/* ------------------------------------------------------------------ numpy pairwise sum */

/* numpy's float reduction order for a contiguous vector (np.mean / np.sum): blocks of <=128
 * elements summed with 8 strided accumulators, recursion by halves above that.  Verified
 * bit-exact against np.mean in this container (tests/test_oracle_golden.py::test_pairwise). */
double rso_pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return rso_pairwise_sum(a, n2) + rso_pairwise_sum(a + n2, n - n2);
    }
}

/* ------------------------------------------------------------------ MCS codeset */

/* MCSCodeset.compute_factors(0.1) (channel_models.py:272-279) */
void rso_mcs_factors(double* A, double* B) {
    double Delta = 0.1;
    double a = 1.0 / Delta;
    double s01 = rs_sigmoid(0.1, 0.0, 1.0), s09 = rs_sigmoid(0.9, 0.0, 1.0);
    a = a * (rs_log(1.0 / s01 - 1.0) - rs_log(1.0 / s09 - 1.0));
    *A = a;
    *B = -rs_log(1.0 / s09 - 1.0);
}

/* MCSCodeset.estimate_rx_prob (channel_models.py:281-286) */
static double estimate_rx_prob(const rs_config* c, double A, double B, int mcs, double snr) {
    double x = A * (snr - c->mcs_snr[mcs]) - B;
    return rs_sigmoid(x, 0.0, 1.0);
}

/* MCSCodeset.mcs_rate_vs_error(snr, 0.1) (channel_models.py:288-295) followed by
 * ue_rate[i] = sym_per_prb * bits_per_sym stored into an int array (schedulers.py:42-44).
 * Q1: returns the FAILING mcs's rate with index mcs-1. */
static void mcs_lookup(const rs_config* c, double A, double B, int64_t e_snr, int* mcs_out, int64_t* rate_out) {
    double rx_prob = 1.0 - 0.1;
    int mcs;
    for (mcs = 0; mcs < c->n_mcs; ++mcs) {
        if (estimate_rx_prob(c, A, B, mcs, (double)e_snr) < rx_prob) {
            *mcs_out = mcs - 1 > 0 ? mcs - 1 : 0;
            *rate_out = (int64_t)((double)c->sym_per_prb * (c->mcs_rate[mcs] * (double)c->mcs_order[mcs]));
            return;
        }
    }
    mcs = c->n_mcs - 1;
    *mcs_out = mcs;
    *rate_out = (int64_t)((double)c->sym_per_prb * (c->mcs_rate[mcs] * (double)c->mcs_order[mcs]));
}

void rso_mcs_lookup(const rs_config* cfg, int e_snr, int* mcs, int* rate) {
    double A, B;
    int64_t r;
    rso_mcs_factors(&A, &B);
    mcs_lookup(cfg, A, B, e_snr, mcs, &r);
    *rate = (int)r;
}

/* MCSCodeset.response (channel_models.py:297-313); snr already includes the nominal SINR */
static double response(const rs_config* c, double A, double B, int mcs, const double* snr, int64_t n) {
    double s;
    if (n > 1) {
        int mod = c->mcs_mod[mcs];
        double x0 = c->mi_x0[mod], k = c->mi_k[mod];
        double mi[1024];
        double* v = n <= 1024 ? mi : (double*)malloc(sizeof(double) * (size_t)n);
        for (int64_t i = 0; i < n; ++i) v[i] = rs_sigmoid(snr[i], x0, k);
        double avg = rso_pairwise_sum(v, n) / (double)n;
        if (v != mi) free(v);
        s = rs_inv_sigmoid(avg, x0, k);
    } else {
        s = snr[0];
    }
    return estimate_rx_prob(c, A, B, mcs, s);
}

double rso_response(const rs_config* cfg, int mcs, const double* snr, int n) {
    double A, B;
    rso_mcs_factors(&A, &B);
    return response(cfg, A, B, mcs, snr, n);
}

/* ------------------------------------------------------------------ propagation */

/* find_y_value (channel_models.py:44-48) */
static double find_y(double x1, double y1, double x2, double y2, double x) {
    double m = (y2 - y1) / (x2 - x1);
    double b = -m * x1 + y1;
    return m * x + b;
}

/* macro_cell (channel_models.py:84-97) given an in-cell point and the shadowing draw */
static double macro_cell_from(const rs_config* c, double x, double y, double LogF) {
    /* location (channel_models.py:62-68), radius = 1/2 */
    double x_t = x - 0.5 / 2;
    double distance = RS_SQRT(x_t * x_t + y * y);
    double cos_theta = x_t / distance;
    double theta = rs_acos(cos_theta);
    theta = theta * RS_RAD2DEG - 60;
    double R = distance * 2 > 0.1 ? distance * 2 : 0.1; /* Rmax = 2 km */
    /* antenna_pattern (channel_models.py:80-82) */
    double t65 = theta / 65;
    double att = 12 * (t65 * t65);
    double G = 15 + (-1 * (att < 20 ? att : 20));
    double lr = rs_log10(R);
    double L = c->prop_A + c->prop_B * lr;
    double gamma = 2.6;
    double FSPL = 20 * rs_log10(2.0) + 92.45 + gamma * 10 * lr;
    L = L > FSPL ? L : FSPL;
    double loss = L + LogF - G;
    double Rx_pw = 30 - (loss > 70 ? loss : 70); /* Tx_pw = 30 dBm, MCL = 70 dB */
    return Rx_pw - (-110) - 9;                    /* IN = -110 dBm, F = 9 dB */
}

static int in_cell(double x, double y) {
    /* generate_xy (channel_models.py:70-76) with lower_left/.../upper_right (:50-60) */
    return (y > find_y(0, 0.5, 0.25, 0, x)) && (y > find_y(0.75, 0, 1, 0.5, x)) &&
           (y < find_y(0, 0.5, 0.25, 1, x)) && (y < find_y(0.75, 1, 1, .5, x));
}

static double macro_cell(rs_oracle* o, rs_stream* s) {
    double x, y;
    int guard = 0;
    do {
        x = dr_random(o, s);
        y = dr_random(o, s);
        if (++guard > 100000) {
            fail(o, RS_ESTATE, "generate_xy did not terminate");
            break;
        }
    } while (!in_cell(x, y));
    double LogF = dr_normal(o, s, 0.0, 10.0);
    return macro_cell_from(&o->cfg, x, y, LogF);
}

double rso_macro_cell(const rs_config* cfg, const double* uv, int n_uv, double normal, int* used) {
    int i = 0;
    double x = 0.1, y = 0.1;
    while (i + 1 < n_uv) {
        x = uv[i];
        y = uv[i + 1];
        i += 2;
        if (in_cell(x, y)) break;
    }
    if (used) *used = i;
    return macro_cell_from(cfg, x, y, normal);
}

/* ------------------------------------------------------------------ fading walker */

int rso_load_fading(rs_oracle* o, int trace_id, const double* data, int rows, int cols) {
    if (trace_id < 0 || trace_id >= RS_N_TRACES || rows <= 0 || cols <= 0) return RS_EINVAL;
    int P = o->cfg.n_prbs > rows ? o->cfg.n_prbs : rows; /* table rows after extension */
    if (o->cfg.n_prbs > 2 * rows) return RS_EINVAL;
    free(o->fad[trace_id]);
    free(o->fad_valid[trace_id]);
    o->fad[trace_id] = (double*)malloc(sizeof(double) * (size_t)P * (size_t)cols);
    o->fad_valid[trace_id] = (uint8_t*)malloc((size_t)cols);
    o->fad_T[trace_id] = cols;
    o->fad_P = P;
    for (int t = 0; t < cols; ++t) {
        /* np.isnan(np.sum(column)) over ALL rows of the extended matrix (channel_models.py:188-189) */
        int bad = 0;
        for (int p = 0; p < P; ++p) {
            double v = data[(size_t)(p % rows) * cols + t]; /* row wrap (channel_models.py:144-148) */
            o->fad[trace_id][(size_t)t * P + p] = v;
            if (v != v) bad = 1;
        }
        o->fad_valid[trace_id][t] = (uint8_t)!bad;
    }
    return RS_OK;
}

static int fad_rows(const rs_oracle* o) { return o->fad_P; }

/* SINRSelectiveFading.insert_user (channel_models.py:163-169) */
static void fading_insert(rs_oracle* o, rso_ue* u) {
    u->ftype = (int)dr_integers(o, &u->st, RS_N_TRACES);
    u->findex = dr_integers(o, &u->st, o->fad_T[u->ftype]);
    u->fstep = dr_pm1(o, &u->st);
    u->nominal = macro_cell(o, &u->st);
}

/* SINRSelectiveFading.get_snr (channel_models.py:171-191): returns the column pointer */
static const double* fading_advance(rs_oracle* o, rso_ue* u) {
    int T = o->fad_T[u->ftype];
    int guard = 0;
    uint32_t attempt = 0;
    for (;;) {
        u->findex += u->fstep;
        if (u->findex >= T || u->findex < 0) {
            if (o->use_tape) {
                u->findex = dr_integers(o, &u->st, T);
                u->fstep = dr_pm1(o, &u->st);
            } else { /* build-defined streams: the redraw is addressed by (slot, attempt), include/rs_philox.h */
                int fi, fs;
                rs_walker_redraw(u->st.key0, u->st.key1, u->st.slice, u->st.serial, (uint32_t)o->now, attempt++, T, &fi,
                                 &fs);
                u->findex = fi;
                u->fstep = fs;
            }
        }
        if (o->fad_valid[u->ftype][u->findex]) break; /* Q10 */
        if (++guard > 4 * T + 16) {
            fail(o, RS_ESTATE, "fading trace has no NaN-free column");
            break;
        }
    }
    return o->fad[u->ftype] + (size_t)u->findex * fad_rows(o);
}

/* ------------------------------------------------------------------ traffic sources */

/* VbrSource.__init__ (traffic_generators.py:62-68) */
static void vbr_init(rs_oracle* o, rso_ue* u) {
    double inter = (1 / o->cfg.vbr_b_rate) / o->cfg.slot_length;
    u->vbr_next = rint(dr_exponential(o, &u->st, inter, RSO_K_GEXP));
    u->n_bursts = 0;
}

/* VbrSource.step (traffic_generators.py:70-99); active bursts are PeriodicSource(period=1)
 * which emit packet_size every slot (traffic_generators.py:24-30).  Q5: a counter that is 0
 * when first decremented goes negative and never fires. */
static double vbr_step(rs_oracle* o, rso_ue* u) {
    double bits = 0;
    int w = 0;
    for (int i = 0; i < u->n_bursts; ++i) {
        u->burst[i] -= 1;
        if (u->burst[i] == 0) continue; /* ending: dropped */
        bits += o->cfg.vbr_p_size;
        u->burst[w++] = u->burst[i];
    }
    u->n_bursts = w;
    u->vbr_next -= 1;
    if (u->vbr_next == 0) {
        double inter = (1 / o->cfg.vbr_b_rate) / o->cfg.slot_length;
        if (u->n_bursts >= o->max_bursts || u->n_bursts >= RSO_MAX_BURSTS) {
            fail(o, RS_EOVERFLOW, "VBR burst capacity exceeded");
        } else {
            u->burst[u->n_bursts++] = rint(dr_exponential(o, &u->st, o->cfg.vbr_b_size, RSO_K_GEXP));
        }
        u->vbr_next = rint(dr_exponential(o, &u->st, inter, RSO_K_GEXP));
    }
    return bits;
}

void rso_vbr_source(const rs_config* cfg, const double* gexp, int64_t n_gexp, int64_t n_slots, double* bits_out,
                    int64_t* used) {
    rs_oracle* o = rso_create(cfg);
    uint8_t* kinds = (uint8_t*)malloc((size_t)n_gexp);
    memset(kinds, RSO_K_GEXP, (size_t)n_gexp);
    rso_set_tape(o, kinds, gexp, n_gexp);
    rso_ue* u = (rso_ue*)calloc(1, sizeof(rso_ue));
    vbr_init(o, u);
    for (int64_t t = 0; t < n_slots; ++t) bits_out[t] = vbr_step(o, u);
    if (used) *used = o->tape_pos;
    free(u);
    free(kinds);
    rso_destroy(o);
}

/* ------------------------------------------------------------------ PF scheduler */

/* ProportionalFair.allocate (schedulers.py:21-76).  snr_col[i] points at the UE's fading column
 * (full PRB axis), the slice occupies [prb_lo, prb_lo+n_prb). */
static void pf_allocate(rs_oracle* o, const rs_config* c, double A, double B, int n_ues, rso_ue** ues,
                        const double** snr_col, int prb_lo, int n_prb) {
    int64_t ue_rbs[RSO_MAX_UE], ue_queue[RSO_MAX_UE], ue_rate[RSO_MAX_UE], ue_bits[RSO_MAX_UE];
    int ue_mcs[RSO_MAX_UE];
    double ue_th[RSO_MAX_UE];
    double b = 1.0 / c->pf_window, a = 1 - b;
    for (int i = 0; i < n_ues; ++i) {
        ue_rbs[i] = 0;
        ue_bits[i] = 0;
        ue_th[i] = ues[i]->th > 1 ? ues[i]->th : 1;
        ue_queue[i] = (int64_t)ues[i]->queue;
        mcs_lookup(c, A, B, ues[i]->e_snr, &ue_mcs[i], &ue_rate[i]);
    }
    for (int r = 0; r < n_prb; r += c->pf_granularity) {
        int64_t prbs = n_prb - r < c->pf_granularity ? n_prb - r : c->pf_granularity;
        /* np.argmax(ue_rate * (ue_queue > 0) / ue_th): first maximum; Q4: all-zero -> 0 */
        int index = 0;
        double best = -1.0;
        for (int i = 0; i < n_ues; ++i) {
            double m = (double)(ue_queue[i] > 0 ? ue_rate[i] : 0) / ue_th[i];
            if (m > best) {
                best = m;
                index = i;
            }
        }
        ue_rbs[index] += prbs;
        int64_t tx = prbs * ue_rate[index] < ue_queue[index] ? prbs * ue_rate[index] : ue_queue[index];
        ue_queue[index] -= tx;
        ue_bits[index] += tx;
        ue_th[index] = a * ue_th[index] + b * (double)ue_bits[index] / c->slot_length;
        if (o) o->counters[2] += 1;
    }
    int64_t prb_i = 0;
    for (int i = 0; i < n_ues; ++i) {
        rso_ue* u = ues[i];
        int64_t prbs = ue_rbs[i];
        u->prbs = prbs;
        u->bits = ue_bits[i];
        if (prbs) {
            double sn[1024];
            double* v = prbs <= 1024 ? sn : (double*)malloc(sizeof(double) * (size_t)prbs);
            for (int64_t k = 0; k < prbs; ++k) v[k] = snr_col[i][prb_lo + prb_i + k] + u->nominal;
            u->p = response(c, A, B, ue_mcs[i], v, prbs);
            if (v != sn) free(v);
        } else {
            u->p = 0;
        }
        prb_i += prbs;
    }
}

void rso_pf_allocate(const rs_config* cfg, int n_ue, int n_prb, const double* th, const double* queue,
                     const int* e_snr, const double* snr, int64_t* prbs, int64_t* bits, double* p) {
    double A, B;
    rso_mcs_factors(&A, &B);
    rso_ue* us = (rso_ue*)calloc((size_t)n_ue, sizeof(rso_ue));
    rso_ue** ptr = (rso_ue**)malloc(sizeof(rso_ue*) * (size_t)n_ue);
    const double** cols = (const double**)malloc(sizeof(double*) * (size_t)n_ue);
    for (int i = 0; i < n_ue; ++i) {
        us[i].th = th[i];
        us[i].queue = queue[i];
        us[i].e_snr = e_snr[i];
        us[i].nominal = 0.0;
        ptr[i] = &us[i];
        cols[i] = snr + (size_t)i * n_prb;
    }
    pf_allocate(NULL, cfg, A, B, n_ue, ptr, cols, 0, n_prb);
    for (int i = 0; i < n_ue; ++i) {
        prbs[i] = us[i].prbs;
        bits[i] = us[i].bits;
        p[i] = us[i].p;
    }
    free(cols);
    free(ptr);
    free(us);
}

/* ------------------------------------------------------------------ eMBB slice */

static void embb_reset_info(rso_embb* e) { /* slice_ran.py:270-273 */
    memset(e->info, 0, sizeof e->info);
    e->slot_counter = 0;
}

static void embb_reset(rs_oracle* o, rso_embb* e, int slice_id) { /* slice_ran.py:182-190, slice_l1.py:145-148 */
    e->n_ue = 0;
    e->cbr_next = 0;
    e->vbr_next = 0;
    e->next_serial = 1;
    e->st.key0 = (uint32_t)o->seed;
    e->st.key1 = (uint32_t)(o->seed >> 32);
    e->st.slice = (uint32_t)slice_id;
    e->st.serial = 0;
    e->st.ctr = 0;
    embb_reset_info(e);
}

static rso_ue* new_ue(rs_oracle* o, rso_embb* e, rso_ue* slot, int type) { /* UE.__init__ slice_ran.py:24-41 */
    memset(slot, 0, sizeof *slot);
    slot->type = type;
    slot->serial = e->next_serial++;
    slot->st = e->st;
    slot->st.serial = slot->serial;
    slot->st.ctr = 0;
    (void)o;
    return slot;
}

/* SliceRANeMBB.cbr_cac (slice_ran.py:195-203) */
static int cbr_cac(const rs_oracle* o, const rso_embb* e) {
    int slots = e->slot_counter > 1 ? e->slot_counter : 1;
    double time = slots * o->cfg.slot_length;
    double cbr_prb = e->info[I_CBR_PRB] / slots;
    double cbr_th = e->info[I_CBR_TH] / time;
    if (cbr_prb >= o->cfg.sla_embb[1] || cbr_th >= o->cfg.sla_embb[0]) return 0;
    return 1;
}

/* SliceRANeMBB.slot's arrival half (slice_ran.py:205-249,263-266) for RAN slice e: new UEs go to pend[] */
static int embb_arrivals(rs_oracle* o, rso_embb* e, int ran, rso_ue* pend) {
    const rs_config* c = &o->cfg;
    int n_pend = 0;
    e->slot_counter += 1;
    /* cbr_arrivals (slice_ran.py:205-227): inter-arrival is drawn BEFORE admission control */
    if (e->cbr_next == 0) {
        double ia = dr_exponential(o, &e->st, 1.0 / c->cbr_lambda, RSO_K_EXP);
        e->cbr_next = rint(ia / c->slot_length);
        if (cbr_cac(o, e)) {
            rso_ue* u = new_ue(o, e, &pend[n_pend++], CBR);
            double hold = dr_exponential(o, &u->st, c->cbr_t_mean, RSO_K_EXP);
            u->hold = rint(hold / c->slot_length);
        }
    } else {
        e->cbr_next -= 1;
    }
    /* vbr_arrivals (slice_ran.py:229-249): source ctor, holding time, THEN next inter-arrival */
    if (e->vbr_next == 0) {
        rso_ue* u = new_ue(o, e, &pend[n_pend++], VBR);
        vbr_init(o, u);
        double hold = dr_exponential(o, &u->st, c->vbr_t_mean, RSO_K_EXP);
        u->hold = rint(hold / c->slot_length);
        double ia = dr_exponential(o, &e->st, 1.0 / c->vbr_lambda, RSO_K_EXP);
        e->vbr_next = rint(ia / c->slot_length);
    } else {
        e->vbr_next -= 1;
    }
    for (int k = 0; k < n_pend; ++k) pend[k].ran = ran;
    return n_pend;
}

/* departures (slice_ran.py:251-261) + extract_users (slice_l1.py:187-191) of RAN slice `ran` on the L1 slice's UE
 * list, then add_users (slice_l1.py:183-185) -> insert_user draws, in arrival order.  Every timer of the RAN slice is
 * decremented, new arrivals included; ==0 departs.  Q5: a timer drawn as 0 never fires. */
static void embb_depart_and_insert(rs_oracle* o, rso_embb* L, int ran, rso_ue* pend, int n_pend) {
    int w = 0;
    for (int i = 0; i < L->n_ue; ++i) {
        if (L->ue[i].ran == ran) {
            L->ue[i].hold -= 1;
            if (L->ue[i].hold == 0) continue;
        }
        if (w != i) L->ue[w] = L->ue[i];
        ++w;
    }
    L->n_ue = w;
    for (int k = 0; k < n_pend; ++k) {
        pend[k].hold -= 1;
        if (pend[k].hold == 0) {
            /* Q13 (build-defined): a UE whose holding time rounds to exactly one slot departs in
             * its arrival slot.  The reference would raise KeyError in extract_user
             * (channel_models.py:193-194) because the user was never inserted; here the UE
             * simply never joins the slice. */
            continue;
        }
        if (L->n_ue >= o->max_ue || L->n_ue >= RSO_MAX_UE) {
            fail(o, RS_EOVERFLOW, "UE capacity exceeded");
            continue;
        }
        rso_ue* u = &L->ue[L->n_ue++];
        *u = pend[k];
        fading_insert(o, u);
    }
}

/* per-UE traffic and channel estimate, scheduling and transmission of one L1 slice (slice_l1.py:200-224) */
static int embb_l1_serve(rs_oracle* o, rso_embb* L) {
    const rs_config* c = &o->cfg;
    const double* col[RSO_MAX_UE];
    rso_ue* ptr[RSO_MAX_UE];
    double queued_data = 0;
    for (int i = 0; i < L->n_ue; ++i) {
        rso_ue* u = &L->ue[i];
        ptr[i] = u;
        /* UE.traffic_step (slice_ran.py:47-49); CbrSource = PeriodicSource(bit_rate*1e-3, period 1) */
        u->new_bits = u->type == CBR ? c->cbr_bit_rate * 1e-3 : vbr_step(o, u);
        u->queue += u->new_bits;
        queued_data += u->queue;
        col[i] = NULL;
        if (L->n_prbs > 0) { /* Q3: with 0 PRBs the walker does not advance and e_snr is stale */
            col[i] = fading_advance(o, u);
            /* UE.estimate_snr (slice_ran.py:43-45): round(np.mean(snr[prb_slice])), half-to-even (Q7) */
            double sn[1024];
            for (int k = 0; k < L->n_prbs; ++k) sn[k] = col[i][L->prb_lo + k] + u->nominal;
            double mean = rso_pairwise_sum(sn, L->n_prbs) / (double)L->n_prbs;
            u->e_snr = (int64_t)rint(mean);
            o->counters[0] += (uint64_t)L->n_prbs;
        }
        o->counters[3] += 1;
    }
    /* ---- scheduling and transmission (slice_l1.py:215-224); Q2: skipped -> stale bits/prbs */
    int scheduled = queued_data > 0 && L->n_prbs > 0;
    if (scheduled) {
        pf_allocate(o, c, o->mcsA, o->mcsB, L->n_ue, ptr, col, L->prb_lo, L->n_prbs);
        double b = 1.0 / c->pf_window, a = 1 - b; /* UE.__init__: b = 1/window, a = 1-b */
        for (int i = 0; i < L->n_ue; ++i) {
            rso_ue* u = &L->ue[i];
            int received = 0;
            if (u->prbs) received = dr_random(o, &u->st) < u->p;
            /* UE.transmission_step (slice_ran.py:51-55) */
            if (!received) u->bits = 0;
            double q = u->queue - (double)u->bits;
            u->queue = q > 0 ? q : 0;
            u->th = a * u->th + b * (double)u->bits / c->slot_length;
        }
    }
    return scheduled;
}

/* SliceRANeMBB.update_info (slice_ran.py:278-305) of RAN slice e over its UEs in the L1 slice's list */
static void embb_update_info(rso_embb* e, int ran, const rso_embb* L) {
    for (int cls = 0; cls < 2; ++cls) {
        double queue = 0, snr = 0;
        int n = 0;
        int base = cls == CBR ? I_CBR_TRAFFIC : I_VBR_TRAFFIC;
        for (int i = 0; i < L->n_ue; ++i) {
            const rso_ue* u = &L->ue[i];
            if (u->type != cls || u->ran != ran) continue;
            e->info[base + 0] += u->new_bits;
            e->info[base + 1] += (double)u->bits;
            e->info[base + 2] += (double)u->prbs;
            queue += u->queue;
            snr += (double)u->e_snr;
            n += 1;
        }
        n = n > 1 ? n : 1;
        e->info[base + 3] += queue / n;
        e->info[base + 4] += snr / n;
    }
}

static void embb_trace(const rs_oracle* o, const rso_embb* L, int scheduled, rs_alloc_rec* trace) {
    for (int i = 0; i < o->max_ue; ++i) {
        rs_alloc_rec* r = &trace[i];
        memset(r, 0, sizeof *r);
        if (i >= L->n_ue) continue;
        const rso_ue* u = &L->ue[i];
        r->serial = (int32_t)u->serial;
        r->type = u->type | (u->ran << 8); /* bits 8..: RAN slice (0 unless slices are multiplexed) */
        r->e_snr = (int32_t)u->e_snr;
        r->prbs = (int32_t)u->prbs;
        r->bits = u->bits;
        r->queue = u->queue;
        r->th = u->th;
        r->p = scheduled ? u->p : 0.0; /* this slot's allocation only */
    }
}

/* one slot of SliceL1eMBB (slice_l1.py:193-228) with its single SliceRANeMBB (L1_level=True) */
static void embb_slot(rs_oracle* o, rso_embb* e, rs_alloc_rec* trace) {
    rso_ue pend[2];
    const int n_pend = embb_arrivals(o, e, 0, pend);
    embb_depart_and_insert(o, e, 0, pend, n_pend);
    const int scheduled = embb_l1_serve(o, e);
    embb_update_info(e, 0, e);
    if (trace) embb_trace(o, e, scheduled, trace);
}

/* one slot of the multiplexed SliceL1eMBB (L1_level=False): slice_l1.py:193-228 with several slices_ran */
static void mux_embb_slot(rs_oracle* o, rs_alloc_rec* trace) {
    rso_embb* L = &o->mux_embb;
    for (int m = 0; m < o->cfg.n_embb; ++m) { /* arrivals, departures, extract, add -- RAN slice by RAN slice */
        rso_ue pend[2];
        const int n_pend = embb_arrivals(o, &o->embb[m], m, pend);
        embb_depart_and_insert(o, L, m, pend, n_pend);
    }
    const int scheduled = embb_l1_serve(o, L);
    for (int m = 0; m < o->cfg.n_embb; ++m) embb_update_info(&o->embb[m], m, L);
    if (trace) embb_trace(o, L, scheduled, trace);
}

/* SliceRANeMBB.compute_reward (slice_ran.py:307-319) */
static int embb_violation(const rs_oracle* o, const rso_embb* e) {
    const rs_config* c = &o->cfg;
    double observation_time = c->slots_per_step * c->slot_length;
    int cbr_th = e->info[I_CBR_TH] / observation_time > c->sla_embb[0];
    int cbr_prb = e->info[I_CBR_PRB] / c->slots_per_step > c->sla_embb[1];
    int cbr_queue = e->info[I_CBR_QUEUE] / c->slots_per_step < c->sla_embb[2];
    int vbr_th = e->info[I_VBR_TH] / observation_time > c->sla_embb[3];
    int vbr_prb = e->info[I_VBR_PRB] / c->slots_per_step > c->sla_embb[4];
    int vbr_queue = e->info[I_VBR_QUEUE] / c->slots_per_step < c->sla_embb[5];
    int cbr_ok = cbr_th || cbr_prb || cbr_queue;
    int vbr_ok = vbr_th || vbr_prb || vbr_queue;
    return !(cbr_ok && vbr_ok);
}

/* ------------------------------------------------------------------ mMTC slice */

static void mmtc_reset(rs_oracle* o, rso_mmtc* m, int slice_id) { /* slice_l1.py:29-38, slice_ran.py:91-101 */
    const rs_config* c = &o->cfg;
    m->time = 0;
    m->n_users = 0;
    memset(m->info, 0, sizeof m->info);
    m->st.key0 = (uint32_t)o->seed;
    m->st.key1 = (uint32_t)(o->seed >> 32);
    m->st.slice = (uint32_t)slice_id;
    m->st.serial = 0;
    m->st.ctr = 0;
    for (int i = 0; i < m->n_dev; ++i) {
        m->dev_rep[i] = dr_choice_set(o, &m->st, c->mtc_rep_set, c->mtc_n_rep);
        m->period[i] = dr_choice_set(o, &m->st, c->mtc_period_set, c->mtc_n_period);
        m->t_to_arrival[i] = 1 + dr_choice_arange(o, &m->st, m->period[i]);
    }
}

/* SliceL1mMTC.slot (slice_l1.py:87-125) with its single SliceRANmMTC (slice_ran.py:103-121) */
static void mmtc_slot(rs_oracle* o, rso_mmtc* m) {
    m->time += 1;
    for (int i = 0; i < m->n_dev; ++i) {
        m->t_to_arrival[i] -= 1;
        if (m->t_to_arrival[i] == 0) {
            if (m->n_users >= m->cap) {
                fail(o, RS_EOVERFLOW, "mMTC queue capacity exceeded");
            } else {
                m->q_rep[m->n_users] = m->dev_rep[i];
                m->q_start[m->n_users] = m->time;
                m->n_users += 1;
            }
            m->t_to_arrival[i] = m->period[i];
        }
    }
    int n_tx = m->n_prbs < m->n_users ? m->n_prbs : m->n_users; /* one NB-IoT carrier per PRB */
    for (int i = 0; i < n_tx; ++i) m->q_rep[i] -= 1;
    int w = 0;
    for (int i = 0; i < m->n_users; ++i) {
        if (m->q_rep[i] > 0) {
            m->q_rep[w] = m->q_rep[i];
            m->q_start[w] = m->q_start[i];
            ++w;
        }
    }
    m->n_users = w;
    double delay = 0, avg_rep = 0;
    if (w > 0) {
        /* integer sums are exact in f64, so numpy's reduction order is irrelevant here */
        int64_t sd = 0, sr = 0;
        for (int i = 0; i < w; ++i) {
            int64_t d = m->time - m->q_start[i];
            sd += d > 0 ? d : 0;
            sr += m->q_rep[i];
        }
        delay = (double)sd / (double)w;
        avg_rep = rint((double)sr / (double)w);
    }
    /* SliceRANmMTC.update_info (slice_ran.py:139-142) */
    m->info[0] += delay;
    m->info[1] += avg_rep;
    m->info[2] += (double)w;
}

/* one slot of the multiplexed SliceL1mMTC (L1_level=False, slice_l1.py:87-125 with several slices_ran): one FIFO;
 * arrivals are appended RAN slice by RAN slice (device order inside a slice), every entry remembers its slice, and
 * each slice's delay / repetitions / devices are taken over its own entries */
static void mux_mmtc_slot(rs_oracle* o) {
    rso_mmtc* Q = &o->mux_mmtc;
    Q->time += 1;
    for (int s = 0; s < o->cfg.n_mmtc; ++s) {
        rso_mmtc* m = &o->mmtc[s];
        for (int i = 0; i < m->n_dev; ++i) {
            m->t_to_arrival[i] -= 1;
            if (m->t_to_arrival[i] == 0) {
                if (Q->n_users >= Q->cap) {
                    fail(o, RS_EOVERFLOW, "mMTC queue capacity exceeded");
                } else {
                    Q->q_rep[Q->n_users] = m->dev_rep[i];
                    Q->q_start[Q->n_users] = Q->time;
                    o->mq_ran[Q->n_users] = s;
                    Q->n_users += 1;
                }
                m->t_to_arrival[i] = m->period[i];
            }
        }
    }
    int n_tx = Q->n_prbs < Q->n_users ? Q->n_prbs : Q->n_users; /* one NB-IoT carrier per PRB */
    for (int i = 0; i < n_tx; ++i) Q->q_rep[i] -= 1;
    int w = 0;
    for (int i = 0; i < Q->n_users; ++i) {
        if (Q->q_rep[i] > 0) {
            Q->q_rep[w] = Q->q_rep[i];
            Q->q_start[w] = Q->q_start[i];
            o->mq_ran[w] = o->mq_ran[i];
            ++w;
        }
    }
    Q->n_users = w;
    for (int s = 0; s < o->cfg.n_mmtc; ++s) {
        int64_t sd = 0, sr = 0, n = 0;
        for (int i = 0; i < w; ++i) {
            if (o->mq_ran[i] != s) continue;
            int64_t d = Q->time - Q->q_start[i];
            sd += d > 0 ? d : 0;
            sr += Q->q_rep[i];
            n += 1;
        }
        double delay = 0, avg_rep = 0;
        if (n > 0) {
            delay = (double)sd / (double)n;
            avg_rep = rint((double)sr / (double)n);
        }
        o->mmtc[s].info[0] += delay;
        o->mmtc[s].info[1] += avg_rep;
        o->mmtc[s].info[2] += (double)n;
    }
}

/* ------------------------------------------------------------------ env */

static void set_defaults(rs_oracle* o) {
    o->mux = o->cfg.l1_multiplex != 0;
    o->max_ue = o->cfg.max_ue > 0 ? o->cfg.max_ue : (o->mux ? 64 : 32);
    o->max_bursts = o->cfg.max_bursts > 0 ? o->cfg.max_bursts : 16;
    o->max_queue = o->cfg.max_mtc_queue > 0 ? o->cfg.max_mtc_queue : 1024;
}

rs_oracle* rso_create(const rs_config* cfg) {
    rs_oracle* o = (rs_oracle*)calloc(1, sizeof *o);
    o->cfg = *cfg;
    set_defaults(o);
    o->n_slices = cfg->n_embb + cfg->n_mmtc;
    o->n_vars = cfg->n_embb * RS_N_EMBB_VARS + cfg->n_mmtc * RS_N_MMTC_VARS;
    o->embb = (rso_embb*)calloc((size_t)(cfg->n_embb > 0 ? cfg->n_embb : 1), sizeof(rso_embb));
    for (int i = 0; i < cfg->n_embb; ++i) {
        o->embb[i].ue = (rso_ue*)calloc(RSO_MAX_UE, sizeof(rso_ue));
        o->embb[i].n_prbs = 20; /* scenario_creator.py:160 */
    }
    o->mmtc = (rso_mmtc*)calloc((size_t)(cfg->n_mmtc > 0 ? cfg->n_mmtc : 1), sizeof(rso_mmtc));
    for (int i = 0; i < cfg->n_mmtc; ++i) {
        rso_mmtc* m = &o->mmtc[i];
        m->n_dev = cfg->mtc_n_devices;
        m->n_prbs = 5; /* scenario_creator.py:165 */
        m->period = (int64_t*)calloc((size_t)m->n_dev, 8);
        m->t_to_arrival = (int64_t*)calloc((size_t)m->n_dev, 8);
        m->dev_rep = (int64_t*)calloc((size_t)m->n_dev, 8);
        m->cap = o->max_queue;
        m->q_rep = (int64_t*)calloc((size_t)m->cap, 8);
        m->q_start = (int64_t*)calloc((size_t)m->cap, 8);
    }
    if (o->mux) {
        o->mux_embb.ue = (rso_ue*)calloc(RSO_MAX_UE, sizeof(rso_ue));
        o->mux_embb.n_prbs = 20;
        o->mux_mmtc.n_prbs = 5;
        o->mux_mmtc.cap = o->max_queue * (cfg->n_mmtc > 0 ? cfg->n_mmtc : 1);
        o->mux_mmtc.q_rep = (int64_t*)calloc((size_t)o->mux_mmtc.cap, 8);
        o->mux_mmtc.q_start = (int64_t*)calloc((size_t)o->mux_mmtc.cap, 8);
        o->mq_ran = (int64_t*)calloc((size_t)o->mux_mmtc.cap, 8);
    }
    rso_mcs_factors(&o->mcsA, &o->mcsB);
    return o;
}

void rso_destroy(rs_oracle* o) {
    if (!o) return;
    for (int i = 0; i < o->cfg.n_embb; ++i) free(o->embb[i].ue);
    for (int i = 0; i < o->cfg.n_mmtc; ++i) {
        free(o->mmtc[i].period);
        free(o->mmtc[i].t_to_arrival);
        free(o->mmtc[i].dev_rep);
        free(o->mmtc[i].q_rep);
        free(o->mmtc[i].q_start);
    }
    free(o->embb);
    free(o->mmtc);
    free(o->mux_embb.ue);
    free(o->mux_mmtc.q_rep);
    free(o->mux_mmtc.q_start);
    free(o->mq_ran);
    for (int t = 0; t < RS_N_TRACES; ++t) {
        free(o->fad[t]);
        free(o->fad_valid[t]);
    }
    free(o);
}

void rso_set_tape(rs_oracle* o, const uint8_t* kind, const double* val, int64_t n) {
    o->use_tape = 1;
    o->tape_kind = kind;
    o->tape_val = val;
    o->tape_n = n;
    o->tape_pos = 0;
}
int64_t rso_tape_pos(const rs_oracle* o) { return o->tape_pos; }
void rso_set_seed(rs_oracle* o, uint64_t seed) {
    o->use_tape = 0;
    o->seed = seed;
}
const char* rso_error(const rs_oracle* o) { return o->errmsg; }
void rso_get_counters(const rs_oracle* o, uint64_t counters[4]) { memcpy(counters, o->counters, sizeof o->counters); }
int rso_get_mtc_queue(const rs_oracle* o, int s, int64_t* rep, int64_t* start, int cap, int64_t* time) {
    if (!o || s < 0 || s >= o->cfg.n_mmtc) return -1;
    const rso_mmtc* m = o->mux ? &o->mux_mmtc : &o->mmtc[s]; /* multiplexed: the one FIFO, whatever s */
    for (int i = 0; i < m->n_users && i < cap; ++i) {
        rep[i] = m->q_rep[i];
        start[i] = m->q_start[i];
    }
    if (time) *time = m->time;
    return m->n_users;
}

int rso_max_ue(const rs_oracle* o) { return o->max_ue; }

/* NodeB.reset (node_b.py:17-22) */
int rso_reset(rs_oracle* o) {
    o->err = 0;
    o->errmsg[0] = 0;
    memset(o->counters, 0, sizeof o->counters);
    o->slots_done = 0;
    o->now = 0;
    for (int i = 0; i < o->cfg.n_embb; ++i) embb_reset(o, &o->embb[i], i);
    for (int i = 0; i < o->cfg.n_mmtc; ++i) mmtc_reset(o, &o->mmtc[i], o->cfg.n_embb + i);
    o->mux_embb.n_ue = 0;
    o->mux_mmtc.time = 0;
    o->mux_mmtc.n_users = 0;
    return o->err;
}

/* RanSlice.step (ran_slice.py:38-54) over NodeB.step (node_b.py:59-91) */
int rso_step(rs_oracle* o, const int32_t* action, float* obs, double* reward, int32_t* labels,
             int32_t* violations, double* info, rs_alloc_rec* trace) {
    const rs_config* c = &o->cfg;
    if (o->err) return o->err;
    const int n_l1 = o->mux ? (c->n_embb > 0) + (c->n_mmtc > 0) : o->n_slices; /* L1 slices = action entries */
    int64_t total = 0;
    for (int s = 0; s < n_l1; ++s) {
        if (action[s] < 0) {
            fail(o, RS_EINVAL, "negative action");
            return o->err;
        }
        total += action[s];
    }
    if (total > c->n_prbs) { /* Q9: the reference silently mis-slices; the build rejects */
        fail(o, RS_EINVAL, "sum(action) > n_prbs");
        return o->err;
    }
    for (int t = 0; t < RS_N_TRACES; ++t)
        if (c->n_embb > 0 && !o->fad[t]) {
            fail(o, RS_ESTATE, "fading traces not loaded");
            return o->err;
        }
    /* reset_info + set_prbs (node_b.py:64-74) */
    int i_prb = 0;
    for (int s = 0; s < c->n_embb; ++s) embb_reset_info(&o->embb[s]);
    for (int s = 0; s < c->n_mmtc; ++s) memset(o->mmtc[s].info, 0, sizeof o->mmtc[s].info);
    if (o->mux) {
        int a = 0;
        if (c->n_embb > 0) {
            o->mux_embb.prb_lo = 0;
            o->mux_embb.n_prbs = action[a];
            i_prb += action[a++];
        }
        if (c->n_mmtc > 0) o->mux_mmtc.n_prbs = action[a];
    } else {
        for (int s = 0; s < c->n_embb; ++s) {
            o->embb[s].prb_lo = i_prb;
            o->embb[s].n_prbs = action[s];
            i_prb += action[s];
        }
        for (int s = 0; s < c->n_mmtc; ++s) o->mmtc[s].n_prbs = action[c->n_embb + s];
    }
    /* slots (node_b.py:77-78): slot-major, slices in order (they share one rng in the reference) */
    for (int t = 0; t < c->slots_per_step; ++t) {
        o->now = o->slots_done + t + 1;
        if (o->mux) { /* trace: [slots_per_step][max_ue] of the one eMBB L1 slice */
            if (c->n_embb > 0) mux_embb_slot(o, trace ? trace + (size_t)t * o->max_ue : NULL);
            if (c->n_mmtc > 0) mux_mmtc_slot(o);
            continue;
        }
        for (int s = 0; s < c->n_embb; ++s)
            embb_slot(o, &o->embb[s], trace ? trace + ((size_t)s * c->slots_per_step + t) * o->max_ue : NULL);
        for (int s = 0; s < c->n_mmtc; ++s) mmtc_slot(o, &o->mmtc[s]);
    }
    o->slots_done += c->slots_per_step;
    o->counters[1] += 1;
    /* get_state (node_b.py:40-44; slice_ran.py:321-325, 133-137): f64 ratio stored as f32 */
    int64_t tv = 0;
    int v = 0;
    int mux_viol[2] = {0, 0};
    for (int s = 0; s < c->n_embb; ++s) {
        rso_embb* e = &o->embb[s];
        for (int k = 0; k < RS_N_EMBB_VARS; ++k) {
            if (obs) obs[v] = (float)(e->info[k] / c->norm_embb[k]);
            ++v;
        }
        int viol = embb_violation(o, e);
        if (o->mux) { /* SliceL1eMBB.compute_reward sums its RAN slices' breaches (slice_l1.py:160-171) */
            mux_viol[0] += viol;
        } else {
            if (labels) labels[s] = viol == 0 ? 1 : -1; /* slice_l1.py:160-171 */
            if (violations) violations[s] = viol;
        }
        tv += viol;
        if (info) memcpy(info + (size_t)s * 10, e->info, sizeof e->info);
    }
    for (int s = 0; s < c->n_mmtc; ++s) {
        rso_mmtc* m = &o->mmtc[s];
        /* state order: devices, avg_rep, delay (scenario_creator.py:92) */
        if (obs) {
            obs[v + 0] = (float)(m->info[2] / c->norm_mmtc[0]);
            obs[v + 1] = (float)(m->info[1] / c->norm_mmtc[1]);
            obs[v + 2] = (float)(m->info[0] / c->norm_mmtc[2]);
        }
        v += 3;
        /* SliceRANmMTC.compute_reward (slice_ran.py:145-148) */
        int ok = m->info[0] / c->slots_per_step < c->sla_mtc_delay;
        int viol = !ok;
        int S = c->n_embb + s;
        if (o->mux) {
            mux_viol[1] += viol;
        } else {
            if (labels) labels[S] = viol == 0 ? 1 : -1;
            if (violations) violations[S] = viol;
        }
        tv += viol;
        if (info) {
            double* d = info + (size_t)S * 10;
            memset(d, 0, 10 * sizeof(double));
            d[0] = m->info[0];
            d[1] = m->info[1];
            d[2] = m->info[2];
        }
    }
    if (o->mux) {
        int a = 0;
        if (c->n_embb > 0) {
            if (labels) labels[a] = mux_viol[0] == 0 ? 1 : -1;
            if (violations) violations[a] = mux_viol[0];
            ++a;
        }
        if (c->n_mmtc > 0) {
            if (labels) labels[a] = mux_viol[1] == 0 ? 1 : -1;
            if (violations) violations[a] = mux_viol[1];
        }
    }
    /* reward (ran_slice.py:45-52) */
    if (reward) {
        if (tv > 0)
            *reward = (double)(-1 * c->penalty * (double)tv);
        else
            *reward = (double)(c->n_prbs - total > 0 ? c->n_prbs - total : 0);
    }
    return o->err;
}

/* bench action script: per PRB one categorical draw over S slices + "unused"
 * (SURVEY.md §8d config 2).  Philox counter = (prb, replica_lo, step_lo, step_hi^replica_hi). */
void rso_random_actions(const rs_config* cfg, uint64_t seed, uint64_t step_index, int64_t replica,
                        int32_t* action) {
    int S = cfg->n_embb + cfg->n_mmtc;
    for (int s = 0; s < S; ++s) action[s] = 0;
    for (int p = 0; p < cfg->n_prbs; ++p) {
        uint32_t a, b;
        rs_philox4x32_10((uint32_t)p, (uint32_t)replica, (uint32_t)step_index,
                         (uint32_t)(step_index >> 32) ^ (uint32_t)((uint64_t)replica >> 32) ^ 0x5bd1e995u,
                         (uint32_t)seed, (uint32_t)(seed >> 32), &a, &b);
        uint32_t bin = (uint32_t)(((uint64_t)a * (uint64_t)(S + 1)) >> 32);
        if ((int)bin < S) action[bin] += 1;
    }
}

/* cpu_baseline leg of bench.py: n_steps of the bench workload (random action script) for one
 * replica, entirely in C.  Returns the sum of rewards as a checksum. */
double rso_bench_run(rs_oracle* o, uint64_t action_seed, int64_t replica, uint64_t step0, int64_t n_steps) {
    int32_t action[64];
    float obs[1024];
    int32_t labels[64], viol[64];
    double reward = 0, acc = 0;
    for (int64_t i = 0; i < n_steps; ++i) {
        rso_random_actions(&o->cfg, action_seed, step0 + (uint64_t)i, replica, action);
        if (rso_step(o, action, obs, &reward, labels, viol, NULL, NULL) != 0) return NAN;
        acc += reward;
    }
    return acc;
}

double rso_exp(double x) { return rs_exp(x); }
double rso_log(double x) { return rs_log(x); }
double rso_acos(double x) { return rs_acos(x); }

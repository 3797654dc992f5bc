/* kb_oracle.c -- CPU restatement of the reference KBRL agent (see kb_oracle.h).
 * TEST INFRASTRUCTURE: not part of the product.  Build with -ffp-contract=off. */
#include "kb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rs_philox.h"

double rso_pairwise_sum(const double* a, int64_t n); /* rs_oracle.c: numpy's reduction order */

typedef struct {
    int d;    /* len(x) */
    int m;    /* SVvariable.counter == Projectron.counter */
    int cap;
    double* L;     /* landmarks [cap][d] */
    double* coeff; /* [cap] */
    double* Kinv;  /* [cap][cap], leading dimension cap */
    double f;      /* cached by predict (projectron.py:34) */
    double* K_f;
    int kf_n;
    double* dstar;
    rs_stream st;
} learner;

struct kb_oracle {
    int S, n_prbs, nv;
    double alfa, lo, hi, gamma, eta;
    learner* ln;
    int32_t* off; /* start of learner s's slice of the state */
    int32_t* action;
    int32_t* security;
    int32_t* margins;
    double* acc; /* [S][n_prbs] */
    const double* tape;
    int64_t tape_n, tape_pos;
    int use_tape;
    int64_t n_predict, n_mistakes;
    int err;
    int saturated_now; /* set by kbo_update when a full dictionary had to project a sample it would have added */
};

kb_oracle* kbo_create(int n_slices, const int32_t* dims, int n_prbs, double alfa, double acc_lo, double acc_hi,
                      const int32_t* initial_action, const int32_t* security_factor, double gamma, double eta,
                      int capacity) {
    kb_oracle* a = (kb_oracle*)calloc(1, sizeof *a);
    a->S = n_slices;
    a->n_prbs = n_prbs;
    a->alfa = alfa;
    a->lo = acc_lo;
    a->hi = acc_hi;
    a->gamma = gamma;
    a->eta = eta;
    a->ln = (learner*)calloc((size_t)n_slices, sizeof(learner));
    a->off = (int32_t*)calloc((size_t)n_slices, 4);
    a->action = (int32_t*)calloc((size_t)n_slices, 4);
    a->security = (int32_t*)calloc((size_t)n_slices, 4);
    a->margins = (int32_t*)calloc((size_t)n_slices, 4);
    a->acc = (double*)calloc((size_t)n_slices * n_prbs, 8);
    int o = 0;
    for (int s = 0; s < n_slices; ++s) {
        learner* l = &a->ln[s];
        l->d = dims[s] + 1;
        l->cap = capacity;
        l->L = (double*)calloc((size_t)capacity * l->d, 8);
        l->coeff = (double*)calloc((size_t)capacity, 8);
        l->Kinv = (double*)calloc((size_t)capacity * capacity, 8);
        l->K_f = (double*)calloc((size_t)capacity + 1, 8);
        l->dstar = (double*)calloc((size_t)capacity + 1, 8);
        a->off[s] = o;
        o += dims[s];
        a->action[s] = initial_action[s];
        a->security[s] = security_factor[s];
        /* kbrl_control.py:38-39 */
        for (int c = 0; c < n_prbs; ++c) a->acc[(size_t)s * n_prbs + c] = (acc_lo + acc_hi) / 2;
    }
    a->nv = o;
    return a;
}

void kbo_destroy(kb_oracle* a) {
    if (!a) return;
    for (int s = 0; s < a->S; ++s) {
        free(a->ln[s].L);
        free(a->ln[s].coeff);
        free(a->ln[s].Kinv);
        free(a->ln[s].K_f);
        free(a->ln[s].dstar);
    }
    free(a->ln);
    free(a->off);
    free(a->action);
    free(a->security);
    free(a->margins);
    free(a->acc);
    free(a);
}

void kbo_set_tape(kb_oracle* a, const double* val, int64_t n) {
    a->use_tape = 1;
    a->tape = val;
    a->tape_n = n;
    a->tape_pos = 0;
}
void kbo_set_seed(kb_oracle* a, uint64_t seed) {
    a->use_tape = 0;
    for (int s = 0; s < a->S; ++s) {
        rs_stream st = {(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)s, 0xFFFFFFFFu, 0u};
        a->ln[s].st = st;
    }
}
int64_t kbo_tape_pos(const kb_oracle* a) { return a->tape_pos; }
int kbo_error(const kb_oracle* a) { return a->err; }

/* GaussianKernel.k (kernel.py:13-20): row-wise squared distance summed in numpy's order */
static void kernel_row(const kb_oracle* a, learner* l, const double* x) {
    double dist[64];
    for (int j = 0; j < l->m; ++j) {
        const double* lj = l->L + (size_t)j * l->d;
        for (int q = 0; q < l->d; ++q) {
            double t = lj[q] - x[q];
            dist[q] = t * t;
        }
        double s = rso_pairwise_sum(dist, l->d);
        double k = rs_exp(-a->gamma * s);
        /* with a single 1-D landmark numpy builds np.array([k_eval], dtype=np.float32) (kernel.py:16) */
        l->K_f[j] = l->m == 1 ? (double)(float)k : k;
    }
    l->kf_n = l->m;
}

/* Projectron.predict (projectron.py:32-37) over GaussianKernel.predict (kernel.py:22-28) */
int kbo_predict(kb_oracle* a, int s, const double* x, double* f_out) {
    learner* l = &a->ln[s];
    a->n_predict += 1;
    int y;
    if (l->m > 0) {
        kernel_row(a, l, x);
        double f = 0.0;
        if (l->m == 1) {
            f = (double)(float)((float)l->K_f[0] * (float)l->coeff[0]); /* float32 dot */
        } else {
            for (int j = 0; j < l->m; ++j) f += l->K_f[j] * l->coeff[j];
        }
        l->f = f;
        y = f > 0 ? 1 : (f < 0 ? -1 : 0);
        if (y == 0) { /* np.random.choice([-1, 1]) (Q11) */
            if (a->use_tape) {
                if (a->tape_pos >= a->tape_n) {
                    a->err = 1;
                    y = 1;
                } else {
                    y = (int)a->tape[a->tape_pos++];
                }
            } else {
                y = rs_stream_pm1(&l->st);
            }
        }
    } else { /* empty dictionary (projectron.py:35-36) */
        y = 0;
        l->f = 0.0;
        l->K_f[0] = 0.0;
        l->kf_n = 1;
    }
    if (f_out) *f_out = l->f;
    return y;
}

/* Projectron.update (projectron.py:39-60); uses f / K_f cached by the preceding predict (Q12) */
int kbo_update(kb_oracle* a, int s, const double* x, int y, double* delta_out) {
    learner* l = &a->ln[s];
    if (!(l->f * y <= 0)) return 0;
    a->n_mistakes += 1;
    const int m = l->m, ld = l->cap;
    double Kii = 1.0; /* k_eval(x, x) = exp(-gamma * 0) */
    double dot = 0.0;
    if (m <= 1) {
        /* Kinv is the 1-element float32 array [0.0] (m == 0) or [1/Kii] (m == 1); K_f is float32 */
        float kinv = m == 0 ? 0.0f : 1.0f;
        float ds = kinv * (float)l->K_f[0];
        l->dstar[0] = (double)ds;
        dot = (double)(float)(ds * (float)l->K_f[0]);
    } else {
        for (int i = 0; i < m; ++i) {
            double acc = 0.0;
            for (int j = 0; j < m; ++j) acc += l->Kinv[(size_t)i * ld + j] * l->K_f[j];
            l->dstar[i] = acc;
        }
        for (int i = 0; i < m; ++i) dot += l->dstar[i] * l->K_f[i];
    }
    double delta = Kii - dot;
    delta = delta > 0 ? delta : 0;
    if (delta_out) *delta_out = delta;
    /* build-defined: a dictionary at its capacity projects every further sample onto its span instead of growing
     * (the reference's SVvariable grows without bound); err = 2 records that it happened */
    if (delta > a->eta && m >= l->cap) {
        a->err = 2;
        a->saturated_now = 1; /* tells update_control to stop augmenting this learner for this step */
    }
    if (delta <= a->eta || m >= l->cap) {
        /* SVvariable.update (projectron.py:13-14); the single-landmark coeff array is float32 */
        if (m == 1)
            l->coeff[0] = (double)(float)(l->coeff[0] + (double)y * l->dstar[0]);
        else
            for (int i = 0; i < m; ++i) l->coeff[i] += (double)y * l->dstar[i];
        return 1;
    }
    /* SVvariable.extend / insert (projectron.py:7-21) */
    l->coeff[m] = (double)y;
    memcpy(l->L + (size_t)m * l->d, x, sizeof(double) * (size_t)l->d);
    l->m = m + 1;
    if (l->m > 1) {
        /* Kinv <- [[Kinv, 0], [0, 0]] + outer([d*, -1], [d*, -1]) / delta (projectron.py:54-58) */
        for (int i = 0; i <= m; ++i) {
            l->Kinv[(size_t)i * ld + m] = 0.0;
            l->Kinv[(size_t)m * ld + i] = 0.0;
        }
        l->dstar[m] = -1.0;
        for (int i = 0; i <= m; ++i)
            for (int j = 0; j <= m; ++j) l->Kinv[(size_t)i * ld + j] += (l->dstar[i] * l->dstar[j]) / delta;
    } else {
        l->Kinv[0] = (double)(float)(1.0 / Kii);
    }
    return 2;
}

int kbo_set_size(const kb_oracle* a, int s) { /* projectron.py:62-64: landmarks.shape[0] */
    const learner* l = &a->ln[s];
    return l->m == 1 ? l->d : l->m;
}
int kbo_m(const kb_oracle* a, int s) { return a->ln[s].m; }
const double* kbo_coeff(const kb_oracle* a, int s) { return a->ln[s].coeff; }
const double* kbo_landmarks(const kb_oracle* a, int s) { return a->ln[s].L; }
const double* kbo_kinv(const kb_oracle* a, int s, int* ld) {
    if (ld) *ld = a->ln[s].cap;
    return a->ln[s].Kinv;
}
const int32_t* kbo_margins(const kb_oracle* a) { return a->margins; }
const int32_t* kbo_security_factors(const kb_oracle* a) { return a->security; }
const double* kbo_accuracies(const kb_oracle* a) { return a->acc; }
int64_t kbo_n_predict(const kb_oracle* a) { return a->n_predict; }
int64_t kbo_n_mistakes(const kb_oracle* a) { return a->n_mistakes; }

static void make_x(const kb_oracle* a, int s, const float* state, int prbs, double* x) {
    const learner* l = &a->ln[s];
    /* np.append(l1_state (float32), l1_prbs / n_prbs) -> float64 (kbrl_control.py:55) */
    for (int q = 0; q < l->d - 1; ++q) x[q] = (double)state[a->off[s] + q];
    x[l->d - 1] = (double)prbs / (double)a->n_prbs;
}

/* KBRL_Control.select_action (kbrl_control.py:41-73) */
int kbo_select_action(kb_oracle* a, const float* state, int32_t* action) {
    double x[64];
    int adjusted = 0;
    for (int i = 0; i < a->S; ++i) {
        int offset = a->security[i];
        int margin = 0;
        int l1_prbs = 0;
        int lo = 0 - offset > 0 ? 0 - offset : 0;
        for (l1_prbs = lo; l1_prbs <= a->n_prbs; ++l1_prbs) {
            make_x(a, i, state, l1_prbs, x);
            int prediction = kbo_predict(a, i, x, NULL);
            if (prediction == 1) {
                int aa = a->n_prbs < l1_prbs + offset ? a->n_prbs : l1_prbs + offset;
                margin = aa - l1_prbs;
                l1_prbs = aa;
                break;
            }
        }
        if (l1_prbs > a->n_prbs) l1_prbs = a->n_prbs; /* loop ran out: Python keeps the last value */
        action[i] = l1_prbs;
        a->margins[i] = margin;
    }
    int64_t assigned = 0;
    for (int i = 0; i < a->S; ++i) assigned += action[i];
    if (assigned > a->n_prbs) { /* adjust_action (kbrl_control.py:75-78) */
        adjusted = 1;
        for (int i = 0; i < a->S; ++i) {
            double p = (double)action[i] / (double)assigned;
            int na = (int)(int16_t)floor((double)a->n_prbs * p);
            int diff = action[i] - na;
            action[i] = na;
            a->margins[i] = (int)(int16_t)(a->margins[i] - diff);
        }
    }
    for (int i = 0; i < a->S; ++i) a->action[i] = action[i];
    return adjusted;
}

/* KBRL_Control.update_control (kbrl_control.py:80-114) */
void kbo_update_control(kb_oracle* a, const float* state, const int32_t* action, const int32_t* labels,
                        int adjusted, int32_t* hits) {
    double x[64];
    const int n = a->n_prbs;
    for (int i = 0; i < a->S; ++i) {
        int l1_action = action[i];
        make_x(a, i, state, l1_action, x);
        int y_pred = kbo_predict(a, i, x, NULL);
        int y = labels[i];
        int hit = y == y_pred;
        int margin = a->margins[i] > 0 ? a->margins[i] : 0;
        double* acc = a->acc + (size_t)i * n;
        if (y_pred == 1) {
            if (hit == 0) {
                for (int c = 0; c < margin + 1 && c < n; ++c) acc[c] = (1 - a->alfa) * acc[c];
            } else {
                for (int c = margin; c < n; ++c) acc[c] = (1 - a->alfa) * acc[c] + a->alfa;
            }
        }
        if (!adjusted) { /* np.argmax(acc > accuracy_range[0]) */
            int sf = 0;
            for (int c = 0; c < n; ++c)
                if (acc[c] > a->lo) {
                    sf = c;
                    break;
                }
            a->security[i] = sf;
        }
        hits[i] = hit;
        /* sample augmentation (kbrl_control.py:102-112) */
        int from = y == 1 ? l1_action : 0;
        int to = y == 1 ? n : l1_action;
        for (int c = from; c <= to; ++c) {
            make_x(a, i, state, c, x);
            (void)kbo_predict(a, i, x, NULL);
            a->saturated_now = 0;
            (void)kbo_update(a, i, x, y, NULL);
            /* build-defined: a FULL dictionary that met a sample it would have added cannot represent this region;
             * the remaining candidates of the range would meet the same wall one O(m^2) projection at a time, so
             * the augmentation of this learner stops for this step (the reference's dictionary is unbounded) */
            if (a->saturated_now) break;
        }
    }
}

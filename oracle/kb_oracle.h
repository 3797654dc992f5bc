/* kb_oracle.h -- CPU restatement of the reference's KBRL agent.  TEST INFRASTRUCTURE.
 *
 * Restates, in plain C and double precision (with the float32 roundings numpy applies while
 * the dictionary holds a single landmark),
 *   GaussianKernel.k / predict      (reference algorithms/kernel.py:8-28)
 *   SVvariable, Projectron          (reference algorithms/projectron.py:3-64)
 *   KBRL_Control.select_action / adjust_action / update_control (reference kbrl_control.py:41-114)
 * Never linked into or called from the product.  Parity status: PINNED against teacher-forced
 * golden sequences recorded from the reference (fixtures G9, G10; tests/test_kbrl_oracle_golden.py)
 * with a stated tolerance on f, delta, coeff, Kinv (numpy's BLAS summation order is not
 * reproducible) and exact agreement on every decision whose margin exceeds that tolerance.
 */
#ifndef KB_ORACLE_H
#define KB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kb_oracle kb_oracle;

/* dims[s]: number of state variables of learner s (its x has dims[s]+1 entries); the state slice
 * of learner s starts at sum(dims[:s]) (scenario_creator.py:224-235) */
kb_oracle* kbo_create(int n_slices, const int32_t* dims, int n_prbs, double alfa, double acc_lo, double acc_hi,
                      const int32_t* initial_action, const int32_t* security_factor, double gamma, double eta,
                      int capacity);
void kbo_destroy(kb_oracle* a);
/* tie-break draws of GaussianKernel.predict (kernel.py:26-27): tape of +-1, or a Philox seed */
void kbo_set_tape(kb_oracle* a, const double* val, int64_t n);
void kbo_set_seed(kb_oracle* a, uint64_t seed);
int64_t kbo_tape_pos(const kb_oracle* a);

/* Projectron.predict / update on learner s (x has dims[s]+1 doubles) */
int kbo_predict(kb_oracle* a, int s, const double* x, double* f_out);
/* returns 0 = no update (f*y > 0), 1 = coefficient projection, 2 = dictionary grew; delta_out valid if != 0 */
int kbo_update(kb_oracle* a, int s, const double* x, int y, double* delta_out);
int kbo_set_size(const kb_oracle* a, int s);       /* Projectron.get_set_size() quirk included */
int kbo_m(const kb_oracle* a, int s);
const double* kbo_coeff(const kb_oracle* a, int s);
const double* kbo_landmarks(const kb_oracle* a, int s);
const double* kbo_kinv(const kb_oracle* a, int s, int* ld);

/* KBRL_Control.select_action(state) -> action[n_slices] (int16 semantics), returns adjusted */
int kbo_select_action(kb_oracle* a, const float* state, int32_t* action);
/* KBRL_Control.update_control(state, action, labels) -> hits[n_slices]; `adjusted` is the flag
 * the run loop stored from the previous select_action (kbrl_control.py:133) */
void kbo_update_control(kb_oracle* a, const float* state, const int32_t* action, const int32_t* labels,
                        int adjusted, int32_t* hits);
const int32_t* kbo_margins(const kb_oracle* a);
const int32_t* kbo_security_factors(const kb_oracle* a);
const double* kbo_accuracies(const kb_oracle* a); /* [n_slices][n_prbs] */
int64_t kbo_n_predict(const kb_oracle* a);
int64_t kbo_n_mistakes(const kb_oracle* a);
int kbo_error(const kb_oracle* a);

#ifdef __cplusplus
}
#endif
#endif

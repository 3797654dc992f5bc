/* ranslice.h -- C ABI of libranslice.so, the MI355X-native batched RAN-slice simulator.
 *
 * The reference (jjalcaraz-upct/network-slicing) has no FFI: its boundary for this path is
 * the Python class surface gym_ran_slice.RanSlice.reset()/step() (reference
 * gym-ran_slice/gym_ran_slice/ran_slice.py:30-54) over NodeB.reset()/step()
 * (reference node_b.py:17-22, 59-91).  The entry points below are what a ctypes binding of
 * that surface needs; each cites the reference method it stands in for.  All pointers are
 * plain host pointers unless the name ends in _device; no torch types appear.
 *
 * Every function returns RS_OK (0) or a negative error code; rs_last_error() gives text.
 * A handle is bound to one HIP device and one stream and is not thread-safe.
 */
#ifndef RANSLICE_H
#define RANSLICE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_OK 0
#define RS_EINVAL (-1)    /* bad argument / action shape or sum (reference Q9: silently mis-slices) */
#define RS_EOVERFLOW (-2) /* a fixed capacity (UEs per slice, bursts per UE, mMTC queue) was exceeded */
#define RS_EHIP (-3)      /* HIP runtime error */
#define RS_ESTATE (-4)    /* call order error (e.g. step before reset / fading not loaded) */

#define RS_N_EMBB_VARS 10 /* reference scenario_creator.py:80-82 */
#define RS_N_MMTC_VARS 3  /* reference scenario_creator.py:92 */
#define RS_MAX_MCS 32
#define RS_MAX_SET 8
#define RS_N_TRACES 3 /* reference channel_models.py:29-33 */

/* Immutable description of one scenario.  Field values come from the reference's
 * module-level constants (scenario_creator.py:26-96,115-134; channel_models.py:21-27,
 * 268-270; schedulers.py:13; datasets/mcs_codeset.csv); the Python host fills it. */
typedef struct rs_config {
    int32_t n_envs;         /* independent env replicas simulated by this handle */
    int32_t n_prbs;         /* scenario_creator.py:26-50 */
    int32_t n_embb;         /* eMBB slices come first, then mMTC (scenario_creator.py:158-166) */
    int32_t n_mmtc;
    int32_t slots_per_step; /* scenario_creator.py:100 (50); 1 .. 63 (the per-step statistics of a task are packed) */
    int32_t max_ue;         /* capacity: UEs per eMBB slice (0 -> 32) */
    int32_t max_bursts;     /* capacity: VBR bursts running at once per UE (0 -> 16) */
    int32_t max_mtc_queue;  /* capacity: backlogged mMTC devices per slice (0 -> 1024) */
    double slot_length;     /* 1e-3 s */
    double penalty;         /* ran_slice.py:19 */
    /* eMBB traffic (scenario_creator.py:55-69) */
    double cbr_lambda, cbr_t_mean, cbr_bit_rate;
    double vbr_lambda, vbr_t_mean, vbr_p_size, vbr_b_size, vbr_b_rate;
    /* eMBB SLA (scenario_creator.py:71-78): cbr_th, cbr_prb, cbr_queue, vbr_th, vbr_prb, vbr_queue */
    double sla_embb[6];
    /* normalisation of the 10 eMBB state variables, in state order (scenario_creator.py:115-126) */
    double norm_embb[RS_N_EMBB_VARS];
    /* mMTC (scenario_creator.py:86-96,130-134) */
    int32_t mtc_n_devices;
    int32_t mtc_n_rep, mtc_n_period;
    int32_t mtc_rep_set[RS_MAX_SET];
    int32_t mtc_period_set[RS_MAX_SET];
    double sla_mtc_delay;
    double norm_mmtc[RS_N_MMTC_VARS]; /* devices, avg_rep, delay */
    /* propagation (channel_models.py:84-97,121-124): L = A + B log10(R) */
    double prop_A, prop_B;
    /* proportional-fair scheduler (schedulers.py:13) */
    int32_t pf_granularity, pf_window, sym_per_prb;
    /* MCS table (datasets/mcs_codeset.csv; channel_models.py:260-270) */
    int32_t n_mcs;
    double mcs_rate[RS_MAX_MCS];
    double mcs_snr[RS_MAX_MCS];
    int32_t mcs_order[RS_MAX_MCS];
    int32_t mcs_mod[RS_MAX_MCS]; /* 0 qpsk, 1 16qam, 2 64qam */
    double mi_x0[3], mi_k[3];    /* mutual-information sigmoid parameters per modulation */
    /* create_env(..., L1_level) (scenario_creator.py:156-177).  0 = L1_level=True: one L1 slice per RAN slice, the
     * action has n_embb + n_mmtc entries.  1 = L1_level=False: the eMBB RAN slices share ONE L1 slice (one UE list,
     * one PF scheduler, one PRB range) and the mMTC RAN slices ONE FIFO; the action has one entry per L1 slice
     * ((n_embb > 0) + (n_mmtc > 0)), labels/violations likewise (violations = RAN slices in breach), the observation
     * and rs_get_info keep one block per RAN slice; UE capacity is 64 per L1 slice. */
    int32_t l1_multiplex;
    int32_t reserved_;
} rs_config;

typedef struct rs_handle rs_handle;

/* per-slot, per-UE record for bit-exact allocation checks (debug / parity only) */
typedef struct rs_alloc_rec {
    int32_t serial; /* arrival serial of the UE inside its slice, >= 1; 0 = empty entry */
    int32_t type;   /* 0 CBR, 1 VBR */
    int32_t e_snr;  /* UE.e_snr after the slot (slice_ran.py:45) */
    int32_t prbs;   /* UE.prbs after the slot (schedulers.py:68) */
    int64_t bits;   /* UE.bits after transmission_step (slice_ran.py:51-55) */
    double queue;   /* UE.queue after the slot */
    double th;      /* UE.th after the slot */
    double p;       /* UE.p, reception probability of this slot's allocation (0 if none) */
} rs_alloc_rec;

/* Build the simulator for cfg->n_envs replicas on HIP device `device`.
 * Stands in for scenario_creator.create_env (scenario_creator.py:100-183) minus file I/O. */
int rs_create(const rs_config* cfg, int device, rs_handle** out);

/* Upload fading trace `trace_id` (0..2), given in the reference's CSV layout: row-major
 * [rows = PRB][cols = time], dB, NaN allowed.  Rows are wrapped up to n_prbs as in
 * SINRSelectiveFading.__init__ (channel_models.py:141-150).  Data is copied. */
int rs_load_fading(rs_handle* h, int trace_id, const double* data, int rows, int cols);

/* RanSlice.reset()/NodeB.reset() (ran_slice.py:30-36, node_b.py:17-22) for every replica.
 * seeds[n_envs]: one 64-bit stream seed per replica (Evaluator.evaluate's default_rng(seed=i),
 * experiments_kbrl.py:46).  obs (may be NULL) receives zeros [n_envs][n_vars]. */
int rs_reset(rs_handle* h, const uint64_t* seeds, float* obs);

/* RanSlice.step(action) for every replica (ran_slice.py:38-54, node_b.py:59-91).
 * actions [n_envs][n_slices] PRBs per slice; must be >= 0 with row sums <= n_prbs.
 * Outputs (any may be NULL): obs f32 [n_envs][n_vars]; reward f64 [n_envs];
 * labels i32 [n_envs][n_slices] (+1/-1, node_b.py:51-57); violations i32 [n_envs][n_slices]. */
int rs_step(rs_handle* h, const int32_t* actions, float* obs, double* reward, int32_t* labels,
            int32_t* violations);

/* Same step, but actions are already resident in the handle's device action buffer
 * (filled by rs_random_actions) and nothing is copied back: the timed path of bench.py. */
int rs_step_resident(rs_handle* h);

/* Fill the device action buffer with i.i.d. multinomial(n_prbs; 1/(S+1) per slice and one
 * "unused" bin) draws per replica, from Philox stream (seed, step_index).  Used by bench.py
 *  and reproduced by the oracle for the CPU baseline. */
int rs_random_actions(rs_handle* h, uint64_t seed, uint64_t step_index);

/* Copy the results of the last step (resident or not) to host buffers (any may be NULL). */
int rs_fetch(rs_handle* h, int32_t* actions, float* obs, double* reward, int32_t* labels,
             int32_t* violations);

/* info['l1_info'] accumulators of the last step (node_b.py:46-49): f64 [n_envs][n_slices][10]
 * (eMBB: cbr_traffic, cbr_th, cbr_prb, cbr_queue, cbr_snr, vbr_...; mMTC: delay, avg_rep,
 * devices, rest 0). */
int rs_get_info(rs_handle* h, double* info);

/* n_steps x { rs_random_actions(seed, step_index0 + i); rs_step_resident(); } enqueued by one call.  With
 * use_graph != 0 the loop body is captured once as a hipGraph (two consecutive steps) and replayed; the slot
 * clock and the script index live in device memory, so results are identical with and without the graph. */
int rs_run_random(rs_handle* h, uint64_t seed, uint64_t step_index0, int n_steps, int use_graph);

/* Enable (capacity > 0) or disable per-slot allocation tracing.  When enabled every step
 * records [n_envs][n_embb][slots_per_step][max_ue] rs_alloc_rec entries. */
int rs_set_alloc_trace(rs_handle* h, int enable);
int rs_get_alloc_trace(rs_handle* h, rs_alloc_rec* out);

/* Counters maintained by the step kernels since the last rs_reset: [0] = sum over replicas,
 * slots, eMBB slices of n_ue * n_prbs (fading samples read; SURVEY.md §8d algorithmic bytes),
 * [1] = env-steps executed, [2] = PF loop iterations, [3] = UE-slots. */
int rs_get_counters(rs_handle* h, uint64_t counters[4]);

/* The reception step (slice_l1.py:219-224: `rng.random() < mcs_codeset.response(mcs, snr)`, channel_models.py:297-313) is
 * decided without forming the probability whenever a float32 evaluation of both sides leaves no doubt (rs_embb.hip:
 * fast_sigmoid; the outcome is the exact comparison's in every case).  Since the last rs_reset, per-slice step kernels
 * without allocation tracing: out[0] = reception tests, out[1] = those that evaluated the exact probability, out[2] = 1 if
 * the short test is available for this configuration (mcsA > 0, every MI slope k > 0, 1 <= mcsA / k <= 1e4).  The two counts
 * share one 64-bit word per task (32 bits each: they wrap after ~2e7 steps of one task). */
int rs_get_rx_stats(rs_handle* h, uint64_t out[3]);

/* Average device time of the dominant step kernel over the launches since the last call,
 * measured with HIP events on the handle's stream (bench.py roofline leg). */
int rs_kernel_time_ms(rs_handle* h, double* avg_ms, int64_t* launches);
/* The same with the spread: out = {mean, min, max} ms over those launches (a 20-step driver run rests on 20 of them). */
int rs_kernel_time_stats_ms(rs_handle* h, double out[3], int64_t* launches);
int rs_set_kernel_timing(rs_handle* h, int enable);

/* Lanes per (replica, eMBB slice) task in the primary step launch: 8, 16 or 32 (default: 32 up to 6144 tasks,
 * 16 above).  Tasks that need
 * more UE lanes are replayed by the 32-lane instance; results are identical for every setting. */
int rs_set_group_size(rs_handle* h, int lanes);

/* Scheduling hint.  mode 1: allocations come from a learning agent (the carrier is concentrated on few, wide slices);
 * the 16-lane step uses its BLOCK instance, which hands out the RB pairs of wide contested slices in block rounds
 * (every backlogged UE steps ahead in parallel) instead of one leader run at a time.  mode 0: the plain instance
 * (trip loop only; faster on the 30-50-RB slices of an even split).  mode < 0 (default): automatic -- rs_step looks at
 * the allocations it is handed, kb_step_resident asks for the BLOCK instance for the environment it drives, the
 * on-device random script goes by its average slice width.  Results do not depend on it (both are exact against the
 * reference loop, tests/test_gpu_parity.py::test_block_round_allocations). */
int rs_set_schedule_hint(rs_handle* h, int mode);

/* Developer aid: cycle sums per code section of the eMBB step kernel (zeros in normal builds). */
int rs_get_section_profile(rs_handle* h, uint64_t out[16]);
/* Developer aid (profile builds): per eMBB task [n_envs * n_embb][4] = cycles of the task's wave in the last step,
 * UEs and RBs at its start, contested PF loop trips; then 16 more values: the section cycle sums of the slowest
 * wave seen so far (out must hold 4 * n_tasks + 16 values). */
int rs_get_task_profile(rs_handle* h, uint64_t* out);

int rs_synchronize(rs_handle* h);
/* HIP devices visible to this process, or RS_EHIP */
int rs_device_count(void);
/* Free and total bytes of a device's memory (hipMemGetInfo), e.g. to size kb_config.pool_bytes. */
int rs_device_mem_info(int device, uint64_t* free_bytes, uint64_t* total_bytes);
int rs_n_vars(const rs_handle* h);
/* Checkpoint / restore (no reference counterpart: a run of experiments_kbrl.py that dies starts over).  The state of a handle
 * -- every device array behind it except the tables (fading traces, constants), plus its slot clock -- as one blob of
 * rs_state_bytes bytes; rs_load_state accepts a blob saved by a handle of the same configuration, on any device.  Loading
 * and stepping on reproduces the steps the saving handle would have made, bit for bit. */
int rs_state_bytes(rs_handle* h, uint64_t* bytes);
int rs_save_state(rs_handle* h, void* blob, uint64_t bytes);
int rs_load_state(rs_handle* h, const void* blob, uint64_t bytes);
int rs_n_slices(const rs_handle* h);
const char* rs_last_error(const rs_handle* h);
void rs_destroy(rs_handle* h);


/* ------------------------------------------------------------------------------------------
 * KBRL agent (hot path B).  Stands in for kbrl_control.KBRL_Control over
 * algorithms.projectron.Projectron(GaussianKernel(SVvariable)) (reference kbrl_control.py:23-114,
 * algorithms/projectron.py:23-64, algorithms/kernel.py:3-34), one independent agent per replica.
 * ------------------------------------------------------------------------------------------ */
#define KB_MAX_SLICES 8
#define KB_CAPACITY_MAX 65536

typedef struct kb_config {
    int32_t n_envs;               /* agents (one per env replica) */
    int32_t n_slices;             /* learners per agent */
    int32_t n_prbs;               /* <= 255: the candidates 0 .. n_prbs of a learner fill four 64-lane groups (rs_create takes carriers
                                     of up to 256 PRBs; an agent for a 256-PRB carrier is refused by kb_create with RS_EINVAL) */
    int32_t capacity;             /* most landmarks a dictionary may hold (<= KB_CAPACITY_MAX); storage is taken from the
                                     pool 64 landmarks at a time as dictionaries grow, so this is a limit, not a reservation */
    int32_t dims[KB_MAX_SLICES];  /* state variables of learner s: 10 eMBB / 3 mMTC (scenario_creator.py:209-235) */
    double alfa;                  /* scenario_creator.py:187 */
    double acc_lo, acc_hi;        /* accuracy_range */
    double gamma, eta;            /* scenario_creator.py:218, projectron.py:25 */
    int32_t shared_dictionary;    /* 0: one agent per replica (the reference); 1: one dictionary per slice shared by all
                                     replicas (build-defined extension) */
    int32_t first_env;            /* shared mode: global id of this handle's replica 0 (rank * n_envs) */
    int64_t pool_bytes;           /* device memory all dictionaries of the handle grow in (landmarks, coefficients, Kinv:
                                     SVvariable / Projectron.Kinv, projectron.py:3-30, which the reference grows without
                                     bound).  0: the smallest of every dictionary at its capacity, 1 GB + 2 MB per
                                     dictionary, and half of the free device memory */
} kb_config;

typedef struct kb_handle kb_handle;

int kb_create(const kb_config* cfg, int device, kb_handle** out);
void kb_destroy(kb_handle* k);
const char* kb_last_error(const kb_handle* k);
/* KBRL_Control.__init__ state (kbrl_control.py:28-39): initial_action / security_factor [n_envs][S];
 * seeds[n_envs] feed the tie-break stream of GaussianKernel.predict (kernel.py:26-27). */
int kb_reset(kb_handle* k, const int32_t* initial_action, const int32_t* security_factor, const uint64_t* seeds);
/* KBRL_Control.update_control(state, action, labels) -> hits (kbrl_control.py:80-114), all agents */
int kb_update_control(kb_handle* k, const float* state, const int32_t* action, const int32_t* labels, int32_t* hits);
/* KBRL_Control.select_action(state) -> (action, adjusted) (kbrl_control.py:41-78), all agents */
int kb_select_action(kb_handle* k, const float* state, int32_t* action, int32_t* adjusted);
/* One closed-loop agent step entirely on the device (KBRL_Control.run body, kbrl_control.py:129-134):
 * update_control(previous obs, the action just executed by `env`, its SLA labels) followed by
 * select_action(new obs); the selected action is written into env's device action buffer. */
int kb_step_resident(kb_handle* k, rs_handle* env);
/* n_steps of that loop body followed each by the simulator's step -- n x (kb_step_resident(k, env); rs_step_resident(env)) --
 * enqueued by one call; with use_graph two consecutive steps are captured once into a hipGraph and replayed (one graph launch
 * per two steps instead of ~50 kernel launches).  Identical results either way. */
int kb_run_resident(kb_handle* k, rs_handle* env, int n_steps, int use_graph);
/* Projectron.predict(x) / update(x, y) on learner `s` of agent `e` (projectron.py:32-60).
 * branch: 0 none, 1 projection, 2 dictionary grew. */
int kb_predict(kb_handle* k, int e, int s, const double* x, int32_t* y_pred, double* f);
int kb_update(kb_handle* k, int e, int s, const double* x, int32_t y, int32_t* branch, double* delta);
/* GaussianKernel.k(x) (kernel.py:13-20) of the last kb_predict on learner (e, s): its kernel row, m entries (the third
 * return value of GaussianKernel.predict, kernel.py:28).  row may be NULL to ask for m only. */
int kb_get_kernel_row(kb_handle* k, int e, int s, int32_t* m, double* row);
/* Dictionary of learner (e, s): m landmarks [m][dims+1], coeff [m], Kinv [m][m] (any may be NULL) */
int kb_get_learner(kb_handle* k, int e, int s, int32_t* m, double* landmarks, double* coeff, double* kinv);
/* margins / security_factors / current action [n_envs][S], adjusted [n_envs], accuracies [n_envs][S][n_prbs] */
int kb_get_control(kb_handle* k, int32_t* margins, int32_t* security, int32_t* action, int32_t* adjusted,
                   double* accuracies);
int kb_set_adjusted(kb_handle* k, const int32_t* adjusted);
/* ---- shared-dictionary mode (kb_config.shared_dictionary = 1).  One learning step = round 0, 1, ...:
 *   kb_shared_scan   every replica looks for its first mistake (augmentation order, kbrl_control.py:103-112)
 *                    against the frozen shared dictionaries; round 0 also does update_control's bookkeeping
 *                    (hits, accuracies, security factors) and needs state/action/labels (later rounds: NULL).
 *                    counts[S] = local proposers per slice; props[S][budget][KB_PROP_WIDTH] = the first
 *                    `budget` of them in replica order (global replica id, packed candidate/label, state).
 *   (caller)         all-gather props/counts over RCCL, merge by global replica id, keep the first `budget`.
 *   kb_shared_apply  apply a merged list (same on every rank) in order through Projectron.predict/update.
 *   kb_shared_commit n_accept[S] = how many of this handle's proposers (in replica order) were in the merged
 *                    list: they move on to their next candidate; the others propose again next round. */
#define KB_PROP_WIDTH 18
int kb_shared_scan(kb_handle* k, const float* state, const int32_t* action, const int32_t* labels, int32_t round,
                   int32_t budget, int32_t* hits, int32_t* counts, double* props);
int kb_shared_apply(kb_handle* k, const int32_t* counts, const double* props, int32_t budget);
int kb_shared_commit(kb_handle* k, const int32_t* n_accept);

/* The same learning step with the exchange kept on the device: every round scans, packs this rank's proposals,
 * all-gathers the blocks of all ranks with ncclAllGather (RCCL, bound at run time, on the agent's stream), merges by
 * global replica id, applies and commits -- until no rank proposes anything or max_rounds rounds were made
 * (rounds_out).  Only a 4-byte "anything left" flag per round returns to the host.  kb_comm_unique_id: 128 bytes
 * from ncclGetUniqueId, generated by ONE rank and handed to the others by the launcher; kb_comm_init joins the
 * communicator (rank = the handle's index among the `world` agents that share their dictionaries; kb_config.first_env
 * must be rank * n_envs).  Without kb_comm_init the handle is its own world (no RCCL needed). */
int kb_comm_unique_id(void* id128);
int kb_comm_init(kb_handle* k, const void* id128, int rank, int world);
/* What the live communicator itself reports (ncclCommUserRank / ncclCommCount); (0, 1) for a handle that never joined one.
 * After the communicator was aborted (a failed or timed-out exchange: kb_shared_step returned RS_EHIP) this and every
 * shared step return RS_ESTATE until kb_comm_init joins a new one -- the handle does not quietly become a world of its own. */
int kb_comm_info(kb_handle* k, int* rank, int* world);
int kb_shared_step(kb_handle* k, const float* state, const int32_t* action, const int32_t* labels, int32_t budget,
                   int32_t max_rounds, int32_t* hits, int32_t* rounds_out);
/* kb_step_resident for a shared-dictionary agent: the learning step above on the simulator's own device buffers (previous
 * observation, the action `env` just executed, its labels), then select_action of the new observation into the
 * simulator's action buffer.  Only the per-round "anything left" flag crosses PCIe; with rounds_out == NULL the host
 * does not wait for the last permitted round's flag either (the whole step is enqueued and the call returns). */
int kb_shared_step_resident(kb_handle* k, rs_handle* env, int32_t budget, int32_t max_rounds, int32_t* rounds_out);
/* The merge of kb_shared_step on a caller-supplied gathered buffer (what ncclAllGather delivers): `world` blocks of
 * [S proposer counts][S][budget][KB_PROP_WIDTH] doubles -> merged proposals [S][budget][KB_PROP_WIDTH], their counts [S],
 * how many of rank `me`'s made it [S], and the proposers of all ranks.  Lets a single process check the device merge
 * for any world size against the host rule (ranslice.kbrl_dev.merge_proposals). */
int kb_shared_merge(kb_handle* k, const double* gathered, int32_t world, int32_t me, int32_t budget, double* merged,
                    int32_t* counts, int32_t* taken, int32_t* total);

/* Histories of KBRL_Control.run (kbrl_control.py:119-124,135-141) kept on the device: after kb_history_begin(steps)
 * every kb_step_resident records one column per replica -- reward f64, resources (sum of the newly selected action),
 * hits [S], adjusted, SLA (sum of labels), violation (total), all int16 as in the reference -- so a whole run needs no
 * per-step read-back.  kb_history_fetch: [n_envs][steps] arrays (hits [n_envs][S][steps]) and the columns recorded. */
int kb_history_begin(kb_handle* k, int32_t steps);
int kb_history_fetch(kb_handle* k, double* reward, int16_t* resources, int16_t* hits, int16_t* adjusted, int16_t* sla,
                     int16_t* violation, int32_t* n_recorded);

/* sums over learners since kb_reset: [0] predicts, [1] mistakes, [2] insertions, [3] kernel evaluations */
int kb_get_stats(kb_handle* k, uint64_t stats[4]);
/* landmarks in every dictionary: i32 [n_envs][S] (one agent per replica) or [S] (shared dictionaries) */
int kb_get_sizes(kb_handle* k, int32_t* m);
/* the dictionary pool: bytes in use / in total, replicas with a dictionary at its capacity, replicas that found the pool
 * exhausted (both keep learning by projection -- build-defined, the reference's SVvariable is unbounded; any may be NULL) */
int kb_get_pool(kb_handle* k, uint64_t* used_bytes, uint64_t* total_bytes, int32_t* n_saturated, int32_t* n_pool_full);
/* per-replica flag words [n_envs] behind kb_get_pool's counts: bit 8 a dictionary of the replica reached its capacity, bit 16
 * it found the pool exhausted */
int kb_get_flags(kb_handle* k, int32_t* flags);
/* bytes behind the repair rounds of the large dictionaries since kb_reset (projectron.py:42 Kinv @ K_f, :54-58 the rank-1
 * update), as their kernels count them: work[0] tiles of Kinv the mat-vec kernel read (32,768 bytes each), work[1] units of
 * the rank-1 kernel (8,192 bytes read + 8,192 written each), work[2] / work[3] launches of either that had work; work[4] scoring passes that
 * evaluated landmarks' exponentials one by one (outlier states, off-grid landmarks), work[5] the landmarks they evaluated --
 * both counted by developer builds only (-DKB_COUNT_DIRECT), 0 otherwise */
int kb_get_repair_work(kb_handle* k, uint64_t work[8]);
int kb_kernel_time_ms(kb_handle* k, double* avg_ms, int64_t* launches);
/* the same per phase: ms[0] / n[0] the update phase (update_control_kernel and its repair kernels; shared mode: the scan
 * kernels), ms[1] / n[1] select_kernel */
int kb_phase_times_ms(kb_handle* k, double ms[2], int64_t n[2]);
int kb_set_kernel_timing(kb_handle* k, int enable);
/* mean duration of one launch of the two kernels that stream Kinv in the repair rounds -- ms[0] / n[0] the mat-vec, ms[1] /
 * n[1] the rank-1 update -- over the span the last kb_phase_times_ms call covered (for their HBM roofline) */
int kb_repair_times_ms(kb_handle* k, double ms[2], int64_t n[2]);
/* the same for every kernel that has a roofline entry of its own in the bench record: mean duration of ONE launch over that
 * span -- [2] heavy_matvec_kernel, [3] heavy_rank1_kernel, [4] the binning pass of select_action (select_bin_kernel alone, or select_bin_big_kernel + select_bin_kernel + big_list_kernel once large learners are listed), [5] heavy_finish_kernel, [6] select_gemm_kernel,
 * [7] update_small_kernel; [0], [1] repeat the two phases of kb_phase_times_ms */
int kb_kernel_times_ms(kb_handle* k, double ms[8], int64_t n[8]);
/* Checkpoint / restore of the agents: the per-learner tables, the control state, the part of the dictionary pool in use and
 * the recorded histories, as one blob (kb_state_bytes waits for the stream and sizes it); kb_load_state takes a blob saved by
 * a handle of the same configuration whose pool is no larger than this handle's. */
int kb_state_bytes(kb_handle* k, uint64_t* bytes);
int kb_save_state(kb_handle* k, void* blob, uint64_t bytes);
int kb_load_state(kb_handle* k, const void* blob, uint64_t bytes);
/* waits for the agent's stream and reports an internal error flag raised by any kernel since kb_reset (the
 * device-resident loop kb_step_resident does not check on its own); dictionaries at capacity are not errors (kb_get_pool) */
int kb_synchronize(kb_handle* k);

#ifdef __cplusplus
}
#endif

#endif /* RANSLICE_H */

// rs_api.hip -- host side of libranslice.so: the C ABI declared in include/ranslice.h.
// Owns device memory, the stream and the launches.  No torch, no CPU fallback: without a HIP
// device rs_create fails with RS_EHIP.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// Developer knobs (sweep switches, the guard bands, the fault injector of the shared step's abort path) are read from the
// environment only by the test build (make dev: -DRS_DEV -> build/libranslice_dev.so, loaded explicitly by the tests that turn
// them: ranslice._lib.load(dev=True)).  The production library reads none of them.
#ifdef RS_DEV
static inline const char* dev_env(const char* name) { return getenv(name); }
#else
static inline const char* dev_env(const char*) { return nullptr; }
#endif

#include "rs_embb.hip"
#include "rs_mmtc.hip"
#include "rs_order.hip"
#include "rs_mux.hip"

using namespace rs;

// ---- guard bands (RANSLICE_GUARD=1, developer knob): every handle-owned device buffer gets 64 KB of 0xA5 on both
// sides; rs_destroy / kb_destroy read them back and abort with the buffer's index if a kernel wrote outside.
struct GuardedAlloc {
    void* base;
    size_t bytes;
};
static const size_t kGuard = 64 * 1024;
static bool guards_on() {
    static const bool on = dev_env("RANSLICE_GUARD") != nullptr;
    return on;
}
static hipError_t guarded_malloc(void** q, size_t bytes, std::vector<GuardedAlloc>* reg) {
    if (!guards_on()) return hipMalloc(q, bytes);
    void* b = nullptr;
    const size_t padded = (bytes + 255) / 256 * 256;
    hipError_t e = hipMalloc(&b, padded + 2 * kGuard);
    if (e != hipSuccess) return e;
    // the bands on both sides (and, for buffers of ordinary size, the payload too, so that reads of unwritten memory
    // show; a multi-gigabyte dictionary pool only gets its bands)
    if (padded <= ((size_t)1 << 30)) {
        e = hipMemset(b, 0xA5, padded + 2 * kGuard);
    } else {
        e = hipMemset(b, 0xA5, kGuard);
        if (e == hipSuccess) e = hipMemset((char*)b + kGuard + padded, 0xA5, kGuard);
    }
    if (e != hipSuccess) return e;
    reg->push_back({b, padded});
    *q = (char*)b + kGuard;
    if (const char* v = dev_env("RANSLICE_GUARD"))
        if (v[0] == '2') fprintf(stderr, "RANSLICE_GUARD: buffer #%zu at %p, %zu bytes\n", reg->size() - 1, *q, bytes);
    return hipSuccess;
}
static void check_guards(const std::vector<GuardedAlloc>& reg, const char* who) {
    std::vector<unsigned char> buf(kGuard);
    for (size_t i = 0; i < reg.size(); ++i)
        for (int side = 0; side < 2; ++side) {
            const char* src = (const char*)reg[i].base + (side ? kGuard + reg[i].bytes : 0);
            if (hipMemcpy(buf.data(), src, kGuard, hipMemcpyDeviceToHost) != hipSuccess) continue;
            for (size_t k = 0; k < kGuard; ++k)
                if (buf[k] != 0xA5) {
                    fprintf(stderr, "RANSLICE_GUARD: %s buffer #%zu (%zu bytes): %s guard overwritten at offset %zu\n", who, i,
                            reg[i].bytes, side ? "upper" : "lower", k);
                    abort();
                }
        }
}

struct rs_handle {
    rs_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;      // the mMTC slices' kernel of a step runs here, beside the eMBB kernels (they share nothing but
                                     // the step's inputs; finalize_kernel waits for both).  RANSLICE_MTC_STREAM=0: one stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // The split step: the head of the cost ranking (and every task with more UEs than eight lanes hold) on the 16-lane instance,
    // one task per wave, on a stream of its own BESIDE the 8-lane launch that takes the rest eight tasks to a wave.
    int mixed = 0;                   // 0 off; n: the first 1/n of the ranking goes to a 16-lane launch of one task per wave
    int mixed_light = 256;           // the last mixed_light/256 of the ranking go to the 8-lane launch (256: all but the head)
    int mixed_ue = 8;                // tasks with this many UEs or more lead the ranking
    hipStream_t side2 = nullptr, side3 = nullptr;
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr, ev_join3 = nullptr;
    RsDev hdev;            // host copy of the device constants
    RsDev* ddev = nullptr;
    RsState st;
    RsState* d_st = nullptr;  // device copy of `st`
    MtcState mst;
    std::vector<void*> allocs;
    std::vector<GuardedAlloc> guarded;
    std::vector<std::pair<void*, size_t>> regions;  // every device array behind the handle but the tables (rs_save_state)
    double* fad = nullptr;
    uint8_t* fad_valid = nullptr;
    double rx_band0 = 0.0;       // the reception test's guard band per RB before the table-dependent term (rx_fast_setup)
    float* fad32 = nullptr;      // the same samples in float32 (the reception test by guard band: rs_embb.hip, fast_sigmoid)
    double* fps = nullptr;       // per column the prefix sums of its samples, P + 1 entries (channel estimates by guard band)
    bool fad_loaded[RS_N_TRACES] = {false, false, false};
    std::vector<double> fad_host[RS_N_TRACES];
    std::vector<uint8_t> valid_host[RS_N_TRACES];
    int32_t* d_actions = nullptr;
    float* d_obs = nullptr;
    double* d_reward = nullptr;
    int32_t* d_labels = nullptr;
    int32_t* d_viol = nullptr;
    double* d_info = nullptr;
    uint64_t* d_counters = nullptr;   // [n_tasks][4]
    uint64_t* d_counter_sum = nullptr;  // [4]
    rs_alloc_rec* d_trace = nullptr;
    uint64_t* d_sections = nullptr;
    unsigned long long* d_pace = nullptr;  // [4] wave pace accumulators of the eMBB step kernel (dynamic priority)
    int32_t* d_redo = nullptr;   // [n_tasks] tasks the fast (G < 32) launch handed to the G = 32 replay
    int32_t* d_order = nullptr;  // [n_tasks] launch order of the step tasks (rs_order.hip)
    uint64_t* d_oslot = nullptr; // [n_tasks] counting-sort scratch
    int* d_ohist = nullptr;      // [2][RS_ORDER_BINS] bin counters, alternating between steps
    int order_par = 0;           // which half of d_ohist the next step counts into
    int order_mode = 6;          // 0: task index order; 1..3: cost keys of rs_order.hip (RANSLICE_ORDER)
    int order_pair = 256;        // modes 4..: share (/256) of the waves led by one heavy task (RANSLICE_PAIR)
    int block_hint = 0;          // 1: wide contested slices are expected, the 16-lane step uses its BLOCK instance (rs_set_schedule_hint)
    int spread_mode = -1;        // one task per wave (StepArgs::spread): -1 = when the batch has at most spread_max tasks, 0 never, 1 always
    int spread_max = 1024;       // SIMDs of the device (rs_create)
    int key_w[4] = {16, 16, 0, 0};  // weights of the cost key in sixteenths (RANSLICE_KEY_W, developer knob; rs_order.hip)
    int rot_mask = 0;             // rounds rotated by half a round (RANSLICE_SNAKE_ROT, developer knob)
    int snake_mask = 0x2aaaaaaa;  // the rounds dealt backwards (bit k = round k): every second one; RANSLICE_SNAKE_MASK (developer knob)
    int snake = 1;               // every second round of waves in reverse cost order (rs_order.hip); RANSLICE_SNAKE=0: off, > 1: the length of a round in waves (developer knob)
    bool hint_auto = true;      // block_hint follows the scenario / the driving agent until the caller sets it
    int group = 16;              // lanes per task of the primary launch: 8, 16 or 32 (profiles/HISTORY.md)
    bool trace_on = false;
    int n_slices = 0, n_vars = 0, n_tasks = 0;   // n_slices = action / label entries per replica
    int n_ran = 0;                                // RAN slices (info rows): n_embb + n_mmtc
    bool mux = false;                             // rs_config.l1_multiplex
    int64_t* d_run = nullptr;    // device-side run state read by the step kernels: [0] slots since reset,
                                 // [1] step index and [2] seed of the on-device action script (rs_run_random)
    hipGraph_t graph = nullptr;  // two captured steps (one per parity of the order counters) of rs_run_random
    hipGraphExec_t gexec = nullptr;
    int graph_par = 0, graph_sig = -1;
    uint64_t launch_sig = 0;     // bumped whenever the launch shape of a step changes (drop_graph): a graph another handle captured
                                 // around this one's steps (kb_run_resident) is stale then
    int32_t clock = 0;     // slots since reset (host mirror of d_run[0])
    uint64_t steps = 0;
    bool is_reset = false;
    // kernel timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t ev_used = 0;
    std::string err;
};

static int auto_hint(const rs_handle* h);

static void drop_graph(rs_handle* h);

#define HIPCHK(h, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
            return RS_EHIP;                                                                          \
        }                                                                                            \
    } while (0)

template <class T>
static int dalloc(rs_handle* h, T** p, size_t n) {
    void* q = nullptr;
    size_t bytes = sizeof(T) * (n ? n : 1);
    HIPCHK(h, guarded_malloc(&q, bytes, &h->guarded));
    HIPCHK(h, hipMemsetAsync(q, 0, bytes, h->stream));
    if (!guards_on()) h->allocs.push_back(q);
    h->regions.emplace_back(q, bytes);
    *p = (T*)q;
    return RS_OK;
}

// ------------------------------------------------------------------ mMTC host helpers

// dynamic LDS of mtc_mux_step_kernel: the shared FIFO (two words per entry) and the device tables of every mMTC RAN slice
static size_t mtc_mux_lds_bytes(int cap, int n_mmtc) {
    return ((size_t)2 * cap * n_mmtc + (size_t)n_mmtc * MTC_DEV_MAX) * sizeof(int32_t);
}

static int mtc_alloc(rs_handle* h, rs::MtcState* m, size_t n_tasks, const RsDev& d) {
    memset(m, 0, sizeof *m);
    m->n_tasks = n_tasks;
    m->cap = d.mtc_cap;
    if (n_tasks == 0) return RS_OK;
    if (d.mtc_n_dev > MTC_DEV_MAX || d.mtc_cap > MTC_CAP_MAX || d.mtc_n_rep <= 0 || d.mtc_n_rep > 8 ||
        d.mtc_n_period <= 0 || d.mtc_n_period > 8) {
        h->err = "rs_create: mMTC configuration out of range (<=1024 devices, queue capacity <=2048)";
        return RS_EINVAL;
    }
    if (d.mux) {
        // mtc_mux_step_kernel keeps the FIFO and the device tables of ALL mMTC RAN slices of a replica in LDS: ask for
        // what it needs now (the default limit on dynamic LDS is 64 KB; six slices at capacity 1024 need 72 KB) instead of
        // failing at the first step
        const size_t lds = mtc_mux_lds_bytes(d.mtc_cap, d.n_mmtc);
        if (lds > 160 * 1024 - 4096) {
            h->err = "rs_create: l1_multiplex with this many mMTC RAN slices and this queue capacity needs more LDS than a CU has";
            return RS_EINVAL;
        }
        if (lds > 48 * 1024)
            HIPCHK(h, hipFuncSetAttribute((const void*)rs::mtc_mux_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    for (int i = 0; i < d.mtc_n_period; ++i)
        if (d.mtc_period_set[i] < d.slots) {
            h->err = "rs_create: mMTC periods shorter than one observation period are not supported";
            return RS_EINVAL;
        }
    int rc;
    if ((rc = dalloc(h, &m->n_users, n_tasks)) != RS_OK) return rc;
    if ((rc = dalloc(h, &m->s_start, n_tasks)) != RS_OK) return rc;
    if ((rc = dalloc(h, &m->s_rep, n_tasks)) != RS_OK) return rc;
    if ((rc = dalloc(h, &m->dev_next, n_tasks * MTC_DEV_MAX)) != RS_OK) return rc;
    if ((rc = dalloc(h, &m->dev_period, n_tasks * MTC_DEV_MAX)) != RS_OK) return rc;
    if ((rc = dalloc(h, &m->dev_rep, n_tasks * MTC_DEV_MAX)) != RS_OK) return rc;
    if ((rc = dalloc(h, &m->q_rep, n_tasks * (size_t)m->cap)) != RS_OK) return rc;
    if ((rc = dalloc(h, &m->q_start, n_tasks * (size_t)m->cap)) != RS_OK) return rc;
    return RS_OK;
}

static int mtc_reset(rs_handle* h, rs::MtcState* m) {
    if (m->n_tasks == 0) return RS_OK;
    hipLaunchKernelGGL(rs::mtc_reset_kernel, dim3((unsigned)m->n_tasks), dim3(256), 0, h->stream, h->ddev, *m,
                       h->st.seeds);
    return RS_OK;
}

static int mtc_step(rs_handle* h, rs::MtcState* m, hipStream_t stream) {
    if (m->n_tasks == 0) return RS_OK;
    rs::MtcArgs a;
    a.D = h->ddev;
    a.M = *m;
    a.actions = h->d_actions;
    a.run = h->d_run;
    a.obs = h->d_obs;
    a.labels = h->d_labels;
    a.violations = h->d_viol;
    a.info = h->d_info;
    a.err = h->st.err;
    size_t lds = (size_t)4 * 2 * m->cap * sizeof(int32_t);
    hipLaunchKernelGGL(rs::mtc_step_kernel, dim3((unsigned)((m->n_tasks + 3) / 4)), dim3(256), lds, stream, a);
    return RS_OK;
}

// ------------------------------------------------------------------ small kernels

namespace rs {

__global__ void reset_kernel(const RsDev* D, RsState S, const uint64_t* seeds_in) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n_tasks = D->n_envs * D->n_embb;
    if (i < n_tasks) {
        S.t_n_ue[i] = 0;
        S.t_cbr_at[i] = 1;  // cbr_steps_next_arrival = 0 (slice_ran.py:185): fires in the first slot
        S.t_vbr_at[i] = 1;
        S.t_ctr[i] = 0;
        S.t_serial[i] = 1;
        S.t_cost[i] = 0;
    }
    if (i < D->n_envs) {
        S.seeds[i] = seeds_in[i];
        S.err[i] = 0;
    }
}

// reward of RanSlice.step (ran_slice.py:45-52)
__global__ void finalize_kernel(const RsDev* D, const int32_t* actions, const int32_t* viol, double* reward,
                                int64_t* run) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= D->n_envs) return;
    if (r == 0) {  // last kernel of the step: advance the device-side clock and the action-script index
        run[0] += D->slots;
        run[1] += 1;
    }
    int S = D->n_act;
    long tv = 0, ta = 0;
    for (int s = 0; s < S; ++s) {
        tv += viol[r * S + s];
        ta += actions[r * S + s];
    }
    double rew;
    if (tv > 0)
        rew = -1 * D->penalty * (double)tv;
    else
        rew = (double)(D->n_prbs - ta > 0 ? D->n_prbs - ta : 0);
    reward[r] = rew;
}

// bench action script: one wave per replica, one categorical draw
// per PRB over S slices + "unused"; identical to rso_random_actions.
// With `run` the seed and the step index come from the device-side run state (graph-captured loops).
__global__ __launch_bounds__(256) void random_actions_kernel(const RsDev* D, int32_t* actions, uint64_t seed,
                                                           uint64_t step, const int64_t* run) {
    if (run) {
        step = (uint64_t)run[1];
        seed = (uint64_t)run[2];
    }
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= D->n_envs) return;
    const int S = D->n_act;
    int my_bin[8];
    int cnt = 0;
    for (int p = lane; p < D->n_prbs; p += 64) {
        uint32_t a, b;
        rs_philox4x32_10((uint32_t)p, (uint32_t)r, (uint32_t)step,
                         (uint32_t)(step >> 32) ^ (uint32_t)((uint64_t)r >> 32) ^ 0x5bd1e995u, (uint32_t)seed,
                         (uint32_t)(seed >> 32), &a, &b);
        my_bin[cnt++ & 7] = (int)(((uint64_t)a * (uint64_t)(S + 1)) >> 32);
    }
    for (int s = 0; s < S; ++s) {
        int c = 0;
        for (int k = 0; k < cnt; ++k) c += my_bin[k] == s ? 1 : 0;
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
        if (lane == 0) actions[r * S + s] = c;
    }
}

__global__ void set_run_kernel(int64_t* run, uint64_t seed, uint64_t step) {
    run[1] = (int64_t)step;
    run[2] = (int64_t)seed;
}

__global__ void counter_sum_kernel(const uint64_t* per_task, int n_tasks, uint64_t* out) {
    __shared__ unsigned long long acc[4];
    if (threadIdx.x < 4) acc[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long loc[4] = {0ull, 0ull, 0ull, 0ull};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_tasks; i += gridDim.x * blockDim.x)
        for (int k = 0; k < 4; ++k) loc[k] += per_task[(size_t)i * 4 + k];
    for (int k = 0; k < 4; ++k) atomicAdd(&acc[k], loc[k]);
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd((unsigned long long*)&out[threadIdx.x], acc[threadIdx.x]);
}

}  // namespace rs

// ------------------------------------------------------------------ create / destroy

static void mcs_factors(double* A, double* B) {  // MCSCodeset.compute_factors (channel_models.py:272-279)
    double Delta = 0.1;
    double a = 1.0 / Delta;
    double s01 = rs_sigmoid(0.1, 0.0, 1.0), s09 = rs_sigmoid(0.9, 0.0, 1.0);
    a = a * (rs_log(1.0 / s01 - 1.0) - rs_log(1.0 / s09 - 1.0));
    *A = a;
    *B = -rs_log(1.0 / s09 - 1.0);
}

// mcs_rate_vs_error (channel_models.py:288-295) + int truncation of the rate (schedulers.py:44)
static void mcs_lookup(const rs_config* c, double A, double B, int e_snr, int* mcs_out, int* rate_out) {
    double rx_prob = 1.0 - 0.1;
    int mcs;
    for (mcs = 0; mcs < c->n_mcs; ++mcs) {
        double x = A * ((double)e_snr - c->mcs_snr[mcs]) - B;
        if (rs_sigmoid(x, 0.0, 1.0) < rx_prob) {
            *mcs_out = mcs - 1 > 0 ? mcs - 1 : 0;
            *rate_out = (int)((double)c->sym_per_prb * (c->mcs_rate[mcs] * (double)c->mcs_order[mcs]));
            return;
        }
    }
    mcs = c->n_mcs - 1;
    *mcs_out = mcs;
    *rate_out = (int)((double)c->sym_per_prb * (c->mcs_rate[mcs] * (double)c->mcs_order[mcs]));
}

// The reception test by guard band (rs_embb.hip, fast_sigmoid): u < p_rx  <=>  S > S*(u) for A, k > 0, both sides formed in
// float32 and compared with a band that covers their errors.  Per RB of the span (absolute, in units of one sigmoid):
//   S~  : 4e-7  = argument (three float roundings: 1.8e-7 |t|, through t sigma'(t) <= 0.224) + v_exp_f32 and v_rcp_f32 at two
//                 ulps each + the rounding of 1 + e;
//   S*~ : 4e-7 for its sigmoid + 0.25 k x 6.2e-6 / A for s* (u and 1 - u in float, v_rcp_f32, v_log_f32 to 4e-6 absolute in
//                 log2 units over |log2| <= 13.3 -- the draw is within [1e-4, 1 - 1e-4] or the UE takes the exact path --,
//                 the float products with ln 2 and 1/A);
//   the exact chain itself (channel_models.py:35-41,297-313 in f64: y = S/n, 1/y - 1, log, A (s - ref) - B, exp, 1/(1+e))
//                 equals the real-valued p at an S perturbed by < 1e-15 n, times (1 +- 1e-12); p rises by at least
//                 u (1 - u) / e x 4 (A/k) x 2e-6 >= 3e-10 (A/k) over 2e-6 n of S, which is what the last term buys.
//   band = 6e-6 + 6e-6 k/A per RB (needed: 2.8e-6 + 1.6e-6 k/A).  Single-RB spans compare s - ref with s* - ref in dB:
//   6.2e-6/A of error + 8e-6/A of margin -> 4e-5/A + 1e-6.
// Off (band 0: every UE evaluated exactly) unless A > 0, every k > 0 and 1 <= A/k <= 1e4.
static void rx_fast_setup(RsDev& d, double* band0) {
    const double A = d.mcsA;
    double kmax = 0.0, kmin = 1e300;
    bool ok = std::isfinite(A) && A > 0.0 && std::isfinite(d.mcsB);
    for (int m = 0; m < 3; ++m) {
        const double k = d.mi_k[m];
        ok = ok && std::isfinite(k) && k > 0.0 && std::isfinite(d.mi_x0[m]);
        kmax = k > kmax ? k : kmax;
        kmin = k < kmin ? k : kmin;
        d.rx_c1[m] = (float)(-k * RS_INV_LN2);
    }
    ok = ok && A / kmax >= 1.0 && A / kmin <= 1.0e4;
    d.rx_invA = ok ? (float)(1.0 / A) : 0.0f;
    d.rx_B = (float)d.mcsB;
    d.rx_band = 0.0;  // set when the tables are loaded (upload_fading adds the float32 samples' term)
    *band0 = ok ? 6.0e-6 + 6.0e-6 * (kmax / A) : 0.0;
    d.rx_band1 = ok ? 4.0e-5 / A + 1.0e-6 : 0.0;
    if (dev_env("RANSLICE_RX_EXACT")) *band0 = 0.0;  // test build: the exact probability for every UE
    if (const char* e = dev_env("RANSLICE_RX_BAND_SCALE")) {
        const double f = atof(e);
        if (f >= 1.0) d.rx_band1 *= f;
    }
}

extern "C" int rs_create(const rs_config* cfg, int device, rs_handle** out) {
    if (!cfg || !out) return RS_EINVAL;
    *out = nullptr;
    rs_handle* h = new rs_handle();
    *out = h;  // returned even on failure so that rs_last_error works; caller destroys it
    h->cfg = *cfg;
    h->device = device;
    if (cfg->n_envs <= 0 || cfg->n_prbs <= 0 || cfg->n_prbs > RS_MAX_PRBS || cfg->n_embb < 0 || cfg->n_mmtc < 0 ||
        cfg->n_embb + cfg->n_mmtc <= 0 || cfg->slots_per_step <= 0 || cfg->slots_per_step > 63 || cfg->n_mcs <= 0 || cfg->n_mcs > 32 ||
        cfg->pf_granularity <= 0 || (cfg->max_ue != 0 && cfg->max_ue != (cfg->l1_multiplex ? RS_MUX_UE : RS_GROUP)) ||
        (cfg->l1_multiplex && (cfg->n_embb > 6 || cfg->n_mmtc > RS_MUX_RAN)) ||
        (cfg->max_bursts != 0 && cfg->max_bursts != RS_BURSTS)) {
        h->err = "rs_create: unsupported configuration (n_prbs <= 256; slots_per_step <= 63; max_ue 0 or 32, 64 with l1_multiplex; max_bursts 0 or 16; "
                 "l1_multiplex: at most 6 eMBB and 8 mMTC RAN slices)";
        return RS_EINVAL;
    }
    if (cfg->l1_multiplex && cfg->n_embb == 0) {
        // the reference always creates the eMBB L1 slice in this mode, even with no eMBB RAN slice to serve
        // (scenario_creator.py:170-172), which gives the action an entry nothing uses; not mirrored
        h->err = "rs_create: l1_multiplex needs at least one eMBB RAN slice";
        return RS_EINVAL;
    }
    {
        // per-UE sums of arrived bits are folded out of order (they must be exactly representable)
        const double cb = cfg->cbr_bit_rate * 1e-3;
        if (cb != std::floor(cb) || cfg->vbr_p_size != std::floor(cfg->vbr_p_size) || cb < 0 || cfg->vbr_p_size < 0 ||
            cb * cfg->slots_per_step > 1e9 || cfg->vbr_p_size * RS_BURSTS * cfg->slots_per_step > 1e9) {
            h->err = "rs_create: CBR bits per slot and VBR packet size must be non-negative integers (< 1e9 per step)";
            return RS_EINVAL;
        }
    }
    int ndev = 0;
    HIPCHK(h, hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        h->err = "rs_create: no such HIP device";
        return RS_EHIP;
    }
    HIPCHK(h, hipSetDevice(device));
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    {
        const char* e = dev_env("RANSLICE_MTC_STREAM");
        if (!(e && atoi(e) == 0)) {
            HIPCHK(h, hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
    }
    h->mux = cfg->l1_multiplex != 0;
    h->n_ran = cfg->n_embb + cfg->n_mmtc;
    h->n_slices = h->mux ? (cfg->n_embb > 0) + (cfg->n_mmtc > 0) : h->n_ran;
    h->n_vars = cfg->n_embb * RS_N_EMBB_VARS + cfg->n_mmtc * RS_N_MMTC_VARS;
    h->n_tasks = cfg->n_envs * cfg->n_embb;

    RsDev& d = h->hdev;
    memset(&d, 0, sizeof d);
    d.n_envs = cfg->n_envs;
    d.n_prbs = cfg->n_prbs;
    d.n_embb = cfg->n_embb;
    d.n_mmtc = cfg->n_mmtc;
    d.n_slices = h->n_ran;
    d.n_act = h->n_slices;
    d.mux = h->mux ? 1 : 0;
    d.slots = cfg->slots_per_step;
    d.n_vars = h->n_vars;
    d.slot_length = cfg->slot_length;
    d.cbr_bits = cfg->cbr_bit_rate * 1e-3;
    d.cbr_ia_scale = 1.0 / cfg->cbr_lambda;
    d.cbr_hold_scale = cfg->cbr_t_mean;
    d.vbr_ia_scale = 1.0 / cfg->vbr_lambda;
    d.vbr_hold_scale = cfg->vbr_t_mean;
    d.vbr_p_size = cfg->vbr_p_size;
    d.vbr_b_size = cfg->vbr_b_size;
    d.vbr_inter = (1 / cfg->vbr_b_rate) / cfg->slot_length;
    for (int i = 0; i < 6; ++i) d.sla[i] = cfg->sla_embb[i];
    for (int i = 0; i < 10; ++i) d.norm[i] = cfg->norm_embb[i];
    d.prop_A = cfg->prop_A;
    d.prop_B = cfg->prop_B;
    mcs_factors(&d.mcsA, &d.mcsB);
    d.pf_b = 1.0 / cfg->pf_window;
    d.pf_a = 1 - d.pf_b;
    d.gran = cfg->pf_granularity;
    d.penalty = cfg->penalty;
    // e_snr -> (mcs, rate): outside [lo, hi] the answer is constant (first / no failing MCS)
    {
        int lo = (int)std::floor(cfg->mcs_snr[0]) - 4, hi = (int)std::ceil(cfg->mcs_snr[cfg->n_mcs - 1]) + 4;
        if (hi - lo + 1 > RS_LUT_MAX) {
            h->err = "rs_create: MCS table spans too many dB for the lookup";
            return RS_EINVAL;
        }
        d.lut_lo = lo;
        d.lut_n = hi - lo + 1;
        for (int e = lo; e <= hi; ++e) mcs_lookup(cfg, d.mcsA, d.mcsB, e, &d.lut_mcs[e - lo], &d.lut_rate[e - lo]);
        int m, r;
        mcs_lookup(cfg, d.mcsA, d.mcsB, lo - 50, &m, &r);
        bool ok = m == d.lut_mcs[0] && r == d.lut_rate[0];
        mcs_lookup(cfg, d.mcsA, d.mcsB, hi + 50, &m, &r);
        ok = ok && m == d.lut_mcs[d.lut_n - 1] && r == d.lut_rate[d.lut_n - 1];
        if (!ok) {
            h->err = "rs_create: MCS lookup is not constant outside the tabulated range";
            return RS_EINVAL;
        }
    }
    {
        // The PF metric update divides pf_b * bits by the slot length once per granted RB pair, inside a chain
        // of dependent operations.  With the correctly rounded reciprocal, q = x * rc; r = fma(-q, d, x);
        // q' = fma(r, rc, q) is the correctly rounded quotient for almost every x; `bits` is an integer no
        // larger than n_prbs * max rate, so instead of arguing about the exceptions every reachable value is
        // compared with the true divide here, and the kernel uses the short form only if all of them agree.
        int max_rate = 0;
        for (int i = 0; i < d.lut_n; ++i) max_rate = d.lut_rate[i] > max_rate ? d.lut_rate[i] : max_rate;
        if (max_rate >= 65536) {
            h->err = "rs_create: bits per PRB do not fit the packed MCS lookup";
            return RS_EINVAL;
        }
        const double dl = cfg->slot_length, rc = 1.0 / dl;
        bool same = std::isfinite(rc) && dl > 0.0;
        const long long top = (long long)cfg->n_prbs * max_rate;
        for (long long b = 0; same && b <= top; ++b) {
            const double x = d.pf_b * (double)b;
            const double q = x * rc;
            const double r = std::fma(-q, dl, x);
            same = std::fma(r, rc, q) == x / dl;
        }
        d.slot_rc = rc;
        d.pf_div_fast = (same && !dev_env("RANSLICE_EXACT_DIV")) ? 1 : 0;  // the variable forces the divide (tests)
    }
    for (int m = 0; m < cfg->n_mcs; ++m) {
        d.mcs_ref[m] = cfg->mcs_snr[m];
        if (cfg->mcs_mod[m] < 0 || cfg->mcs_mod[m] > 2) {
            h->err = "rs_create: modulation index out of range";
            return RS_EINVAL;
        }
        d.mcs_mod[m] = cfg->mcs_mod[m];
    }
    for (int m = 0; m < 3; ++m) {
        d.mi_x0[m] = cfg->mi_x0[m];
        d.mi_k[m] = cfg->mi_k[m];
    }
    rx_fast_setup(d, &h->rx_band0);
    d.mtc_n_dev = cfg->mtc_n_devices;
    d.mtc_cap = cfg->max_mtc_queue > 0 ? cfg->max_mtc_queue : 1024;
    d.mtc_n_rep = cfg->mtc_n_rep;
    d.mtc_n_period = cfg->mtc_n_period;
    for (int i = 0; i < 8; ++i) {
        d.mtc_rep_set[i] = cfg->mtc_rep_set[i];
        d.mtc_period_set[i] = cfg->mtc_period_set[i];
    }
    d.sla_mtc_delay = cfg->sla_mtc_delay;
    for (int i = 0; i < 3; ++i) d.norm_mmtc[i] = cfg->norm_mmtc[i];

    int rc;
    const size_t T = (size_t)h->n_tasks, N = (size_t)cfg->n_envs;
    const size_t U = T * RS_GROUP;
    RsState& s = h->st;
#define DA(p, n)                                   \
    if ((rc = dalloc(h, &(p), (n))) != RS_OK) return rc
    DA(h->ddev, 1);
    DA(s.t_n_ue, T); DA(s.t_cbr_at, T); DA(s.t_vbr_at, T); DA(s.t_ctr, T); DA(s.t_serial, T); DA(s.t_cost, T);
    DA(s.u_queue, U); DA(s.u_th, U); DA(s.u_nominal, U);
    DA(s.u_hold_at, U); DA(s.u_e_snr, U); DA(s.u_findex, U); DA(s.u_bits, U); DA(s.u_prbs, U);
    DA(s.u_vbr_at, U); DA(s.u_ctr, U); DA(s.u_serial, U); DA(s.u_flags, U);
    DA(s.u_burst, U * RS_BURSTS);
    DA(s.seeds, N); DA(s.err, N);
    DA(h->d_actions, N * h->n_slices);
    DA(h->d_obs, N * h->n_vars);
    DA(h->d_reward, N);
    DA(h->d_labels, N * h->n_slices);
    DA(h->d_viol, N * h->n_slices);
    DA(h->d_info, N * h->n_ran * 10);
    DA(h->d_counters, (T ? T : 1) * 4);
    DA(h->d_counter_sum, 4);
    DA(h->d_sections, 16 + 4 * (T ? T : 1) + 16);
    DA(h->d_redo, T ? T : 1);
    DA(h->d_order, T ? T : 1);
    DA(h->d_oslot, T ? T : 1);
    DA(h->d_ohist, 2 * RS_ORDER_BINS);
    HIPCHK(h, hipMemset(h->d_ohist, 0, sizeof(int) * 2 * RS_ORDER_BINS));
    if (const char* e = dev_env("RANSLICE_ORDER")) h->order_mode = atoi(e);
    {
        // One heavy task per wave pays while the whole batch is co-resident (the launch ends with its heaviest wave;
        // 1.13 vs 1.27 ms at 4096 replicas).  A batch of several rounds of waves is bound by the instructions issued
        // instead, and waves of like tasks issue a sixth fewer (3.9 vs 3.4 M env-steps/s at 8192 replicas, 4.3 vs 3.6
        // at 16384; profiles/HISTORY.md).
        int cus = 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus = 256;
        (void)hipGetLastError();
        const long long resident_waves = (long long)cus * 4 * RS_OCC;
        h->order_pair = (long long)h->n_tasks / 4 > resident_waves + resident_waves / 4 ? 0 : 256;
    }
    if (const char* e = dev_env("RANSLICE_PAIR")) h->order_pair = atoi(e);
    // Lanes per task: with few tasks the step is pure latency and the 32-lane instance (more lanes per sum and per
    // RB pass, all its waves co-resident at 3 per SIMD up to 6144 tasks) is faster; from there on 16 lanes
    // (4 tasks per wave, 5 waves per SIMD) carry more tasks in flight (profiles/HISTORY.md).
    h->group = h->n_tasks <= 6144 ? 32 : 16;
    if (const char* e = dev_env("RANSLICE_GROUP")) {  // developer knob (profiles/HISTORY.md)
        const int g = atoi(e);
        if (g == 8 || g == 16 || g == 32) h->group = g;
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) h->spread_max = cus * 4;
        (void)hipGetLastError();
        if (const char* e = dev_env("RANSLICE_SPREAD")) h->spread_mode = atoi(e);  // developer knob
        if (const char* e = dev_env("RANSLICE_SNAKE")) h->snake = atoi(e);
        if (const char* e = dev_env("RANSLICE_KEY_W")) (void)sscanf(e, "%d,%d,%d,%d", &h->key_w[0], &h->key_w[1], &h->key_w[2], &h->key_w[3]);
        if (const char* e = dev_env("RANSLICE_SNAKE_MASK")) h->snake_mask = (int)strtol(e, nullptr, 0);
        if (const char* e = dev_env("RANSLICE_SNAKE_ROT")) h->rot_mask = (int)strtol(e, nullptr, 0);
    }
    // The split step pays where the batch is several rounds of waves -- the launch is then bound by the instructions issued, and
    // eight like tasks per wave issue a sixth fewer than four -- and loses where every wave is resident at once and the launch ends
    // with its slowest wave, whose chain eight tasks in lockstep lengthen (round 5, 4096 / 8192 / 16384 / 65536 replicas: step
    // kernel 1.08 -> 1.19-1.31 ms, 1.98 -> 1.95, 3.55 -> 2.96, 13.2 -> 11.5 with a head; profiles/r05_c_mixed.txt).  With round 6's
    // kernel (profiles/r06_x_split_threshold.txt): 8192 replicas of five slices even, 10,240 replicas 5.29 -> 5.79 M env-steps/s.  From
    // 49,152 tasks on: the lightest 7/8 of the ranking on the 8-lane instance.
    if (h->group == 16 && h->n_tasks >= 49152) h->mixed_light = 224;
    if (const char* e = dev_env("RANSLICE_MIXED")) h->mixed = atoi(e);
    if (const char* e = dev_env("RANSLICE_MIXED_UE")) h->mixed_ue = atoi(e);
    if (const char* e = dev_env("RANSLICE_MIXED_LIGHT")) h->mixed_light = atoi(e);
    if (h->mixed > 0 || h->mixed_light < 256) {
        // the split step's two extra streams exist only on handles that split: several handles side by side (evaluate_grid: six cells)
        // must stay within the hardware queues of the runtime -- with two idle streams more per handle the grid took 220 s against 135
        HIPCHK(h, hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork2, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_join2, hipEventDisableTiming));
        HIPCHK(h, hipStreamCreateWithFlags(&h->side3, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_join3, hipEventDisableTiming));
    }
    h->block_hint = auto_hint(h);
    if (const char* e = dev_env("RANSLICE_HINT")) {  // developer knob (profiles/HISTORY.md): as rs_set_schedule_hint
        h->hint_auto = atoi(e) < 0;
        if (!h->hint_auto) h->block_hint = atoi(e) ? 1 : 0;
    }
    DA(h->d_st, 1);
    DA(h->d_run, 4);
    DA(h->d_pace, 4);
    if ((rc = mtc_alloc(h, &h->mst, N * (size_t)cfg->n_mmtc, d)) != RS_OK) return rc;
#undef DA
    HIPCHK(h, hipMemcpyAsync(h->d_st, &h->st, sizeof(RsState), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RS_OK;
}

extern "C" void rs_destroy(rs_handle* h) {
    if (!h) return;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    drop_graph(h);
    if (guards_on()) check_guards(h->guarded, "rs");
    for (auto& g : h->guarded) (void)hipFree(g.base);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->fad) (void)hipFree(h->fad);
    if (h->fad_valid) (void)hipFree(h->fad_valid);
    if (h->fad32) (void)hipFree(h->fad32);
    if (h->fps) (void)hipFree(h->fps);
    if (h->d_trace) (void)hipFree(h->d_trace);
    for (auto& e : h->ev) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->side2) (void)hipStreamDestroy(h->side2);
    if (h->side3) (void)hipStreamDestroy(h->side3);
    if (h->ev_join3) (void)hipEventDestroy(h->ev_join3);
    if (h->ev_fork2) (void)hipEventDestroy(h->ev_fork2);
    if (h->ev_join2) (void)hipEventDestroy(h->ev_join2);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" const char* rs_last_error(const rs_handle* h) { return h ? h->err.c_str() : "null handle"; }
extern "C" int rs_n_vars(const rs_handle* h) { return h ? h->n_vars : RS_EINVAL; }
extern "C" int rs_n_slices(const rs_handle* h) { return h ? h->n_slices : RS_EINVAL; }
/* HIP devices visible to this process (a launcher picks `device = local_rank % rs_device_count()`) */
extern "C" int rs_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : RS_EHIP;
}

/* free and total bytes of a device's memory (hipMemGetInfo): callers size the KBRL pool from it (bench.py, tools/run_length.py) */
extern "C" int rs_device_mem_info(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
    if (!free_bytes || !total_bytes) return RS_EINVAL;
    size_t f = 0, t = 0;
    if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) return RS_EHIP;
    *free_bytes = (uint64_t)f;
    *total_bytes = (uint64_t)t;
    return RS_OK;
}

// ------------------------------------------------------------------ fading tables

static int upload_fading(rs_handle* h) {
    RsDev& d = h->hdev;
    size_t elems = 0, vbytes = 0;
    d.has_nan = 0;
    for (int f = 0; f < RS_N_TRACES; ++f) {
        d.fad_off[f] = (int64_t)elems;
        d.valid_off[f] = (int64_t)vbytes;
        elems += h->fad_host[f].size();
        vbytes += h->valid_host[f].size();
        bool any_valid = false;
        for (uint8_t v : h->valid_host[f]) {
            if (!v) d.has_nan = 1;
            any_valid = any_valid || v;
        }
        if (!any_valid) {
            h->err = "rs_load_fading: a trace has no NaN-free column";
            return RS_EINVAL;
        }
    }
    if (elems + (size_t)RS_MAX_PRBS >= (size_t)1 << 31) {
        h->err = "rs_load_fading: tables exceed 2^31 samples";
        return RS_EINVAL;
    }
    if (h->fad) (void)hipFree(h->fad);
    if (h->fad_valid) (void)hipFree(h->fad_valid);
    if (h->fad32) (void)hipFree(h->fad32);
    if (h->fps) (void)hipFree(h->fps);
    h->fad32 = nullptr;
    h->fps = nullptr;
    // tail padding so that a subgroup's strided reads never leave the allocation
    HIPCHK(h, hipMalloc((void**)&h->fad, sizeof(double) * (elems + 16)));
    HIPCHK(h, hipMalloc((void**)&h->fad_valid, vbytes + 16));
    if (const char* v = dev_env("RANSLICE_GUARD"))
        if (v[0] == '2') fprintf(stderr, "RANSLICE_GUARD: fading table at %p, %zu bytes\n", (void*)h->fad, sizeof(double) * (elems + 16));
    for (int f = 0; f < RS_N_TRACES; ++f) {
        HIPCHK(h, hipMemcpyAsync(h->fad + d.fad_off[f], h->fad_host[f].data(), sizeof(double) * h->fad_host[f].size(),
                                 hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->fad_valid + d.valid_off[f], h->valid_host[f].data(), h->valid_host[f].size(),
                                 hipMemcpyHostToDevice, h->stream));
    }
    {
        // Two derived tables for the decisions the step kernel takes by guard band (rs_embb.hip): the samples in float32
        // for the MI sums of the reception test, and per column the prefix sums of its samples (accumulated in long double,
        // rounded once) for the channel estimates, round(mean(snr over the slice's RBs)) = round((PS[hi] - PS[lo]) / n + nominal)
        // unless the mean lies within est_band of a half-integer.  Error of that mean against the f64 pairwise sum the exact
        // path forms: 2^-52 P smax (the two prefix entries) + 1.1e-15 (smax + |nominal|) (the pairwise sum) + the roundings of
        // the quotient and the sum < 1.1e-10 for smax <= 1e3, |mean| <= 3e4 (checked by the kernel) -> est_band 1e-9.
        const int P = d.P;
        std::vector<float> t32(elems + 64, 0.0f);
        std::vector<double> ps((elems / (size_t)P) * (size_t)(P + 1) + 16, 0.0);
        double smax = 0.0;
        bool finite = true;
        for (int f = 0; f < RS_N_TRACES; ++f) {
            const std::vector<double>& tab = h->fad_host[f];
            const size_t cols = tab.size() / (size_t)P, c0 = (size_t)d.fad_off[f] / (size_t)P;
            d.col_off[f] = (int32_t)c0;
            for (size_t t = 0; t < cols; ++t) {
                long double acc = 0.0L;
                double* row = &ps[(c0 + t) * (size_t)(P + 1)];
                row[0] = 0.0;
                const bool ok = h->valid_host[f][t] != 0;
                for (int p = 0; p < P; ++p) {
                    const double v = tab[t * (size_t)P + p];
                    t32[(size_t)d.fad_off[f] + t * (size_t)P + p] = (float)v;
                    acc += (long double)v;
                    row[p + 1] = (double)acc;
                    if (ok) {
                        finite = finite && std::isfinite(v);
                        smax = std::fabs(v) > smax ? std::fabs(v) : smax;
                    }
                }
            }
        }
        const bool small = finite && smax <= 1.0e3;
        // (the kernel addresses the prefix table with 32-bit element offsets, as it does the samples)
        const bool ps_fits = ps.size() + (size_t)RS_MAX_PRBS < ((size_t)1 << 31);
        d.est_band = (small && ps_fits && !dev_env("RANSLICE_EST_EXACT")) ? 1.0e-9 : 0.0;
        // the float32 sample adds 2^-24 smax to the sigmoid's argument in dB: 0.25 k 2^-24 smax per RB, doubled
        double kmax = 0.0;
        for (int m = 0; m < 3; ++m) kmax = d.mi_k[m] > kmax ? d.mi_k[m] : kmax;
        d.rx_band = (small && h->rx_band0 > 0.0) ? h->rx_band0 + 0.5 * kmax * 5.97e-8 * smax : 0.0;
        if (const char* e = dev_env("RANSLICE_RX_BAND_SCALE")) {  // test build: a wider band sends more UEs down the exact path
            const double fsc = atof(e);
            if (fsc >= 1.0) d.rx_band *= fsc;
        }
        if (const char* e = dev_env("RANSLICE_EST_BAND")) {  // test build: the estimates' band itself (0.2: two in five by the pairwise sum)
            const double b = atof(e);
            if (d.est_band > 0.0 && b > 0.0 && b < 0.5) d.est_band = b;
        }
        HIPCHK(h, hipMalloc((void**)&h->fad32, sizeof(float) * t32.size()));
        HIPCHK(h, hipMalloc((void**)&h->fps, sizeof(double) * ps.size()));
        HIPCHK(h, hipMemcpy(h->fad32, t32.data(), sizeof(float) * t32.size(), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->fps, ps.data(), sizeof(double) * ps.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(h, hipMemcpyAsync(h->ddev, &d, sizeof d, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int f = 0; f < RS_N_TRACES; ++f) {
        std::vector<double>().swap(h->fad_host[f]);
        std::vector<uint8_t>().swap(h->valid_host[f]);
    }
    return RS_OK;
}

extern "C" int rs_load_fading(rs_handle* h, int trace_id, const double* data, int rows, int cols) {
    if (!h) return RS_EINVAL;
    if (trace_id < 0 || trace_id >= RS_N_TRACES || !data || rows <= 0 || cols <= 0 || h->cfg.n_prbs > 2 * rows) {
        h->err = "rs_load_fading: bad arguments";
        return RS_EINVAL;
    }
    HIPCHK(h, hipSetDevice(h->device));
    drop_graph(h);
    const int P = h->cfg.n_prbs > rows ? h->cfg.n_prbs : rows;
    if (h->hdev.P != 0 && h->hdev.P != P) {
        h->err = "rs_load_fading: all traces must have the same number of rows";
        return RS_EINVAL;
    }
    h->hdev.P = P;
    h->hdev.T[trace_id] = cols;
    // transpose to [time][PRB] with the reference's row wrap (channel_models.py:144-148); a column
    // is invalid if np.isnan(np.sum(column)) over all rows (channel_models.py:188-189)
    std::vector<double>& tab = h->fad_host[trace_id];
    std::vector<uint8_t>& val = h->valid_host[trace_id];
    tab.assign((size_t)P * cols, 0.0);
    val.assign((size_t)cols, 1);
    for (int t = 0; t < cols; ++t) {
        bool bad = false;
        for (int p = 0; p < P; ++p) {
            double v = data[(size_t)(p % rows) * cols + t];
            tab[(size_t)t * P + p] = v;
            bad = bad || (v != v);
        }
        val[t] = bad ? 0 : 1;
    }
    h->fad_loaded[trace_id] = true;
    if (h->fad_loaded[0] && h->fad_loaded[1] && h->fad_loaded[2]) return upload_fading(h);
    return RS_OK;
}

// ------------------------------------------------------------------ reset / step

static bool fading_ready(const rs_handle* h) {
    return h->cfg.n_embb == 0 || (h->fad != nullptr && h->fad_loaded[0] && h->fad_loaded[1] && h->fad_loaded[2]);
}

extern "C" int rs_reset(rs_handle* h, const uint64_t* seeds, float* obs) {
    if (!h || !seeds) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    if (!fading_ready(h)) {
        h->err = "rs_reset: fading traces not loaded";
        return RS_ESTATE;
    }
    const int N = h->cfg.n_envs;
    if (h->cfg.n_embb == 0) HIPCHK(h, hipMemcpyAsync(h->ddev, &h->hdev, sizeof(RsDev), hipMemcpyHostToDevice, h->stream));
    uint64_t* tmp = nullptr;
    HIPCHK(h, hipMalloc((void**)&tmp, sizeof(uint64_t) * N));
    HIPCHK(h, hipMemcpyAsync(tmp, seeds, sizeof(uint64_t) * N, hipMemcpyHostToDevice, h->stream));
    int n = h->n_tasks > N ? h->n_tasks : N;
    hipLaunchKernelGGL(reset_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->ddev, h->st, tmp);
    if (h->cfg.n_mmtc > 0) mtc_reset(h, &h->mst);
    HIPCHK(h, hipMemsetAsync(h->d_counters, 0, sizeof(uint64_t) * 4 * (h->n_tasks ? h->n_tasks : 1), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_obs, 0, sizeof(float) * N * h->n_vars, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_run, 0, sizeof(int64_t) * 4, h->stream));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    (void)hipFree(tmp);
    h->clock = 0;
    h->steps = 0;
    h->is_reset = true;
    if (obs) memset(obs, 0, sizeof(float) * (size_t)N * h->n_vars);
    return RS_OK;
}

static int launch_step(rs_handle* h) {
    if (!h->is_reset) {
        h->err = "rs_step: call rs_reset first";
        return RS_ESTATE;
    }
    if (h->clock > 2000000000 - h->cfg.slots_per_step) {
        h->err = "rs_step: slot clock would overflow; reset the environment";
        return RS_ESTATE;
    }
    if (h->mux) {
        if (h->n_tasks > 0) {
            StepArgs a;
            memset(&a, 0, sizeof a);
            a.D = h->ddev;
            a.S = h->d_st;
            a.fad = h->fad;
            a.fad32 = h->fad32;
            a.fps = h->fps;
            a.fad_valid = h->fad_valid;
            a.actions = h->d_actions;
            a.run = h->d_run;
            a.obs = h->d_obs;
            a.labels = h->d_labels;
            a.violations = h->d_viol;
            a.info = h->d_info;
            a.counters = h->d_counters;
            a.trace = h->d_trace;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (h->timing) {
                if (h->ev_used == h->ev.size()) {
                    hipEvent_t a0, a1;
                    HIPCHK(h, hipEventCreate(&a0));
                    HIPCHK(h, hipEventCreate(&a1));
                    h->ev.emplace_back(a0, a1);
                }
                e0 = h->ev[h->ev_used].first;
                e1 = h->ev[h->ev_used].second;
                h->ev_used++;
                HIPCHK(h, hipEventRecord(e0, h->stream));
            }
            if (h->trace_on) hipLaunchKernelGGL((embb_mux_step_kernel<true>), dim3((unsigned)h->cfg.n_envs), dim3(64), 0, h->stream, a);
            else hipLaunchKernelGGL((embb_mux_step_kernel<false>), dim3((unsigned)h->cfg.n_envs), dim3(64), 0, h->stream, a);
            if (h->timing) HIPCHK(h, hipEventRecord(e1, h->stream));
        }
        if (h->cfg.n_mmtc > 0) {
            rs::MtcArgs ma;
            ma.D = h->ddev;
            ma.M = h->mst;
            ma.actions = h->d_actions;
            ma.run = h->d_run;
            ma.obs = h->d_obs;
            ma.labels = h->d_labels;
            ma.violations = h->d_viol;
            ma.info = h->d_info;
            ma.err = h->st.err;
            const size_t lds = mtc_mux_lds_bytes(h->mst.cap, h->cfg.n_mmtc);
            hipLaunchKernelGGL(rs::mtc_mux_step_kernel, dim3((unsigned)h->cfg.n_envs), dim3(64), lds, h->stream, ma);
        }
        hipLaunchKernelGGL(finalize_kernel, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, h->stream, h->ddev,
                           h->d_actions, h->d_viol, h->d_reward, h->d_run);
        HIPCHK(h, hipGetLastError());
        h->clock += h->cfg.slots_per_step;
        h->steps += 1;
        return RS_OK;
    }
    // the mMTC slices step beside the eMBB ones on the side stream (fork here, join before finalize_kernel; inside a
    // stream capture the pair becomes two branches of the graph)
    const bool forked = h->side && h->cfg.n_mmtc > 0 && h->n_tasks > 0 && h->mst.n_tasks > 0;
    if (forked) {
        HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
        mtc_step(h, &h->mst, h->side);
        HIPCHK(h, hipEventRecord(h->ev_join, h->side));
    }
    if (h->n_tasks > 0) {
        StepArgs a;
        a.D = h->ddev;
        a.S = h->d_st;
        a.fad = h->fad;
        a.fad32 = h->fad32;
        a.fps = h->fps;
        a.fad_valid = h->fad_valid;
        a.actions = h->d_actions;
        a.run = h->d_run;
        a.obs = h->d_obs;
        a.labels = h->d_labels;
        a.violations = h->d_viol;
        a.info = h->d_info;
        a.counters = h->d_counters;
        a.trace = h->d_trace;
        a.sections = h->d_sections;
        a.redo = h->d_redo;
        a.pace = h->d_pace;
        a.replay = 0;
        a.order = nullptr;
        a.spread = 0;
        a.order_off = 0;
        a.order_cnt = -1;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (h->timing) {
            if (h->ev_used == h->ev.size()) {
                hipEvent_t a0, a1;
                HIPCHK(h, hipEventCreate(&a0));
                HIPCHK(h, hipEventCreate(&a1));
                h->ev.emplace_back(a0, a1);
            }
            e0 = h->ev[h->ev_used].first;
            e1 = h->ev[h->ev_used].second;
            h->ev_used++;
        }
        hipStream_t lstream = h->stream;
        auto launch = [&](int g) {
            // at most one wave per SIMD of the chip: one task per wave (StepArgs::spread)
            if (a.order_cnt < 0) a.spread = (h->spread_mode == 1 || (h->spread_mode < 0 && h->n_tasks <= h->spread_max)) ? 1 : 0;
            const int per_block = a.spread ? 4 : 256 / g;
            const int n_mine = a.order_cnt >= 0 ? a.order_cnt : h->n_tasks;
            dim3 grid((n_mine + per_block - 1) / per_block), block(256);
            const bool tr = h->trace_on;
            // BLOCK instances hand out the RB pairs of wide contested slices in block rounds; the plain 16-lane one
            // carries the trip loop alone (rs_set_schedule_hint)
#define RS_LAUNCH_STEP(G_, TR_, BL_)                                                                              \
    do {                                                                                                          \
        if (h->hdev.pf_div_fast)                                                                                  \
            hipLaunchKernelGGL((embb_step_kernel<G_, TR_, BL_, true>), grid, block, 0, lstream, a);             \
        else                                                                                                      \
            hipLaunchKernelGGL((embb_step_kernel<G_, TR_, BL_, false>), grid, block, 0, lstream, a);            \
    } while (0)
            if (g == 8) {
                if (tr) RS_LAUNCH_STEP(8, true, true);
                else if (h->block_hint) RS_LAUNCH_STEP(8, false, true);
                else hipLaunchKernelGGL((embb_step_kernel<8, false, false, true>), grid, block, 0, lstream, a);
            } else if (g == 16) {
                if (tr) RS_LAUNCH_STEP(16, true, true);
                else if (h->block_hint) RS_LAUNCH_STEP(16, false, true);
                else hipLaunchKernelGGL((embb_step_kernel<16, false, false, true>), grid, block, 0, lstream, a);  // (run-time flag inside)
            } else {
                if (tr) RS_LAUNCH_STEP(32, true, true);
                else RS_LAUNCH_STEP(32, false, true);
            }
#undef RS_LAUNCH_STEP
        };
        // primary launch with h->group lanes per task; tasks that do not fit raise their redo flag and are
        // replayed from their untouched state by the 32-lane instance (waves without flagged tasks exit)
        int seg_head = 0, seg_light = 0;
        if (h->order_mode > 0) {
            const int par = h->order_par;
            h->order_par ^= 1;
            const unsigned nb = (unsigned)((h->n_tasks + 255) / 256);
            const bool split = h->side2 != nullptr && (h->mixed > 0 || h->mixed_light < 256) && !h->trace_on && h->group == 16 && h->n_tasks >= 4096;
            // the three stretches of the ranking (heaviest first): [0, head) one task per 16-lane wave, [head, n - light) four per
            // 16-lane wave, composed among themselves as without a split, [n - light, n) eight per 8-lane wave
            if (split) {
                seg_head = h->mixed > 0 ? (h->n_tasks / h->mixed) & ~3 : 0;
                seg_light = h->mixed_light >= 256 ? h->n_tasks - seg_head : ((int)(((long long)h->n_tasks * h->mixed_light) >> 8)) & ~7;
                if (((h->n_tasks - seg_head - seg_light) & 3) != 0) seg_light += (h->n_tasks - seg_head - seg_light) & 3;
            }
            const int seg_mid = h->n_tasks - seg_head - seg_light;
            hipLaunchKernelGGL(order_key_kernel, dim3(nb), dim3(256), 0, h->stream, h->ddev, h->d_st, h->d_actions,
                               h->order_mode, h->d_ohist + par * RS_ORDER_BINS, h->d_oslot, make_int4(h->key_w[0], h->key_w[1], h->key_w[2], h->key_w[3]),
                               split ? h->mixed_ue : 0);
            hipLaunchKernelGGL(order_scatter_kernel, dim3(nb), dim3(256), 0, h->stream, h->ddev,
                               h->d_ohist + par * RS_ORDER_BINS, h->d_ohist + (1 - par) * RS_ORDER_BINS, h->d_oslot,
                               h->d_order, h->order_mode > 3 ? h->order_pair : 0, 64 / h->group,  // modes 4.. = keys 1.. with heavy+light pairing
                               h->snake == 1 ? h->spread_max : h->snake, h->snake_mask, h->rot_mask, seg_head, seg_head + seg_mid);
            a.order = h->d_order;
        }
        // the event pair brackets the step launches of the step (without a split: what rocprofv3 lists as embb_step_kernel<G,...>)
        if (h->timing) HIPCHK(h, hipEventRecord(e0, h->stream));
        if (a.order && seg_head + seg_light > 0) {
            const int seg_mid = h->n_tasks - seg_head - seg_light;
            HIPCHK(h, hipEventRecord(h->ev_fork2, h->stream));
            if (seg_head > 0) {
                HIPCHK(h, hipStreamWaitEvent(h->side2, h->ev_fork2, 0));
                lstream = h->side2;
                a.order_off = 0;
                a.order_cnt = seg_head;
                a.spread = 1;
                launch(16);
                HIPCHK(h, hipEventRecord(h->ev_join2, h->side2));
            }
            if (seg_mid > 0 && seg_light > 0) {
                HIPCHK(h, hipStreamWaitEvent(h->side3, h->ev_fork2, 0));
                lstream = h->side3;
                a.order_off = seg_head + seg_mid;
                a.order_cnt = seg_light;
                a.spread = 0;
                launch(8);
                HIPCHK(h, hipEventRecord(h->ev_join3, h->side3));
            }
            lstream = h->stream;
            a.spread = 0;
            if (seg_mid > 0) {
                a.order_off = seg_head;
                a.order_cnt = seg_mid;
                launch(16);
            } else {
                a.order_off = seg_head;
                a.order_cnt = seg_light;
                launch(8);
            }
            if (seg_head > 0) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join2, 0));
            if (seg_mid > 0 && seg_light > 0) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join3, 0));
            a.order_off = 0;
            a.order_cnt = -1;
        } else {
            launch(h->group);
        }
        if (h->timing) HIPCHK(h, hipEventRecord(e1, h->stream));
        a.order = nullptr;
        if (h->group < 32) {
            a.replay = 1;
            launch(32);
        }
    }
    if (forked) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
    else if (h->cfg.n_mmtc > 0) mtc_step(h, &h->mst, h->stream);
    hipLaunchKernelGGL(finalize_kernel, dim3((h->cfg.n_envs + 255) / 256), dim3(256), 0, h->stream, h->ddev,
                       h->d_actions, h->d_viol, h->d_reward, h->d_run);
    HIPCHK(h, hipGetLastError());
    h->clock += h->cfg.slots_per_step;
    h->steps += 1;
    return RS_OK;
}

// Default of the scheduling hint: an even split of the carrier (what the on-device random script deals out on average)
// gives slices wide enough for block rounds?  rs_step looks at the allocations it is handed, kb_step_resident asks for
// the BLOCK instance outright.
static int auto_hint(const rs_handle* h) {
    const int gran = h->cfg.pf_granularity > 0 ? h->cfg.pf_granularity : 1;
    return h->cfg.n_prbs / (h->n_slices + 1) >= RS_HINT_PAIRS * gran ? 1 : 0;
}

static int check_errors(rs_handle* h) {
    std::vector<int32_t> e((size_t)h->cfg.n_envs);
    HIPCHK(h, hipMemcpyAsync(e.data(), h->st.err, sizeof(int32_t) * e.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < e.size(); ++i)
        if (e[i]) {
            h->err = std::string("capacity exceeded in replica ") + std::to_string(i) + ":" +
                     ((e[i] & 1) ? " UEs per slice" : "") + ((e[i] & 2) ? " active VBR bursts per UE" : "") +
                     ((e[i] & 4) ? " backlogged mMTC devices" : "");
            return RS_EOVERFLOW;
        }
    return RS_OK;
}

extern "C" int rs_fetch(rs_handle* h, int32_t* actions, float* obs, double* reward, int32_t* labels,
                        int32_t* violations) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t N = (size_t)h->cfg.n_envs, S = (size_t)h->n_slices;
    if (actions) HIPCHK(h, hipMemcpyAsync(actions, h->d_actions, sizeof(int32_t) * N * S, hipMemcpyDeviceToHost, h->stream));
    if (obs) HIPCHK(h, hipMemcpyAsync(obs, h->d_obs, sizeof(float) * N * h->n_vars, hipMemcpyDeviceToHost, h->stream));
    if (reward) HIPCHK(h, hipMemcpyAsync(reward, h->d_reward, sizeof(double) * N, hipMemcpyDeviceToHost, h->stream));
    if (labels) HIPCHK(h, hipMemcpyAsync(labels, h->d_labels, sizeof(int32_t) * N * S, hipMemcpyDeviceToHost, h->stream));
    if (violations) HIPCHK(h, hipMemcpyAsync(violations, h->d_viol, sizeof(int32_t) * N * S, hipMemcpyDeviceToHost, h->stream));
    return check_errors(h);
}

extern "C" int rs_step(rs_handle* h, const int32_t* actions, float* obs, double* reward, int32_t* labels,
                       int32_t* violations) {
    if (!h || !actions) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t N = (size_t)h->cfg.n_envs, S = (size_t)h->n_slices;
    size_t wide = 0;  // eMBB slices wide enough for block rounds of the PF allocation
    const int wide_prbs = RS_HINT_PAIRS * (h->cfg.pf_granularity > 0 ? h->cfg.pf_granularity : 1);
    for (size_t r = 0; r < N; ++r) {  // Q9: the reference silently mis-slices; the build rejects
        long tot = 0;
        for (size_t s = 0; s < S; ++s) {
            if (actions[r * S + s] < 0) {
                h->err = "rs_step: negative action";
                return RS_EINVAL;
            }
            tot += actions[r * S + s];
            wide += actions[r * S + s] >= wide_prbs;
        }
        if (tot > h->cfg.n_prbs) {
            h->err = "rs_step: sum(action) > n_prbs in replica " + std::to_string(r);
            return RS_EINVAL;
        }
    }
    if (h->hint_auto) {  // these allocations are in plain sight: pick the instance for them (a hint, same results)
        const int want = wide * 16 >= N ? 1 : 0;
        if (want != h->block_hint) {
            h->block_hint = want;
            drop_graph(h);
        }
    }
    HIPCHK(h, hipMemcpyAsync(h->d_actions, actions, sizeof(int32_t) * N * S, hipMemcpyHostToDevice, h->stream));
    int rc = launch_step(h);
    if (rc != RS_OK) return rc;
    return rs_fetch(h, nullptr, obs, reward, labels, violations);
}

extern "C" int rs_step_resident(rs_handle* h) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    return launch_step(h);
}

extern "C" int rs_random_actions(rs_handle* h, uint64_t seed, uint64_t step_index) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->n_slices > 8) {
        h->err = "rs_random_actions: at most 8 slices";
        return RS_EINVAL;
    }
    if (h->cfg.n_prbs > 512) return RS_EINVAL;
    hipLaunchKernelGGL(random_actions_kernel, dim3((h->cfg.n_envs + 3) / 4), dim3(256), 0, h->stream, h->ddev,
                       h->d_actions, seed, step_index, (const int64_t*)nullptr);
    HIPCHK(h, hipGetLastError());
    return RS_OK;
}

static void drop_graph(rs_handle* h) {
    h->launch_sig += 1;
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    h->gexec = nullptr;
    h->graph = nullptr;
}

// one step of the on-device action script: actions from (run[2], run[1]), then RanSlice.step
static int enqueue_scripted_step(rs_handle* h) {
    hipLaunchKernelGGL(random_actions_kernel, dim3((h->cfg.n_envs + 3) / 4), dim3(256), 0, h->stream, h->ddev,
                       h->d_actions, (uint64_t)0, (uint64_t)0, (const int64_t*)h->d_run);
    return launch_step(h);
}

// n_steps x (rs_random_actions(seed, step_index0 + i); rs_step_resident()) enqueued by one call.  With use_graph
// the loop body is captured once into a hipGraph of two consecutive steps (the order counters alternate
// between two buffers) and replayed; the slot clock and the action-script index live in device memory
// (d_run) so that the captured kernels need no per-step arguments.  Results are identical either way.
extern "C" int rs_run_random(rs_handle* h, uint64_t seed, uint64_t step_index0, int n_steps, int use_graph) {
    if (!h || n_steps < 0) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->n_slices > 8 || h->cfg.n_prbs > 512) {
        h->err = "rs_run_random: at most 8 slices and 512 PRBs";
        return RS_EINVAL;
    }
    if (!h->is_reset) {
        h->err = "rs_run_random: call rs_reset first";
        return RS_ESTATE;
    }
    if ((int64_t)h->clock + (int64_t)n_steps * h->cfg.slots_per_step > 2000000000) {
        h->err = "rs_run_random: slot clock would overflow; reset the environment";
        return RS_ESTATE;
    }
    hipLaunchKernelGGL(set_run_kernel, dim3(1), dim3(1), 0, h->stream, h->d_run, seed, step_index0);
    int done = 0, rc;
    if (use_graph && !h->timing && n_steps >= 2) {
        if (h->gexec && h->graph_par != h->order_par) {  // realign with the parity the graph was captured at
            if ((rc = enqueue_scripted_step(h)) != RS_OK) return rc;
            done += 1;
        }
        if (!h->gexec) {
            const int32_t clock0 = h->clock;
            const uint64_t steps0 = h->steps;
            h->graph_par = h->order_par;
            HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            const int rc1 = enqueue_scripted_step(h);
            const int rc2 = rc1 == RS_OK ? enqueue_scripted_step(h) : rc1;
            const hipError_t ec = hipStreamEndCapture(h->stream, &h->graph);
            h->clock = clock0;
            h->steps = steps0;
            h->order_par = h->graph_par;
            if (rc2 != RS_OK || ec != hipSuccess) {
                drop_graph(h);
                if (rc2 == RS_OK) h->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(ec);
                return rc2 != RS_OK ? rc2 : RS_EHIP;
            }
            HIPCHK(h, hipGraphInstantiate(&h->gexec, h->graph, nullptr, nullptr, 0));
        }
        while (n_steps - done >= 2) {
            HIPCHK(h, hipGraphLaunch(h->gexec, h->stream));
            h->clock += 2 * h->cfg.slots_per_step;
            h->steps += 2;
            done += 2;
        }
    }
    for (; done < n_steps; ++done)
        if ((rc = enqueue_scripted_step(h)) != RS_OK) return rc;
    return RS_OK;
}

extern "C" int rs_get_info(rs_handle* h, double* info) {
    if (!h || !info) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(info, h->d_info, sizeof(double) * (size_t)h->cfg.n_envs * h->n_ran * 10,
                             hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RS_OK;
}

extern "C" int rs_set_alloc_trace(rs_handle* h, int enable) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    if (enable && !h->d_trace) {
        size_t n = (size_t)h->n_tasks * h->cfg.slots_per_step * RS_GROUP;
        if (h->mux) n = (size_t)h->cfg.n_envs * h->cfg.slots_per_step * RS_MUX_UE;
        HIPCHK(h, hipMalloc((void**)&h->d_trace, sizeof(rs_alloc_rec) * (n ? n : 1)));
    }
    h->trace_on = enable != 0;
    drop_graph(h);
    return RS_OK;
}

extern "C" int rs_get_alloc_trace(rs_handle* h, rs_alloc_rec* out) {
    if (!h || !out || !h->d_trace) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    size_t n = (size_t)h->n_tasks * h->cfg.slots_per_step * RS_GROUP;
    if (h->mux) n = (size_t)h->cfg.n_envs * h->cfg.slots_per_step * RS_MUX_UE;  // [n_envs][slots][64]
    HIPCHK(h, hipMemcpyAsync(out, h->d_trace, sizeof(rs_alloc_rec) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RS_OK;
}

extern "C" int rs_get_counters(rs_handle* h, uint64_t counters[4]) {
    if (!h || !counters) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemsetAsync(h->d_counter_sum, 0, sizeof(uint64_t) * 4, h->stream));
    if (h->n_tasks > 0)
        hipLaunchKernelGGL(counter_sum_kernel, dim3(64), dim3(256), 0, h->stream, h->d_counters, h->n_tasks,
                           h->d_counter_sum);
    HIPCHK(h, hipMemcpyAsync(counters, h->d_counter_sum, sizeof(uint64_t) * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    counters[1] = h->steps * (uint64_t)h->cfg.n_envs;
    return RS_OK;
}

extern "C" int rs_get_rx_stats(rs_handle* h, uint64_t out[3]) {
    if (!h || !out) return RS_EINVAL;
    uint64_t c[4];
    const int rc = rs_get_counters(h, c);
    if (rc != RS_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(c, h->d_counter_sum, sizeof(uint64_t) * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    out[0] = c[1] >> 32;
    out[1] = c[1] & 0xffffffffull;
    out[2] = h->hdev.rx_band > 0.0 ? 1u : 0u;
    return RS_OK;
}

// mode 1: allocations come from a learning agent, which concentrates the carrier on few wide slices: the 16-lane step
// uses its BLOCK instance (block rounds of the contested PF allocation, rs_embb.hip).  mode 0: the plain instance
// (trip loop only).  mode < 0: automatic (auto_hint; rs_step goes by the allocations it is handed, kb_step_resident
// switches BLOCK on for the environment it drives).  A scheduling hint only: results are identical.
extern "C" int rs_set_schedule_hint(rs_handle* h, int mode) {
    if (!h) return RS_EINVAL;
    h->hint_auto = mode < 0;
    h->block_hint = mode < 0 ? auto_hint(h) : (mode ? 1 : 0);
    drop_graph(h);
    return RS_OK;
}

extern "C" int rs_set_group_size(rs_handle* h, int lanes) {
    if (!h || (lanes != 8 && lanes != 16 && lanes != 32)) return RS_EINVAL;
    h->group = lanes;
    drop_graph(h);
    return RS_OK;
}

// cycle sums per code section of embb_step_kernel; all zero unless built with -DRS_SECTION_PROFILE
extern "C" int rs_get_section_profile(rs_handle* h, uint64_t out[16]) {
    if (!h || !out) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(out, h->d_sections, sizeof(uint64_t) * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RS_OK;
}

extern "C" int rs_get_task_profile(rs_handle* h, uint64_t* out) {
    if (!h || !out) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    // [n_tasks][4] per-task records followed by the 16 section sums of the slowest wave seen so far
    HIPCHK(h, hipMemcpyAsync(out, h->d_sections + 16, sizeof(uint64_t) * (4 * (size_t)h->n_tasks + 16),
                             hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RS_OK;
}

extern "C" int rs_set_kernel_timing(rs_handle* h, int enable) {
    if (!h) return RS_EINVAL;
    h->timing = enable != 0;
    h->ev_used = 0;
    return RS_OK;
}

extern "C" int rs_kernel_time_stats_ms(rs_handle* h, double out[3], int64_t* launches) {
    if (!h || !out) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double tot = 0.0, mn = 0.0, mx = 0.0;
    for (size_t i = 0; i < h->ev_used; ++i) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev[i].first, h->ev[i].second));
        tot += ms;
        mn = (i == 0 || ms < mn) ? ms : mn;
        mx = ms > mx ? ms : mx;
    }
    out[0] = h->ev_used ? tot / (double)h->ev_used : 0.0;
    out[1] = mn;
    out[2] = mx;
    if (launches) *launches = (int64_t)h->ev_used;
    h->ev_used = 0;
    return RS_OK;
}

extern "C" int rs_kernel_time_ms(rs_handle* h, double* avg_ms, int64_t* launches) {
    if (!h || !avg_ms) return RS_EINVAL;
    double st[3];
    const int rc = rs_kernel_time_stats_ms(h, st, launches);
    if (rc == RS_OK) *avg_ms = st[0];
    return rc;
}

extern "C" int rs_synchronize(rs_handle* h) {
    if (!h) return RS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RS_OK;
}

// ------------------------------------------------------------------ checkpoint / restore (the reference has
// none -- a run of experiments_kbrl.py that dies starts over; here a 50,400-step evaluation can be cut and resumed)
// The state of a handle is every device array behind it (the structure-of-arrays simulator state, the outputs of the last
// step, the counters, the launch-order scratch) plus a few host words; the tables (fading traces, constants) are inputs and
// are not saved.  A blob only fits a handle of the same configuration (checked).
struct rs_state_header {
    uint64_t magic, n_regions, total_bytes, cfg_hash;
    int64_t clock, steps;
    int32_t order_par, graph_par, block_hint, hint_auto, is_reset, pad;
};
static const uint64_t kRsStateMagic = 0x52534c4943453034ull;  // "RSLICE04"
static uint64_t rs_cfg_hash(const rs_handle* h) {
    uint64_t x = 1469598103934665603ull;
    const unsigned char* p = (const unsigned char*)&h->cfg;
    for (size_t i = 0; i < sizeof h->cfg; ++i) x = (x ^ p[i]) * 1099511628211ull;
    for (auto& r : h->regions) x = (x ^ (uint64_t)r.second) * 1099511628211ull;
    return x;
}
extern "C" int rs_state_bytes(rs_handle* h, uint64_t* bytes) {
    if (!h || !bytes) return RS_EINVAL;
    uint64_t t = sizeof(rs_state_header);
    for (auto& r : h->regions) t += r.second;
    *bytes = t;
    return RS_OK;
}
extern "C" int rs_save_state(rs_handle* h, void* blob, uint64_t bytes) {
    uint64_t need = 0;
    if (!h || !blob || rs_state_bytes(h, &need) != RS_OK) return RS_EINVAL;
    if (bytes < need) {
        h->err = "rs_save_state: buffer smaller than rs_state_bytes";
        return RS_EINVAL;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->side) HIPCHK(h, hipStreamSynchronize(h->side));
    if (h->side2) HIPCHK(h, hipStreamSynchronize(h->side2));
    if (h->side3) HIPCHK(h, hipStreamSynchronize(h->side3));
    rs_state_header hd = {kRsStateMagic, (uint64_t)h->regions.size(), need, rs_cfg_hash(h), (int64_t)h->clock, (int64_t)h->steps,
                          h->order_par, h->graph_par, h->block_hint, h->hint_auto ? 1 : 0, h->is_reset ? 1 : 0, 0};
    memcpy(blob, &hd, sizeof hd);
    char* o = (char*)blob + sizeof hd;
    for (auto& r : h->regions) {
        HIPCHK(h, hipMemcpy(o, r.first, r.second, hipMemcpyDeviceToHost));
        o += r.second;
    }
    return RS_OK;
}
extern "C" int rs_load_state(rs_handle* h, const void* blob, uint64_t bytes) {
    uint64_t need = 0;
    if (!h || !blob || rs_state_bytes(h, &need) != RS_OK) return RS_EINVAL;
    rs_state_header hd;
    if (bytes < sizeof hd) return RS_EINVAL;
    memcpy(&hd, blob, sizeof hd);
    if (hd.magic != kRsStateMagic || hd.n_regions != h->regions.size() || hd.total_bytes != need || bytes < need ||
        hd.cfg_hash != rs_cfg_hash(h)) {
        h->err = "rs_load_state: the blob was not saved by a handle of this configuration";
        return RS_EINVAL;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->side) HIPCHK(h, hipStreamSynchronize(h->side));
    if (h->side2) HIPCHK(h, hipStreamSynchronize(h->side2));
    if (h->side3) HIPCHK(h, hipStreamSynchronize(h->side3));
    drop_graph(h);
    const char* o = (const char*)blob + sizeof hd;
    for (auto& r : h->regions) {
        if (r.first != (void*)h->ddev && r.first != (void*)h->d_st)  // (constants and the pointer table belong to THIS handle)
            HIPCHK(h, hipMemcpy(r.first, o, r.second, hipMemcpyHostToDevice));
        o += r.second;
    }
    h->clock = (int32_t)hd.clock;
    h->steps = (uint64_t)hd.steps;
    h->order_par = hd.order_par;
    h->graph_par = hd.graph_par;
    h->graph_sig = -1;
    h->block_hint = hd.block_hint;
    h->hint_auto = hd.hint_auto != 0;
    h->is_reset = hd.is_reset != 0;
    return RS_OK;
}
#include "kb_api.hip"

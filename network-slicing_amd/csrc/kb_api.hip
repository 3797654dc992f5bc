// kb_api.hip -- host side of the KBRL C ABI (include/ranslice.h, kb_*).  Included by rs_api.hip so
// that kb_step_resident can read the simulator's device buffers.
#pragma once
#include "kb_kbrl.hip"

struct kb_handle {
    kb_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_order = nullptr;  // orders the agent's stream against the simulator's (kb_step_resident)
    kb::KbDev D;
    kb::KbState K;
    std::vector<void*> allocs;
    std::vector<GuardedAlloc> guarded;
    float* d_state = nullptr;      // staging for host-provided states
    float* d_prev_state = nullptr; // resident loop: obs the executed action was chosen in
    int32_t* d_action = nullptr;
    int32_t* d_labels = nullptr;
    int32_t* d_hits = nullptr;
    double* d_out = nullptr;  // [4]
    int32_t* d_cursor = nullptr;   // shared mode [T]
    int32_t* d_cstar = nullptr;    // shared mode [T]
    double* d_props = nullptr;     // shared mode [S][budget_cap][KB_PROP_W]
    int32_t* d_counts = nullptr;   // [S]
    uint64_t* d_gstats = nullptr;  // [4] shared-dictionary updates
    int budget_cap = 256;
    int n_dict = 0;
    int T = 0, nv = 0;
    bool is_reset = false;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t ev_used = 0;
    std::string err;
};

template <class Tp>
static int kalloc(kb_handle* k, Tp** p, size_t n, bool zero = true) {
    void* q = nullptr;
    size_t bytes = sizeof(Tp) * (n ? n : 1);
    HIPCHK(k, guarded_malloc(&q, bytes, &k->guarded));
    if (zero) HIPCHK(k, hipMemsetAsync(q, 0, bytes, k->stream));
    if (!guards_on()) k->allocs.push_back(q);
    *p = (Tp*)q;
    return RS_OK;
}

extern "C" int kb_create(const kb_config* cfg, int device, kb_handle** out) {
    if (!cfg || !out) return RS_EINVAL;
    kb_handle* k = new kb_handle();
    *out = k;
    k->cfg = *cfg;
    k->device = device;
    if (cfg->n_envs <= 0 || cfg->n_slices <= 0 || cfg->n_slices > KB_MAX_SLICES || cfg->n_prbs <= 0 ||
        cfg->n_prbs > 256 || cfg->capacity < 2 || cfg->capacity > 1024) {
        k->err = "kb_create: unsupported configuration (<= 8 learners, n_prbs <= 256, 2 <= capacity <= 1024)";
        return RS_EINVAL;
    }
    int ndev = 0;
    HIPCHK(k, hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        k->err = "kb_create: no such HIP device";
        return RS_EHIP;
    }
    HIPCHK(k, hipSetDevice(device));
    HIPCHK(k, hipStreamCreateWithFlags(&k->stream, hipStreamNonBlocking));
    kb::KbDev& D = k->D;
    memset(&D, 0, sizeof D);
    D.n_envs = cfg->n_envs;
    D.S = cfg->n_slices;
    D.n_prbs = cfg->n_prbs;
    D.cap = cfg->capacity;
    int o = 0;
    for (int s = 0; s < cfg->n_slices; ++s) {
        if (cfg->dims[s] <= 0 || cfg->dims[s] + 1 > KB_DMAX) {
            k->err = "kb_create: learner dimension out of range";
            return RS_EINVAL;
        }
        D.dims[s] = cfg->dims[s];
        D.off[s] = o;
        o += cfg->dims[s];
    }
    D.nv = o;
    D.alfa = cfg->alfa;
    D.lo = cfg->acc_lo;
    D.hi = cfg->acc_hi;
    D.gamma = cfg->gamma;
    D.eta = cfg->eta;
    D.shared = cfg->shared_dictionary ? 1 : 0;
    D.first_env = cfg->first_env;
    k->nv = o;
    k->T = cfg->n_envs * cfg->n_slices;
    const size_t T = (size_t)k->T, N = (size_t)cfg->n_envs, cap = (size_t)cfg->capacity;
    const size_t ND = D.shared ? (size_t)cfg->n_slices : T;  // dictionaries
    k->n_dict = (int)ND;
    int rc;
    kb::KbState& K = k->K;
#define KA(p, n, z) if ((rc = kalloc(k, &(p), (n), (z))) != RS_OK) return rc
    KA(K.m, ND, true);
    KA(K.L, ND * KB_DMAX * cap, true);
    KA(K.coeff, ND * cap, true);
    KA(K.Kinv, ND * cap * cap, false);  // entries are written before they are read
    KA(K.kf, T * cap, true);
    KA(K.f_last, T, true);
    KA(K.m_last, T, true);
    KA(K.tie_ctr, T, true);
    KA(K.seeds, N, true);
    KA(K.action, T, true);
    KA(K.security, T, true);
    KA(K.margins, T, true);
    KA(K.adjusted, N, true);
    KA(K.acc, T * (size_t)cfg->n_prbs, true);
    KA(K.err, N, true);
    KA(K.stats, T * 4, true);
    KA(k->d_state, N * (size_t)k->nv, true);
    KA(k->d_prev_state, N * (size_t)k->nv, true);
    KA(k->d_action, T, true);
    KA(k->d_labels, T, true);
    KA(k->d_hits, T, true);
    KA(k->d_out, 4, true);
    KA(k->d_cursor, T, true);
    KA(k->d_cstar, T, true);
    KA(k->d_props, (size_t)cfg->n_slices * k->budget_cap * KB_PROP_W, true);
    KA(k->d_counts, (size_t)cfg->n_slices, true);
    KA(k->d_gstats, 4, true);
#undef KA
    HIPCHK(k, hipStreamSynchronize(k->stream));
    return RS_OK;
}

extern "C" void kb_destroy(kb_handle* k) {
    if (!k) return;
    if (k->stream) (void)hipStreamSynchronize(k->stream);
    if (guards_on()) check_guards(k->guarded, "kb");
    for (auto& g : k->guarded) (void)hipFree(g.base);
    for (void* p : k->allocs) (void)hipFree(p);
    for (auto& e : k->ev) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (k->ev_order) (void)hipEventDestroy(k->ev_order);
    if (k->stream) (void)hipStreamDestroy(k->stream);
    delete k;
}

extern "C" const char* kb_last_error(const kb_handle* k) { return k ? k->err.c_str() : "null handle"; }

extern "C" int kb_reset(kb_handle* k, const int32_t* initial_action, const int32_t* security_factor,
                        const uint64_t* seeds) {
    if (!k || !initial_action || !security_factor || !seeds) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    const size_t T = (size_t)k->T, N = (size_t)k->cfg.n_envs;
    uint64_t* dseed = nullptr;
    HIPCHK(k, hipMalloc((void**)&dseed, sizeof(uint64_t) * N));
    HIPCHK(k, hipMemcpyAsync(k->d_action, initial_action, sizeof(int32_t) * T, hipMemcpyHostToDevice, k->stream));
    HIPCHK(k, hipMemcpyAsync(k->d_labels, security_factor, sizeof(int32_t) * T, hipMemcpyHostToDevice, k->stream));
    HIPCHK(k, hipMemcpyAsync(dseed, seeds, sizeof(uint64_t) * N, hipMemcpyHostToDevice, k->stream));
    size_t n = T > N ? T : N;
    hipLaunchKernelGGL(kb::kb_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, k->stream, k->D, k->K,
                       k->d_action, k->d_labels, dseed);
    HIPCHK(k, hipMemsetAsync(k->d_prev_state, 0, sizeof(float) * N * k->nv, k->stream));
    HIPCHK(k, hipMemsetAsync(k->K.m, 0, sizeof(int32_t) * (size_t)k->n_dict, k->stream));
    HIPCHK(k, hipMemsetAsync(k->d_gstats, 0, sizeof(uint64_t) * 4, k->stream));
    HIPCHK(k, hipGetLastError());
    HIPCHK(k, hipStreamSynchronize(k->stream));
    (void)hipFree(dseed);
    k->is_reset = true;
    return RS_OK;
}

static int kb_check(kb_handle* k) {
    std::vector<int32_t> e((size_t)k->cfg.n_envs);
    HIPCHK(k, hipMemcpyAsync(e.data(), k->K.err, sizeof(int32_t) * e.size(), hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    for (size_t i = 0; i < e.size(); ++i)
        if (e[i] & ~8) {  // bit 8 = a dictionary is saturated (projects instead of growing): reported, not an error
            k->err = "KBRL agent " + std::to_string(i) + ": internal error flag " + std::to_string(e[i]);
            return RS_EOVERFLOW;
        }
    return RS_OK;
}

static int kb_time_begin(kb_handle* k, hipEvent_t* e1) {
    *e1 = nullptr;
    if (!k->timing) return RS_OK;
    if (k->ev_used == k->ev.size()) {
        hipEvent_t a0, a1;
        HIPCHK(k, hipEventCreate(&a0));
        HIPCHK(k, hipEventCreate(&a1));
        k->ev.emplace_back(a0, a1);
    }
    HIPCHK(k, hipEventRecord(k->ev[k->ev_used].first, k->stream));
    *e1 = k->ev[k->ev_used].second;
    k->ev_used++;
    return RS_OK;
}

static int launch_update_control(kb_handle* k, const float* d_state, const int32_t* d_action, const int32_t* d_labels) {
    kb::CtlArgs a;
    a.D = k->D;
    a.K = k->K;
    a.state = d_state;
    a.action = d_action;
    a.labels = d_labels;
    a.hits = k->d_hits;
    hipEvent_t e1;
    int rc = kb_time_begin(k, &e1);
    if (rc != RS_OK) return rc;
    hipLaunchKernelGGL(kb::update_control_kernel, dim3((unsigned)k->T), dim3(256), kb::kb_lds_bytes(k->cfg.capacity), k->stream, a);
    if (e1) HIPCHK(k, hipEventRecord(e1, k->stream));
    return RS_OK;
}

static int launch_select(kb_handle* k, const float* d_state, int32_t* d_action_out) {
    kb::SelArgs a;
    a.D = k->D;
    a.K = k->K;
    a.state = d_state;
    hipEvent_t e1;
    int rc = kb_time_begin(k, &e1);
    if (rc != RS_OK) return rc;
    hipLaunchKernelGGL(kb::select_kernel, dim3((unsigned)k->T), dim3(256), kb::kb_lds_bytes(k->cfg.capacity), k->stream, a);
    if (e1) HIPCHK(k, hipEventRecord(e1, k->stream));
    hipLaunchKernelGGL(kb::adjust_kernel, dim3((unsigned)((k->cfg.n_envs + 255) / 256)), dim3(256), 0, k->stream, k->D,
                       k->K, d_action_out);
    return RS_OK;
}

extern "C" int kb_update_control(kb_handle* k, const float* state, const int32_t* action, const int32_t* labels,
                                 int32_t* hits) {
    if (!k || !state || !action || !labels) return RS_EINVAL;
    if (!k->is_reset) {
        k->err = "kb_update_control: call kb_reset first";
        return RS_ESTATE;
    }
    if (k->D.shared) {
        k->err = "kb_update_control: shared-dictionary handles learn through kb_shared_scan/apply/commit";
        return RS_ESTATE;
    }
    HIPCHK(k, hipSetDevice(k->device));
    const size_t T = (size_t)k->T, N = (size_t)k->cfg.n_envs;
    for (size_t i = 0; i < T; ++i)
        if (action[i] < 0 || action[i] > k->cfg.n_prbs || (labels[i] != 1 && labels[i] != -1)) {
            k->err = "kb_update_control: action out of [0, n_prbs] or label not +-1";
            return RS_EINVAL;
        }
    HIPCHK(k, hipMemcpyAsync(k->d_state, state, sizeof(float) * N * k->nv, hipMemcpyHostToDevice, k->stream));
    HIPCHK(k, hipMemcpyAsync(k->d_action, action, sizeof(int32_t) * T, hipMemcpyHostToDevice, k->stream));
    HIPCHK(k, hipMemcpyAsync(k->d_labels, labels, sizeof(int32_t) * T, hipMemcpyHostToDevice, k->stream));
    int rc = launch_update_control(k, k->d_state, k->d_action, k->d_labels);
    if (rc != RS_OK) return rc;
    HIPCHK(k, hipGetLastError());
    if (hits) HIPCHK(k, hipMemcpyAsync(hits, k->d_hits, sizeof(int32_t) * T, hipMemcpyDeviceToHost, k->stream));
    return kb_check(k);
}

extern "C" int kb_select_action(kb_handle* k, const float* state, int32_t* action, int32_t* adjusted) {
    if (!k || !state) return RS_EINVAL;
    if (!k->is_reset) {
        k->err = "kb_select_action: call kb_reset first";
        return RS_ESTATE;
    }
    HIPCHK(k, hipSetDevice(k->device));
    const size_t T = (size_t)k->T, N = (size_t)k->cfg.n_envs;
    HIPCHK(k, hipMemcpyAsync(k->d_state, state, sizeof(float) * N * k->nv, hipMemcpyHostToDevice, k->stream));
    int rc = launch_select(k, k->d_state, nullptr);
    if (rc != RS_OK) return rc;
    HIPCHK(k, hipGetLastError());
    if (action) HIPCHK(k, hipMemcpyAsync(action, k->K.action, sizeof(int32_t) * T, hipMemcpyDeviceToHost, k->stream));
    if (adjusted) HIPCHK(k, hipMemcpyAsync(adjusted, k->K.adjusted, sizeof(int32_t) * N, hipMemcpyDeviceToHost, k->stream));
    return kb_check(k);
}

extern "C" int kb_step_resident(kb_handle* k, rs_handle* env) {
    if (!k || !env) return RS_EINVAL;
    if (!k->is_reset || env->cfg.n_envs != k->cfg.n_envs || env->n_slices != k->cfg.n_slices || env->n_vars != k->nv ||
        env->device != k->device) {
        k->err = "kb_step_resident: agent and environment do not match (or kb_reset missing)";
        return RS_EINVAL;
    }
    HIPCHK(k, hipSetDevice(k->device));
    if (env->grant_auto && !env->grant_mode) {  // the next steps take agent-made allocations
        rs_set_schedule_hint(env, 1);
        env->grant_auto = true;
    }
    // order after the simulator's step on its own stream
    if (!k->ev_order) HIPCHK(k, hipEventCreateWithFlags(&k->ev_order, hipEventDisableTiming));
    hipEvent_t done = k->ev_order;
    HIPCHK(k, hipEventRecord(done, env->stream));
    HIPCHK(k, hipStreamWaitEvent(k->stream, done, 0));
    int rc = launch_update_control(k, k->d_prev_state, env->d_actions, env->d_labels);
    if (rc != RS_OK) return rc;
    rc = launch_select(k, env->d_obs, env->d_actions);
    if (rc != RS_OK) return rc;
    HIPCHK(k, hipMemcpyAsync(k->d_prev_state, env->d_obs, sizeof(float) * (size_t)k->cfg.n_envs * k->nv,
                             hipMemcpyDeviceToDevice, k->stream));
    HIPCHK(k, hipEventRecord(done, k->stream));
    HIPCHK(k, hipStreamWaitEvent(env->stream, done, 0));
    HIPCHK(k, hipGetLastError());
    return RS_OK;
}

static int kb_one(kb_handle* k, int e, int s, const double* x, int y, bool update, double out[4]) {
    if (!k || !x || e < 0 || e >= k->cfg.n_envs || s < 0 || s >= k->cfg.n_slices) return RS_EINVAL;
    if (!k->is_reset) {
        k->err = "kb_predict/kb_update: call kb_reset first";
        return RS_ESTATE;
    }
    HIPCHK(k, hipSetDevice(k->device));
    kb::OneArgs a;
    a.D = k->D;
    a.K = k->K;
    a.task = e * k->cfg.n_slices + s;
    a.y = y;
    memset(a.x, 0, sizeof a.x);
    for (int q = 0; q < k->cfg.dims[s] + 1; ++q) a.x[q] = x[q];
    a.out = k->d_out;
    if (update)
        hipLaunchKernelGGL(kb::update_one_kernel, dim3(1), dim3(256), kb::kb_lds_bytes(k->cfg.capacity), k->stream, a);
    else
        hipLaunchKernelGGL(kb::predict_one_kernel, dim3(1), dim3(256), kb::kb_lds_bytes(k->cfg.capacity), k->stream, a);
    HIPCHK(k, hipGetLastError());
    HIPCHK(k, hipMemcpyAsync(out, k->d_out, sizeof(double) * 4, hipMemcpyDeviceToHost, k->stream));
    return kb_check(k);
}

extern "C" int kb_predict(kb_handle* k, int e, int s, const double* x, int32_t* y_pred, double* f) {
    double out[4] = {0, 0, 0, 0};
    int rc = kb_one(k, e, s, x, 0, false, out);
    if (rc != RS_OK) return rc;
    if (y_pred) *y_pred = (int32_t)out[0];
    if (f) *f = out[1];
    return RS_OK;
}

extern "C" int kb_update(kb_handle* k, int e, int s, const double* x, int32_t y, int32_t* branch, double* delta) {
    if (y != 1 && y != -1) return RS_EINVAL;
    double out[4] = {0, 0, 0, 0};
    int rc = kb_one(k, e, s, x, y, true, out);
    if (rc != RS_OK) return rc;
    if (out[2] < 0.0) {
        k->err = "kb_update: the dictionary changed since the kb_predict whose (f, K_f) this update would use "
                 "(projectron.py:40-42 caches them; update_control / select_action grew the shared dictionary)";
        return RS_ESTATE;
    }
    if (branch) *branch = (int32_t)out[2];
    if (delta) *delta = out[3];
    return RS_OK;
}

extern "C" int kb_get_learner(kb_handle* k, int e, int s, int32_t* m_out, double* landmarks, double* coeff, double* kinv) {
    if (!k || e < 0 || e >= k->cfg.n_envs || s < 0 || s >= k->cfg.n_slices) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    const size_t task = k->D.shared ? (size_t)s : (size_t)e * k->cfg.n_slices + s, cap = (size_t)k->cfg.capacity;
    int32_t m = 0;
    HIPCHK(k, hipMemcpyAsync(&m, k->K.m + task, sizeof m, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    if (m_out) *m_out = m;
    const int d = k->cfg.dims[s] + 1;
    if (landmarks && m > 0) {
        std::vector<double> tmp((size_t)KB_DMAX * cap);
        HIPCHK(k, hipMemcpyAsync(tmp.data(), k->K.L + task * KB_DMAX * cap, sizeof(double) * tmp.size(),
                                 hipMemcpyDeviceToHost, k->stream));
        HIPCHK(k, hipStreamSynchronize(k->stream));
        for (int j = 0; j < m; ++j)
            for (int q = 0; q < d; ++q) landmarks[(size_t)j * d + q] = tmp[(size_t)q * cap + j];
    }
    if (coeff && m > 0)
        HIPCHK(k, hipMemcpyAsync(coeff, k->K.coeff + task * cap, sizeof(double) * m, hipMemcpyDeviceToHost, k->stream));
    if (kinv && m > 0)
        HIPCHK(k, hipMemcpy2DAsync(kinv, sizeof(double) * m, k->K.Kinv + task * cap * cap, sizeof(double) * cap,
                                   sizeof(double) * m, m, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    return RS_OK;
}

extern "C" int kb_get_control(kb_handle* k, int32_t* margins, int32_t* security, int32_t* action, int32_t* adjusted,
                              double* accuracies) {
    if (!k) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    const size_t T = (size_t)k->T, N = (size_t)k->cfg.n_envs;
    if (margins) HIPCHK(k, hipMemcpyAsync(margins, k->K.margins, sizeof(int32_t) * T, hipMemcpyDeviceToHost, k->stream));
    if (security) HIPCHK(k, hipMemcpyAsync(security, k->K.security, sizeof(int32_t) * T, hipMemcpyDeviceToHost, k->stream));
    if (action) HIPCHK(k, hipMemcpyAsync(action, k->K.action, sizeof(int32_t) * T, hipMemcpyDeviceToHost, k->stream));
    if (adjusted) HIPCHK(k, hipMemcpyAsync(adjusted, k->K.adjusted, sizeof(int32_t) * N, hipMemcpyDeviceToHost, k->stream));
    if (accuracies)
        HIPCHK(k, hipMemcpyAsync(accuracies, k->K.acc, sizeof(double) * T * k->cfg.n_prbs, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    return RS_OK;
}

extern "C" int kb_set_adjusted(kb_handle* k, const int32_t* adjusted) {
    if (!k || !adjusted) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    HIPCHK(k, hipMemcpyAsync(k->K.adjusted, adjusted, sizeof(int32_t) * (size_t)k->cfg.n_envs, hipMemcpyHostToDevice,
                             k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    return RS_OK;
}

extern "C" int kb_get_stats(kb_handle* k, uint64_t stats[4]) {
    if (!k || !stats) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    std::vector<uint64_t> tmp((size_t)k->T * 4);
    HIPCHK(k, hipMemcpyAsync(tmp.data(), k->K.stats, sizeof(uint64_t) * tmp.size(), hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    for (int q = 0; q < 4; ++q) stats[q] = 0;
    for (size_t i = 0; i < (size_t)k->T; ++i)
        for (int q = 0; q < 4; ++q) stats[q] += tmp[i * 4 + q];
    uint64_t g[4] = {0, 0, 0, 0};
    HIPCHK(k, hipMemcpy(g, k->d_gstats, sizeof g, hipMemcpyDeviceToHost));
    stats[1] += g[1];
    stats[2] += g[2];
    return RS_OK;
}

// GaussianKernel.k(x) of the last kb_predict on learner (e, s) (kernel.py:13-20): the cached row K_f, m entries
extern "C" int kb_get_kernel_row(kb_handle* k, int e, int s, int32_t* m_out, double* row) {
    if (!k || e < 0 || e >= k->cfg.n_envs || s < 0 || s >= k->cfg.n_slices) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    const size_t task = (size_t)e * k->cfg.n_slices + s, cap = (size_t)k->cfg.capacity;
    int32_t m = 0;
    HIPCHK(k, hipMemcpyAsync(&m, k->K.m_last + task, sizeof m, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    if (m_out) *m_out = m;
    if (row && m > 0) {
        HIPCHK(k, hipMemcpyAsync(row, k->K.kf + task * cap, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, k->stream));
        HIPCHK(k, hipStreamSynchronize(k->stream));
    }
    return RS_OK;
}

// landmarks held by every dictionary: [n_envs][S] (per-replica agents) or [S] (shared dictionaries)
extern "C" int kb_get_sizes(kb_handle* k, int32_t* m_out) {
    if (!k || !m_out) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    HIPCHK(k, hipMemcpyAsync(m_out, k->K.m, sizeof(int32_t) * (size_t)k->n_dict, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    return RS_OK;
}

extern "C" int kb_set_kernel_timing(kb_handle* k, int enable) {
    if (!k) return RS_EINVAL;
    k->timing = enable != 0;
    k->ev_used = 0;
    return RS_OK;
}

extern "C" int kb_kernel_time_ms(kb_handle* k, double* avg_ms, int64_t* launches) {
    if (!k || !avg_ms) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    double tot = 0.0;
    for (size_t i = 0; i < k->ev_used; ++i) {
        float ms = 0.f;
        HIPCHK(k, hipEventElapsedTime(&ms, k->ev[i].first, k->ev[i].second));
        tot += ms;
    }
    *avg_ms = k->ev_used ? tot / (double)k->ev_used : 0.0;
    if (launches) *launches = (int64_t)k->ev_used;
    k->ev_used = 0;
    return RS_OK;
}

// Waits for the agent's stream and reports a dictionary overflow raised by any kernel since kb_reset -- the resident
// loop (kb_step_resident) never reads the flag itself, so this is where a device-driven run learns about it.
extern "C" int kb_synchronize(kb_handle* k) {
    if (!k) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    return kb_check(k);
}


// ------------------------------------------------------------------ shared-dictionary mode

extern "C" int kb_shared_scan(kb_handle* k, const float* state, const int32_t* action, const int32_t* labels,
                              int32_t round, int32_t budget, int32_t* hits, int32_t* counts, double* props) {
    if (!k || !counts || !props || budget <= 0) return RS_EINVAL;
    if (!k->D.shared || !k->is_reset) {
        k->err = "kb_shared_scan: handle is not a reset shared-dictionary agent";
        return RS_ESTATE;
    }
    if (budget > k->budget_cap) {
        k->err = "kb_shared_scan: budget too large (<= 256)";
        return RS_EINVAL;
    }
    HIPCHK(k, hipSetDevice(k->device));
    const size_t T = (size_t)k->T, N = (size_t)k->cfg.n_envs, S = (size_t)k->cfg.n_slices;
    if (round == 0) {
        if (!state || !action || !labels) return RS_EINVAL;
        for (size_t i = 0; i < T; ++i)
            if (action[i] < 0 || action[i] > k->cfg.n_prbs || (labels[i] != 1 && labels[i] != -1)) {
                k->err = "kb_shared_scan: action out of [0, n_prbs] or label not +-1";
                return RS_EINVAL;
            }
        HIPCHK(k, hipMemcpyAsync(k->d_state, state, sizeof(float) * N * k->nv, hipMemcpyHostToDevice, k->stream));
        HIPCHK(k, hipMemcpyAsync(k->d_action, action, sizeof(int32_t) * T, hipMemcpyHostToDevice, k->stream));
        HIPCHK(k, hipMemcpyAsync(k->d_labels, labels, sizeof(int32_t) * T, hipMemcpyHostToDevice, k->stream));
    }
    kb::ScanArgs a;
    a.D = k->D;
    a.K = k->K;
    a.state = k->d_state;
    a.action = k->d_action;
    a.labels = k->d_labels;
    a.hits = k->d_hits;
    a.cursor = k->d_cursor;
    a.cstar = k->d_cstar;
    a.round = round;
    hipEvent_t e1;
    int rc = kb_time_begin(k, &e1);
    if (rc != RS_OK) return rc;
    hipLaunchKernelGGL(kb::shared_scan_kernel, dim3((unsigned)k->T), dim3(256), kb::kb_lds_bytes(k->cfg.capacity), k->stream, a);
    if (e1) HIPCHK(k, hipEventRecord(e1, k->stream));
    hipLaunchKernelGGL(kb::shared_collect_kernel, dim3((unsigned)S), dim3(64), 0, k->stream, k->D, k->d_state, k->d_labels,
                       k->d_cstar, (int)budget, k->d_props, k->d_counts);
    HIPCHK(k, hipGetLastError());
    if (hits && round == 0)
        HIPCHK(k, hipMemcpyAsync(hits, k->d_hits, sizeof(int32_t) * T, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipMemcpyAsync(counts, k->d_counts, sizeof(int32_t) * S, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipMemcpyAsync(props, k->d_props, sizeof(double) * S * budget * KB_PROP_W, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    return RS_OK;
}

extern "C" int kb_shared_apply(kb_handle* k, const int32_t* counts, const double* props, int32_t budget) {
    if (!k || !counts || !props || budget <= 0 || budget > k->budget_cap) return RS_EINVAL;
    if (!k->D.shared || !k->is_reset) {
        k->err = "kb_shared_apply: handle is not a reset shared-dictionary agent";
        return RS_ESTATE;
    }
    HIPCHK(k, hipSetDevice(k->device));
    const size_t S = (size_t)k->cfg.n_slices;
    HIPCHK(k, hipMemcpyAsync(k->d_counts, counts, sizeof(int32_t) * S, hipMemcpyHostToDevice, k->stream));
    HIPCHK(k, hipMemcpyAsync(k->d_props, props, sizeof(double) * S * budget * KB_PROP_W, hipMemcpyHostToDevice, k->stream));
    hipLaunchKernelGGL(kb::shared_apply_kernel, dim3((unsigned)S), dim3(1024), kb::kb_lds_bytes(k->cfg.capacity), k->stream, k->D, k->K, k->d_props,
                       k->d_counts, (int)budget, k->d_gstats);
    HIPCHK(k, hipGetLastError());
    return kb_check(k);
}

extern "C" int kb_shared_commit(kb_handle* k, const int32_t* n_accept) {
    if (!k || !n_accept) return RS_EINVAL;
    if (!k->D.shared || !k->is_reset) return RS_ESTATE;
    HIPCHK(k, hipSetDevice(k->device));
    const size_t S = (size_t)k->cfg.n_slices;
    HIPCHK(k, hipMemcpyAsync(k->d_counts, n_accept, sizeof(int32_t) * S, hipMemcpyHostToDevice, k->stream));
    hipLaunchKernelGGL(kb::shared_commit_kernel, dim3((unsigned)S), dim3(64), 0, k->stream, k->D, k->d_cstar, k->d_counts,
                       k->d_cursor);
    HIPCHK(k, hipGetLastError());
    HIPCHK(k, hipStreamSynchronize(k->stream));
    return RS_OK;
}

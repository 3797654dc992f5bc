// kb_api.hip -- host side of the KBRL C ABI (include/ranslice.h, kb_*).  Included by rs_api.hip so
// that kb_step_resident can read the simulator's device buffers.
#pragma once
#include "kb_kbrl.hip"

struct kb_handle {
    kb_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_order = nullptr;  // orders the agent's stream against the simulator's (kb_step_resident)
    hipEvent_t ev_join = nullptr;   // kb_run_resident: the agent's stream waits for the graph launches on the simulator's
    // The repairs of the small dictionaries (update_small_kernel: a workgroup per learner, latency-bound) run on a side stream BESIDE
    // the chip-wide rounds of the large ones (they touch different learners; the pool's allocator and the error flags are atomics)
    hipStream_t side = nullptr;
    hipEvent_t ev_sfork = nullptr, ev_sjoin = nullptr;
    kb::KbDev D;
    kb::KbState K;
    std::vector<void*> allocs;
    std::vector<GuardedAlloc> guarded;
    std::vector<std::pair<void*, size_t>> regions;  // every device array behind the handle (kb_save_state)
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
    // device-resident exchange (kb_shared_step)
    double* d_block = nullptr;     // this rank's proposal block
    double* d_gather = nullptr;    // [world][block]
    double* d_mprops = nullptr;    // merged proposals [S][budget_cap][KB_PROP_W]
    int32_t* d_mcounts = nullptr;  // [S]
    int32_t* d_taken = nullptr;    // [S]
    int32_t* d_total = nullptr;    // [2] proposers of all ranks in the round; 1 when a rank marked its block as failed
    // histories of the resident loop (kb_history_begin)
    double* h_reward = nullptr;
    int16_t *h_resources = nullptr, *h_hits = nullptr, *h_adjusted = nullptr, *h_sla = nullptr, *h_violation = nullptr;
    int32_t* h_cursor = nullptr;
    int32_t h_steps = 0;
    void* comm = nullptr;          // ncclComm_t (RCCL), bound at run time
    int comm_rank = 0, comm_world = 1;
    bool comm_aborted = false;     // the communicator was aborted: shared steps refuse until kb_comm_init forms a new one
    int32_t* h_total = nullptr;    // pinned: the per-round (proposals left, failure mark) pair of the shared step
    int budget_cap = 256;
    int heavy_blocks = 256;
    int mv_grid = 2048, r1_grid = 2048;  // one co-resident round of workgroups of the two Kinv-streaming kernels (kb_create)
    int heavy_rounds = 3;
    int rounds_gate = 200;         // tiles of Kinv queued per step from which the rounds are enqueued (eight learners of 320 landmarks;
                                   // KBRL_ROUNDS_GATE)
    hipGraph_t graph = nullptr;    // two captured closed-loop steps (agent step + simulator step) of kb_run_resident
    hipGraphExec_t gexec = nullptr;
    int g_big_par = 0, g_order_par = 0;  // the parities of the two alternating buffers the graph was captured at
    rs_handle* g_env = nullptr;
    uint64_t g_sig = 0;            // env->launch_sig at the capture
    bool gemm_fresh = false;       // shared, resident loop: workF / workE hold the scores of d_prev_state against the dictionaries as they are
    int big_par = 0;               // the large-learner list the next launches are ordered by (per-replica agents)
    int32_t* h_seen = nullptr;     // pinned, device-visible: large learners queued in a recent step (heavy_reset_kernel)
    bool rounds_always = false;
    int n_dict = 0;
    int T = 0, nv = 0;
    bool is_reset = false;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<int> ev_kind;  // 0 update phase, 1 select phase; ONE launch of: 2 heavy_matvec_kernel, 3 heavy_rank1_kernel, 4 select_bin_kernel,
                               // 5 heavy_finish_kernel, 6 select_gemm_kernel, 7 update_small_kernel
    size_t ev_used = 0;
    double kind_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // mean per kind over the span the last kb_phase_times_ms call summed up
    int64_t kind_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::string err;
};

template <class Tp>
static int kalloc(kb_handle* k, Tp** p, size_t n, bool zero = true) {
    void* q = nullptr;
    size_t bytes = sizeof(Tp) * (n ? n : 1);
    HIPCHK(k, guarded_malloc(&q, bytes, &k->guarded));
    if (zero) HIPCHK(k, hipMemsetAsync(q, 0, bytes, k->stream));
    if (!guards_on()) k->allocs.push_back(q);
    k->regions.emplace_back(q, bytes);
    *p = (Tp*)q;
    return RS_OK;
}

// ------------------------------------------------------------------ RCCL, bound directly at run time
// The only collective of the build: one all-gather of the proposal blocks per exchange round of the
// shared-dictionary mode, on the agent's own HIP stream, between device buffers.  librccl.so is opened on first
// use, so libranslice.so itself has no link-time dependency on it (replica-sharded runs never need it).
#include <dlfcn.h>
#include <chrono>
#include <unistd.h>
namespace rccl {
typedef struct { char internal[128]; } UniqueId;
typedef int (*GetUniqueId_t)(UniqueId*);
typedef int (*CommInitRank_t)(void**, int, UniqueId, int);
typedef int (*CommDestroy_t)(void*);
typedef int (*AllGather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*GetErrorString_t)(int);
typedef int (*CommAbort_t)(void*);
typedef int (*CommGetAsyncError_t)(void*, int*);
typedef int (*CommCount_t)(void*, int*);
static void* lib = nullptr;
static CommAbort_t CommAbort = nullptr;
static CommGetAsyncError_t CommGetAsyncError = nullptr;
static CommCount_t CommCount = nullptr, CommUserRank = nullptr;
static GetUniqueId_t GetUniqueId = nullptr;
static CommInitRank_t CommInitRank = nullptr;
static CommDestroy_t CommDestroy = nullptr;
static AllGather_t AllGather = nullptr;
static GetErrorString_t GetErrorString = nullptr;
static const int kDouble = 8;  // ncclFloat64 / ncclDouble
static bool load() {
    if (lib) return true;
    // RANSLICE_RCCL_LIB names the library to bind (a process that has already loaded an RCCL of its own -- e.g. through
    // torch.distributed's nccl backend -- should point this at the same file so that one copy serves both)
    const char* names[] = {getenv("RANSLICE_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    // one copy per process: a library of that name the process has already mapped (torch.distributed's nccl backend brings
    // its own) is taken as it is -- RTLD_NOLOAD finds it without loading anything -- before a second one would be opened
    for (const char* n : names)
        if (n && (lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;
    if (!lib)
        for (const char* n : names)
            if (n && (lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) return false;
    GetUniqueId = (GetUniqueId_t)dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (CommInitRank_t)dlsym(lib, "ncclCommInitRank");
    CommDestroy = (CommDestroy_t)dlsym(lib, "ncclCommDestroy");
    AllGather = (AllGather_t)dlsym(lib, "ncclAllGather");
    GetErrorString = (GetErrorString_t)dlsym(lib, "ncclGetErrorString");
    CommAbort = (CommAbort_t)dlsym(lib, "ncclCommAbort");                          // (optional: the abort path)
    CommGetAsyncError = (CommGetAsyncError_t)dlsym(lib, "ncclCommGetAsyncError");  // (optional)
    CommCount = (CommCount_t)dlsym(lib, "ncclCommCount");                          // (optional: kb_comm_info)
    CommUserRank = (CommCount_t)dlsym(lib, "ncclCommUserRank");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllGather) {
        dlclose(lib);
        lib = nullptr;
        return false;
    }
    return true;
}
}  // namespace rccl

static void kb_history_release(kb_handle* k) {
    void* ps[] = {k->h_reward, k->h_resources, k->h_hits, k->h_adjusted, k->h_sla, k->h_violation, k->h_cursor};
    for (void* p : ps)
        if (p) (void)hipFree(p);
    k->h_reward = nullptr;
    k->h_resources = k->h_hits = k->h_adjusted = k->h_sla = k->h_violation = nullptr;
    k->h_cursor = nullptr;
    k->h_steps = 0;
}

static void kb_comm_release(kb_handle* k) {
    if (k->comm && rccl::CommDestroy) (void)rccl::CommDestroy(k->comm);
    k->comm = nullptr;
}

// ncclCommAbort: the handle is NOT a world of its own afterwards -- its dictionaries are the replicated ones of a group it
// can no longer reach, so every shared step refuses (RS_ESTATE) until kb_comm_init joins a new communicator
static void kb_comm_abort(kb_handle* k) {
    if (k->comm && rccl::CommAbort) (void)rccl::CommAbort(k->comm);
    k->comm = nullptr;
    k->comm_aborted = true;
}

/* 128 bytes for kb_comm_init, generated by ONE rank (ncclGetUniqueId) and handed to the others by the launcher */
extern "C" int kb_comm_unique_id(void* id128) {
    if (!id128) return RS_EINVAL;
    if (!rccl::load()) return RS_ESTATE;
    rccl::UniqueId id;
    if (rccl::GetUniqueId(&id) != 0) return RS_EHIP;
    memcpy(id128, &id, sizeof id);
    return RS_OK;
}

/* Join the communicator of the `world` agents that share their dictionaries; rank = this handle's index */
extern "C" int kb_comm_init(kb_handle* k, const void* id128, int rank, int world) {
    if (!k || !id128 || world < 1 || rank < 0 || rank >= world || world > 64) return RS_EINVAL;
    if (!rccl::load()) {
        k->err = "kb_comm_init: librccl.so not found";
        return RS_ESTATE;
    }
    HIPCHK(k, hipSetDevice(k->device));
    kb_comm_release(k);
    rccl::UniqueId id;
    memcpy(&id, id128, sizeof id);
    const int rc = rccl::CommInitRank(&k->comm, world, id, rank);
    if (rc != 0) {
        k->comm = nullptr;
        k->err = std::string("ncclCommInitRank: ") + (rccl::GetErrorString ? rccl::GetErrorString(rc) : "error");
        return RS_EHIP;
    }
    k->comm_rank = rank;
    k->comm_world = world;
    k->comm_aborted = false;
    if (k->d_gather) (void)hipFree(k->d_gather);  // (sized by the world: allocated again on the first exchange)
    k->d_gather = nullptr;
    return RS_OK;
}

/* What the live communicator itself says (ncclCommUserRank / ncclCommCount): rank and number of ranks; (0, 1) for a handle
 * that never joined one.  RS_ESTATE after an abort. */
extern "C" int kb_comm_info(kb_handle* k, int* rank, int* world) {
    if (!k || !rank || !world) return RS_EINVAL;
    if (k->comm_aborted) {
        k->err = "kb_comm_info: the communicator was aborted (kb_comm_init forms a new one)";
        return RS_ESTATE;
    }
    *rank = 0;
    *world = 1;
    if (!k->comm) return RS_OK;
    if (!rccl::CommCount || !rccl::CommUserRank || rccl::CommCount(k->comm, world) != 0 || rccl::CommUserRank(k->comm, rank) != 0) {
        k->err = "kb_comm_info: ncclCommCount / ncclCommUserRank failed";
        return RS_EHIP;
    }
    return RS_OK;
}

extern "C" int kb_create(const kb_config* cfg, int device, kb_handle** out) {
    if (!cfg || !out) return RS_EINVAL;
    kb_handle* k = new kb_handle();
    *out = k;
    k->cfg = *cfg;
    k->device = device;
    if (cfg->n_envs <= 0 || cfg->n_slices <= 0 || cfg->n_slices > KB_MAX_SLICES || cfg->n_prbs <= 0 ||
        cfg->n_prbs > KB_NPRB_MAX || cfg->capacity < 2 || cfg->capacity > KB_CAPACITY_MAX) {
        k->err = "kb_create: unsupported configuration (<= 8 learners, n_prbs <= 255, 2 <= capacity <= 65536)";
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
    // (off unless the test build is asked for it: beside the rounds the small repairs gain 0.03 ms per step late in learning, but inside
    // the captured graph of kb_run_resident the extra branch cost the EARLY point 0.66 ms per step -- 2.31 against 1.65,
    // profiles/r05_o_bench_full.json -- for reasons not pursued)
    if (!cfg->shared_dictionary && dev_env("KBRL_SIDE_STREAM") && atoi(dev_env("KBRL_SIDE_STREAM")) != 0) {
        HIPCHK(k, hipStreamCreateWithFlags(&k->side, hipStreamNonBlocking));
        HIPCHK(k, hipEventCreateWithFlags(&k->ev_sfork, hipEventDisableTiming));
        HIPCHK(k, hipEventCreateWithFlags(&k->ev_sjoin, hipEventDisableTiming));
    }
    kb::KbDev& D = k->D;
    memset(&D, 0, sizeof D);
    D.n_envs = cfg->n_envs;
    D.S = cfg->n_slices;
    D.n_prbs = cfg->n_prbs;
    D.cap = cfg->capacity;
    D.max_shells = (cfg->capacity + KB_CH - 1) / KB_CH;
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
    D.tri = D.shared ? 0 : 1;  // one agent per replica: the lower block triangle of Kinv only (kb_kbrl.hip, "Storage")
    D.heavy_m = dev_env("KBRL_HEAVY_M") ? atoi(dev_env("KBRL_HEAVY_M")) : 0;  // developer knob; results do not depend on it
    if (dev_env("KBRL_ROUNDS")) {  // developer knob (tests): that many rounds, always enqueued; results do not depend on it
        k->heavy_rounds = atoi(dev_env("KBRL_ROUNDS"));
        k->rounds_always = true;
    }
    if (dev_env("KBRL_ROUNDS_GATE")) k->rounds_gate = atoi(dev_env("KBRL_ROUNDS_GATE"));
    if (hipHostMalloc((void**)&k->h_seen, 2 * sizeof(int32_t), hipHostMallocMapped) != hipSuccess) k->h_seen = nullptr;
    if (k->h_seen) k->h_seen[0] = k->h_seen[1] = 0;
    D.serial_apply = dev_env("KBRL_SERIAL_APPLY") ? 1 : 0;  // test knob: the batched apply of full dictionaries off
    D.first_env = cfg->first_env;
    k->nv = o;
    k->T = cfg->n_envs * cfg->n_slices;
    const size_t T = (size_t)k->T, N = (size_t)cfg->n_envs;
    const size_t ND = D.shared ? (size_t)cfg->n_slices : T;  // dictionaries
    k->n_dict = (int)ND;
    int rc;
    kb::KbState& K = k->K;
#define KA(p, n, z) if ((rc = kalloc(k, &(p), (n), (z))) != RS_OK) return rc
    KA(K.m, ND, true);
    KA(K.shell, ND * (size_t)D.max_shells, true);
    KA(K.head, ND * KB_HEAD, true);
    KA(K.pool_top, 1, true);
    KA(K.gtab, KB_GTAB, true);
    KA(K.kf_owner, ND, true);
    KA(K.heavy, T + 4, true);
    KA(K.hv_cfrom, T, true); KA(K.hv_cstar, T, true); KA(K.hv_state, T, true); KA(K.hv_grew, T, true); KA(K.hv_m, T, true);
    KA(K.hv_pend, 2 * T, true); KA(K.hv_delta, T, true); KA(K.hv_f, 256 * T, true);
    KA(K.hv_mvbase, T + 1, true); KA(K.hv_r1base, T + 1, true); KA(K.hv_work, 8, true);
    KA(K.big, 2 * (1 + KB_BIG_MAX), true); KA(K.isbig, 2 * T, true);
    {
        // The pool every dictionary of the handle grows in (kb_kbrl.hip, "Storage").  Upper bound on what can ever be
        // asked for: every dictionary at its capacity.  kb_config.pool_bytes == 0 picks a default below it.
        unsigned long long need_all = 64;
        for (int b = 0; b < D.max_shells; ++b) need_all += (unsigned long long)ND * kb::kb_shell_doubles(b, D.tri);
        unsigned long long want = cfg->pool_bytes > 0 ? (unsigned long long)cfg->pool_bytes / 8 : need_all;
        if (want > need_all) want = need_all;
        if (cfg->pool_bytes <= 0) {
            // default: what every dictionary at capacity would need, but no more than 1 GB + 2 MB per dictionary (a mean of
            // ~500 landmarks) nor than half of what the device has left; callers that know better pass pool_bytes
            size_t free_b = 0, total_b = 0;
            HIPCHK(k, hipMemGetInfo(&free_b, &total_b));
            const unsigned long long modest = ((1ull << 30) + (unsigned long long)ND * (2ull << 20)) / 8;
            const unsigned long long avail = (unsigned long long)free_b / 2 / 8;
            if (want > modest) want = modest;
            if (want > avail) want = avail;
        }
        const unsigned long long least = 64 + (unsigned long long)ND * kb::kb_shell_doubles(0, D.tri);
        if (want < least) {
            k->err = "kb_create: the dictionary pool cannot hold even one shell (48 KB) per dictionary";
            return RS_EINVAL;
        }
        D.pool_doubles = want;
        KA(K.pool, (size_t)want, false);  // every entry is written before it is read
    }
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
    KA(K.workb, D.shared ? (size_t)cfg->n_slices * 2 * k->budget_cap * kb::kb_capr(cfg->capacity) : 1, true);
    KA(K.offgrid, ND, true);
    KA(K.f32bad, ND, true);
    KA(K.F, D.shared ? 1 : T * 256, true);
    KA(K.fstate, D.shared ? 1 : T * 16, true);
    KA(K.fver, T, true);
    KA(K.Wg, D.shared ? 1 : T * 256, true);
    KA(K.fdirect, T, true);
    KA(K.dlist, D.shared ? 1 : T * KB_DLIST * 3, true);
    KA(K.ver, ND, true);
    KA(K.workq, D.shared ? (size_t)cfg->n_slices * 16 * kb::kb_capr(cfg->capacity) * 16 : 1, true);
    KA(K.workF, D.shared ? (size_t)cfg->n_slices * KB_GEMM_KS * N * 256 : 1, true);
    KA(K.workE, D.shared ? (size_t)cfg->n_slices * KB_GEMM_KS * N : 1, true);
    KA(K.workg, D.shared ? (size_t)cfg->n_slices * k->budget_cap * k->budget_cap : 1, true);
    KA(K.workf, D.shared ? (size_t)cfg->n_slices * k->budget_cap : 1, true);
    KA(k->d_props, (size_t)cfg->n_slices * k->budget_cap * KB_PROP_W, true);
    KA(k->d_counts, (size_t)cfg->n_slices, true);
    KA(k->d_gstats, 32, true);
    KA(k->d_block, (size_t)cfg->n_slices * (1 + (size_t)k->budget_cap * KB_PROP_W), true);
    KA(k->d_mprops, (size_t)cfg->n_slices * k->budget_cap * KB_PROP_W, true);
    KA(k->d_mcounts, (size_t)cfg->n_slices, true);
    KA(k->d_taken, (size_t)cfg->n_slices, true);
    KA(k->d_total, 2, true);
#undef KA
    if (D.shared) {
        // shared_apply_kernel keeps the coefficient column of a full dictionary and the proposals' Gram block in LDS
        const size_t lds = sizeof(double) * kb::kb_apply_lds_doubles(cfg->capacity, k->budget_cap);
        if (lds > 150 * 1024) {
            k->err = "kb_create: shared dictionaries hold at most ~14,000 landmarks (the batched apply keeps a column in LDS)";
            return RS_EINVAL;
        }
        if (lds > 48 * 1024)
            HIPCHK(k, hipFuncSetAttribute((const void*)kb::shared_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    {
        // The two kernels that stream Kinv split their work line evenly over the waves of the launch: exactly one resident
        // round of workgroups, so that no second, partly filled round trails the first (KBRL_STREAM_GRID overrides both)
        int cus = 256, b1 = 0, b2 = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b1, kb::heavy_matvec_kernel, 256, 0) == hipSuccess && b1 > 0) k->mv_grid = b1 * cus;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b2, kb::heavy_rank1_kernel, 256, 0) == hipSuccess && b2 > 0) k->r1_grid = b2 * cus;
        if (dev_env("KBRL_STREAM_GRID")) k->mv_grid = k->r1_grid = atoi(dev_env("KBRL_STREAM_GRID"));
    }
    hipLaunchKernelGGL(kb::kb_gtab_kernel, dim3((KB_GTAB + 255) / 256), dim3(256), 0, k->stream, k->D, k->K);
    HIPCHK(k, hipGetLastError());
    HIPCHK(k, hipStreamSynchronize(k->stream));
    return RS_OK;
}

static void kb_drop_graph(kb_handle* k) {
    if (k->gexec) (void)hipGraphExecDestroy(k->gexec);
    if (k->graph) (void)hipGraphDestroy(k->graph);
    k->gexec = nullptr;
    k->graph = nullptr;
    k->g_env = nullptr;
}

extern "C" void kb_destroy(kb_handle* k) {
    if (!k) return;
    if (k->stream) (void)hipStreamSynchronize(k->stream);
    kb_drop_graph(k);
    if (k->D.shared && dev_env("KBRL_APPLY_TIMES") && k->d_gstats) {  // developer aid: where shared_apply_kernel spends its time
        uint64_t g[32];
        if (hipMemcpy(g, k->d_gstats, sizeof g, hipMemcpyDeviceToHost) == hipSuccess)
            for (int s = 0; s < k->cfg.n_slices; ++s)
                fprintf(stderr, "shared_apply slice %d: %.3f ms in total, %llu samples applied of %llu proposed\n", s,
                        (double)g[8 + s] / 1e5, (unsigned long long)g[16 + s], (unsigned long long)g[24 + s]);
    }
    if (guards_on()) check_guards(k->guarded, "kb");
    for (auto& g : k->guarded) (void)hipFree(g.base);
    for (void* p : k->allocs) (void)hipFree(p);
    for (auto& e : k->ev) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (k->ev_order) (void)hipEventDestroy(k->ev_order);
    if (k->ev_join) (void)hipEventDestroy(k->ev_join);
    if (k->side) (void)hipStreamSynchronize(k->side);
    if (k->side) (void)hipStreamDestroy(k->side);
    if (k->ev_sfork) (void)hipEventDestroy(k->ev_sfork);
    if (k->ev_sjoin) (void)hipEventDestroy(k->ev_sjoin);
    kb_comm_release(k);
    if (k->d_gather) (void)hipFree(k->d_gather);
    kb_history_release(k);
    if (k->h_seen) (void)hipHostFree(k->h_seen);
    if (k->h_total) (void)hipHostFree(k->h_total);
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
    if ((size_t)k->n_dict > n) n = (size_t)k->n_dict;
    hipLaunchKernelGGL(kb::kb_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, k->stream, k->D, k->K,
                       k->d_action, k->d_labels, dseed, k->n_dict);
    HIPCHK(k, hipMemsetAsync(k->d_prev_state, 0, sizeof(float) * N * k->nv, k->stream));
    HIPCHK(k, hipMemsetAsync(k->K.m, 0, sizeof(int32_t) * (size_t)k->n_dict, k->stream));
    HIPCHK(k, hipMemsetAsync(k->K.shell, 0, sizeof(uint64_t) * (size_t)k->n_dict * k->D.max_shells, k->stream));
    HIPCHK(k, hipMemsetAsync(k->K.head, 0xFF, sizeof(int32_t) * (size_t)k->n_dict * KB_HEAD, k->stream));
    HIPCHK(k, hipMemsetAsync(k->K.offgrid, 0, sizeof(int32_t) * (size_t)k->n_dict, k->stream));
    HIPCHK(k, hipMemsetAsync(k->K.f32bad, 0, sizeof(int32_t) * (size_t)k->n_dict, k->stream));
    HIPCHK(k, hipMemsetAsync(k->K.ver, 0, sizeof(int32_t) * (size_t)k->n_dict, k->stream));
    HIPCHK(k, hipMemsetAsync(k->K.fver, 0xFF, sizeof(int32_t) * T, k->stream));  // -1: no stored scores
    HIPCHK(k, hipMemsetAsync(k->d_gstats, 0, sizeof(uint64_t) * 32, k->stream));
    HIPCHK(k, hipMemsetAsync(k->K.hv_work, 0, sizeof(unsigned long long) * 8, k->stream));
    HIPCHK(k, hipGetLastError());
    HIPCHK(k, hipStreamSynchronize(k->stream));
    if (k->h_seen) k->h_seen[0] = k->h_seen[1] = 0;
    k->gemm_fresh = false;
    k->big_par = 0;
    kb_drop_graph(k);
    HIPCHK(k, hipMemset(k->K.big, 0, sizeof(int32_t) * 2 * (1 + KB_BIG_MAX)));
    HIPCHK(k, hipMemset(k->K.isbig, 0, sizeof(int32_t) * 2 * T));
    (void)hipFree(dseed);
    k->is_reset = true;
    return RS_OK;
}

static int kb_check(kb_handle* k) {
    std::vector<int32_t> e((size_t)k->cfg.n_envs);
    HIPCHK(k, hipMemcpyAsync(e.data(), k->K.err, sizeof(int32_t) * e.size(), hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    for (size_t i = 0; i < e.size(); ++i)
        if (e[i] & ~(8 | 16)) {  // bit 8: a dictionary is at its capacity, bit 16: the pool is exhausted (it projects
                                 // instead of growing): reported by kb_get_pool / kb_get_sizes, not an error
            k->err = "KBRL agent " + std::to_string(i) + ": internal error flag " + std::to_string(e[i]);
            return RS_EOVERFLOW;
        }
    return RS_OK;
}

// apply a proposal list per slice: the wide kernels prepare what a full dictionary's list needs (they return at once for
// dictionaries that can still grow), then one workgroup per slice walks its list in order
static void launch_shared_apply(kb_handle* k, const double* props, const int32_t* counts, int budget) {
    const unsigned S = (unsigned)k->cfg.n_slices;
    hipLaunchKernelGGL(kb::shared_cols_kernel, dim3(S, KB_COLS_BLOCKS), dim3(256), 0, k->stream, k->D, k->K, props, counts, budget);
    // d* of the whole list: on the matrix cores when the capacity (the size of a full dictionary) is a multiple of 64, by
    // the column walk otherwise -- the same sums bit for bit (KBRL_MATVEC_MFMA=0: the column walk always)
    static const bool mfma_on = !(dev_env("KBRL_MATVEC_MFMA") && atoi(dev_env("KBRL_MATVEC_MFMA")) == 0);
    const int mfma = mfma_on && k->cfg.capacity % 64 == 0 ? 1 : 0;
    if (mfma)
        hipLaunchKernelGGL(kb::shared_matvec_mfma_kernel, dim3(S, KB_MATVEC_BLOCKS), dim3(256), 0, k->stream, k->D, k->K, counts, budget);
    else
        hipLaunchKernelGGL(kb::shared_matvec_kernel, dim3(S, KB_MATVEC_BLOCKS), dim3(256), 0, k->stream, k->D, k->K, counts, budget, 0);
    hipLaunchKernelGGL(kb::shared_gram_kernel, dim3(S, 128), dim3(256), 0, k->stream, k->D, k->K, counts, budget);
    const size_t lds = sizeof(double) * kb::kb_apply_lds_doubles(k->cfg.capacity, k->budget_cap);
    hipLaunchKernelGGL(kb::shared_apply_kernel, dim3(S), dim3(1024), lds, k->stream, k->D, k->K, props, counts, budget, k->d_gstats);
}

// the scores of the large shared dictionaries for all replicas at once (F = E Q on MFMA); shared_scan_kernel picks them up
static void launch_shared_gemm(kb_handle* k, const float* d_state) {
    const unsigned S = (unsigned)k->cfg.n_slices, rts = (unsigned)((k->cfg.n_envs + 15) / 16);
    const unsigned chunks = (unsigned)((k->cfg.n_prbs / 16 + 1 + 7) / 8);
    hipLaunchKernelGGL(kb::shared_q_kernel, dim3(S, 64), dim3(256), 0, k->stream, k->D, k->K);
    hipLaunchKernelGGL(kb::shared_fgemm_kernel, dim3(S, rts, chunks * (KB_GEMM_KS / 4)), dim3(256), 0, k->stream, k->D, k->K, d_state);
}

static int kb_time_begin(kb_handle* k, hipEvent_t* e1, int kind = 0) {
    *e1 = nullptr;
    if (!k->timing) return RS_OK;
    if (k->ev_used == k->ev.size()) {
        hipEvent_t a0, a1;
        HIPCHK(k, hipEventCreate(&a0));
        HIPCHK(k, hipEventCreate(&a1));
        k->ev.emplace_back(a0, a1);
        k->ev_kind.push_back(0);
    }
    k->ev_kind[k->ev_used] = kind;
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
    // (update_control_kernel starts from stored scores since round 4: every learner costs the same there, so it runs in task
    // order -- no large-learner list to look up first, no 4,096 empty places in the grid)
    a.big_par = -1;
    const unsigned grid1 = (unsigned)k->T;
    hipEvent_t e1;
    int rc = kb_time_begin(k, &e1);
    if (rc != RS_OK) return rc;
    if (k->D.heavy_m > 0)
        hipLaunchKernelGGL(kb::update_control_kernel<true>, dim3(grid1), dim3(64), 0, k->stream, a);
    else
        hipLaunchKernelGGL(kb::update_control_kernel<false>, dim3(grid1), dim3(64), 0, k->stream, a);
    if (!k->D.shared) {
        // the learners with a mistake to repair: small dictionaries one wave each, all at once; large ones a workgroup
        // each, taken by persistent workgroups
        const unsigned blocks = (unsigned)(k->T < k->heavy_blocks ? k->T : k->heavy_blocks);
        // (with per-kernel event timing on, everything stays on the one stream: the account is per kernel, not per overlap)
        const bool forked = k->side != nullptr && !k->timing;
        hipEvent_t es;
        if ((rc = kb_time_begin(k, &es, 7)) != RS_OK) return rc;
        if (forked) {
            HIPCHK(k, hipEventRecord(k->ev_sfork, k->stream));
            HIPCHK(k, hipStreamWaitEvent(k->side, k->ev_sfork, 0));
        }
        hipLaunchKernelGGL(kb::update_small_kernel, dim3((unsigned)(k->T < 4096 ? k->T : 4096)), dim3(256), 0, forked ? k->side : k->stream, a);
        if (forked) HIPCHK(k, hipEventRecord(k->ev_sjoin, k->side));
        if (es) HIPCHK(k, hipEventRecord(es, k->stream));
        // The rounds are nine launches that do nothing while no dictionary is large (early in learning): they are
        // enqueued only once a recent step has queued a few large learners.  The host reads that count from pinned memory
        // without waiting for the device, so it lags by the depth of the launch queue; until it catches up the clean-up
        // kernel below, which is always launched, repairs them one workgroup each.  Results do not depend on it.
        const bool rounds = k->rounds_always || (k->h_seen && *(volatile int32_t*)k->h_seen >= k->rounds_gate);
        if (rounds && k->heavy_rounds > 0) hipLaunchKernelGGL(kb::heavy_plan_kernel, dim3(1), dim3(1024), 0, k->stream, k->D, k->K);
        for (int r = 0; rounds && r < k->heavy_rounds; ++r) {  // one repair of every pending large learner per round, chip-wide
            hipEvent_t em, er;  // (with kb_set_kernel_timing: each launch of the two streaming kernels on its own, for their roofline)
            if ((rc = kb_time_begin(k, &em, 2)) != RS_OK) return rc;
            hipLaunchKernelGGL(kb::heavy_matvec_kernel, dim3((unsigned)k->mv_grid), dim3(256), 0, k->stream, k->D, k->K);
            if (em) HIPCHK(k, hipEventRecord(em, k->stream));
            hipEvent_t ef;
            if ((rc = kb_time_begin(k, &ef, 5)) != RS_OK) return rc;
            hipLaunchKernelGGL(kb::heavy_finish_kernel, dim3(1024), dim3(256), 0, k->stream, a);
            if (ef) HIPCHK(k, hipEventRecord(ef, k->stream));
            hipLaunchKernelGGL(kb::heavy_plan_kernel, dim3(1), dim3(1024), 0, k->stream, k->D, k->K);
            if ((rc = kb_time_begin(k, &er, 3)) != RS_OK) return rc;
            hipLaunchKernelGGL(kb::heavy_rank1_kernel, dim3((unsigned)k->r1_grid), dim3(256), 0, k->stream, k->D, k->K);
            if (er) HIPCHK(k, hipEventRecord(er, k->stream));
        }
        hipLaunchKernelGGL(kb::update_heavy_kernel, dim3(blocks), dim3(KB_HEAVY_THREADS), 0, k->stream, a);
        if (forked) HIPCHK(k, hipStreamWaitEvent(k->stream, k->ev_sjoin, 0));
        hipLaunchKernelGGL(kb::heavy_reset_kernel, dim3(1), dim3(1), 0, k->stream, k->K, (volatile int32_t*)k->h_seen);
    }
    if (e1) HIPCHK(k, hipEventRecord(e1, k->stream));  // the whole update phase
    return RS_OK;
}

static int launch_select(kb_handle* k, const float* d_state, int32_t* d_action_out) {
    kb::SelArgs a;
    a.D = k->D;
    a.K = k->K;
    a.state = d_state;
    a.big_par = k->D.shared ? -1 : k->big_par;
    a.gemm = k->D.shared ? 1 : 0;
    k->gemm_fresh = false;  // (kb_shared_step_resident sets it again once the state has become d_prev_state)
    hipEvent_t e1;
    int rc = kb_time_begin(k, &e1, 1);
    if (rc != RS_OK) return rc;
    if (a.gemm) launch_shared_gemm(k, d_state);
    if (k->D.shared) {
        hipLaunchKernelGGL(kb::select_kernel, dim3((unsigned)k->T), dim3(64), 0, k->stream, a);
    } else {  // one agent per replica: a wave per learner bins its landmarks, then sixteen learners per workgroup are scored
              // as one product on the matrix cores
        const unsigned slots = (unsigned)k->T + (a.big_par >= 0 ? KB_BIG_MAX : 0);
        hipEvent_t eb, eg;
        if ((rc = kb_time_begin(k, &eb, 4)) != RS_OK) return rc;
        if (a.big_par >= 0) {  // the listed large learners several waves each, the others a wave each
            if (dev_env("KBRL_BIN_TWO_LAUNCHES")) {  // test build: the two kernels one after the other (rounds 4-5; same bits)
                hipLaunchKernelGGL(kb::select_bin_big_kernel, dim3(KB_BINBIG_GRID), dim3(64 * KB_BINBIG_WAVES), 0, k->stream, a);
                hipLaunchKernelGGL(kb::select_bin_kernel, dim3((unsigned)k->T), dim3(64), 0, k->stream, a, (int)KB_BIG_MAX);
            } else {
                hipLaunchKernelGGL(kb::select_bin_all_kernel, dim3(KB_BINBIG_GRID + (unsigned)((k->T + KB_BINBIG_WAVES - 1) / KB_BINBIG_WAVES)),
                                   dim3(64 * KB_BINBIG_WAVES), 0, k->stream, a);
            }
            hipLaunchKernelGGL(kb::big_list_kernel, dim3((unsigned)((k->T + 255) / 256)), dim3(256), 0, k->stream, k->D, k->K, a.big_par);
        } else {
            hipLaunchKernelGGL(kb::select_bin_kernel, dim3(slots), dim3(64), 0, k->stream, a, 0);
        }
        if (eb) HIPCHK(k, hipEventRecord(eb, k->stream));
        if ((rc = kb_time_begin(k, &eg, 6)) != RS_OK) return rc;
        hipLaunchKernelGGL(kb::select_gemm_kernel, dim3((slots + KB_SEL_WAVES - 1) / KB_SEL_WAVES), dim3(256), 0, k->stream, a);
        if (eg) HIPCHK(k, hipEventRecord(eg, k->stream));
    }
    if (e1) HIPCHK(k, hipEventRecord(e1, k->stream));
    hipLaunchKernelGGL(kb::adjust_kernel, dim3((unsigned)((k->cfg.n_envs + 255) / 256)), dim3(256), 0, k->stream, k->D,
                       k->K, d_action_out, a.big_par);
    if (a.big_par >= 0) k->big_par = 1 - k->big_par;  // select_kernel wrote the other list for the launches that follow
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
    if (k->D.shared) {  // (update_control's repair kernels are not launched for shared dictionaries: nothing would learn)
        k->err = "kb_step_resident: shared-dictionary handles step through kb_shared_step_resident";
        return RS_ESTATE;
    }
    HIPCHK(k, hipSetDevice(k->device));
    if (env->hint_auto && !env->block_hint) {  // the next steps take agent-made allocations
        rs_set_schedule_hint(env, 1);
        env->hint_auto = true;
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
    if (k->h_steps > 0) {  // KBRL_Control.run's history columns of this step (kbrl_control.py:135-141)
        kb::HistArgs ha;
        ha.D = k->D;
        ha.K = k->K;
        ha.reward = env->d_reward;
        ha.labels = env->d_labels;
        ha.violations = env->d_viol;
        ha.hits = k->d_hits;
        ha.h_reward = k->h_reward;
        ha.h_resources = k->h_resources;
        ha.h_hits = k->h_hits;
        ha.h_adjusted = k->h_adjusted;
        ha.h_sla = k->h_sla;
        ha.h_violation = k->h_violation;
        ha.cursor = k->h_cursor;
        ha.steps = k->h_steps;
        hipLaunchKernelGGL(kb::history_kernel, dim3((unsigned)((k->cfg.n_envs + 255) / 256)), dim3(256), 0, k->stream, ha);
        hipLaunchKernelGGL(kb::history_advance_kernel, dim3(1), dim3(1), 0, k->stream, k->h_cursor);
    }
    HIPCHK(k, hipMemcpyAsync(k->d_prev_state, env->d_obs, sizeof(float) * (size_t)k->cfg.n_envs * k->nv,
                             hipMemcpyDeviceToDevice, k->stream));
    HIPCHK(k, hipEventRecord(done, k->stream));
    HIPCHK(k, hipStreamWaitEvent(env->stream, done, 0));
    HIPCHK(k, hipGetLastError());
    return RS_OK;
}

// n_steps x (kb_step_resident(k, env); rs_step_resident(env)) -- KBRL_Control.run's loop body (kbrl_control.py:129-134) -- enqueued
// by one call.  With use_graph two consecutive steps are captured once into a hipGraph (the agent's large-learner lists and
// the simulator's order counters alternate between two buffers each; every kernel of the loop reads its step-dependent state
// from device memory) and replayed: ~25 launches per step become one graph launch per two steps, which is what lets several
// handles share a GPU without the host's launch rate becoming the limit (experiments_kbrl.evaluate_grid).  The captured
// sequence always carries the chip-wide repair rounds (they return at once when nothing is queued).  Results are identical
// either way.
static int enqueue_closed_step(kb_handle* k, rs_handle* env) {
    int rc = kb_step_resident(k, env);
    if (rc != RS_OK) return rc;
    rc = launch_step(env);
    if (rc != RS_OK) k->err = env->err;
    return rc;
}

extern "C" int kb_run_resident(kb_handle* k, rs_handle* env, int n_steps, int use_graph) {
    if (!k || !env || n_steps < 0) return RS_EINVAL;
    int done = 0, rc;
    // rocprofv3 (ROCm 7.2) dies with SIGSEGV inside hipGraphLaunch of this loop's graph (kernels, memory copies, memsets; the
    // simulator-only graph of rs_run_random is fine under it): with a profiler tool attached the steps are enqueued one by one,
    // same results.  KBRL_GRAPH_UNDER_PROFILER=1 keeps the graph.
    static const bool no_graph = getenv("ROCP_TOOL_LIBRARIES") != nullptr &&
                                 !(dev_env("KBRL_GRAPH_UNDER_PROFILER") && atoi(dev_env("KBRL_GRAPH_UNDER_PROFILER")) != 0);
    if (no_graph) use_graph = 0;
    if (use_graph && !k->timing && !env->timing && !k->D.shared && n_steps >= 3) {
        if (!k->gexec || k->g_env != env || k->g_sig != env->launch_sig) {
            // (one plain step first: whatever the loop creates lazily -- events, the schedule hint -- exists before the capture)
            kb_drop_graph(k);
            if ((rc = enqueue_closed_step(k, env)) != RS_OK) return rc;
            done += 1;
        } else if (k->big_par != k->g_big_par || env->order_par != k->g_order_par) {
            if ((rc = enqueue_closed_step(k, env)) != RS_OK) return rc;  // realign with the parities of the capture
            done += 1;
            if (k->big_par != k->g_big_par || env->order_par != k->g_order_par) kb_drop_graph(k);  // (out of phase with each other)
        }
        if (!k->gexec && n_steps - done >= 2) {
            const int32_t clock0 = env->clock;
            const uint64_t steps0 = env->steps;
            const bool ra = k->rounds_always, gf = k->gemm_fresh;
            k->g_big_par = k->big_par;
            k->g_order_par = env->order_par;
            k->rounds_always = true;
            HIPCHK(k, hipStreamBeginCapture(env->stream, hipStreamCaptureModeThreadLocal));
            const int rc1 = enqueue_closed_step(k, env);
            const int rc2 = rc1 == RS_OK ? enqueue_closed_step(k, env) : rc1;
            const hipError_t ec = hipStreamEndCapture(env->stream, &k->graph);
            env->clock = clock0;
            env->steps = steps0;
            env->order_par = k->g_order_par;
            k->big_par = k->g_big_par;
            k->rounds_always = ra;
            k->gemm_fresh = gf;
            if (rc2 != RS_OK || ec != hipSuccess) {
                kb_drop_graph(k);
                if (rc2 == RS_OK) k->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(ec);
                return rc2 != RS_OK ? rc2 : RS_EHIP;
            }
            HIPCHK(k, hipGraphInstantiate(&k->gexec, k->graph, nullptr, nullptr, 0));
            k->g_env = env;
            k->g_sig = env->launch_sig;
        }
        bool launched = false;
        while (k->gexec && n_steps - done >= 2) {
            HIPCHK(k, hipGraphLaunch(k->gexec, env->stream));
            env->clock += 2 * env->cfg.slots_per_step;
            env->steps += 2;
            done += 2;
            launched = true;
        }
        if (launched) {
            // The graph carries the agent's kernels but runs on the SIMULATOR's stream: the agent's stream joins it here, so
            // that kb_synchronize / kb_get_stats / kb_save_state (which wait for the agent's stream only) see the loop's end
            // even when the call ends on a graph launch.
            if (!k->ev_join) HIPCHK(k, hipEventCreateWithFlags(&k->ev_join, hipEventDisableTiming));
            HIPCHK(k, hipEventRecord(k->ev_join, env->stream));
            HIPCHK(k, hipStreamWaitEvent(k->stream, k->ev_join, 0));
        }
    }
    for (; done < n_steps; ++done)
        if ((rc = enqueue_closed_step(k, env)) != RS_OK) return rc;
    return RS_OK;
}

static int kb_one(kb_handle* k, int e, int s, const double* x, int y, bool update, double out[4]) {
    if (!k || !x || e < 0 || e >= k->cfg.n_envs || s < 0 || s >= k->cfg.n_slices) return RS_EINVAL;
    k->gemm_fresh = false;
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
        hipLaunchKernelGGL(kb::update_one_kernel, dim3(1), dim3(256), 0, k->stream, a);
    else
        hipLaunchKernelGGL(kb::predict_one_kernel, dim3(1), dim3(256), 0, k->stream, a);
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

// dense copies of dictionary `dict` through a gather kernel (the dictionary is scattered over its shells)
static int kb_gather(kb_handle* k, int dict, int d, int m, double* landmarks, double* coeff, double* kinv, double* kf_row) {
    const size_t nL = landmarks ? (size_t)m * d : 0, nC = coeff ? (size_t)m : 0, nK = kinv ? (size_t)m * m : 0,
                 nF = kf_row ? (size_t)m : 0;
    double* tmp = nullptr;
    HIPCHK(k, hipMalloc((void**)&tmp, sizeof(double) * (nL + nC + nK + nF + 1)));
    double *dL = tmp, *dC = dL + nL, *dK = dC + nC, *dF = dK + nK;
    const unsigned blocks = (unsigned)(nK ? (nK + 255) / 256 > 2048 ? 2048 : (nK + 255) / 256 : 4);
    hipLaunchKernelGGL(kb::gather_learner_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, k->stream, k->D, k->K, dict, d, m,
                       landmarks ? dL : nullptr, coeff ? dC : nullptr, kinv ? dK : nullptr, kf_row ? dF : nullptr);
    int rc = RS_OK;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && nL) e = hipMemcpyAsync(landmarks, dL, sizeof(double) * nL, hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess && nC) e = hipMemcpyAsync(coeff, dC, sizeof(double) * nC, hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess && nK) e = hipMemcpyAsync(kinv, dK, sizeof(double) * nK, hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess && nF) e = hipMemcpyAsync(kf_row, dF, sizeof(double) * nF, hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(k->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) {
        k->err = std::string("kb_gather: ") + hipGetErrorString(e);
        rc = RS_EHIP;
    }
    return rc;
}

extern "C" int kb_get_learner(kb_handle* k, int e, int s, int32_t* m_out, double* landmarks, double* coeff, double* kinv) {
    if (!k || e < 0 || e >= k->cfg.n_envs || s < 0 || s >= k->cfg.n_slices) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    const size_t dict = k->D.shared ? (size_t)s : (size_t)e * k->cfg.n_slices + s;
    int32_t m = 0;
    HIPCHK(k, hipMemcpyAsync(&m, k->K.m + dict, sizeof m, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    if (m_out) *m_out = m;
    if (m > 0 && (landmarks || coeff || kinv)) return kb_gather(k, (int)dict, k->cfg.dims[s] + 1, m, landmarks, coeff, kinv, nullptr);
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
    const size_t task = (size_t)e * k->cfg.n_slices + s;
    const size_t dict = k->D.shared ? (size_t)s : task;
    int32_t m = 0;
    HIPCHK(k, hipMemcpyAsync(&m, k->K.m_last + task, sizeof m, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    if (m_out) *m_out = m;
    if (row && m > 0) return kb_gather(k, (int)dict, k->cfg.dims[s] + 1, m, nullptr, nullptr, nullptr, row);
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

// bytes of the dictionary pool in use / in total, and how many replicas carry the "dictionary at capacity" (bit 8) and
// "pool exhausted" (bit 16) flags
extern "C" int kb_get_pool(kb_handle* k, uint64_t* used_bytes, uint64_t* total_bytes, int32_t* n_saturated, int32_t* n_pool_full) {
    if (!k) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    unsigned long long top = 0;
    std::vector<int32_t> e((size_t)k->cfg.n_envs);
    HIPCHK(k, hipMemcpyAsync(&top, k->K.pool_top, sizeof top, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipMemcpyAsync(e.data(), k->K.err, sizeof(int32_t) * e.size(), hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    if (top > k->D.pool_doubles) top = k->D.pool_doubles;
    if (used_bytes) *used_bytes = (uint64_t)top * 8;
    if (total_bytes) *total_bytes = (uint64_t)k->D.pool_doubles * 8;
    int32_t ns = 0, nf = 0;
    for (int32_t v : e) {
        ns += (v & 8) ? 1 : 0;
        nf += (v & 16) ? 1 : 0;
    }
    if (n_saturated) *n_saturated = ns;
    if (n_pool_full) *n_pool_full = nf;
    return RS_OK;
}

// the per-replica flag words behind kb_get_pool's counts: bit 8 a dictionary of the replica is at its capacity, bit 16 it found
// the pool exhausted (both keep learning by projection)
extern "C" int kb_get_flags(kb_handle* k, int32_t* flags) {
    if (!k || !flags) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    HIPCHK(k, hipMemcpyAsync(flags, k->K.err, sizeof(int32_t) * (size_t)k->cfg.n_envs, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    return RS_OK;
}

// What the chip-wide repair rounds have streamed since kb_reset, counted by the kernels from their own work plan:
// work[0] tiles of Kinv heavy_matvec_kernel read (32,768 bytes each; 128 partial sums written per tile), work[1] units
// of heavy_rank1_kernel (sixteen rows: 8,192 bytes read and 8,192 written each), work[2] / work[3] launches of the two
// that had anything to do.  The algorithmic bytes of projectron.py:42 (Kinv @ K_f) and :54-58 (the rank-1 update).
// work[4] scoring passes that had to evaluate landmarks' exponentials directly (add_direct_terms), work[5] the landmarks they
// evaluated, each for every open candidate group; work[6], work[7] reserved.
extern "C" int kb_get_repair_work(kb_handle* k, uint64_t work[8]) {
    if (!k || !work) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    HIPCHK(k, hipMemcpyAsync(work, k->K.hv_work, sizeof(uint64_t) * 8, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    return RS_OK;
}

extern "C" int kb_set_kernel_timing(kb_handle* k, int enable) {
    if (!k) return RS_EINVAL;
    kb_drop_graph(k);
    k->timing = enable != 0;
    k->ev_used = 0;
    if (enable && k->D.shared && k->d_gstats && dev_env("KBRL_APPLY_TIMES")) {  // (developer aid: count from here on)
        HIPCHK(k, hipSetDevice(k->device));
        HIPCHK(k, hipMemsetAsync(k->d_gstats + 8, 0, sizeof(uint64_t) * 24, k->stream));
    }
    return RS_OK;
}

// mean device time of the timed phases since the last call: [0] the update phase (update_control_kernel + the repair
// kernels; shared mode: the scan kernel), [1] select_kernel; n[i] = how many of each were timed
extern "C" int kb_phase_times_ms(kb_handle* k, double ms[2], int64_t n[2]) {
    if (!k || !ms || !n) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < k->ev_used; ++i) {
        float t = 0.f;
        HIPCHK(k, hipEventElapsedTime(&t, k->ev[i].first, k->ev[i].second));
        tot[k->ev_kind[i] & 7] += t;
        cnt[k->ev_kind[i] & 7] += 1;
    }
    for (int q = 0; q < 8; ++q) {
        k->kind_n[q] = cnt[q];
        k->kind_ms[q] = cnt[q] ? tot[q] / (double)cnt[q] : 0.0;
    }
    for (int q = 0; q < 2; ++q) {
        n[q] = k->kind_n[q];
        ms[q] = k->kind_ms[q];
    }
    k->ev_used = 0;
    return RS_OK;
}

// mean duration of ONE launch, per kernel, over the span the last kb_phase_times_ms / kb_kernel_time_ms call summed up (HIP events
// on the agent's stream around every launch, kb_set_kernel_timing): [2] heavy_matvec_kernel, [3] heavy_rank1_kernel,
// [4] select_bin_kernel, [5] heavy_finish_kernel, [6] select_gemm_kernel, [7] update_small_kernel ([0], [1]: the two phases)
extern "C" int kb_kernel_times_ms(kb_handle* k, double ms[8], int64_t n[8]) {
    if (!k || !ms || !n) return RS_EINVAL;
    for (int q = 0; q < 8; ++q) {
        ms[q] = k->kind_ms[q];
        n[q] = k->kind_n[q];
    }
    return RS_OK;
}

// mean duration of ONE launch of heavy_matvec_kernel (ms[0], n[0] launches) and of heavy_rank1_kernel (ms[1], n[1]) over the
// span the last kb_phase_times_ms / kb_kernel_time_ms call summed up (HIP events on the agent's stream, kb_set_kernel_timing)
extern "C" int kb_repair_times_ms(kb_handle* k, double ms[2], int64_t n[2]) {
    if (!k || !ms || !n) return RS_EINVAL;
    for (int q = 0; q < 2; ++q) {
        ms[q] = k->kind_ms[2 + q];
        n[q] = k->kind_n[2 + q];
    }
    return RS_OK;
}

extern "C" int kb_kernel_time_ms(kb_handle* k, double* avg_ms, int64_t* launches) {
    if (!k || !avg_ms) return RS_EINVAL;
    double ph[2];
    int64_t n[2];
    int rc = kb_phase_times_ms(k, ph, n);
    if (rc != RS_OK) return rc;
    const int64_t tot = n[0] + n[1];
    *avg_ms = tot ? (ph[0] * (double)n[0] + ph[1] * (double)n[1]) / (double)tot : 0.0;
    if (launches) *launches = tot;
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
    k->gemm_fresh = false;
    launch_shared_gemm(k, a.state);
    hipLaunchKernelGGL(kb::shared_scan_kernel, dim3((unsigned)k->T), dim3(64), 0, k->stream, a);
    if (e1) HIPCHK(k, hipEventRecord(e1, k->stream));
    hipLaunchKernelGGL(kb::shared_collect_kernel, dim3((unsigned)S), dim3(KB_RANK_THREADS), 0, k->stream, k->D, k->d_state, k->d_labels,
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
    k->gemm_fresh = false;
    launch_shared_apply(k, k->d_props, k->d_counts, (int)budget);
    HIPCHK(k, hipGetLastError());
    return kb_check(k);
}

extern "C" int kb_shared_commit(kb_handle* k, const int32_t* n_accept) {
    if (!k || !n_accept) return RS_EINVAL;
    if (!k->D.shared || !k->is_reset) return RS_ESTATE;
    HIPCHK(k, hipSetDevice(k->device));
    const size_t S = (size_t)k->cfg.n_slices;
    HIPCHK(k, hipMemcpyAsync(k->d_counts, n_accept, sizeof(int32_t) * S, hipMemcpyHostToDevice, k->stream));
    hipLaunchKernelGGL(kb::shared_commit_kernel, dim3((unsigned)S), dim3(KB_RANK_THREADS), 0, k->stream, k->D, k->d_cstar, k->d_counts,
                       k->d_cursor);
    HIPCHK(k, hipGetLastError());
    HIPCHK(k, hipStreamSynchronize(k->stream));
    return RS_OK;
}

// One learning step of the shared-dictionary mode, exchange included, with every buffer on the device:
//   round r:  scan -> collect into this rank's block -> ncclAllGather of the blocks (RCCL, agent stream) ->
//             merge by global replica id (first `budget` per slice) -> apply -> commit
// until no rank proposes anything or max_rounds is reached.  Only one 4-byte flag per round returns to the host
// (whether any rank still had proposals; not even that after the last permitted round when rounds_out is NULL).
// Without kb_comm_init the handle is its own world.
// the rounds of one shared learning step on device buffers (state / action / labels of the local replicas)
// Wait for the agent's stream.  With a communicator the wait is bounded: while the stream is busy the communicator is asked
// for an asynchronous error (a peer that died: ncclCommGetAsyncError) and a clock runs (KBRL_COLLECTIVE_TIMEOUT_S, default
// 120 s); either one aborts the communicator (ncclCommAbort) and returns RS_EHIP instead of leaving the rank in the
// collective for an outside watchdog to find.
static int shared_wait(kb_handle* k) {
    if (!k->comm) {
        HIPCHK(k, hipStreamSynchronize(k->stream));
        return RS_OK;
    }
    static const double limit = getenv("KBRL_COLLECTIVE_TIMEOUT_S") ? atof(getenv("KBRL_COLLECTIVE_TIMEOUT_S")) : 120.0;
    const auto t0 = std::chrono::steady_clock::now();
    const bool inject = dev_env("KBRL_INJECT_TIMEOUT") != nullptr;  // test build: as if the peers never answered
    for (;;) {
        const hipError_t q = inject ? hipErrorNotReady : hipStreamQuery(k->stream);
        if (q == hipSuccess) return RS_OK;
        const char* why = nullptr;
        int async_err = 0;
        if (inject) why = "timeout injected by KBRL_INJECT_TIMEOUT";
        else if (q != hipErrorNotReady) why = hipGetErrorString(q);
        else if (rccl::CommGetAsyncError && rccl::CommGetAsyncError(k->comm, &async_err) == 0 && async_err != 0)
            why = rccl::GetErrorString ? rccl::GetErrorString(async_err) : "asynchronous RCCL error";
        else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
            why = "no answer from the other ranks within KBRL_COLLECTIVE_TIMEOUT_S";
        if (why) {
            k->err = std::string("kb_shared_step: the exchange did not complete (") + why + "); communicator aborted";
            kb_comm_abort(k);  // (kb_comm_init forms a new one)
            return RS_EHIP;
        }
        usleep(50);
    }
}

// the rounds of one shared learning step on device buffers (state / action / labels of the local replicas)
static int shared_step_core(kb_handle* k, const float* d_state, const int32_t* d_action, const int32_t* d_labels, int32_t budget,
                            int32_t max_rounds, int32_t* hits_host, int32_t* rounds_out) {
    if (k->comm_aborted) {  // not a one-rank world: its dictionaries would silently diverge from the group's
        k->err = "kb_shared_step: the communicator of this handle was aborted; join a new one with kb_comm_init";
        return RS_ESTATE;
    }
    const size_t T = (size_t)k->T, S = (size_t)k->cfg.n_slices;
    const int W = k->comm ? k->comm_world : 1, me = k->comm ? k->comm_rank : 0;
    const size_t blk = S * (1 + (size_t)budget * KB_PROP_W);
    // test knob: behave as if this rank's round r had failed locally (the abort path below, without breaking anything)
    const int inject = dev_env("KBRL_INJECT_FAIL_ROUND") ? atoi(dev_env("KBRL_INJECT_FAIL_ROUND")) : -1;
    int rounds = 0;
    for (int rnd = 0; rnd < max_rounds; ++rnd) {
        // ---- this rank's part of the round.  A failure here (an allocation, a launch) must not leave the other ranks waiting
        // in the collective: the rank still takes part in the all-gather, with the failure mark in the place of its counts, and
        // every rank -- this one included -- learns of it from the merge kernel and returns RS_EHIP after the same round.
        std::string local_err;
        auto local = [&](hipError_t e, const char* what) {
            if (e != hipSuccess && local_err.empty()) local_err = std::string(what) + ": " + hipGetErrorString(e);
        };
        if (!k->d_gather)
            local(hipMalloc((void**)&k->d_gather, sizeof(double) * S * (1 + (size_t)k->budget_cap * KB_PROP_W) * (size_t)W), "hipMalloc of the gather buffer");
        kb::ScanArgs a;
        a.D = k->D;
        a.K = k->K;
        a.state = d_state;
        a.action = d_action;
        a.labels = d_labels;
        a.hits = k->d_hits;
        a.cursor = k->d_cursor;
        a.cstar = k->d_cstar;
        a.round = rnd;
        hipEvent_t e1 = nullptr;
        if (kb_time_begin(k, &e1) != RS_OK) local(hipErrorUnknown, "event for the kernel timing");
        // (resident loop, round 0: select_action of the previous step scored this very state against these very dictionaries)
        if (!(rnd == 0 && k->gemm_fresh && d_state == k->d_prev_state)) launch_shared_gemm(k, a.state);
        k->gemm_fresh = false;
        hipLaunchKernelGGL(kb::shared_scan_kernel, dim3((unsigned)k->T), dim3(64), 0, k->stream, a);
        if (e1) local(hipEventRecord(e1, k->stream), "hipEventRecord");
        hipLaunchKernelGGL(kb::shared_collect_block_kernel, dim3((unsigned)S), dim3(KB_RANK_THREADS), 0, k->stream, k->D, d_state,
                           d_labels, k->d_cstar, (int)budget, k->d_block);
        if (rnd == 0 && hits_host) local(hipMemcpyAsync(hits_host, k->d_hits, sizeof(int32_t) * T, hipMemcpyDeviceToHost, k->stream), "copy of the hits");
        local(hipGetLastError(), "launch of the scan kernels");
        if (rnd == inject) local_err = "failure injected by KBRL_INJECT_FAIL_ROUND";
        if (!local_err.empty() && k->d_gather) {
            static const double mark[KB_MAX_SLICES] = {-1.0, -1.0, -1.0, -1.0, -1.0, -1.0, -1.0, -1.0};
            (void)hipMemcpyAsync(k->d_block, mark, sizeof(double) * S, hipMemcpyHostToDevice, k->stream);
        }
        if (!k->d_gather) {  // nothing to gather into: this rank cannot even say so
            k->err = "kb_shared_step: " + local_err;
            if (k->comm) kb_comm_abort(k);
            return RS_EHIP;
        }
        if (k->comm) {
            const int nrc = rccl::AllGather(k->d_block, k->d_gather, blk, rccl::kDouble, k->comm, k->stream);
            if (nrc != 0) {
                k->err = std::string("ncclAllGather: ") + (rccl::GetErrorString ? rccl::GetErrorString(nrc) : "error");
                kb_comm_abort(k);
                return RS_EHIP;
            }
        } else {
            HIPCHK(k, hipMemcpyAsync(k->d_gather, k->d_block, sizeof(double) * blk, hipMemcpyDeviceToDevice, k->stream));
        }
        HIPCHK(k, hipMemsetAsync(k->d_total, 0, 2 * sizeof(int32_t), k->stream));
        hipLaunchKernelGGL(kb::shared_merge_kernel, dim3((unsigned)S), dim3(256), 0, k->stream, k->D, k->d_gather, W, me,
                           (int)budget, blk, k->d_mprops, k->d_mcounts, k->d_taken, k->d_total);
        launch_shared_apply(k, k->d_mprops, k->d_mcounts, (int)budget);
        hipLaunchKernelGGL(kb::shared_commit_kernel, dim3((unsigned)S), dim3(KB_RANK_THREADS), 0, k->stream, k->D, k->d_cstar, k->d_taken,
                           k->d_cursor);
        HIPCHK(k, hipGetLastError());
        // the last permitted round decides nothing: a single-rank handle whose caller does not ask for the count does not wait
        // for it (with other ranks in the step the failure flag is read after every round, so that all leave together)
        if (rnd + 1 == max_rounds && !rounds_out && !k->comm && local_err.empty()) break;
        // The pair comes back through PINNED memory: a device-to-host copy into pageable memory blocks the calling thread until
        // the stream reaches it -- behind an all-gather a silent peer never completes -- and the bounded wait below would never run.
        if (!k->h_total) HIPCHK(k, hipHostMalloc((void**)&k->h_total, 2 * sizeof(int32_t), hipHostMallocDefault));
        volatile int32_t* total = k->h_total;
        total[0] = total[1] = 0;
        HIPCHK(k, hipMemcpyAsync(k->h_total, k->d_total, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, k->stream));
        const int wrc = shared_wait(k);
        if (wrc != RS_OK) return wrc;
        if (total[1] != 0 || !local_err.empty()) {
            k->err = !local_err.empty() ? "kb_shared_step: this rank failed in round " + std::to_string(rnd) + " (" + local_err +
                                              "); the other ranks were told through the exchange and leave the step with it"
                                        : "kb_shared_step: another rank of the shared-dictionary group reported a failure in round " +
                                              std::to_string(rnd) + "; all ranks leave the step together";
            return RS_EHIP;
        }
        if (total[0] == 0) break;  // identical on every rank: all ranks leave together
        rounds = rnd + 1;
    }
    if (rounds_out) *rounds_out = rounds;
    return RS_OK;
}

// One learning step of the shared-dictionary mode, exchange included, with every buffer on the device:
//   round r:  scan -> collect into this rank's block -> ncclAllGather of the blocks (RCCL, agent stream) ->
//             merge by global replica id (first `budget` per slice) -> apply -> commit
// until no rank proposes anything or max_rounds is reached.  Only one 4-byte flag per round returns to the host
// (whether any rank still had proposals).  Without kb_comm_init the handle is its own world.
// Every argument is validated BEFORE the first collective, so a rank that refuses its input never leaves the others
// waiting in ncclAllGather.  A failure in the middle of the rounds travels with the exchange: the failing rank still
// contributes a block, marked, and every rank returns RS_EHIP after that round (shared_step_core); a rank that has stopped
// answering altogether is met by a bounded wait and ncclCommAbort (shared_wait).
extern "C" int kb_shared_step(kb_handle* k, const float* state, const int32_t* action, const int32_t* labels,
                              int32_t budget, int32_t max_rounds, int32_t* hits, int32_t* rounds_out) {
    if (!k || !state || !action || !labels || budget <= 0 || max_rounds <= 0) return RS_EINVAL;
    if (!k->D.shared || !k->is_reset) {
        k->err = "kb_shared_step: handle is not a reset shared-dictionary agent";
        return RS_ESTATE;
    }
    if (budget > k->budget_cap) {
        k->err = "kb_shared_step: budget too large (<= 256)";
        return RS_EINVAL;
    }
    HIPCHK(k, hipSetDevice(k->device));
    const size_t T = (size_t)k->T, N = (size_t)k->cfg.n_envs;
    for (size_t i = 0; i < T; ++i)
        if (action[i] < 0 || action[i] > k->cfg.n_prbs || (labels[i] != 1 && labels[i] != -1)) {
            k->err = "kb_shared_step: action out of [0, n_prbs] or label not +-1";
            return RS_EINVAL;
        }
    HIPCHK(k, hipMemcpyAsync(k->d_state, state, sizeof(float) * N * k->nv, hipMemcpyHostToDevice, k->stream));
    HIPCHK(k, hipMemcpyAsync(k->d_action, action, sizeof(int32_t) * T, hipMemcpyHostToDevice, k->stream));
    HIPCHK(k, hipMemcpyAsync(k->d_labels, labels, sizeof(int32_t) * T, hipMemcpyHostToDevice, k->stream));
    int rc = shared_step_core(k, k->d_state, k->d_action, k->d_labels, budget, max_rounds, hits, rounds_out);
    if (rc != RS_OK) return rc;
    return kb_check(k);
}

// kb_step_resident for the shared-dictionary agent (KBRL_Control.run body, kbrl_control.py:129-134, with the shared
// learning step in the place of update_control): learns from (previous obs, the action `env` just executed, its labels) in
// the simulator's own device buffers, then selects the next action into them.  Nothing but the per-round 4-byte flag
// crosses PCIe.
extern "C" int kb_shared_step_resident(kb_handle* k, rs_handle* env, int32_t budget, int32_t max_rounds, int32_t* rounds_out) {
    if (!k || !env || budget <= 0 || max_rounds <= 0) return RS_EINVAL;
    if (!k->D.shared || !k->is_reset || env->cfg.n_envs != k->cfg.n_envs || env->n_slices != k->cfg.n_slices ||
        env->n_vars != k->nv || env->device != k->device || budget > k->budget_cap) {
        k->err = "kb_shared_step_resident: agent and environment do not match (or not a reset shared-dictionary agent, or budget > 256)";
        return RS_EINVAL;
    }
    HIPCHK(k, hipSetDevice(k->device));
    if (env->hint_auto && !env->block_hint) {  // the next steps take agent-made allocations
        rs_set_schedule_hint(env, 1);
        env->hint_auto = true;
    }
    if (!k->ev_order) HIPCHK(k, hipEventCreateWithFlags(&k->ev_order, hipEventDisableTiming));
    hipEvent_t done = k->ev_order;
    HIPCHK(k, hipEventRecord(done, env->stream));
    HIPCHK(k, hipStreamWaitEvent(k->stream, done, 0));
    int rc = shared_step_core(k, k->d_prev_state, env->d_actions, env->d_labels, budget, max_rounds, nullptr, rounds_out);
    if (rc != RS_OK) return rc;
    rc = launch_select(k, env->d_obs, env->d_actions);
    if (rc != RS_OK) return rc;
    HIPCHK(k, hipMemcpyAsync(k->d_prev_state, env->d_obs, sizeof(float) * (size_t)k->cfg.n_envs * k->nv,
                             hipMemcpyDeviceToDevice, k->stream));
    HIPCHK(k, hipEventRecord(done, k->stream));
    HIPCHK(k, hipStreamWaitEvent(env->stream, done, 0));
    HIPCHK(k, hipGetLastError());
    k->gemm_fresh = true;  // workF / workE now describe d_prev_state; nothing learns before the next step's first scan
    return RS_OK;
}

extern "C" int kb_shared_merge(kb_handle* k, const double* gathered, int32_t world, int32_t me, int32_t budget, double* merged,
                               int32_t* counts, int32_t* taken, int32_t* total) {
    if (!k || !gathered || world < 1 || world > 64 || me < 0 || me >= world || budget <= 0 || budget > k->budget_cap)
        return RS_EINVAL;
    if (!k->D.shared) {
        k->err = "kb_shared_merge: handle is not a shared-dictionary agent";
        return RS_ESTATE;
    }
    HIPCHK(k, hipSetDevice(k->device));
    const size_t S = (size_t)k->cfg.n_slices, blk = S * (1 + (size_t)budget * KB_PROP_W);
    double* d_g = nullptr;
    HIPCHK(k, hipMalloc((void**)&d_g, sizeof(double) * blk * (size_t)world));
    int rc = RS_OK;
    hipError_t e = hipMemcpyAsync(d_g, gathered, sizeof(double) * blk * (size_t)world, hipMemcpyHostToDevice, k->stream);
    if (e == hipSuccess) e = hipMemsetAsync(k->d_total, 0, 2 * sizeof(int32_t), k->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kb::shared_merge_kernel, dim3((unsigned)S), dim3(256), 0, k->stream, k->D, d_g, (int)world, (int)me,
                           (int)budget, blk, k->d_mprops, k->d_mcounts, k->d_taken, k->d_total);
        e = hipGetLastError();
    }
    if (e == hipSuccess && merged)
        e = hipMemcpyAsync(merged, k->d_mprops, sizeof(double) * S * budget * KB_PROP_W, hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess && counts) e = hipMemcpyAsync(counts, k->d_mcounts, sizeof(int32_t) * S, hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess && taken) e = hipMemcpyAsync(taken, k->d_taken, sizeof(int32_t) * S, hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess && total) e = hipMemcpyAsync(total, k->d_total, sizeof(int32_t), hipMemcpyDeviceToHost, k->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(k->stream);
    (void)hipFree(d_g);
    if (e != hipSuccess) {
        k->err = std::string("kb_shared_merge: ") + hipGetErrorString(e);
        rc = RS_EHIP;
    }
    return rc;
}

// ------------------------------------------------------------------ histories of KBRL_Control.run, kept on the device
// kb_history_begin(steps): from now on every kb_step_resident records one column of reward / resources / hits /
// adjusted / SLA / violation per replica (kbrl_control.py:119-124,135-141), so that a whole run needs no per-step
// read-back; kb_history_fetch returns them ([n_envs][steps], hits [n_envs][S][steps]) and how many were recorded.
extern "C" int kb_history_begin(kb_handle* k, int32_t steps) {
    if (!k || steps <= 0) return RS_EINVAL;
    kb_drop_graph(k);  // (the captured loop carries the history kernels and their buffers)
    HIPCHK(k, hipSetDevice(k->device));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    kb_history_release(k);
    const size_t N = (size_t)k->cfg.n_envs, S = (size_t)k->cfg.n_slices, n = N * (size_t)steps;
    HIPCHK(k, hipMalloc((void**)&k->h_reward, sizeof(double) * n));
    HIPCHK(k, hipMalloc((void**)&k->h_resources, sizeof(int16_t) * n));
    HIPCHK(k, hipMalloc((void**)&k->h_hits, sizeof(int16_t) * n * S));
    HIPCHK(k, hipMalloc((void**)&k->h_adjusted, sizeof(int16_t) * n));
    HIPCHK(k, hipMalloc((void**)&k->h_sla, sizeof(int16_t) * n));
    HIPCHK(k, hipMalloc((void**)&k->h_violation, sizeof(int16_t) * n));
    HIPCHK(k, hipMalloc((void**)&k->h_cursor, sizeof(int32_t)));
    HIPCHK(k, hipMemsetAsync(k->h_reward, 0, sizeof(double) * n, k->stream));
    HIPCHK(k, hipMemsetAsync(k->h_resources, 0, sizeof(int16_t) * n, k->stream));
    HIPCHK(k, hipMemsetAsync(k->h_hits, 0, sizeof(int16_t) * n * S, k->stream));
    HIPCHK(k, hipMemsetAsync(k->h_adjusted, 0, sizeof(int16_t) * n, k->stream));
    HIPCHK(k, hipMemsetAsync(k->h_sla, 0, sizeof(int16_t) * n, k->stream));
    HIPCHK(k, hipMemsetAsync(k->h_violation, 0, sizeof(int16_t) * n, k->stream));
    HIPCHK(k, hipMemsetAsync(k->h_cursor, 0, sizeof(int32_t), k->stream));
    k->h_steps = steps;
    return RS_OK;
}

extern "C" int kb_history_fetch(kb_handle* k, double* reward, int16_t* resources, int16_t* hits, int16_t* adjusted,
                                int16_t* sla, int16_t* violation, int32_t* n_recorded) {
    if (!k) return RS_EINVAL;
    if (k->h_steps <= 0) {
        k->err = "kb_history_fetch: call kb_history_begin first";
        return RS_ESTATE;
    }
    HIPCHK(k, hipSetDevice(k->device));
    const size_t N = (size_t)k->cfg.n_envs, S = (size_t)k->cfg.n_slices, n = N * (size_t)k->h_steps;
    if (reward) HIPCHK(k, hipMemcpyAsync(reward, k->h_reward, sizeof(double) * n, hipMemcpyDeviceToHost, k->stream));
    if (resources) HIPCHK(k, hipMemcpyAsync(resources, k->h_resources, sizeof(int16_t) * n, hipMemcpyDeviceToHost, k->stream));
    if (hits) HIPCHK(k, hipMemcpyAsync(hits, k->h_hits, sizeof(int16_t) * n * S, hipMemcpyDeviceToHost, k->stream));
    if (adjusted) HIPCHK(k, hipMemcpyAsync(adjusted, k->h_adjusted, sizeof(int16_t) * n, hipMemcpyDeviceToHost, k->stream));
    if (sla) HIPCHK(k, hipMemcpyAsync(sla, k->h_sla, sizeof(int16_t) * n, hipMemcpyDeviceToHost, k->stream));
    if (violation) HIPCHK(k, hipMemcpyAsync(violation, k->h_violation, sizeof(int16_t) * n, hipMemcpyDeviceToHost, k->stream));
    int32_t cur = 0;
    HIPCHK(k, hipMemcpyAsync(&cur, k->h_cursor, sizeof cur, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    if (n_recorded) *n_recorded = cur < k->h_steps ? cur : k->h_steps;
    return kb_check(k);
}

// ------------------------------------------------------------------ checkpoint / restore of the agents (with rs_save_state: a
// long evaluation can be cut and resumed; the reference has no counterpart)
// Everything behind the handle: the per-learner tables, the control state, the queues, and of the dictionary pool the part in
// use (offsets inside the pool are relative, so a blob fits any handle of the same configuration and pool size).  The
// histories of kb_history_begin travel too.
struct kb_state_header {
    uint64_t magic, n_regions, cfg_hash, pool_doubles_used, hist_steps, total_bytes;
    int32_t big_par, is_reset, seen0, seen1;
};
static const uint64_t kKbStateMagic = 0x4b42534c49434535ull;      // "KBSLICE5": the configuration hash no longer covers the pool's size
static const uint64_t kKbStateMagicOld = 0x4b42534c49434534ull;   // "KBSLICE4" (rounds 4-5 before that change): refused by name
static uint64_t kb_cfg_hash(const kb_handle* k) {
    uint64_t x = 1469598103934665603ull;
    kb_config c = k->cfg;
    c.pool_bytes = 0;  // (see below: the pool's size is not part of the configuration)
    const unsigned char* p = (const unsigned char*)&c;
    for (size_t i = 0; i < sizeof c; ++i) x = (x ^ p[i]) * 1099511628211ull;
    // (the pool's own size is not part of the configuration: a blob fits any handle whose pool holds what the blob uses --
    // with kb_config.pool_bytes == 0 the pool is sized from the free memory of the moment and differs from process to process)
    for (auto& r : k->regions) x = (x ^ (r.first == (void*)k->K.pool ? 0ull : (uint64_t)r.second)) * 1099511628211ull;
    return x;
}
static size_t kb_hist_bytes(const kb_handle* k, size_t part[7]) {
    const size_t n = (size_t)k->cfg.n_envs * (size_t)k->h_steps, S = (size_t)k->cfg.n_slices;
    const size_t b[7] = {sizeof(double) * n, 2 * n, 2 * n * S, 2 * n, 2 * n, 2 * n, sizeof(int32_t)};
    size_t t = 0;
    for (int i = 0; i < 7; ++i) {
        part[i] = k->h_steps > 0 ? b[i] : 0;
        t += part[i];
    }
    return t;
}
static int kb_pool_used(kb_handle* k, unsigned long long* top) {
    HIPCHK(k, hipMemcpy(top, k->K.pool_top, sizeof *top, hipMemcpyDeviceToHost));
    if (*top > k->D.pool_doubles) *top = k->D.pool_doubles;
    return RS_OK;
}
extern "C" int kb_state_bytes(kb_handle* k, uint64_t* bytes) {
    if (!k || !bytes) return RS_EINVAL;
    HIPCHK(k, hipSetDevice(k->device));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    unsigned long long top = 0;
    int rc = kb_pool_used(k, &top);
    if (rc != RS_OK) return rc;
    size_t part[7];
    uint64_t t = sizeof(kb_state_header) + kb_hist_bytes(k, part);
    for (auto& r : k->regions) t += r.first == (void*)k->K.pool ? (uint64_t)top * 8 : (uint64_t)r.second;
    *bytes = t;
    return RS_OK;
}
extern "C" int kb_save_state(kb_handle* k, void* blob, uint64_t bytes) {
    uint64_t need = 0;
    if (!k || !blob) return RS_EINVAL;
    int rc = kb_state_bytes(k, &need);
    if (rc != RS_OK) return rc;
    if (bytes < need) {
        k->err = "kb_save_state: buffer smaller than kb_state_bytes";
        return RS_EINVAL;
    }
    unsigned long long top = 0;
    if ((rc = kb_pool_used(k, &top)) != RS_OK) return rc;
    kb_state_header hd = {kKbStateMagic, (uint64_t)k->regions.size(), kb_cfg_hash(k), (uint64_t)top, (uint64_t)k->h_steps, need,
                          k->big_par, k->is_reset ? 1 : 0, k->h_seen ? k->h_seen[0] : 0, k->h_seen ? k->h_seen[1] : 0};
    memcpy(blob, &hd, sizeof hd);
    char* o = (char*)blob + sizeof hd;
    for (auto& r : k->regions) {
        const size_t b = r.first == (void*)k->K.pool ? (size_t)top * 8 : r.second;
        HIPCHK(k, hipMemcpy(o, r.first, b, hipMemcpyDeviceToHost));
        o += b;
    }
    size_t part[7];
    (void)kb_hist_bytes(k, part);
    void* hp[7] = {k->h_reward, k->h_resources, k->h_hits, k->h_adjusted, k->h_sla, k->h_violation, k->h_cursor};
    for (int i = 0; i < 7; ++i)
        if (part[i]) {
            HIPCHK(k, hipMemcpy(o, hp[i], part[i], hipMemcpyDeviceToHost));
            o += part[i];
        }
    return RS_OK;
}
extern "C" int kb_load_state(kb_handle* k, const void* blob, uint64_t bytes) {
    if (!k || !blob || bytes < sizeof(kb_state_header)) return RS_EINVAL;
    kb_state_header hd;
    memcpy(&hd, blob, sizeof hd);
    if (hd.magic == kKbStateMagicOld) {
        k->err = "kb_load_state: older checkpoint format (KBSLICE4: its configuration hash covered the pool's size); re-create the "
                 "checkpoint with this build";
        return RS_EINVAL;
    }
    if (hd.magic != kKbStateMagic || hd.n_regions != k->regions.size() || hd.cfg_hash != kb_cfg_hash(k)) {
        k->err = "kb_load_state: the blob was not saved by a handle of this configuration";
        return RS_EINVAL;
    }
    if (hd.pool_doubles_used > k->D.pool_doubles) {
        k->err = "kb_load_state: the blob's dictionaries use " + std::to_string(hd.pool_doubles_used * 8) + " bytes of pool, this handle's pool has " +
                 std::to_string((uint64_t)k->D.pool_doubles * 8) + " (give kb_config.pool_bytes explicitly when checkpoints travel between processes)";
        return RS_EINVAL;
    }
    {   // the size the header's own fields imply: a truncated or corrupt blob is refused before anything is read past its end
        const uint64_t n = (uint64_t)k->cfg.n_envs * hd.hist_steps, S = (uint64_t)k->cfg.n_slices;
        uint64_t expect = sizeof(kb_state_header) + (hd.hist_steps ? 8 * n + 2 * n + 2 * n * S + 2 * n + 2 * n + 2 * n + sizeof(int32_t) : 0);
        for (auto& r : k->regions) expect += r.first == (void*)k->K.pool ? hd.pool_doubles_used * 8 : (uint64_t)r.second;
        if (hd.hist_steps > 0x7fffffffull || hd.total_bytes != expect || bytes < expect) {
            k->err = "kb_load_state: truncated or corrupt blob (" + std::to_string(bytes) + " bytes given, header says " +
                     std::to_string(hd.total_bytes) + ", its fields imply " + std::to_string(expect) + ")";
            return RS_EINVAL;
        }
    }
    HIPCHK(k, hipSetDevice(k->device));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    kb_drop_graph(k);
    if (hd.hist_steps != (uint64_t)k->h_steps) {
        int rc = hd.hist_steps ? kb_history_begin(k, (int32_t)hd.hist_steps) : RS_OK;
        if (rc != RS_OK) return rc;
        if (!hd.hist_steps) kb_history_release(k);
        HIPCHK(k, hipStreamSynchronize(k->stream));
    }
    const char* o = (const char*)blob + sizeof hd;
    for (auto& r : k->regions) {
        const size_t b = r.first == (void*)k->K.pool ? (size_t)hd.pool_doubles_used * 8 : r.second;
        HIPCHK(k, hipMemcpy(r.first, o, b, hipMemcpyHostToDevice));
        o += b;
    }
    size_t part[7];
    (void)kb_hist_bytes(k, part);
    void* hp[7] = {k->h_reward, k->h_resources, k->h_hits, k->h_adjusted, k->h_sla, k->h_violation, k->h_cursor};
    for (int i = 0; i < 7; ++i)
        if (part[i]) {
            HIPCHK(k, hipMemcpy(hp[i], o, part[i], hipMemcpyHostToDevice));
            o += part[i];
        }
    if (hd.pool_doubles_used < (uint64_t)k->D.pool_doubles) {
        // "the pool was exhausted" (err bit 16) described the pool the blob came from; this one has room again
        std::vector<int32_t> e((size_t)k->cfg.n_envs);
        HIPCHK(k, hipMemcpy(e.data(), k->K.err, sizeof(int32_t) * e.size(), hipMemcpyDeviceToHost));
        bool any = false;
        for (auto& v : e) {
            any = any || (v & 16);
            v &= ~16;
        }
        if (any) HIPCHK(k, hipMemcpy(k->K.err, e.data(), sizeof(int32_t) * e.size(), hipMemcpyHostToDevice));
    }
    k->big_par = hd.big_par;
    k->is_reset = hd.is_reset != 0;
    if (k->h_seen) {
        k->h_seen[0] = hd.seen0;
        k->h_seen[1] = hd.seen1;
    }
    k->gemm_fresh = false;
    return RS_OK;
}

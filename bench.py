#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched RanSlice.step on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: RanSlice.step for 4096 env replicas per GPU
(scenario_0: 200 PRBs, 5 eMBB slices, 50 slots per step; BASELINE.json configs[1]) with random
actions generated on the device.  Inputs (fading tables, simulator state, actions) are resident
in HBM when the timed region starts; nothing crosses PCIe inside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: spawns its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

No torch here: the library's own rs_device_count / rs_synchronize, and for N > 1 a plain TCP group on 127.0.0.1
(tools/rank_group.py: barrier, MAX of the elapsed time) found through RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT,
whoever launched the ranks.  Replicas are independent, so GPUs shard them with no data-path collective (weak scaling:
4096 replicas per GPU).

OUTPUT.  The LAST stdout line is the compact record (< 4 KB): metric, value, ms_per_step, config, roofline (with the
kernel's real limiter beside the HBM figure), cpu_baseline, and the summaries `kbrl` (config 3) and `shared_kbrl` (config 4:
the RCCL exchange).  The full record (every sub-record at length) goes to profiles/bench_full_last.json and is printed on an
EARLIER line that starts with "# full record: ".

Before the W warm-up steps the environments are advanced until the UE population is stationary: blocks of 500
steps (25 s of simulated time; the holding times have a 30 s mean) until the mean number of UEs per slice changes
by less than 0.5 % from one block to the next (every rank runs the same number of blocks).  This is environment
set-up, not part of the timed work; the line reports the steps it took and the population.

  roofline     : algorithmic bytes per launch of the dominant kernel (embb_step_kernel) divided by its mean launch
                 duration measured with HIP events on the launch stream, against the 8 TB/s HBM peak (DESIGN.md section 4
                 states the byte model).  `traffic`: HBM bytes per launch from a separate rocprofv3 --pmc run of this
                 command at the same UE population (profiles/hbm_traffic.json; null when the populations differ);
                 `limiter`: what actually bounds the kernel (VALU issue slots), from the SQ counters of the same run;
  cpu_baseline : the CPU oracle (a C port of the reference's numpy path, pinned bit-exact to the reference on golden
                 tapes) timed on this box's host cores on a bounded sample of the same workload -- a reported
                 baseline, never the thing measured above;
  kbrl         : (1 GPU) BASELINE config 3 -- the same 4096 replicas with one KBRL agent each, closed loop on the
                 device, at two points of learning: steps 100-300 and from step 3000 (`late_value`); HBM fractions of
                 the kernels that stream Kinv and the landmarks, MFMA work of select_action, the pool's horizon;
  shared_kbrl  : BASELINE config 4 -- scenario index 2, 4096 replicas per GPU, ONE dictionary per slice shared by all replicas
                 of all ranks, the exchange = ncclAllGather inside libranslice.so (kb_shared_step_resident).  Runs in a child
                 process per rank under a timeout: its failure cannot take the headline with it.  `rccl_ranks` is what the
                 communicator itself reports (kb_comm_info).
  python bench.py --scaling 1,2,4,8 prints ONE line with the curve over N and the CPU baseline.
"""
import argparse
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'network-slicing_amd'), os.path.join(ROOT, 'tools')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

ENVS_PER_GPU = 4096
SCENARIO = 0
FADING_COLS = 10000
ACTION_SEED = 2024
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
F64_PEAK_TFLOPS = 78.6     # MI355X f64 vector / matrix peak
BURN_BLOCK = 500
BURN_MAX = 8000

# bytes of persistent simulator state the step kernel reads and writes per task / per active UE (DESIGN.md §3,
# network-slicing_amd/csrc/rs_device.h): task header 6 x 4 B; UE 3 f64 + 9 i32 + 16 u16 burst end-time codes = 92 B
STATE_TASK_BYTES = 6 * 4
STATE_UE_BYTES = 3 * 8 + 9 * 4 + 16 * 2


def _cpu_worker(args):
    (rank, cfg_kw, burn, timed, barrier_wait, replicas) = args
    from oracle import pyoracle as po
    from ranslice.config import make_config
    from ranslice.fading import synth_fading
    from ranslice.sharding import replica_seed
    cfg = make_config(SCENARIO, n_envs=1, **cfg_kw)
    fading = [synth_fading(t, FADING_COLS) for t in range(3)]
    envs = []
    for r in replicas:
        o = po.OracleEnv(cfg, fading)
        o.set_seed(replica_seed(0, r))
        o.reset()
        o.bench_run(ACTION_SEED, r, 0, burn)
        envs.append((r, o))
    barrier_wait()
    t0 = time.time()
    for r, o in envs:
        chk = o.bench_run(ACTION_SEED, r, burn, timed)
        assert chk == chk, 'oracle reported an error'
    t1 = time.time()
    return (t0, t1, timed * len(envs))


_BARRIER = None


def _barrier_wait():
    _BARRIER.wait()


def _init_pool(b):
    global _BARRIER
    _BARRIER = b


def host_cores():
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # respect a cgroup CPU quota (containers): cpu.max = "<quota> <period>"
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, per = f.read().split()
        if q != 'max':
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    return cores


def cpu_baseline(burn, timed):
    cores = host_cores()
    ctx = mp.get_context('fork')
    barrier = ctx.Barrier(cores)
    with ctx.Pool(cores, initializer=_init_pool, initargs=(barrier,)) as pool:
        jobs = [(i, {}, burn, timed, _barrier_wait, [i]) for i in range(cores)]
        res = pool.map(_cpu_worker, jobs, chunksize=1)
        t0 = min(r[0] for r in res)
        t1 = max(r[1] for r in res)
        total = sum(r[2] for r in res)
    # BASELINE.md section 3 also asks for the single-core number: one replica, one process, the other cores idle
    b1 = ctx.Barrier(1)
    with ctx.Pool(1, initializer=_init_pool, initargs=(b1,)) as pool:
        one = pool.map(_cpu_worker, [(0, {}, burn, timed, _barrier_wait, [0])], chunksize=1)[0]
    return dict(value=total / (t1 - t0), unit='env-steps/s', cores=cores, kind='port',
                single_core_value=one[2] / (one[1] - one[0]), steps=timed, burn_in=burn,
                sample='%d replicas (one per host core) x %d steps of the same scenario_0 workload and action '
                       'script after a %d-step burn-in; C oracle (oracle/rs_oracle.c), 1 thread per replica; '
                       'single_core_value: one replica alone, same steps' % (cores, timed, burn))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


KBRL_CAPACITY = 4096          # a limit, not a reservation: dictionaries take their storage from the pool as they grow
KBRL_POOL_HEADROOM = 24 << 30   # the pool takes the device's free memory less this (ranslice._lib.default_pool_bytes)
KBRL_LATE_STEP = 3000         # second measurement point: dictionaries of several hundred landmarks
SEL_TILE_ROWS = 16            # select_gemm_kernel: v_mfma_f64_16x16x4 tiles of 16 candidates x 16 learners, 4 grid indices per instruction


def kbrl_record(n_envs, device, steps, warmup, late_step=KBRL_LATE_STEP, profile='tdl'):
    """BASELINE config 3 on this GPU: closed loop with one KBRL agent per replica (kb_step_resident), measured twice:
    early in learning (steps `warmup`..`warmup + steps`, the point of rounds 1-2) and from step `late_step` on, where the
    dictionaries hold what a run of that length holds.  `value` is the LATE number (it falls with run length: the step is
    stated)."""
    import ctypes as C
    from ranslice.config import make_config, EMBB_A, EMBB_SEC
    from ranslice.fading import synth_traces
    from ranslice.kbrl_dev import VecKBRL
    from ranslice.vec_env import VecRanSlice
    from ranslice import _lib
    cap = int(os.environ.get('KBRL_CAPACITY', KBRL_CAPACITY))
    cfg = make_config(SCENARIO, n_envs=n_envs)
    env = VecRanSlice(n_envs=n_envs, cfg=cfg, fading=synth_traces(FADING_COLS, profile), device=device)
    pool_bytes = _lib.default_pool_bytes(device, headroom=KBRL_POOL_HEADROOM)   # sized from hipMemGetInfo
    agent = VecKBRL(n_envs, [10] * cfg.n_embb, cfg.n_prbs, accuracy_range=(0.99, 0.999), capacity=cap, device=device,
                    pool_bytes=pool_bytes)
    rng = np.random.default_rng(0)
    ia = rng.integers(EMBB_A[0], EMBB_A[1], size=(n_envs, cfg.n_embb)).astype(np.int32)   # scenario_creator.py:220-221
    sf = rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(n_envs, cfg.n_embb)).astype(np.int32)
    env.reset()
    agent.reset(ia, sf)
    env._check(env.L.rs_step(env.h, ia.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
    done = [0]
    n_learners = n_envs * cfg.n_embb

    def run(k):
        for _ in range(k):
            agent.step_resident(env)
            env.step_resident()
        done[0] += k
    d = 11   # eMBB learner: 10 state variables + the candidate allocation
    # select_gemm_kernel per launch: workgroups of 16 learners (launch slots: the learners + the 4096 places of the
    # large-learner list), 4 waves x 4 candidate tiles x (n_prbs + 4) / 4 instructions each
    sel_blocks = (n_learners + 4096 + 15) // 16
    mfma_per_launch = sel_blocks * 4 * 4 * ((cfg.n_prbs + 4) // 4)

    def point(k):
        # (1) the throughput: k closed-loop steps enqueued by ONE call and replayed from a captured hipGraph (kb_run_resident),
        # no event records in the stream; (2) the account: the next k steps one launch sequence per step with HIP events
        # around the phases and around every launch of the two Kinv-streaming kernels
        env.synchronize()
        agent.synchronize()
        first = done[0]
        t0 = time.perf_counter()
        agent.run_resident(env, k, graph=True)
        done[0] += k
        env.synchronize()
        agent.synchronize()          # raises on a device-side error flag
        dt = time.perf_counter() - t0
        s0 = agent.stats()
        w0 = agent.repair_work()
        p0 = agent.pool()
        agent.set_kernel_timing(True)
        env.set_kernel_timing(True)
        t1 = time.perf_counter()
        run(k)
        env.synchronize()
        agent.synchronize()
        dt_events = time.perf_counter() - t1
        s1 = agent.stats()
        ph = agent.phase_times_ms()
        w1 = agent.repair_work()
        env_ms, _ = env.kernel_time_ms()
        agent.set_kernel_timing(False)
        env.set_kernel_timing(False)
        sizes = agent.dictionary_sizes()
        pool = agent.pool()
        evals = s1[3] - s0[3]

        def stream_roof(kind, ms_key, n_key):
            # ALGORITHMIC bytes (Kinv tiles the work plan of the launch lists: projectron.py:42 reads them, :54-58 reads and
            # rewrites them) over the duration of all launches of the kernel in the window (HIP events on the agent's stream;
            # the later rounds of a step often find nothing left: their launches are in the time), against the HBM peak
            nw = w1[kind + '_launches'] - w0[kind + '_launches']
            if not nw or not ph[ms_key]:
                return None
            total = float(w1[kind + '_bytes'] - w0[kind + '_bytes'])
            ms = ph[ms_key] * ph[n_key]
            gbs = total / (ms * 1e-3) / 1e9
            return {'bound': 'hbm', 'kernel': 'heavy_%s_kernel' % kind, 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': gbs / HBM_PEAK_GBS, 'bytes_per_launch': total / nw, 'bytes_per_step': total / k,
                    'launches_with_work': nw, 'launches_timed': ph[n_key], 'launch_ms_mean': ph[ms_key], 'ms_per_step': ms / k}
        rec = {
            'steps': [first, first + k], 'value': n_envs * k / dt, 'unit': 'env-steps/s', 'ms_per_step': 1e3 * dt / k,
            'loop': 'kb_run_resident: %d steps per call, two captured steps replayed as a hipGraph' % k,
            'account_steps': [first + k, first + 2 * k], 'ms_per_step_with_event_records': 1e3 * dt_events / k,
            'embb_kernel_ms': env_ms, 'kb_update_phase_ms': ph['update_ms'], 'kb_select_ms': ph['select_ms'],
            'kernel_evaluations_per_s': evals / dt_events,
            'predicts_per_env_step': (s1[0] - s0[0]) / (n_envs * k),
            'mistakes_per_env_step': (s1[1] - s0[1]) / (n_envs * k),
            # the reference's cost model of the same predictions ((3 d + 3) flop per landmark and candidate),
            # kept for comparison with rounds 1-3; the build forms them from W[a] (one pass over the landmarks) and a
            # 16 x 204 x 16 product per 16 candidates x 16 learners on the matrix cores
            'direct_model_tflops': evals * (3 * d + 3) / dt_events / 1e12,
            'select_mfma': {'instructions_per_launch': mfma_per_launch, 'flop_per_launch': mfma_per_launch * 2048,
                            'select_phase_ms': ph['select_ms'],
                            'tflops_over_the_select_phase': mfma_per_launch * 2048 / (ph['select_ms'] * 1e-3) / 1e12 if ph['select_ms'] else None,
                            'peak_tflops_f64': F64_PEAK_TFLOPS,
                            'note': 'the select phase is select_bin_kernel (one pass over every landmark: HBM) + '
                                    'select_gemm_kernel (the MFMA product); rocprofv3 counters of the latter in profiled_counters'},
            'direct_exponential_passes_per_step': (w1['direct_passes'] - w0['direct_passes']) / k,
            'dictionary_size_mean': float(np.mean(sizes)), 'dictionary_size_max': int(np.max(sizes)),
            'dictionary_size_p50_p90_p99': [float(np.percentile(sizes, q)) for q in (50, 90, 99)],
            'pool': pool,
            'kinv_streaming': {'rank1': stream_roof('rank1', 'rank1_launch_ms', 'n_rank1'),
                               'matvec': stream_roof('matvec', 'matvec_launch_ms', 'n_matvec')},
            'per_step_ms': {'heavy_matvec': ph['matvec_launch_ms'] * ph['n_matvec'] / k, 'heavy_rank1': ph['rank1_launch_ms'] * ph['n_rank1'] / k,
                            'heavy_finish': ph['finish_launch_ms'] * ph['n_finish'] / k, 'update_small': ph['update_small_launch_ms'] * ph['n_update_small'] / k,
                            'select_bin': ph['select_bin_launch_ms'] * ph['n_select_bin'] / k,
                            'select_gemm': ph['select_gemm_launch_ms'] * ph['n_select_gemm'] / k},
        }
        # select_bin_kernel: ONE pass over every landmark of every learner -- ten coordinate rows, the coefficient and the grid index
        # in (52 B per landmark of an eMBB learner since the coordinates come as float32; 92 B before), the D0 and E rows out (16 B); bytes
        # from the dictionary sizes at the window's end
        if ph['select_bin_launch_ms']:
            # (round 6: the ten state coordinates from their float32 copy, 4 B each)
            sb_bytes = float(np.sum(sizes)) * (4.0 * (d - 1) + 8.0 + 4.0 + 16.0)
            gbs = sb_bytes / (ph['select_bin_launch_ms'] * 1e-3) / 1e9
            rec['select_bin'] = {'bound': 'hbm', 'kernel': 'binning pass of select_action: select_bin_big_kernel + select_bin_kernel (+ big_list_kernel), '
                                 'one HIP-event bracket', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                 'frac': gbs / HBM_PEAK_GBS, 'bytes_per_launch': sb_bytes, 'launch_ms_mean': ph['select_bin_launch_ms'],
                                 'landmarks': int(np.sum(sizes))}
        # the memory horizon: the pool never frees before kb_reset; at the growth of this window it is exhausted at ...
        grow = (pool['used_bytes'] - p0['used_bytes']) / float(k)
        if grow > 0:
            left = (pool['total_bytes'] - pool['used_bytes']) / grow
            rec['pool_horizon'] = {'bytes_per_step': grow, 'steps_left_at_this_rate': left, 'exhausted_near_step': first + 2 * k + left,
                                   'note': 'linear extrapolation of this window; Kinv grows with the square of a dictionary, so a '
                                           'lower bound on the rate and an upper bound on the step.  Past it dictionaries that '
                                           'ask for a shell project instead of growing and their replicas are flagged '
                                           '(kb_get_pool / kb_get_flags; tests/test_gpu_kbrl.py::test_pool_exhaustion_at_batch_size_vs_oracle)'}
        return rec
    def skip(k):     # steps between the measurement points: the graph loop
        agent.run_resident(env, k, graph=True)
        done[0] += k
    skip(warmup)
    early = point(steps)
    late = None
    if late_step and late_step > done[0]:
        skip(late_step - done[0])
        late = point(steps)
    head = late or early
    rec = {
        'workload': 'scenario_0, %d replicas + one KBRL agent per replica, closed loop on the device (%s traces)'
                    % (n_envs, profile),
        'traces': profile, 'dictionary_capacity': cap, 'pool_bytes': pool_bytes,
        'value': head['value'], 'unit': 'env-steps/s', 'ms_per_step': head['ms_per_step'], 'value_at_steps': head['steps'],
        'roofline': (head['kinv_streaming'] or {}).get('rank1'),
        'early': early, 'late': late,
        'scoring': 'f(c) = sum_a G[|a - c|] W[a], W[a] = sum of coeff_j E_j over the landmarks of grid index a: one pass over the '
                   'landmarks per state, then T W for 16 learners at a time on v_mfma_f64_16x16x4 (select_gemm_kernel); '
                   'update_control starts from the scores select_action left (DESIGN.md section 4 KBRL)',
        'peak_tflops_f64': F64_PEAK_TFLOPS,
    }
    # the reference's run length (experiments_kbrl.py:22: 50,400 steps): how many replicas of one GPU get there without a flagged
    # dictionary -- measured by tools/run_length.py, committed under profiles/
    rpath = os.path.join(ROOT, 'profiles', 'run_length.json')
    if os.path.exists(rpath):
        try:
            with open(rpath) as f:
                rec['reference_run_length'] = json.load(f)
        except Exception:
            pass
    ppath = os.path.join(ROOT, 'profiles', 'kbrl_mfma_share.json')
    if os.path.exists(ppath):
        try:
            with open(ppath) as f:
                rec['profiled_counters'] = json.load(f)
        except Exception:
            pass
    env.close()
    agent.close()
    return rec


def run_scaling(args):
    """--scaling 1,2,4,8: the whole report from one command -- the CPU baseline once, then this script once per N (each
    an independent launch), one JSON line with the curve.  Ns beyond the devices of the box are listed as skipped."""
    from ranslice import _lib
    ns = [int(x) for x in args.scaling.split(',') if x]
    ndev = _lib.device_count()
    cpu = None if args.no_cpu_baseline else cpu_baseline(1000, args.cpu_steps)
    curve = []
    for n in ns:
        if n > ndev:
            curve.append({'n_gpus': n, 'skipped': 'the box has %d GPU(s)' % ndev})
            continue
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', str(n), '--steps', str(args.steps), '--warmup',
               str(args.warmup), '--envs-per-gpu', str(args.envs_per_gpu), '--no-cpu-baseline', '--no-kbrl']
        if n == 1 or args.no_shared:
            cmd += ['--no-shared']
        if args.burn_in >= 0:
            cmd += ['--burn-in', str(args.burn_in)]
        if args.graph:
            cmd += ['--graph']
        env = dict(os.environ)
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
            env.pop(k, None)
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        lines = [x for x in out.stdout.splitlines() if x.startswith('{')]
        if out.returncode != 0 or not lines:
            curve.append({'n_gpus': n, 'error': 'rc %d' % out.returncode})
            continue
        line = json.loads(lines[-1])
        pt = {'n_gpus': n, 'value': line['value'], 'ms_per_step': line['ms_per_step'],
              'roofline_frac': line['roofline']['frac'], 'kernel_ms': line['roofline']['kernel_ms'],
              'global_envs': line['config']['global_envs']}
        sh = line.get('shared_kbrl')
        if sh:
            pt['shared_kbrl'] = {k: sh.get(k) for k in ('value', 'ms_per_step', 'rccl_ranks', 'error') if k in sh}
        curve.append(pt)
    base = next((c['value'] for c in curve if c.get('n_gpus') == 1 and 'value' in c), None)
    for c in curve:
        if base and 'value' in c:
            c['per_gpu_vs_1gpu'] = c['value'] / c['n_gpus'] / base
    print(json.dumps({'metric': 'env-steps/sec (batched RanSlice.step, scenario_0)', 'unit': 'env-steps/s',
                      'scaling': 'weak', 'curve': curve, 'cpu_baseline': compact_cpu(cpu), 'dtype': 'f64', 'data': 'synthetic'}),
          flush=True)


# ---------------------------------------------------------------------------------------------- config 4: the RCCL leg
SHARED_SCENARIO = 2           # BASELINE config 4: scenario index 2 (100 PRBs, 1 eMBB + 4 mMTC slices)
SHARED_BUDGET = 256           # proposals per slice and exchange round
SHARED_CAPACITY = 1024        # landmarks per shared dictionary (config 3/4 capacity)
SHARED_TIMEOUT_S = 300
SHARED_FILL_WARMUP = 30       # the first window of the leg starts here: the eMBB dictionary is still filling


def shared_leg(args):
    """One rank of BASELINE config 4 (child process of a bench rank): 4096 scenario_2 replicas on this rank's GPU, ONE KBRL
    dictionary per slice shared by the replicas of all ranks, closed loop on the device (kb_shared_step_resident).  The exchange
    is ncclAllGather inside libranslice.so; this script only hands the 128-byte communicator id from rank 0 to the others.
    Rank 0 prints one JSON line."""
    import ctypes as C
    from rank_group import RankGroup
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (('RANK', 0), ('LOCAL_RANK', 0), ('WORLD_SIZE', 1)))
    from ranslice import _lib
    from ranslice.config import make_config, EMBB_A, EMBB_SEC, MMTC_A, MMTC_SEC
    from ranslice.fading import synth_fading
    from ranslice.kbrl_dev import SharedVecKBRL
    from ranslice.sharding import shard_range, replica_seeds
    from ranslice.vec_env import VecRanSlice
    device = local_rank % _lib.device_count() if os.environ.get('RANSLICE_BENCH_SHARE_GPU') == '1' else local_rank
    group = RankGroup(rank, world, timeout=120.0)
    N = args.envs_per_gpu
    cfg = make_config(SHARED_SCENARIO, n_envs=N)
    first, count = shard_range(world * N, rank, world)
    env = VecRanSlice(n_envs=N, cfg=cfg, fading=[synth_fading(t, FADING_COLS) for t in range(3)], device=device)
    dims = [10] * cfg.n_embb + [3] * cfg.n_mmtc
    agent = SharedVecKBRL(N, dims, cfg.n_prbs, budget=SHARED_BUDGET, max_rounds=1, capacity=SHARED_CAPACITY, device=device,
                          first_env=first)
    # a communicator even for one rank: the exchange then IS ncclAllGather on every box the bench runs on
    uid = group.bcast_bytes(SharedVecKBRL.unique_id() if rank == 0 else None)
    agent.comm_init(uid, rank, world)
    rccl_rank, rccl_ranks = agent.comm_info()
    rng = np.random.default_rng(1000 + rank)
    ia = np.concatenate([rng.integers(EMBB_A[0], EMBB_A[1], size=(N, cfg.n_embb)),
                         rng.integers(MMTC_A[0], MMTC_A[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
    sf = np.concatenate([rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(N, cfg.n_embb)),
                         rng.integers(MMTC_SEC[0], MMTC_SEC[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
    env.reset(seeds=replica_seeds(0, first, count))
    agent.reset(ia, sf, seeds=replica_seeds(7, first, count))
    env._check(env.L.rs_step(env.h, np.ascontiguousarray(ia).ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))

    def run(k):
        for _ in range(k):
            agent.step_resident(env)
            env.step_resident()
    def window(k):
        before = [int(agent.learner(0, s)['m']) for s in range(len(dims))]
        env.synchronize()
        agent.synchronize()
        group.barrier()
        t0 = time.perf_counter()
        run(k)
        env.synchronize()
        agent.synchronize()
        dt_ = group.max(time.perf_counter() - t0)
        sz = [int(agent.learner(0, s)['m']) for s in range(len(dims))]
        return dt_, sz, before

    def regime_of(before):
        # a shared dictionary at its capacity only projects (cheap); one that still grows pays a rank-1 update per insertion.
        # Judged at the START of the window: a dictionary that reaches its capacity inside the window was filling in it.
        return 'saturated' if max(before) >= SHARED_CAPACITY else 'filling'
    # two windows: while the eMBB dictionary still fills, and after it has reached its capacity
    first_w = None
    if args.shared_warmup > SHARED_FILL_WARMUP:
        run(SHARED_FILL_WARMUP)
        dt_a, sz_a, b_a = window(args.shared_steps)
        first_w = {'steps': [SHARED_FILL_WARMUP, SHARED_FILL_WARMUP + args.shared_steps], 'value': world * N * args.shared_steps / dt_a,
                   'ms_per_step': 1e3 * dt_a / args.shared_steps, 'regime': regime_of(b_a), 'dictionary_sizes_start_end': [b_a, sz_a]}
    run(args.shared_warmup)
    dt, sizes, b_main = window(args.shared_steps)
    start = (SHARED_FILL_WARMUP + args.shared_steps if first_w else 0) + args.shared_warmup
    all_sizes = group.allgather(sizes)
    all_ranks = group.allgather([rccl_rank, rccl_ranks])
    if rank == 0:
        S = len(dims)
        blk = 8 * S * (1 + SHARED_BUDGET * 18)          # doubles of one rank's proposal block (ranslice.h: KB_PROP_WIDTH)
        print(json.dumps({
            'workload': 'scenario index %d (%d PRBs, %d eMBB + %d mMTC slices), %d replicas per GPU x %d GPUs, one KBRL dictionary '
                        'per slice shared by all replicas, closed loop on the device' % (SHARED_SCENARIO, cfg.n_prbs, cfg.n_embb,
                                                                                         cfg.n_mmtc, N, world),
            'value': world * N * args.shared_steps / dt, 'unit': 'env-steps/s', 'ms_per_step': 1e3 * dt / args.shared_steps,
            'steps': args.shared_steps, 'warmup': args.shared_warmup, 'n_gpus': world,
            'window': [start, start + args.shared_steps], 'regime': regime_of(b_main), 'filling_window': first_w,
            'rccl_ranks': rccl_ranks, 'rccl_rank_of_each_process': [r[0] for r in all_ranks],
            'collective': 'ncclAllGather (RCCL, bound by libranslice.so: kb_shared_step_resident), one per step on the agent\'s stream',
            'allgather_bytes_per_rank_per_step': blk, 'allgather_bytes_total_per_step': blk * world,
            'budget': SHARED_BUDGET, 'capacity': SHARED_CAPACITY, 'dictionary_sizes': sizes,
            'dictionaries_identical_on_all_ranks': all(x == sizes for x in all_sizes),
        }), flush=True)
    env.close()
    agent.close()
    group.barrier()
    group.close()


def run_shared_leg(args, rank, local_rank, world, tag):
    """start this rank's child of the config-4 leg and wait for it (bounded); rank 0 returns the leg's record"""
    cmd = [sys.executable, os.path.abspath(__file__), '--shared-leg', '--envs-per-gpu', str(args.envs_per_gpu),
           '--shared-steps', str(args.shared_steps), '--shared-warmup', str(args.shared_warmup)]
    env = dict(os.environ)
    env.update({'RANK': str(rank), 'LOCAL_RANK': str(local_rank), 'WORLD_SIZE': str(world),
                'RANSLICE_RDZV_FILE': os.path.join('/tmp', 'ranslice_shared_%s' % tag)})
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('KBRL_COLLECTIVE_TIMEOUT_S', '60')
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=SHARED_TIMEOUT_S)
    except subprocess.TimeoutExpired:
        return {'error': 'the config-4 leg did not finish within %d s (killed)' % SHARED_TIMEOUT_S}
    if rank != 0:
        return None
    lines = [x for x in out.stdout.splitlines() if x.startswith('{')]
    if out.returncode != 0 or not lines:
        err = [x for x in out.stderr.strip().splitlines() if x.strip()]
        return {'error': 'rc %d: %s' % (out.returncode, (err[-1] if err else 'no output')[-300:])}
    return json.loads(lines[-1])


# ---------------------------------------------------------------------------------------------- the compact line
LINE_LIMIT = 4096      # the driver keeps the tail of stdout: the last line must fit with room to spare


def _r(x, nd=4):
    """round to nd significant digits (None and non-numbers pass through)"""
    if isinstance(x, bool) or not isinstance(x, (int, float)):
        return x
    if x == 0 or x != x or x in (float('inf'), float('-inf')):
        return x
    from math import floor, log10
    return round(x, nd - 1 - int(floor(log10(abs(x)))))


def compact_cpu(cpu):
    if not cpu:
        return cpu
    c = {k: _r(cpu.get(k)) for k in ('value', 'unit', 'cores', 'kind', 'single_core_value')}
    c['sample'] = '%s replicas x %s steps, C oracle, 1 thread/replica, same workload' % (cpu.get('cores'), cpu.get('steps', '?'))
    return c


def compact_kbrl(k):
    """config 3 in a dozen numbers"""
    if not k or 'error' in k:
        return k
    late, early = k.get('late') or {}, k.get('early') or {}
    head = late or early
    st = head.get('kinv_streaming') or {}
    c = {'workload': 'config 3: 4096 replicas + one KBRL agent each, closed loop on device, %s traces' % k.get('traces', 'tdl'),
         'early_value': _r(early.get('value')), 'early_steps': early.get('steps'),
         'late_value': _r(late.get('value')), 'late_steps': late.get('steps'), 'late_ms_per_step': _r(late.get('ms_per_step')),
         'unit': 'env-steps/s',
         'late_step_kernel_ms': _r(head.get('embb_kernel_ms')), 'late_update_phase_ms': _r(head.get('kb_update_phase_ms')),
         'late_select_phase_ms': _r(head.get('kb_select_ms')),
         'rank1_frac': _r((st.get('rank1') or {}).get('frac')), 'matvec_frac': _r((st.get('matvec') or {}).get('frac')),
         'select_bin_frac': _r((head.get('select_bin') or {}).get('frac')),
         'hbm_peak_GBs': HBM_PEAK_GBS,
         'mfma_instructions': (head.get('select_mfma') or {}).get('instructions_per_launch'),
         'mfma_kernel': 'select_gemm_kernel (v_mfma_f64_16x16x4)',
         'dictionary_size_mean': _r(head.get('dictionary_size_mean')), 'dictionary_size_max': head.get('dictionary_size_max'),
         'pool_GB_used': _r((head.get('pool') or {}).get('used_bytes', 0) / 1e9), 'pool_GB': _r(k.get('pool_bytes', 0) / 1e9),
         'pool_exhausted_near_step': _r((head.get('pool_horizon') or {}).get('exhausted_near_step'))}
    rl = k.get('reference_run_length') or {}
    if rl:
        c['replicas_reaching_50400_steps_unflagged'] = rl.get('replicas_unflagged')
        c['run_length_source'] = rl.get('source')
    return c


def compact_shared(sh):
    if not sh or 'error' in sh:
        return sh
    keys = ('value', 'unit', 'ms_per_step', 'steps', 'window', 'regime', 'n_gpus', 'rccl_ranks', 'allgather_bytes_total_per_step', 'capacity',
            'dictionary_sizes', 'dictionaries_identical_on_all_ranks')
    c = {k: _r(sh.get(k)) for k in keys}
    fw = sh.get('filling_window')
    if fw:
        c['filling_window'] = {'steps': fw.get('steps'), 'value': _r(fw.get('value')), 'ms_per_step': _r(fw.get('ms_per_step')),
                               'regime': fw.get('regime')}
    c['workload'] = 'config 4: scenario index 2, 4096 replicas/GPU, shared dictionaries, ncclAllGather per step'
    return c


def compact_line(full):
    """The last stdout line: everything the rules credit, under LINE_LIMIT bytes.  Sub-records that would push it over are
    dropped in a fixed order (never the headline, the roofline or the CPU baseline)."""
    roof = dict(full['roofline'])
    for k in ('profiled_traffic',):
        roof.pop(k, None)
    roof = {k: (_r(v, 5) if not isinstance(v, dict) else {kk: _r(vv) for kk, vv in v.items()}) for k, v in roof.items()}
    line = {k: full[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                 'vs_baseline', 'dtype', 'data')}
    line['value'] = _r(line['value'], 6)
    line['ms_per_step'] = _r(line['ms_per_step'], 5)
    line['config'] = dict(full['config'])
    line['roofline'] = roof
    line['cpu_baseline'] = compact_cpu(full.get('cpu_baseline'))
    if 'kbrl' in full:
        line['kbrl'] = compact_kbrl(full['kbrl'])
    if 'shared_kbrl' in full:
        line['shared_kbrl'] = compact_shared(full['shared_kbrl'])
    line['full_record'] = 'profiles/bench_full_last.json (also the stdout line before this one)'
    for victim in ('full_record', 'shared_kbrl', 'kbrl'):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        if victim in line:
            line[victim] = {'dropped': 'line over %d bytes; see the full record' % LINE_LIMIT} if victim != 'full_record' else None
    return line


def spawn_ranks(args):
    """python bench.py --gpus N outside any launcher: time the CPU baseline here, then start one rank per GPU (this script
    again, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment as torch.distributed.run sets them)"""
    cpu_file = None
    if not args.no_cpu_baseline:
        cpu_file = os.path.join('/tmp', 'bench_cpu_%d.json' % os.getpid())
        with open(cpu_file, 'w') as f:
            json.dump(cpu_baseline(1000, args.cpu_steps), f)
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(args.gpus), 'MASTER_ADDR': '127.0.0.1',
                    'MASTER_PORT': str(port)})
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        cmd = [sys.executable, os.path.abspath(__file__)] + list(sys.argv[1:])
        cmd += ['--cpu-baseline-json', cpu_file] if cpu_file else (['--no-cpu-baseline'] if '--no-cpu-baseline' not in sys.argv else [])
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    for pr in procs:
        rc = pr.wait() or rc
    if cpu_file and os.path.exists(cpu_file):
        os.remove(cpu_file)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--burn-in', type=int, default=-1,
                    help='fixed number of burn-in steps; default -1 = until the UE population is stationary')
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kbrl', action='store_true')
    ap.add_argument('--no-shared', action='store_true', help='skip the config-4 leg (shared dictionaries over RCCL)')
    ap.add_argument('--kbrl-steps', type=int, default=200)
    ap.add_argument('--kbrl-sos', action='store_true', help="config 3 also on the fixtures' trace profile (full record only)")
    ap.add_argument('--shared-steps', type=int, default=200)
    ap.add_argument('--shared-warmup', type=int, default=100)
    ap.add_argument('--shared-leg', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--graph', action='store_true',
                    help='replay the timed loop from a captured hipGraph (rs_run_random); the kernel time for the '
                         'roofline is then taken from a separate event-timed pass of 100 steps')
    ap.add_argument('--state-file', default='',
                    help='profiling aid: if the file exists the burn-in is replaced by restoring the environments from it (rs_load_state), '
                         'otherwise it is written after the burn-in -- counter passes of rocprofv3 (20 ms per dispatch) then run at the '
                         'population of the timed run without repeating 3500 burn-in steps each')
    ap.add_argument('--cpu-steps', type=int, default=3000)
    ap.add_argument('--cpu-baseline-json', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--scaling', default='', help='e.g. 1,2,4,8: run every N in turn and print ONE line with the curve and '
                                                  'the CPU baseline (Ns beyond the devices of the box are skipped)')
    args = ap.parse_args()
    if args.scaling:
        run_scaling(args)
        return
    if args.shared_leg:
        shared_leg(args)
        return

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    under_launcher = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    cpu_burn = 1000

    if args.gpus > 1 and not under_launcher:
        sys.exit(spawn_ranks(args))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE (%d) != --gpus (%d)' % (world, args.gpus))

    # the CPU baseline forks worker processes: do it before any HIP initialisation (rank 0 only; the other
    # ranks wait for it in the rendezvous of the rank group)
    cpu_base = None
    if rank == 0:
        if args.cpu_baseline_json and os.path.exists(args.cpu_baseline_json):
            with open(args.cpu_baseline_json) as f:
                cpu_base = json.load(f)
        elif not args.no_cpu_baseline:
            cpu_base = cpu_baseline(cpu_burn, args.cpu_steps)

    from rank_group import RankGroup
    from ranslice import _lib
    from ranslice.config import make_config
    from ranslice.fading import synth_fading
    from ranslice.sharding import shard_range, replica_seeds, aggregate_throughput
    from ranslice.vec_env import VecRanSlice
    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    # RANSLICE_BENCH_SHARE_GPU=1 (developer check of the N > 1 path on a box with fewer GPUs than ranks): ranks wrap around the
    # visible devices; the line says so and is NOT a scaling measurement
    share_gpu = os.environ.get('RANSLICE_BENCH_SHARE_GPU') == '1'
    device = local_rank % ndev if share_gpu else local_rank
    group = RankGroup(rank, world, timeout=900.0)    # (rank 0 may still be timing the CPU baseline when the others arrive)

    n_envs = args.envs_per_gpu
    cfg = make_config(SCENARIO, n_envs=n_envs)
    fading = [synth_fading(t, FADING_COLS) for t in range(3)]
    env = VecRanSlice(n_envs=n_envs, cfg=cfg, fading=fading, device=device)
    # replica ids are global: rank r owns [r*n_envs, (r+1)*n_envs)
    first, count = shard_range(world * n_envs, rank, world)
    assert count == n_envs
    env.reset(seeds=replica_seeds(0, first, count))
    n_tasks = n_envs * cfg.n_embb

    step_idx = 0

    def run(k):
        nonlocal step_idx
        for _ in range(k):
            env.random_actions(ACTION_SEED + rank, step_idx)
            env.step_resident()
            step_idx += 1

    def block_mean_ue(k):
        c_a = env.counters()
        run(k)
        env.synchronize()
        c_b = env.counters()
        return (c_b[3] - c_a[3]) / float(k * cfg.slots_per_step * n_tasks)

    # ---- burn-in to the stationary UE population
    burn_hist = []
    state_file = args.state_file + ('.rank%d' % rank if world > 1 else '') if args.state_file else ''
    if state_file and os.path.exists(state_file + '.npz'):
        z = np.load(state_file + '.npz')
        env.load_state(z['env'])
        step_idx = int(z['step_idx'])
        burn_hist = [float(x) for x in z['burn_hist']]
    elif args.burn_in >= 0:
        run(args.burn_in)
    else:
        prev = None
        while step_idx < BURN_MAX:
            cur = block_mean_ue(BURN_BLOCK)
            burn_hist.append(round(cur, 4))
            done = prev is not None and abs(cur - prev) <= 0.005 * prev
            # every rank must run the same number of blocks: continue while ANY rank is still moving
            done = group.max(0.0 if done else 1.0) == 0.0
            prev = cur
            if done:
                break
    burn_steps = step_idx
    if state_file and not os.path.exists(state_file + '.npz'):
        env.synchronize()
        np.savez(state_file, env=env.save_state(), step_idx=np.int64(step_idx), burn_hist=np.asarray(burn_hist, dtype=np.float64))
    run(args.warmup)
    env.synchronize()
    c0 = env.counters()
    env.set_kernel_timing(True)

    group.barrier()
    env.synchronize()
    t0 = time.perf_counter()
    if args.graph:
        env.set_kernel_timing(False)
        env.run_random(ACTION_SEED + rank, step_idx, args.steps, graph=True)
        step_idx += args.steps
    else:
        run(args.steps)
    env.synchronize()
    group.barrier()
    t1 = time.perf_counter()

    c1 = env.counters()
    if args.graph:
        env.set_kernel_timing(True)
        run(100)
        env.synchronize()
    (kern_ms, kern_min, kern_max), launches = env.kernel_time_stats_ms()
    env.set_kernel_timing(False)
    rx_tests, rx_exact, rx_short = env.rx_stats()
    out = env.fetch()  # also surfaces capacity-overflow errors
    assert np.isfinite(out['reward']).all()

    elapsed = group.max(t1 - t0)
    env.close()

    full = None
    if rank == 0:
        value = aggregate_throughput(n_envs * args.steps, world, elapsed)
        # ---- roofline of the dominant kernel (per launch = one step of n_envs replicas)
        samples = (c1[0] - c0[0]) / args.steps          # fading samples read per launch
        ue_slots = (c1[3] - c0[3]) / args.steps
        mean_ue = ue_slots / (n_tasks * cfg.slots_per_step)
        nv = cfg.n_embb * 10 + cfg.n_mmtc * 3
        n_slices = cfg.n_embb + cfg.n_mmtc
        b_fading = 8.0 * samples
        b_state = 2.0 * n_tasks * (STATE_TASK_BYTES + mean_ue * STATE_UE_BYTES)
        b_io = n_envs * (4.0 * n_slices + 4.0 * nv + 8.0 + 8.0 * n_slices)
        alg_bytes = b_fading + b_state + b_io
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        roof = {
            'bound': 'hbm', 'kernel': 'embb_step_kernel' if n_tasks < 49152 else
            'embb_step_kernel<8> + embb_step_kernel<16> side by side (the split step of large batches; kernel_ms spans both)',
            'achieved': achieved, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'traffic': None,
            'algorithmic_bytes_per_launch': alg_bytes, 'kernel_ms': kern_ms, 'kernel_ms_min': kern_min, 'kernel_ms_max': kern_max,
            'launches_timed': launches,
            'bytes_per_env_step': alg_bytes / n_envs, 'mean_ues_per_slice': mean_ue,
            'pf_iterations_per_env_step': (c1[2] - c0[2]) / args.steps / n_envs,
            # the byte model above is the REFERENCE algorithm's (every UE reads its slice's samples every slot, 8 B each:
            # channel_models.py:171-191); since round 6 the kernel takes the channel estimates from per-column prefix sums (16 B
            # each) and the reception sums from a float32 mirror, so what it moves (`traffic`) is less than it
            'byte_model': 'reference algorithm: 8 B x fading samples + 2 x state + I/O (SURVEY 8d)',
            # reception tests decided in float32 by guard band / of which fell inside the band and formed the f64 probability
            'rx_tests_since_reset': rx_tests, 'rx_exact_share': (rx_exact / rx_tests) if rx_tests else None, 'rx_short_test': rx_short,
        }
        # Counters of a separate rocprofv3 --pmc run of this command (tools/profile_round.sh), NOT of this process: the HBM bytes per
        # launch are comparable with algorithmic_bytes_per_launch only at the same UE population, so `traffic` is filled in only then.
        # `limiter` is what the SQ counters of that run say bounds the kernel: VALU issue slots, not HBM.
        tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    prof = json.load(f)
                pop = prof.get('mean_ues_per_slice')
                same = pop is not None and abs(pop - mean_ue) <= 0.03 * mean_ue and n_envs == prof.get('n_envs', ENVS_PER_GPU)
                if same:
                    roof['traffic'] = prof.get('embb_step_kernel_bytes_per_launch')
                    roof['traffic_over_algorithmic'] = roof['traffic'] / alg_bytes if roof['traffic'] and alg_bytes else None
                    roof['traffic_from'] = 'committed profile (%s; FETCH_SIZE x %.1f, WRITE_SIZE x %.1f per profiles/%s), not this run' % (
                        prof.get('file', 'profiles/hbm_traffic.json'), prof.get('fetch_correction', 1.0), prof.get('write_correction', 1.0),
                        prof.get('calibration_file', '?'))
                roof['profiled_traffic'] = {'bytes_per_launch': prof.get('embb_step_kernel_bytes_per_launch'),
                                            'mean_ues_per_slice': pop, 'same_population_as_this_run': bool(same),
                                            'source': 'profiles/hbm_traffic.json (%s)' % prof.get('source', 'see file')}
                if prof.get('valu_issue_frac') is not None:
                    roof['limiter'] = {'bound': 'valu_issue', 'frac': prof['valu_issue_frac'],
                                       'valu_insts_per_launch': prof.get('valu_insts_per_launch'),
                                       'source': prof.get('valu_issue_file', 'profiles/hbm_traffic.json')}
            except Exception:
                pass
        full = {
            'metric': 'env-steps/sec (batched RanSlice.step, scenario_0)',
            'value': value,
            'unit': 'env-steps/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': 'scenario_0 (200 PRBs, 5 eMBB slices, 50 slots/step), %d env replicas per GPU, step() only, '
                            'random multinomial actions generated on device' % n_envs,
                'envs_per_gpu': n_envs, 'global_envs': world * n_envs,
                'burn_in_steps': burn_steps,
                'burn_in': 'fixed' if args.burn_in >= 0 else 'until the mean UEs/slice of a %d-step block moves < 0.5 %%' % BURN_BLOCK,
                'fading': '3 synthetic traces x %d samples x 200 PRB, f64' % FADING_COLS,
                'parallelism': 'replica-sharded x%d, no collective in step' % world +
                               (' (RANSLICE_BENCH_SHARE_GPU: the ranks share %d device(s) -- a check of the launch path, not a '
                                'scaling measurement)' % ndev if share_gpu else ''),
                'loop': 'hipGraph replay (rs_run_random)' if args.graph else 'one launch sequence per step from the host',
            },
            'roofline': roof,
            'burn_in_history_mean_ues': burn_hist,
        }
        full['cpu_baseline'] = cpu_base
        if world == 1 and not args.no_kbrl:
            try:
                # config 3 on the 'tdl' synthetic trace profile (tapped delay lines, the construction of the ns-3 traces the
                # reference was run on; the profile the reference comparison of DESIGN.md section 7 was recorded on);
                # --kbrl-sos: the fixtures' profile beside it (rounds 1-3's numbers)
                full['kbrl'] = kbrl_record(n_envs, device, args.kbrl_steps, 100, profile=os.environ.get('KBRL_TRACES', 'tdl'))
                if args.kbrl_sos:
                    full['kbrl_sos'] = kbrl_record(n_envs, device, args.kbrl_steps, 100, profile='sos')
            except Exception as e:  # the headline line must not depend on the agent's sub-record
                full.setdefault('kbrl', {'error': repr(e)[:300]})

    # ---- config 4: the exchange over RCCL, every rank's child process under a timeout (a failure stays inside the sub-record)
    if not args.no_shared:
        tag = '%s_%d' % (os.environ.get('MASTER_PORT', '0'), os.getppid() if world > 1 else os.getpid())
        try:
            sh = run_shared_leg(args, rank, local_rank, world, tag)
        except Exception as e:
            sh = {'error': repr(e)[:300]}
        if full is not None:
            full['shared_kbrl'] = sh
    try:
        group.barrier()
    except Exception:
        pass
    group.close()

    if full is not None:
        try:
            with open(os.path.join(ROOT, 'profiles', 'bench_full_last.json'), 'w') as f:
                json.dump(full, f, indent=1)
        except OSError:
            pass
        print('# full record: ' + json.dumps(full), flush=True)
        print(json.dumps(compact_line(full)), flush=True)


if __name__ == '__main__':
    main()

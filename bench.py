#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched RanSlice.step on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: RanSlice.step for 4096 env replicas per GPU
(scenario_0: 200 PRBs, 5 eMBB slices, 50 slots per step; BASELINE.json configs[1]) with random
actions generated on the device.  Inputs (fading tables, simulator state, actions) are resident
in HBM when the timed region starts; nothing crosses PCIe inside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: spawns its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Replicas are independent, so GPUs shard them with no data-path collective (weak scaling: 4096
replicas per GPU).  torch is used only for the process group (barrier, max-over-ranks).

Before the W warm-up steps the environments are advanced until the UE population is stationary: blocks of 500
steps (25 s of simulated time; the holding times have a 30 s mean) until the mean number of UEs per slice changes
by less than 0.5 % from one block to the next (every rank runs the same number of blocks).  This is environment
set-up, not part of the timed work; the line reports the steps it took and the population.

The JSON line also carries
  roofline     : algorithmic bytes per launch of the dominant kernel (embb_step_kernel) divided by
                 its mean launch duration measured with HIP events on the launch stream, against
                 the 8 TB/s HBM peak (DESIGN.md §Measurement states the byte model).  `traffic` is null: this
                 process does not read PMC counters; the HBM bytes of a separate rocprofv3 --pmc run of this
                 same command are reported as `profiled_traffic` with the file they come from;
  cpu_baseline : the CPU oracle (a C port of the reference's numpy path, pinned bit-exact to the
                 reference on golden tapes) timed on this box's host cores on a bounded sample of
                 the same workload -- a reported baseline, never the thing measured above;
  kbrl         : (1 GPU) BASELINE config 3 -- the same 4096 replicas with one KBRL agent each, closed loop on
                 the device, at two points of learning: steps 100-300 (dictionaries of tens of landmarks) and from
                 step 3000 (hundreds): env-steps/s, per-phase kernel times, kernel evaluations/s, dictionary sizes
                 and the pool in use.
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
LDS_PAIRS_PEAK = 256 * 16 * 2.4e9   # 8-byte LDS reads per second: 256 CUs x 128 B/clk x 2.4 GHz
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
    return dict(value=total / (t1 - t0), unit='env-steps/s', cores=cores, kind='port',
                sample='%d replicas (one per host core) x %d steps of the same scenario_0 workload and action '
                       'script after a %d-step burn-in; C oracle (oracle/rs_oracle.c), 1 thread per replica'
                       % (cores, timed, burn))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


KBRL_CAPACITY = 4096          # a limit, not a reservation: dictionaries take their storage from the pool as they grow
KBRL_POOL_BYTES = 64 << 30
KBRL_LATE_STEP = 3000         # second measurement point: dictionaries of several hundred landmarks


def kbrl_record(n_envs, device, steps, warmup, late_step=KBRL_LATE_STEP):
    """BASELINE config 3 on this GPU: closed loop with one KBRL agent per replica (kb_step_resident), measured twice:
    early in learning (steps `warmup`..`warmup + steps`, the point of rounds 1-2) and from step `late_step` on, where the
    dictionaries hold what a run of that length holds."""
    import ctypes as C
    from ranslice.config import make_config, EMBB_A, EMBB_SEC
    from ranslice.fading import synth_traces
    from ranslice.kbrl_dev import VecKBRL
    from ranslice.vec_env import VecRanSlice
    profile = os.environ.get('KBRL_TRACES', 'sos')        # (developer knob) synthetic trace profile of this leg
    cap = int(os.environ.get('KBRL_CAPACITY', KBRL_CAPACITY))
    cfg = make_config(SCENARIO, n_envs=n_envs)
    env = VecRanSlice(n_envs=n_envs, cfg=cfg, fading=synth_traces(FADING_COLS, profile), device=device)
    agent = VecKBRL(n_envs, [10] * cfg.n_embb, cfg.n_prbs, accuracy_range=(0.99, 0.999), capacity=cap, device=device,
                    pool_bytes=KBRL_POOL_BYTES)
    rng = np.random.default_rng(0)
    ia = rng.integers(EMBB_A[0], EMBB_A[1], size=(n_envs, cfg.n_embb)).astype(np.int32)   # scenario_creator.py:220-221
    sf = rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(n_envs, cfg.n_embb)).astype(np.int32)
    env.reset()
    agent.reset(ia, sf)
    env._check(env.L.rs_step(env.h, ia.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
    done = [0]

    def run(k):
        for _ in range(k):
            agent.step_resident(env)
            env.step_resident()
        done[0] += k
    d = 11   # eMBB learner: 10 state variables + the candidate allocation

    def point(k):
        env.synchronize()
        agent.synchronize()
        s0 = agent.stats()
        agent.set_kernel_timing(True)
        env.set_kernel_timing(True)
        first = done[0]
        t0 = time.perf_counter()
        run(k)
        env.synchronize()
        agent.synchronize()          # raises on a device-side error flag
        dt = time.perf_counter() - t0
        s1 = agent.stats()
        ph = agent.phase_times_ms()
        env_ms, _ = env.kernel_time_ms()
        agent.set_kernel_timing(False)
        env.set_kernel_timing(False)
        sizes = agent.dictionary_sizes()
        evals = s1[3] - s0[3]
        # what the table-factorised scoring actually executes per (landmark, candidate) pair: one FMA + one LDS read
        # (2 flop); per landmark and pass one exp and the 3 (d - 1) flop of its distance.  SURVEY.md 8d's model -- the
        # (3 d + 3) flop of a direct evaluation per pair -- is kept beside it for comparison with rounds 1-2.
        passes = 2.0 + (s1[1] - s0[1]) / float(n_envs * cfg.n_embb * k)   # update pass, select pass, one per repair
        exps = passes * float(sizes.sum()) * k
        return {
            'steps': [first, first + k], 'value': n_envs * k / dt, 'unit': 'env-steps/s', 'ms_per_step': 1e3 * dt / k,
            'embb_kernel_ms': env_ms, 'kb_update_phase_ms': ph['update_ms'], 'kb_select_ms': ph['select_ms'],
            'kernel_evaluations_per_s': evals / dt, 'exps_per_s_estimate': exps / dt,
            'predicts_per_env_step': (s1[0] - s0[0]) / (n_envs * k),
            'mistakes_per_env_step': (s1[1] - s0[1]) / (n_envs * k),
            # what bounds the table-factorised scoring per (landmark, candidate) pair is one 8-byte LDS read (128 B/clk/CU) and
            # three VALU instructions per 64 pairs, not flops: the LDS-side bound is 256 CUs x 16 pairs/clk x 2.4 GHz
            'scoring_bound': {'bound': 'lds', 'achieved_pairs_per_s': evals / dt, 'peak_pairs_per_s': LDS_PAIRS_PEAK,
                              'frac': evals / dt / LDS_PAIRS_PEAK,
                              'note': 'the one-wave kernels are latency-bound (waves wait ~70 % of their cycles, '
                                      'profiles/kbrl_mfma_share.json), not throughput-bound'},
            'direct_model_tflops': evals * (3 * d + 3) / dt / 1e12,
            'direct_model_frac_of_f64_peak': evals * (3 * d + 3) / dt / 1e12 / F64_PEAK_TFLOPS,
            'dictionary_size_mean': float(np.mean(sizes)), 'dictionary_size_max': int(np.max(sizes)),
            'dictionary_size_p50_p90_p99': [float(np.percentile(sizes, q)) for q in (50, 90, 99)],
            'pool': agent.pool(),
        }
    run(warmup)
    early = point(steps)
    late = None
    if late_step and late_step > done[0]:
        run(late_step - done[0])
        late = point(steps)
    rec = {
        'workload': 'scenario_0, %d replicas + one KBRL agent per replica, closed loop on the device (%s traces)'
                    % (n_envs, profile),
        'dictionary_capacity': cap, 'pool_bytes': KBRL_POOL_BYTES,
        'value': early['value'], 'unit': 'env-steps/s', 'ms_per_step': early['ms_per_step'],
        'early': early, 'late': late,
        'scoring': 'k(l_j, x_c) = E_j G[|a_j - c|]: one exp per landmark and pass, one FMA + one LDS read per (landmark, '
                   'candidate); no MFMA in these kernels (a one-column product; DESIGN.md §4 KBRL)',
        'peak_tflops_f64': F64_PEAK_TFLOPS,
    }
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
    an independent launch: N = 1 in a child process, N > 1 through the launcher), one JSON line with the curve.  Ns
    beyond the devices of the box are listed as skipped."""
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
        curve.append({'n_gpus': n, 'value': line['value'], 'ms_per_step': line['ms_per_step'],
                      'roofline_frac': line['roofline']['frac'], 'kernel_ms': line['roofline']['kernel_ms'],
                      'global_envs': line['config']['global_envs']})
    base = next((c['value'] for c in curve if c.get('n_gpus') == 1 and 'value' in c), None)
    for c in curve:
        if base and 'value' in c:
            c['per_gpu_vs_1gpu'] = c['value'] / c['n_gpus'] / base
    print(json.dumps({'metric': 'env-steps/sec (batched RanSlice.step, scenario_0)', 'unit': 'env-steps/s',
                      'scaling': 'weak', 'curve': curve, 'cpu_baseline': cpu, 'dtype': 'f64', 'data': 'synthetic'}),
          flush=True)


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
    ap.add_argument('--kbrl-steps', type=int, default=200)
    ap.add_argument('--graph', action='store_true',
                    help='replay the timed loop from a captured hipGraph (rs_run_random); the kernel time for the '
                         'roofline is then taken from a separate event-timed pass of 100 steps')
    ap.add_argument('--cpu-steps', type=int, default=3000)
    ap.add_argument('--cpu-baseline-json', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--scaling', default='', help='e.g. 1,2,4,8: run every N in turn and print ONE line with the curve and '
                                                  'the CPU baseline (Ns beyond the devices of the box are skipped)')
    args = ap.parse_args()
    if args.scaling:
        run_scaling(args)
        return

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    under_launcher = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    cpu_burn = 1000

    if args.gpus > 1 and not under_launcher:
        # one command for the whole report: time the CPU baseline here, then launch one rank per GPU
        cpu_file = None
        if not args.no_cpu_baseline:
            cpu_file = os.path.join('/tmp', 'bench_cpu_%d.json' % os.getpid())
            with open(cpu_file, 'w') as f:
                json.dump(cpu_baseline(cpu_burn, args.cpu_steps), f)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)]
        cmd += [a for a in sys.argv[1:]]
        if cpu_file:
            cmd += ['--cpu-baseline-json', cpu_file]
        else:
            cmd += ['--no-cpu-baseline'] if '--no-cpu-baseline' not in sys.argv else []
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        rc = subprocess.call(cmd, env=env)
        if cpu_file and os.path.exists(cpu_file):
            os.remove(cpu_file)
        sys.exit(rc)
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE (%d) != --gpus (%d)' % (world, args.gpus))

    # the CPU baseline forks worker processes: do it before any HIP/torch initialisation (rank 0 only; the other
    # ranks wait for it in the process-group rendezvous)
    cpu_base = None
    if rank == 0:
        if args.cpu_baseline_json and os.path.exists(args.cpu_baseline_json):
            with open(args.cpu_baseline_json) as f:
                cpu_base = json.load(f)
        elif not args.no_cpu_baseline:
            cpu_base = cpu_baseline(cpu_burn, args.cpu_steps)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))

    def barrier():
        if world > 1:
            dist.barrier()

    from dist_util import max_over_ranks
    from ranslice.config import make_config
    from ranslice.fading import synth_fading
    from ranslice.sharding import shard_range, replica_seeds, aggregate_throughput
    from ranslice.vec_env import VecRanSlice

    n_envs = args.envs_per_gpu
    cfg = make_config(SCENARIO, n_envs=n_envs)
    fading = [synth_fading(t, FADING_COLS) for t in range(3)]
    env = VecRanSlice(n_envs=n_envs, cfg=cfg, fading=fading, device=local_rank)
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
    if args.burn_in >= 0:
        run(args.burn_in)
    else:
        prev = None
        while step_idx < BURN_MAX:
            cur = block_mean_ue(BURN_BLOCK)
            burn_hist.append(round(cur, 4))
            done = prev is not None and abs(cur - prev) <= 0.005 * prev
            # every rank must run the same number of blocks: continue while ANY rank is still moving
            if world > 1:
                done = max_over_ranks(0.0 if done else 1.0, device='cuda') == 0.0
            prev = cur
            if done:
                break
    burn_steps = step_idx
    run(args.warmup)
    env.synchronize()
    c0 = env.counters()
    env.set_kernel_timing(True)

    barrier()
    torch.cuda.synchronize()
    env.synchronize()
    t0 = time.perf_counter()
    if args.graph:
        env.set_kernel_timing(False)
        env.run_random(ACTION_SEED + rank, step_idx, args.steps, graph=True)
        step_idx += args.steps
    else:
        run(args.steps)
    env.synchronize()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()

    c1 = env.counters()
    if args.graph:
        env.set_kernel_timing(True)
        run(100)
        env.synchronize()
    kern_ms, launches = env.kernel_time_ms()
    env.set_kernel_timing(False)
    out = env.fetch()  # also surfaces capacity-overflow errors
    assert np.isfinite(out['reward']).all()

    elapsed = max_over_ranks(t1 - t0, device='cuda')
    env.close()

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
            'bound': 'hbm', 'kernel': 'embb_step_kernel', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'traffic': None,   # not measured by this process (needs rocprofv3 --pmc); see profiled_traffic
            'algorithmic_bytes_per_launch': alg_bytes, 'kernel_ms': kern_ms, 'launches_timed': launches,
            'bytes_per_env_step': alg_bytes / n_envs, 'mean_ues_per_slice': mean_ue,
            'pf_iterations_per_env_step': (c1[2] - c0[2]) / args.steps / n_envs,
        }
        tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    prof = json.load(f)
                # numbers of a separate rocprofv3 --pmc run of this command, NOT of this process
                roof['profiled_traffic'] = {'bytes_per_launch': prof.get('embb_step_kernel_bytes_per_launch'),
                                            'valu_issue_frac': prof.get('valu_issue_frac'),
                                            'source': 'profiles/hbm_traffic.json (%s)' % prof.get('source', 'see file')}
            except Exception:
                pass
        line = {
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
                'burn_in': ('fixed' if args.burn_in >= 0 else
                            'until stationary: mean UEs/slice per %d-step block %s' % (BURN_BLOCK, burn_hist)),
                'fading': '3 synthetic traces x %d samples x 200 PRB, f64' % FADING_COLS,
                'parallelism': 'replica-sharded x%d, no collective in step' % world,
                'loop': 'hipGraph replay (rs_run_random)' if args.graph else 'one launch sequence per step from the host',
            },
            'roofline': roof,
        }
        line['cpu_baseline'] = cpu_base
        if world == 1 and not args.no_kbrl:
            try:
                line['kbrl'] = kbrl_record(n_envs, local_rank, args.kbrl_steps, 100)
            except Exception as e:  # the headline line must not depend on the agent's sub-record
                line['kbrl'] = {'error': repr(e)}
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

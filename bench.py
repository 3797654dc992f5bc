#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched RanSlice.step on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: RanSlice.step for 4096 env replicas per GPU
(scenario_0: 200 PRBs, 5 eMBB slices, 50 slots per step; BASELINE.json configs[1]) with random
actions generated on the device.  Inputs (fading tables, simulator state, actions) are resident
in HBM when the timed region starts; nothing crosses PCIe inside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Replicas are independent, so GPUs shard them with no data-path collective (weak scaling: 4096
replicas per GPU).  torch is used only for the process group (barrier, max-over-ranks).

Before the W warm-up steps the environments are advanced `--burn-in` steps (default 1000 = 50 s of
simulated time) so that the UE population is at its steady state (~3.5 UEs per slice) instead of
the 2-UE state right after reset; this is environment set-up, not part of the timed work.

The JSON line also carries
  roofline     : algorithmic bytes per launch of the dominant kernel (embb_step_kernel) divided by
                 its mean launch duration measured with HIP events on the launch stream, against
                 the 8 TB/s HBM peak (DESIGN.md §Measurement states the byte model);
  cpu_baseline : the CPU oracle (a C port of the reference's numpy path, pinned bit-exact to the
                 reference on golden tapes) timed on this box's host cores on a bounded sample of
                 the same workload -- a reported baseline, never the thing measured above.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'network-slicing_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

ENVS_PER_GPU = 4096
SCENARIO = 0
FADING_COLS = 10000
ACTION_SEED = 2024
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

# bytes of persistent simulator state the step kernel reads and writes per task / per active UE
# (network-slicing_amd/csrc/rs_device.h): header 5 x 4 B; UE 3 f64 + 9 i32 + 8 burst i32
STATE_TASK_BYTES = 20
STATE_UE_BYTES = 3 * 8 + 9 * 4 + 8 * 4


def _cpu_worker(args):
    (rank, cfg_kw, burn, timed, barrier_wait, replicas) = args
    from oracle import pyoracle as po
    from ranslice.config import make_config
    from ranslice.fading import synth_fading
    cfg = make_config(SCENARIO, n_envs=1, **cfg_kw)
    fading = [synth_fading(t, FADING_COLS) for t in range(3)]
    envs = []
    for r in replicas:
        o = po.OracleEnv(cfg, fading)
        o.set_seed(r)
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


def cpu_baseline(burn, timed):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--burn-in', type=int, default=1000)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true',
                    help='replay the timed loop from a captured hipGraph (rs_run_random); the kernel time for the '
                         'roofline is then taken from a separate event-timed pass of 100 steps')
    ap.add_argument('--cpu-steps', type=int, default=3000)
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d '
                             '--master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d ...' % (args.gpus, args.gpus))
        raise SystemExit('WORLD_SIZE (%d) != --gpus (%d)' % (world, args.gpus))

    # the CPU baseline forks worker processes: do it before any HIP/torch initialisation
    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(args.burn_in, args.cpu_steps)

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

    from ranslice.config import make_config
    from ranslice.fading import synth_fading
    from ranslice.vec_env import VecRanSlice

    n_envs = args.envs_per_gpu
    cfg = make_config(SCENARIO, n_envs=n_envs)
    fading = [synth_fading(t, FADING_COLS) for t in range(3)]
    env = VecRanSlice(n_envs=n_envs, cfg=cfg, fading=fading, device=local_rank)
    # replica ids are global: rank r owns [r*n_envs, (r+1)*n_envs)
    from ranslice.sharding import shard_range, replica_seeds, max_over_ranks, aggregate_throughput
    first, count = shard_range(world * n_envs, rank, world)
    assert count == n_envs
    env.reset(seeds=replica_seeds(0, first, count))

    step_idx = 0

    def run(k):
        nonlocal step_idx
        for _ in range(k):
            env.random_actions(ACTION_SEED + rank, step_idx)
            env.step_resident()
            step_idx += 1

    run(args.burn_in)
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

    if rank == 0:
        value = aggregate_throughput(n_envs * args.steps, world, elapsed)
        # ---- roofline of the dominant kernel (per launch = one step of n_envs replicas)
        samples = (c1[0] - c0[0]) / args.steps          # fading samples read per launch
        ue_slots = (c1[3] - c0[3]) / args.steps
        n_tasks = n_envs * cfg.n_embb
        mean_ue = ue_slots / (n_tasks * cfg.slots_per_step)
        nv = cfg.n_embb * 10 + cfg.n_mmtc * 3
        n_slices = cfg.n_embb + cfg.n_mmtc
        b_fading = 8.0 * samples
        b_state = 2.0 * n_tasks * (STATE_TASK_BYTES + mean_ue * STATE_UE_BYTES)
        b_io = n_envs * (4.0 * n_slices + 4.0 * nv + 8.0 + 8.0 * n_slices)
        alg_bytes = b_fading + b_state + b_io
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic = None
        valu_frac = None
        tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    prof = json.load(f)
                    traffic = prof.get('embb_step_kernel_bytes_per_launch')
                    valu_frac = prof.get('valu_issue_frac')
            except Exception:
                traffic = None
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
                'envs_per_gpu': n_envs, 'global_envs': world * n_envs, 'burn_in_steps': args.burn_in,
                'fading': '3 synthetic traces x %d samples x 200 PRB, f64' % FADING_COLS,
                'parallelism': 'replica-sharded x%d, no collective in step' % world,
                'loop': 'hipGraph replay (rs_run_random)' if args.graph else 'one launch sequence per step from the host',
            },
            'roofline': {
                'bound': 'hbm', 'kernel': 'embb_step_kernel', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                'algorithmic_bytes_per_launch': alg_bytes, 'kernel_ms': kern_ms, 'launches_timed': launches,
                'bytes_per_env_step': alg_bytes / n_envs, 'mean_ues_per_slice': mean_ue,
                'pf_iterations_per_env_step': (c1[2] - c0[2]) / args.steps / n_envs,
                # what actually limits the kernel: f64 VALU issue share from the committed SQ counters
                'valu_issue_frac_profiled': valu_frac,
            },
        }
        line['cpu_baseline'] = cpu_base
        print(json.dumps(line), flush=True)

    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

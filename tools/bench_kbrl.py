#!/usr/bin/env python3
"""BASELINE config 3: scenario_0, N env replicas on one MI355X with one KBRL agent per replica, closed loop
entirely on the device (kb_step_resident + rs_step_resident).  Reports env-steps/s with the agent in the loop,
kernel evaluations/s of the RBF scoring (kb_get_stats[3]) and the device time of the agent kernels.

  python tools/bench_kbrl.py [--envs 4096] [--steps 300] [--warmup 100] [--capacity 256]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
import numpy as np  # noqa: E402
from ranslice.config import make_config, EMBB_A, EMBB_SEC  # noqa: E402
from ranslice.fading import synth_traces  # noqa: E402
from ranslice.kbrl_dev import VecKBRL  # noqa: E402
from ranslice.vec_env import VecRanSlice  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--capacity', type=int, default=4096)
    ap.add_argument('--pool-gb', type=float, default=64.0)
    ap.add_argument('--profile', default='sos', help='synthetic trace profile: sos | tdl')
    ap.add_argument('--save-state', default='', help='after the warm-up: write the environment and agent checkpoints to this directory and exit')
    ap.add_argument('--load-state', default='', help='instead of the warm-up: restore the checkpoints of --save-state (same --envs / --capacity / '
                                                     '--pool-gb / --profile): profilers then see the late point of learning only')
    ap.add_argument('--random', action='store_true',
                    help="no agent: the on-device random script drives the same environment (the step kernel's workload beside the agent's)")
    args = ap.parse_args()
    N = args.envs
    cfg = make_config(0, n_envs=N)
    env = VecRanSlice(n_envs=N, cfg=cfg, fading=synth_traces(10000, args.profile))
    agent = VecKBRL(N, [10] * 5, cfg.n_prbs, accuracy_range=(0.99, 0.999), capacity=args.capacity,
                    pool_bytes=int(args.pool_gb * 2 ** 30))
    rng = np.random.default_rng(0)
    ia = rng.integers(EMBB_A[0], EMBB_A[1], size=(N, 5)).astype(np.int32)   # scenario_creator.py:220-221
    sf = rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(N, 5)).astype(np.int32)
    env.reset()
    agent.reset(ia, sf)
    import ctypes as C
    env._check(env.L.rs_step(env.h, ia.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))

    step_no = [0]

    def run(k):
        for _ in range(k):
            if args.random:
                env.random_actions(7, step_no[0])
                step_no[0] += 1
            else:
                agent.step_resident(env)
            env.step_resident()
    if args.load_state:
        env.load_state(np.load(os.path.join(args.load_state, 'env.npy'), mmap_mode='r'))
        agent.load_state(np.load(os.path.join(args.load_state, 'agent.npy'), mmap_mode='r'))
    elif args.random:
        run(args.warmup)
    else:
        agent.run_resident(env, args.warmup, graph=True)   # (the graph-replayed loop: same results, a third faster to get there)
    env.synchronize(); agent.synchronize()
    if args.save_state:
        os.makedirs(args.save_state, exist_ok=True)
        np.save(os.path.join(args.save_state, 'env.npy'), env.save_state())
        np.save(os.path.join(args.save_state, 'agent.npy'), agent.save_state())
        print(json.dumps({'saved': args.save_state, 'after_steps': args.warmup, 'dictionary_size_mean': float(agent.dictionary_sizes().mean())}))
        return
    s0 = agent.stats()
    w0 = agent.repair_work()
    c0 = env.counters()
    agent.set_kernel_timing(True)
    env.set_kernel_timing(True)
    t0 = time.perf_counter()
    run(args.steps)
    env.synchronize(); agent.synchronize()
    dt = time.perf_counter() - t0
    s1 = agent.stats()
    c1 = env.counters()
    per = lambda i: float(c1[i] - c0[i]) / (N * args.steps)
    ph = agent.phase_times_ms()
    w1 = agent.repair_work()
    kb_n = ph['n_update'] + ph['n_select']
    kb_ms = (ph['update_ms'] * ph['n_update'] + ph['select_ms'] * ph['n_select']) / max(1, kb_n)

    def roof(kind, ms_key, n_key):
        # bytes from the kernels' own work plan over the time of ALL their launches in the window (HIP events; the later rounds
        # of a step often find nothing left and return at once: they are in the time and in the count), against 8 TB/s
        nw = w1[kind + '_launches'] - w0[kind + '_launches']
        if not nw or not ph[ms_key]:
            return None
        total = w1[kind + '_bytes'] - w0[kind + '_bytes']
        gbs = total / (ph[ms_key] * ph[n_key] * 1e-3) / 1e9
        return {'bytes_per_step': total / args.steps, 'launch_ms_mean': ph[ms_key], 'launches_timed': ph[n_key], 'launches_with_work': nw,
                'ms_per_step': ph[ms_key] * ph[n_key] / args.steps, 'achieved_GBs': gbs, 'frac_of_8TBs': gbs / 8000.0}
    env_ms, _ = env.kernel_time_ms()
    out = env.fetch()
    evals = s1[3] - s0[3]
    sizes = agent.dictionary_sizes()
    pool = agent.pool()
    print(json.dumps({
        'config': 'scenario_0, %d envs + KBRL agent per env, closed loop on device, dictionary capacity %d' % (N, args.capacity),
        'env_steps_per_s': N * args.steps / dt, 'ms_per_step': 1e3 * dt / args.steps,
        'embb_kernel_ms': env_ms, 'kb_kernel_ms_mean_of_update_and_select': kb_ms, 'kb_launches': kb_n,
        'kernel_evaluations_per_s': evals / dt, 'predicts_per_env_step': (s1[0] - s0[0]) / (N * args.steps),
        'mistakes_per_env_step': (s1[1] - s0[1]) / (N * args.steps),
        'dictionary_size_mean': float(sizes.mean()), 'dictionary_size_max': int(sizes.max()),
        'dictionary_size_p50_p90_p99': [float(np.percentile(sizes, q)) for q in (50, 90, 99)], 'pool': pool,
        'mean_action_sum': float(out['actions'].sum(axis=1).mean()),
        'action_per_slice_p10_p50_p90_max': [float(np.percentile(out['actions'], q)) for q in (10, 50, 90, 100)],
        'step_workload_per_env_step': {'fading_samples': per(0), 'pf_iterations': per(2), 'ue_slots': per(3)},
        'kb_update_phase_ms': ph['update_ms'], 'kb_select_ms': ph['select_ms'],
        'per_step_ms': {k2: ph[k2 + '_launch_ms'] * ph['n_' + k2] / args.steps for k2 in ('matvec', 'rank1', 'finish', 'update_small', 'select_bin', 'select_gemm')},
        'select_bin_GBs': (float(sizes.sum()) * 68.0 / (ph['select_bin_launch_ms'] * 1e-3) / 1e9) if ph['select_bin_launch_ms'] else None,
        'kinv_streaming': {'heavy_matvec_kernel': roof('matvec', 'matvec_launch_ms', 'n_matvec'), 'heavy_rank1_kernel': roof('rank1', 'rank1_launch_ms', 'n_rank1')},
        'stamps_raw_w4_w7': [int(x) for x in agent._repair_raw()[4:8]],
        'direct_passes_per_step': (w1['direct_passes'] - w0['direct_passes']) / args.steps,
        'direct_landmarks_per_step': (w1['direct_landmarks'] - w0['direct_landmarks']) / args.steps,
        'driver': 'random script' if args.random else 'KBRL agents',
        'violations_per_env_step_last': float(out['violations'].sum(axis=1).mean()),
    }))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""BASELINE config 3 at the REFERENCE's run length (experiments_kbrl.py:22: 50,400 steps): N replicas of scenario_0 with one
KBRL agent each, closed loop on the device, the dictionary pool sized from the device's free memory -- how many replicas of
one GPU get through a run of that length without a dictionary that had to project for want of storage?

  python tools/run_length.py [--envs 2048] [--steps 50400] [--budget-s 900] [--profile tdl] > profiles/<tag>_run_length.txt
Prints one progress line per --report steps and a JSON summary: steps done, wall time, env-steps/s over the whole run and over
the last window, dictionary sizes, pool use, replicas flagged (kb_get_flags: capacity / pool).  Stops early at --budget-s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
import numpy as np  # noqa: E402
from ranslice import _lib  # noqa: E402
from ranslice.config import make_config, EMBB_A, EMBB_SEC  # noqa: E402
from ranslice.fading import synth_traces  # noqa: E402
from ranslice.kbrl_dev import VecKBRL  # noqa: E402
from ranslice.vec_env import VecRanSlice  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=2048)
    ap.add_argument('--steps', type=int, default=50400)
    ap.add_argument('--report', type=int, default=4200)
    ap.add_argument('--chunk', type=int, default=600)
    ap.add_argument('--capacity', type=int, default=16384)
    ap.add_argument('--headroom-gb', type=float, default=24.0)
    ap.add_argument('--budget-s', type=float, default=900.0)
    ap.add_argument('--profile', default='tdl')
    args = ap.parse_args()
    import ctypes as C
    N = args.envs
    cfg = make_config(0, n_envs=N)
    env = VecRanSlice(n_envs=N, cfg=cfg, fading=synth_traces(10000, args.profile))
    free, total = _lib.device_mem_info(0)
    pool_bytes = _lib.default_pool_bytes(0, headroom=int(args.headroom_gb * 2 ** 30))
    agent = VecKBRL(N, [10] * 5, cfg.n_prbs, accuracy_range=(0.99, 0.999), capacity=args.capacity, pool_bytes=pool_bytes)
    rng = np.random.default_rng(0)
    ia = rng.integers(EMBB_A[0], EMBB_A[1], size=(N, 5)).astype(np.int32)   # scenario_creator.py:220-221
    sf = rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(N, 5)).astype(np.int32)
    env.reset()
    agent.reset(ia, sf)
    env._check(env.L.rs_step(env.h, ia.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
    print('# %d replicas x %d steps, scenario_0 + one KBRL agent per replica (%s traces); device memory %.1f GB free of %.1f, pool %.1f GB, capacity %d'
          % (N, args.steps, args.profile, free / 1e9, total / 1e9, pool_bytes / 1e9, args.capacity), flush=True)
    t0 = time.perf_counter()
    done, last_t, last_done = 0, t0, 0
    windows = []
    while done < args.steps:
        k = min(args.chunk, args.steps - done)
        agent.run_resident(env, k, graph=True)
        done += k
        if done % args.report == 0 or done == args.steps:
            env.synchronize()
            agent.synchronize()
            now = time.perf_counter()
            sizes = agent.dictionary_sizes()
            pool = agent.pool()
            w = dict(step=done, wall_s=now - t0, ms_per_step=1e3 * (now - last_t) / (done - last_done),
                     env_steps_per_s=N * (done - last_done) / (now - last_t), dict_mean=float(sizes.mean()), dict_max=int(sizes.max()),
                     pool_used_gb=pool['used_bytes'] / 1e9, saturated=pool['saturated'], pool_full=pool['pool_full'])
            windows.append(w)
            print('step %6d  %7.1f s  %.3f ms/step  %.3g env-steps/s  dictionaries mean %.0f max %d  pool %.1f GB  flagged: capacity %d pool %d'
                  % (done, w['wall_s'], w['ms_per_step'], w['env_steps_per_s'], w['dict_mean'], w['dict_max'], w['pool_used_gb'],
                     w['saturated'], w['pool_full']), flush=True)
            last_t, last_done = now, done
            if now - t0 > args.budget_s:
                break
    env.synchronize()
    agent.synchronize()
    wall = time.perf_counter() - t0
    fl = agent.flagged_replicas()
    out = env.fetch()
    assert np.isfinite(out['reward']).all()
    print(json.dumps({'replicas': N, 'steps_done': done, 'steps_asked': args.steps, 'wall_s': wall, 'env_steps_per_s_whole_run': N * done / wall,
                      'ms_per_step_whole_run': 1e3 * wall / done, 'last_window': windows[-1] if windows else None,
                      'pool_bytes': pool_bytes, 'replicas_flagged_capacity': len(fl['saturated']), 'replicas_flagged_pool': len(fl['pool_full']),
                      'replicas_unflagged': N - len(set(fl['saturated']) | set(fl['pool_full']))}))
    env.close()
    agent.close()


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Large oracle-vs-HIP comparison of the production step instance (no trace): every replica, bit for bit.
usage: python tests/soak_check.py [scenario] [n_envs] [steps] [seed]   (needs a GPU; developer tool)"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def oracle_run(args):
    scenario, seed, acts, churn = args
    from oracle import pyoracle as po
    from ranslice.config import make_config
    from test_gpu_parity import _churn
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'fading_small.npz'))
    cfg = make_config(scenario, n_envs=1)
    if churn:
        _churn(cfg)
    o = po.OracleEnv(cfg, [g['t0'], g['t1'], g['t2']])
    o.set_seed(seed)
    o.reset()
    out = []
    for a in acts:
        r = o.step(a)
        out.append((r['obs'].copy(), r['reward'], r['labels'].copy(), r['violations'].copy(), r['info'].copy()))
    return out


def main():
    scenario = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 9000
    churn = os.environ.get('SOAK_CHURN', '1') == '1'
    from ranslice.config import make_config
    from ranslice.vec_env import VecRanSlice
    from test_gpu_parity import _actions, _churn
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'fading_small.npz'))
    cfg = make_config(scenario, n_envs=n)
    if churn:
        _churn(cfg)
    rng = np.random.default_rng(seed0)
    ns = cfg.n_embb + cfg.n_mmtc
    acts = [_actions(rng, n, ns, cfg.n_prbs, i) for i in range(steps)]
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        fut = ex.map(oracle_run, [(scenario, seed0 + r, [a[r] for a in acts], churn) for r in range(n)], chunksize=4)
        env = VecRanSlice(n_envs=n, cfg=cfg, fading=[g['t0'], g['t1'], g['t2']], seed=seed0)
        if os.environ.get('SOAK_GROUP'):
            env.set_group_size(int(os.environ['SOAK_GROUP']))
        env.reset()
        hip = []
        for a in acts:
            obs, rew, done, info = env.step(a)
            hip.append((obs, rew, info['SLA_labels'], info['violations'], env.l1_info()))
        bad = 0
        for r, ref in enumerate(fut):
            for i in range(steps):
                o, w, l, v, inf = ref[i]
                h = hip[i]
                if (h[0][r].tobytes() != o.tobytes() or h[1][r] != w or (h[2][r] != l).any() or (h[3][r] != v).any()
                        or h[4][r].tobytes() != inf.tobytes()):
                    bad += 1
                    print('MISMATCH replica %d step %d' % (r, i))
                    break
    print('scenario %d: %d replicas x %d steps, %d mismatching replicas, %.1f s' % (scenario, n, steps, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())

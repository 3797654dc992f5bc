"""Where do two identical runs of the shared-dictionary loop first differ? (developer check)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
from ranslice.config import make_config, EMBB_A, EMBB_SEC, MMTC_A, MMTC_SEC  # noqa: E402
from ranslice.fading import synth_fading  # noqa: E402
from ranslice.kbrl_dev import SharedVecKBRL  # noqa: E402
from ranslice.vec_env import VecRanSlice  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cfg = make_config(2, n_envs=N)
fading = [synth_fading(t, 10000) for t in range(3)]
dims = [10] * cfg.n_embb + [3] * cfg.n_mmtc


def make():
    env = VecRanSlice(n_envs=N, cfg=cfg, fading=fading)
    agent = SharedVecKBRL(N, dims, cfg.n_prbs, budget=64, max_rounds=4, capacity=256)
    rng = np.random.default_rng(1000)
    ia = np.concatenate([rng.integers(EMBB_A[0], EMBB_A[1], size=(N, cfg.n_embb)),
                         rng.integers(MMTC_A[0], MMTC_A[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
    sf = np.concatenate([rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(N, cfg.n_embb)),
                         rng.integers(MMTC_SEC[0], MMTC_SEC[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
    state = env.reset()
    agent.reset(ia, sf)
    return dict(env=env, agent=agent, state=state, action=ia.copy())


A, B = make(), make()
for i in range(STEPS):
    res = []
    for R in (A, B):
        obs, rew, _, info = R['env'].step(R['action'])
        hits = R['agent'].update_control(R['state'], R['action'], info['SLA_labels'])
        sizes = [R['agent'].learner(0, s)['m'] for s in range(len(dims))]
        coeff = [np.asarray(R['agent'].learner(0, s)['coeff']).copy() for s in range(len(dims))]
        action, adj = R['agent'].select_action(obs)
        R['state'], R['action'] = obs, action
        res.append(dict(obs=obs, hits=hits, sizes=sizes, coeff=coeff, action=action, rounds=R['agent'].rounds_last))
    a, b = res
    diffs = []
    if a['obs'].tobytes() != b['obs'].tobytes():
        diffs.append('obs')
    if (a['hits'] != b['hits']).any():
        diffs.append('hits')
    if a['sizes'] != b['sizes']:
        diffs.append('sizes %s %s' % (a['sizes'], b['sizes']))
    for s in range(len(dims)):
        if a['coeff'][s].tobytes() != b['coeff'][s].tobytes():
            diffs.append('coeff[%d] maxdiff %.3g' % (s, np.abs(a['coeff'][s] - b['coeff'][s]).max() if a['coeff'][s].shape == b['coeff'][s].shape else -1))
    if (a['action'] != b['action']).any():
        diffs.append('action (%d replicas)' % int((a['action'] != b['action']).any(axis=1).sum()))
    if diffs:
        print('step %d: first difference: %s' % (i, '; '.join(diffs)))
        sys.exit(1)
print('no difference in %d steps; sizes %s' % (STEPS, res[0]['sizes']))

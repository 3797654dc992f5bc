import os, sys, time, numpy as np
sys.path.insert(0, '/root/repo/network-slicing_amd')
from ranslice.config import make_config, EMBB_A, EMBB_SEC, MMTC_A, MMTC_SEC
from ranslice.fading import synth_fading
from ranslice.kbrl_dev import SharedVecKBRL, merge_proposals
from ranslice.vec_env import VecRanSlice
import ranslice.kbrl_dev as kd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = make_config(2, n_envs=N)
env = VecRanSlice(n_envs=N, cfg=cfg, fading=[synth_fading(t, 10000) for t in range(3)])
dims = [10] * cfg.n_embb + [3] * cfg.n_mmtc
CAP = int(os.environ.get('SHARED_CAP', '1024'))
agent = SharedVecKBRL(N, dims, cfg.n_prbs, budget=int(os.environ.get('SHARED_BUDGET', '64')), max_rounds=int(os.environ.get('SHARED_ROUNDS', '4')), capacity=CAP)
rng = np.random.default_rng(1000)
ia = np.concatenate([rng.integers(EMBB_A[0], EMBB_A[1], size=(N, cfg.n_embb)), rng.integers(MMTC_A[0], MMTC_A[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
sf = np.concatenate([rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(N, cfg.n_embb)), rng.integers(MMTC_SEC[0], MMTC_SEC[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
state = env.reset(); agent.reset(ia, sf); action = ia.copy()
T = dict(step=0.0, update=0.0, select=0.0)
# wrap library calls for timing
tl = {}
L = agent.L
def wrap(name):
    f = getattr(L, name)
    def g(*a):
        t = time.perf_counter(); r = f(*a); tl[name] = tl.get(name, 0.0) + time.perf_counter() - t; return r
    return g
class P:  # proxy
    def __getattr__(self, n):
        return wrap(n) if n.startswith('kb_shared') else getattr(L, n)
agent.L = P()
om = kd.merge_proposals
def mm(*a):
    t = time.perf_counter(); r = om(*a); tl['merge'] = tl.get('merge', 0.0) + time.perf_counter() - t; return r
kd.merge_proposals = mm
steps = 60
for i in range(steps):
    t = time.perf_counter(); obs, rew, _, info = env.step(action); T['step'] += time.perf_counter() - t
    t = time.perf_counter(); agent.update_control(state, action, info['SLA_labels']); T['update'] += time.perf_counter() - t
    t = time.perf_counter(); action, adj = agent.select_action(obs); T['select'] += time.perf_counter() - t
    state = obs
print('N', N, {k: round(1e3 * v / steps, 3) for k, v in T.items()}, 'ms/step;', {k: round(1e3 * v / steps, 3) for k, v in tl.items()}, 'sizes', [agent.learner(0, s)['m'] for s in range(len(dims))])

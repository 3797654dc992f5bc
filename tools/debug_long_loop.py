"""debug aid: replica R of the 4096-replica long closed loop, device (several repair paths) vs oracle"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import pyoracle as po
from ranslice.config import make_config
from ranslice.fading import synth_fading
from ranslice.kbrl_dev import VecKBRL
from ranslice.sharding import replica_seed, replica_seeds
from ranslice.vec_env import VecRanSlice

R, steps = int(sys.argv[1]), int(sys.argv[2])
N0 = 4096
rng = np.random.default_rng(12)
ia = rng.integers(10, 35, size=(N0, 5)).astype(np.int32)
sf = rng.integers(2, 8, size=(N0, 5)).astype(np.int32)
lo = (R // 64) * 64
n = 64
fading = [synth_fading(t, 10000) for t in range(3)]
# oracle
e = po.OracleEnv(make_config(0), fading); e.set_seed(replica_seed(301, R)); e.reset()
a = po.OracleKBRL([10] * 5, 200, ia[R], sf[R], capacity=2048); a.set_seed(9 + R)
state = np.zeros(e.n_vars, dtype=np.float32); action = ia[R].copy()
ref = []
for i in range(steps):
    r = e.step(action)
    a.update_control(state, action, r['labels'])
    na, adj = a.select_action(r['obs']); a.adjusted = adj
    ref.append((na.copy(), [a.m(s) for s in range(5)], a.margins.copy(), a.security_factors.copy()))
    state, action = r['obs'], na
for name, envv in (('default', {}), ('rounds3', {'KBRL_ROUNDS': '3'}), ('rounds0', {'KBRL_ROUNDS': '0'}), ('inline', {'KBRL_HEAVY_M': '1000000'})):
    for k in ('KBRL_ROUNDS', 'KBRL_HEAVY_M'):
        os.environ.pop(k, None)
    os.environ.update(envv)
    env = VecRanSlice(n_envs=n, cfg=make_config(0, n_envs=n), fading=fading)
    ag = VecKBRL(n, [10] * 5, 200, capacity=2048, pool_bytes=8 << 30)
    env.reset(seeds=np.array([replica_seed(301, lo + r) for r in range(n)], dtype=np.uint64))
    ag.reset(ia[lo:lo + n], sf[lo:lo + n], seeds=np.arange(lo, lo + n, dtype=np.uint64) + 9)
    a0 = np.ascontiguousarray(ia[lo:lo + n])
    env._check(env.L.rs_step(env.h, a0.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
    bad = None
    for i in range(steps):
        ag.step_resident(env)
        nxt = env.fetch()['actions'][R - lo]
        sizes = ag.dictionary_sizes()[R - lo].tolist()
        if bad is None and ((nxt != ref[i][0]).any() or sizes != ref[i][1]):
            c = ag.control(with_accuracies=False)
            bad = (i, nxt.tolist(), ref[i][0].tolist(), sizes, ref[i][1], c['margins'][R - lo].tolist(), ref[i][2].tolist(),
                   c['security_factors'][R - lo].tolist(), ref[i][3].tolist())
            break
        if i + 1 < steps:
            env.step_resident()
    print(name, 'first mismatch:', bad)
    env.close(); ag.close()

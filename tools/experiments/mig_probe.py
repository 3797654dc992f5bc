#!/usr/bin/env python3
"""Developer aid: step time and task sets handed over per step (migration in embb_step_kernel) on the bench's random script.
(with tools/experiments/step_migration.patch applied) RANSLICE_MIGRATE=0|1 python tools/experiments/mig_probe.py [--kbrl]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
import numpy as np
from ranslice.config import make_config
from ranslice.fading import synth_fading
from ranslice.vec_env import VecRanSlice
N = int(os.environ.get('PROFILE_ENVS', '4096'))
KBRL = '--kbrl' in sys.argv
env = VecRanSlice(n_envs=N, cfg=make_config(0, n_envs=N), fading=[synth_fading(t, 10000) for t in range(3)])
env.reset()
if KBRL:
    env.set_schedule_hint(1)
    from ranslice.kbrl_dev import VecKBRL
    agent = VecKBRL(N, [10] * 5, 200, capacity=512)
    rng = np.random.default_rng(0)
    ia = rng.integers(4, 20, size=(N, 5)).astype(np.int32)
    agent.reset(ia, rng.integers(2, 8, size=(N, 5)).astype(np.int32))
    env._check(env.L.rs_step(env.h, ia.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
def advance(i):
    if KBRL: agent.step_resident(env)
    else: env.random_actions(2024, i)
    env.step_resident()
for i in range(300 if KBRL else 1500): advance(i)
env.synchronize()
a = (C.c_uint64 * 16)(); env.L.rs_get_section_profile(env.h, a); b0 = a[0]
K = 300
t = time.time()
for i in range(K): advance(5000 + i)
env.synchronize()
dt = time.time() - t
env.L.rs_get_section_profile(env.h, a)
print('MIGRATE=%s %s: %.3f ms/step (wall, %d steps), task sets handed over per step %.1f' % (os.environ.get('RANSLICE_MIGRATE', '1'), 'kbrl' if KBRL else 'random', 1e3 * dt / K, K, (a[0] - b0) / K))

"""developer probe: ms/step of the device-resident KBRL loop over time (where does a long run slow down?)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
from ranslice.config import make_config, EMBB_A, EMBB_SEC  # noqa: E402
from ranslice.kbrl_dev import VecKBRL  # noqa: E402
from ranslice.vec_env import VecRanSlice  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
BLOCKS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = make_config(0, n_envs=N)
env = VecRanSlice(n_envs=N, cfg=cfg)
agent = VecKBRL(N, [10] * 5, 200, capacity=1024)
rng = np.random.default_rng(0)
ia = rng.integers(EMBB_A[0], EMBB_A[1], size=(N, 5)).astype(np.int32)
sf = rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(N, 5)).astype(np.int32)
env.reset()
agent.reset(ia, sf)
env._check(env.L.rs_step(env.h, ia.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
for b in range(BLOCKS):
    t0 = time.time()
    agent.set_kernel_timing(True)
    env.set_kernel_timing(True)
    for i in range(500):
        agent.step_resident(env)
        env.step_resident()
    env.synchronize()
    agent.synchronize()
    dt = time.time() - t0
    kb, _ = agent.kernel_time_ms()
    em, _ = env.kernel_time_ms()
    f = env.fetch()
    c = env.counters()
    print('steps %5d: %.2f ms/step (embb kernel %.2f, kb kernels %.2f each); dict max %d mean %.1f; mean PRBs %.1f; UE-slots/step/task %.2f'
          % ((b + 1) * 500, 2 * dt, em, kb, agent.dictionary_sizes().max(), agent.dictionary_sizes().mean(),
             f['actions'].sum(axis=1).mean(), c[3] / ((b + 1) * 500 * 50 * N * 5)), flush=True)

#!/usr/bin/env python3
"""kernel time of the eMBB step for each group size and batch size (developer tool)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
from ranslice.config import make_config
from ranslice.fading import synth_fading
from ranslice.vec_env import VecRanSlice
fading = [synth_fading(t, 10000) for t in range(3)]
SC = int(os.environ.get('SWEEP_SCENARIO', '0'))
GROUPS = [int(g) for g in os.environ.get('SWEEP_GROUPS', '8,16,32').split(',')]
for N in [int(x) for x in (sys.argv[1:] or ['4096', '16384'])]:
    for g in GROUPS:
        env = VecRanSlice(n_envs=N, cfg=make_config(SC, n_envs=N), fading=fading)
        env.set_group_size(g)
        env.reset()
        for i in range(400):
            env.random_actions(2024, i); env.step_resident()
        env.synchronize()
        env.set_kernel_timing(True)
        t0 = time.perf_counter()
        K = 100
        for i in range(K):
            env.random_actions(2024, 400 + i); env.step_resident()
        env.synchronize()
        dt = time.perf_counter() - t0
        ms, n = env.kernel_time_ms()
        print('scenario %d ' % SC, end='')
        print('N=%6d group=%2d : %.3f ms/step (embb kernels %.3f ms)  %.2f M env-steps/s' % (N, g, 1e3 * dt / K, ms, N * K / dt / 1e6), flush=True)
        env.close()

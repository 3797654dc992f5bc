#!/usr/bin/env python3
"""Per-section cycle shares of embb_step_kernel (needs `make -C network-slicing_amd/csrc profile`).
RANSLICE_LIB=network-slicing_amd/csrc/build/libranslice_prof.so python tools/section_profile.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
from ranslice.config import make_config  # noqa: E402
from ranslice.fading import synth_fading  # noqa: E402
from ranslice.vec_env import VecRanSlice  # noqa: E402

N = 4096
env = VecRanSlice(n_envs=N, cfg=make_config(0, n_envs=N), fading=[synth_fading(t, 10000) for t in range(3)])
env.reset()
for i in range(400):
    env.random_actions(2024, i)
    env.step_resident()
env.synchronize()
a = (C.c_uint64 * 8)()
env.L.rs_get_section_profile(env.h, a)
base = list(a)
K = 100
for i in range(K):
    env.random_actions(2024, 400 + i)
    env.step_resident()
env.synchronize()
env.L.rs_get_section_profile(env.h, a)
d = [a[i] - base[i] for i in range(8)]
names = ['arrivals+departures', 'traffic_step', 'fading walker + e_snr', 'PF loop', 'RB scan + response',
         'reception + tx_step', 'update_info', 'unused']
trips = d[7]
d[7] = 0
tot = sum(d) or 1
waves = N * 5 / 2
for n, v in zip(names, d):
    print('%-24s %6.2f%%   %9.0f cycles/wave/step' % (n, 100.0 * v / tot, v / K / waves))
print('total %.0f cycles/wave/step' % (tot / K / waves))
print('PF loop trips per wave per slot: %.2f' % (trips / K / waves / 50))

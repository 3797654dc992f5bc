#!/usr/bin/env python3
"""Mean wave pace (cycles per slot) per XCD of the eMBB step kernel; needs a -DRS_PACE_XCC build:
RANSLICE_LIB=network-slicing_amd/csrc/build/libranslice_xcc.so python tools/xcc_pace.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
from ranslice.config import make_config  # noqa: E402
from ranslice.fading import synth_fading  # noqa: E402
from ranslice.vec_env import VecRanSlice  # noqa: E402

N = int(os.environ.get('PROFILE_ENVS', '4096'))
env = VecRanSlice(n_envs=N, cfg=make_config(0, n_envs=N), fading=[synth_fading(t, 10000) for t in range(3)])
env.reset()
for i in range(1500):
    env.random_actions(2024, i)
    env.step_resident()
env.synchronize()
a = (C.c_uint64 * 16)()
env.L.rs_get_section_profile(env.h, a)
base = list(a)
for i in range(200):
    env.random_actions(2024, 1500 + i)
    env.step_resident()
env.synchronize()
env.L.rs_get_section_profile(env.h, a)
d = [a[i] - base[i] for i in range(16)]
for x in range(8):
    print('XCC %d: waves/launch %.0f mean pace %.0f cycles/slot' % (x, d[2 * x + 1] / 200, d[2 * x] / max(1, d[2 * x + 1])))

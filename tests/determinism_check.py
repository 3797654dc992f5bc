"""Run-to-run determinism of the resident step loop (developer check): the same seeds and action script twice in
one process and the hash printed for comparison across processes.  usage: python tests/determinism_check.py [scenario] [n]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
from ranslice.config import make_config  # noqa: E402
from ranslice.fading import synth_fading  # noqa: E402
from ranslice.vec_env import VecRanSlice  # noqa: E402

sc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
fading = [synth_fading(t, 10000) for t in range(3)]
hs = []
for rep in range(2):
    env = VecRanSlice(n_envs=N, cfg=make_config(sc, n_envs=N), fading=fading)
    env.reset()
    h = hashlib.sha256()
    for i in range(150):
        env.random_actions(2024, i)
        env.step_resident()
        if i % 10 == 9:
            f = env.fetch()
            for k in ('obs', 'reward', 'labels', 'violations'):
                h.update(f[k].tobytes())
            h.update(env.l1_info().tobytes())
    hs.append(h.hexdigest())
    env.close()
print('scenario %d N=%d: %s %s' % (sc, N, hs[0][:16], 'same' if hs[0] == hs[1] else 'DIFFERENT ' + hs[1][:16]))

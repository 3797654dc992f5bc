"""Statistical sanity check (last paragraph): the REFERENCE's own KBRL loop on the build's synthetic
fading traces, in this container only -- violations/step, mean PRBs, adjusted rate, hit rate and dictionary sizes
after STEPS steps, to set beside experiments_kbrl.BatchedEvaluator's numbers for the same length (they are not
comparable draw for draw: different random streams).  usage: python tools/ref_kbrl_stats.py [steps] [seed] [profile: sos | tdl]"""
import io
import contextlib
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'network-slicing_amd'))
import refharness as rh  # noqa: E402
from ranslice.fading import synth_traces  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
profile = sys.argv[3] if len(sys.argv) > 3 else 'sos'
tabs = synth_traces(10000, profile)
for k in [m for m in sys.modules if m.startswith('ranslice')]:
    del sys.modules[k]
sys.path.remove(os.path.join(os.path.dirname(HERE), 'network-slicing_amd'))
rh.setup(tempfile.mkdtemp(prefix='refstats_'), tabs)
import scenario_creator as sc  # noqa: E402  (the reference's)
np.random.seed(seed)
rng = np.random.default_rng(seed)
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    env = sc.create_env(rng, 0)
    agent = sc.create_kbrl_agent(rng, 0, accuracy_range=[0.99, 0.999])
    res = agent.run(env, steps)
dt = time.time() - t0
sizes = [h.algorithm.sv.landmarks.shape[0] if np.ndim(h.algorithm.sv.landmarks) > 1 else 1 for h in agent.learners]
half = steps // 2
print('reference KBRL, scenario_0, traces %s, seed %d, %d steps (%.0f s): violations/step %.4f (second half %.4f), mean PRBs %.1f, '
      'adjusted %.3f, hit rate %.3f, dictionary sizes %s'
      % (profile, seed, steps, dt, res['violation'].mean(), res['violation'][half:].mean(), res['resources'].mean(),
         res['adjusted'].mean(), res['hits'].mean(), sizes))

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

N = int(os.environ.get('PROFILE_ENVS', '4096'))
KBRL = '--kbrl' in sys.argv  # drive the env with one KBRL agent per replica instead of random actions
# --load-state DIR [--traces tdl|sos]: the closed loop from a checkpoint of tools/bench_kbrl.py --save-state (the late point of learning)
LOAD = sys.argv[sys.argv.index('--load-state') + 1] if '--load-state' in sys.argv else ''
TRACES = sys.argv[sys.argv.index('--traces') + 1] if '--traces' in sys.argv else 'sos'
if LOAD:
    import numpy as np
    from ranslice.fading import synth_traces
    from ranslice.kbrl_dev import VecKBRL
    KBRL = True
    env = VecRanSlice(n_envs=N, cfg=make_config(0, n_envs=N), fading=synth_traces(10000, TRACES))
    agent = VecKBRL(N, [10] * 5, 200, accuracy_range=(0.99, 0.999), capacity=4096, pool_bytes=64 << 30)
    env.reset()
    agent.reset(np.full((N, 5), 10, np.int32), np.full((N, 5), 3, np.int32))
    env.load_state(np.load(os.path.join(LOAD, 'env.npy'), mmap_mode='r'))
    agent.load_state(np.load(os.path.join(LOAD, 'agent.npy'), mmap_mode='r'))
else:
    env = VecRanSlice(n_envs=N, cfg=make_config(0, n_envs=N), fading=[synth_fading(t, 10000) for t in range(3)])
    env.reset()
if KBRL and not LOAD:
    env.set_schedule_hint(1)  # what kb_step_resident selects: the BLOCK instance
    import numpy as np
    from ranslice.kbrl_dev import VecKBRL
    agent = VecKBRL(N, [10] * 5, 200, capacity=512)
    rng = np.random.default_rng(0)
    ia = rng.integers(4, 20, size=(N, 5)).astype(np.int32)
    agent.reset(ia, rng.integers(2, 8, size=(N, 5)).astype(np.int32))
    env._check(env.L.rs_step(env.h, ia.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))


def advance(i):
    if KBRL:
        agent.step_resident(env)
    else:
        env.random_actions(2024, i)
    env.step_resident()


for i in range(20 if LOAD else (300 if KBRL else 1000)):
    advance(i)
env.synchronize()
a = (C.c_uint64 * 16)()
env.L.rs_get_section_profile(env.h, a)
base = list(a)
K = 100
for i in range(K):
    advance(1000 + i)
env.synchronize()
env.L.rs_get_section_profile(env.h, a)
d = [a[i] - base[i] for i in range(16)]
# section i accumulates the time from the previous mark to mark i
names = {9: 'slice arrivals', 11: 'PF trip: leader / runner-up reductions', 12: 'PF trip: leader run or closed form', 0: 'timer events', 1: 'traffic_step', 7: 'channel estimates ahead (chunk prologue)', 2: 'walker step + table read',
         8: 'PF set-up (MCS lookup, first metric)', 3: 'PF trip: take broadcast + loop control', 
         10: 'RB scan + R1 + R2 (MI of the evaluated spans, pairwise sums)', 4: 'R3 (inv_sigmoid + rx prob)', 5: 'reception draw + tx_step',
         6: 'update_info'}
trips = d[15]
runs = d[13]
d[13] = 0
d[15] = 0
print('leader-run iterations per task per slot: %.2f' % (runs / K / (N * 5) / 50))
slowest = a[14]
d[14] = 0
tot = sum(d) or 1
waves = N * 5 / 4  # 16 lanes per task
for i in (9, 7, 0, 1, 2, 8, 11, 12, 3, 10, 4, 5, 6):
    print('%-38s %6.2f%%   %9.0f cycles/wave/slot' % (names[i], 100.0 * d[i] / tot, d[i] / K / waves / 50))
print('slowest wave of any launch: %.0f cycles/slot (mean wave: %.0f)' % (slowest / 50, tot / K / waves / 50))
print('total %.0f cycles/wave/slot; PF loop trips per wave per slot: %.2f' % (tot / K / waves / 50, trips / K / waves / 50))

# per-task view of the last step: which tasks sit in the slowest waves?
import numpy as np
raw = np.zeros(N * 5 * 4 + 16, dtype=np.uint64)
env.L.rs_get_task_profile(env.h, raw.ctypes.data_as(C.POINTER(C.c_uint64)))
tp = raw[:N * 5 * 4].reshape(N * 5, 4).copy()
xcc = (tp[:, 1] >> np.uint64(32)).astype(np.int64) & 0xf
hwid = (tp[:, 2] >> np.uint64(32)).astype(np.int64)
start = (tp[:, 3] >> np.uint64(24)).astype(np.float64)
tp[:, 3] &= np.uint64(0xffffff)
tp[:, 1] &= np.uint64(0xffffffff)
tp[:, 2] &= np.uint64(0xffffffff)
sw = raw[N * 5 * 4:].astype(np.float64)
print('section split of the slowest wave of the run (cycles/slot):')
for i in (9, 7, 0, 1, 2, 8, 11, 12, 3, 10, 4, 5, 6):
    print('   %-38s %9.0f  %5.1f%%' % (names[i], sw[i] / 50, 100.0 * sw[i] / max(1.0, sw[:13].sum())))
cyc = tp[:, 0].astype(np.float64) / 50
print('last step: wave cycles/slot percentiles 50/90/99/99.9/max: %s' % np.percentile(cyc, [50, 90, 99, 99.9, 100]).round(0))
idx = np.argsort(-cyc)
seen = set()
print('slowest waves (cycles/slot : [n_ue, n_prb, pf_trips] of their tasks)')
for i in idx:
    c = cyc[i]
    if c in seen:
        continue
    seen.add(c)
    members = np.nonzero(cyc == c)[0]
    print('  %8.0f : %s' % (c, ' '.join('[%d,%d,%d]' % (tp[m, 1], tp[m, 2], tp[m, 3]) for m in members)))
    if len(seen) >= 12:
        break
# how well do simple predictors explain a task's wave time?
for name, v in (('n_ue*n_prb', tp[:, 1] * tp[:, 2]), ('pf_trips', tp[:, 3]), ('n_ue', tp[:, 1]), ('n_prb', tp[:, 2])):
    print('corr(wave cycles, %s) = %.3f' % (name, np.corrcoef(cyc, v.astype(np.float64))[0, 1]))

# where did the time go: is a wave slow because of its own tasks or because of the SIMD it shared?
simd = (hwid >> 4) & 3
cu = (hwid >> 8) & 0xf
se = (hwid >> 12) & 0xf  # sh + se bits
slot = ((xcc * 16 + se) * 16 + cu) * 4 + simd
# one entry per wave: the tasks of a wave (not consecutive under the cost-ranked order) share start time and place
_, wave_first = np.unique(np.stack([start, hwid.astype(np.float64), xcc.astype(np.float64)], axis=1), axis=0, return_index=True)
wc, ws = cyc[wave_first], slot[wave_first]
print('distinct SIMDs used: %d, waves %d' % (len(np.unique(ws)), len(wc)))
means = {}
for k in np.unique(ws):
    means[k] = wc[ws == k]
between = np.var([v.mean() for v in means.values()])
within = np.mean([v.var() for v in means.values()])
print('wave cycles/slot: overall var %.3g = between-SIMD %.3g + within-SIMD %.3g' % (wc.var(), between, within))
sm = np.array([v.mean() for v in means.values()])
smax = np.array([v.max() for v in means.values()])
cnt = np.array([len(v) for v in means.values()])
print('waves per SIMD: min %d max %d; per-SIMD mean cycles/slot pct 5/50/95/max: %s; per-SIMD max pct 5/50/95: %s' % (
    cnt.min(), cnt.max(), np.percentile(sm, [5, 50, 95, 100]).round(0), np.percentile(smax, [5, 50, 95]).round(0)))
cum = {}
for k in np.unique(slot // 4):
    cum[k] = wc[(ws // 4) == k]
cm = np.array([v.mean() for v in cum.values()])
print('per-CU mean cycles/slot pct 5/50/95/max: %s (CUs %d)' % (np.percentile(cm, [5, 50, 95, 100]).round(0), len(cm)))
xm = [wc[(ws // 1024) == x].mean() for x in np.unique(ws // 1024)]
print('per-XCC mean cycles/slot: %s' % np.round(xm, 0))

st = start[wave_first].copy()
wx = xcc[wave_first]
for x in np.unique(wx):  # the clock s_memtime reads is per XCD
    st[wx == x] -= st[wx == x].min()
en = st + wc * 50
print('wave start offsets (cycles) pct 50/90/99/max: %s; end pct 1/10/50/90/99/max: %s' % (
    np.percentile(st, [50, 90, 99, 100]).round(0), np.percentile(en, [1, 10, 50, 90, 99, 100]).round(0)))
late = st > 0.2 * en.max()
print('waves starting after 20%% of the kernel: %d' % late.sum())
for x in np.unique(wx):
    e = en[wx == x]
    print('  XCC %d: waves %d, late starters %d, end pct 50/max: %s' % (x, (wx == x).sum(), (late & (wx == x)).sum(), np.percentile(e, [50, 100]).round(0)))
cnt_by_cu = {}
for k in np.unique(ws // 4):
    cnt_by_cu[k] = ((ws // 4) == k).sum()
v = np.array(list(cnt_by_cu.values()))
print('waves per CU: min %d median %d max %d' % (v.min(), np.median(v), v.max()))

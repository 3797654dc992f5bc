#!/usr/bin/env python3
"""Developer aid: when and where every wave of the PRODUCTION step kernel ran (needs a -DRS_WAVE_LOG build of the library:
hipcc <Makefile flags> -DRS_WAVE_LOG -shared -o build/libranslice_wlog.so rs_api.hip;
RANSLICE_LIB=.../libranslice_wlog.so python tools/wave_log.py [--kbrl]).  Prints how the waves' durations split between and
within SIMDs / CUs / XCDs and how stable the block -> CU placement is from launch to launch."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
import numpy as np  # noqa: E402
from ranslice.config import make_config  # noqa: E402
from ranslice.fading import synth_fading  # noqa: E402
from ranslice.vec_env import VecRanSlice  # noqa: E402

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
    if KBRL:
        agent.step_resident(env)
    else:
        env.random_actions(2024, i)
    env.step_resident()


for i in range(int(os.environ.get('WARM', '300')) if KBRL else 1500):
    advance(i)
env.synchronize()
T = N * 5
W = (T + 3) // 4
raw = np.zeros(T * 4 + 16, dtype=np.uint64)
place = []
for k in range(6):
    advance(2000 + k)
    env.synchronize()
    env.L.rs_get_task_profile(env.h, raw.ctypes.data_as(C.POINTER(C.c_uint64)))
    w16 = raw[:W * 16].reshape(W, 16).astype(np.int64)
    w = w16[:, :4]
    dur, t0, hw, xcc = w[:, 0], w[:, 1], w[:, 2] & 0xffffffff, (w[:, 2] >> 32) & 7
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    sid = cuid * 4 + simd
    place.append(cuid.copy())
    # [1], [3]: s_memrealtime (100 MHz, one clock for the chip) at the wave's start and end; [0]: its s_memtime cycles
    start = (t0 - t0.min()).astype(np.float64)
    end = (w[:, 3] - t0.min()).astype(np.float64)
    print('launch %d: waves %d, span %.0f ticks of 10 ns; wave duration mean %.0f (pct 5/50/95/max %s); start pct 50/99/max %s' % (
        k, W, end.max(), dur.mean(), np.percentile(dur, [5, 50, 95, 100]).round(), np.percentile(start, [50, 99, 100]).round()))
    us = np.unique(sid)
    per = np.array([end[sid == s_].max() for s_ in us])
    cnt = np.array([(sid == s_).sum() for s_ in us])
    print('   SIMDs used %d (waves per SIMD min %d max %d); last end per SIMD pct 5/50/95/max %s  -> mean/max = %.3f' % (
        len(us), cnt.min(), cnt.max(), np.percentile(per, [5, 50, 95, 100]).round(), per.mean() / per.max()))
    tot = dur.var()
    means = np.array([dur[sid == s_].mean() for s_ in us])
    busy = np.array([np.sort(end[sid == s_]) for s_ in us])  # per SIMD: the ends of its waves in order
    span = end.max()
    # SIMD-time idle before the launch ends, and the time with fewer than 5 / fewer than 3 waves resident
    print('   idle SIMD-time %.1f %% of the launch; SIMD-time with < 5 waves %.1f %%, < 3 waves %.1f %%; waves ending in the last 10 %% of the launch: %d on %d SIMDs' % (
        100 * (1 - per.mean() / span), 100 * (1 - busy[:, 0].mean() / span), 100 * (1 - busy[:, 2].mean() / span),
        (end > 0.9 * span).sum(), len(np.unique(sid[end > 0.9 * span]))))
    print('   wave ends pct 1/10/50/90/99/100 of the span: %s' % (np.percentile(end, [1, 10, 50, 90, 99, 100]) / span).round(3))
    print('   duration variance: between SIMDs %.3g of %.3g; per-XCD mean end %s' % (
        ((means - dur.mean()) ** 2).mean(), tot, [int(end[xcc == x].mean()) for x in range(8)]))
    slow = np.argsort(-dur)[:8]
    print('   slowest waves (end as a share of the span : [UEs, RBs, PF rounds of the step, UE-slots] of its tasks):')
    for i_ in slow:
        tk = w16[i_, 4:8]
        print('      %.3f : %s' % (end[i_] / span, ' '.join('[%d,%d,%d,%d]' % (x & 0xff, (x >> 8) & 0xff, (x >> 16) & 0xffff, x >> 32) for x in tk)))
    # classes of waves that share a SIMD if the dispatcher deals round by round: wave index mod (number of SIMDs)
    R_ = len(us)
    if W % R_ == 0 and W > R_:
        cls = np.arange(W) % R_
        blk = np.arange(W) // 4
        same_simd = np.mean([len(np.unique(blk[sid == s_] % (R_ // 4))) == 1 for s_ in us[::37]])  # the blocks of a SIMD: one residue
        cend = np.array([end[cls == c_].max() for c_ in range(R_)])
        oct_ = [cend[i * R_ // 8:(i + 1) * R_ // 8].mean() / span for i in range(8)]
        print('   SIMDs whose five waves come from blocks b, b + %d, b + 2 x %d, ...: %.0f %% of those sampled; last end of the waves w, w + %d, ... by octile of w (share of the span): %s' % (
            R_ // 4, R_ // 4, 100 * same_simd, R_, ' '.join('%.3f' % v for v in oct_)))
    if k == 5:
        print('   wave indices per SIMD (six SIMDs): %s' % '; '.join(str(sorted(np.nonzero(sid == s_)[0].tolist())) for s_ in us[:3].tolist() + us[500:503].tolist()))
    ucu = np.unique(cuid)
    print('   CUs used %d; blocks per CU min %d max %d' % (len(ucu), min((cuid[::4] == c).sum() for c in ucu), max((cuid[::4] == c).sum() for c in ucu)))
# how the dispatcher deals the blocks of one XCD over its CUs (launch order = cost rank, heaviest first): the CU of each of the
# XCD's first 48 blocks, and per CU the ranks (within the XCD) of the five blocks it got
bx = xcc[::4]
bcu = (cuid[::4] % 256)
for x_ in (0, 5):
    ids = np.nonzero(bx == x_)[0]
    seq = bcu[ids]
    print('XCD %d: CU of its blocks in launch order (first 48): %s' % (x_, ' '.join('%d' % c for c in seq[:48])))
    per = {}
    for r_, c in enumerate(seq):
        per.setdefault(int(c), []).append(r_)
    some = sorted(per.items())[:6]
    print('   ranks of the blocks per CU (first six CUs): %s' % '; '.join('CU %d: %s' % (c, v) for c, v in some))
    sums = np.array([np.sum(v) for v in per.values()])
    print('   sum of ranks per CU: min %d max %d (even dealing: all equal %d)' % (sums.min(), sums.max(), 5 * (len(seq) - 1) // 2))
same = [(place[k] == place[k + 1]).mean() for k in range(len(place) - 1)]
print('block -> CU placement equal between consecutive launches (share of waves):', np.round(same, 3))
print('blocks 0..15 of the last launch: XCD', [int(x) for x in ((raw[:W * 16].reshape(W, 16)[::4, 2] >> np.uint64(32)) & np.uint64(7))[:16]])

"""GPU parity: the HIP simulator (through the C ABI, via VecRanSlice) against the CPU oracle on the
same Philox streams.  Everything is compared bit-for-bit: observations (f32 bits), rewards, SLA
labels, violation counts, the ten info accumulators per slice (f64 bits) and -- with allocation
tracing on -- every UE's e_snr / RBs / bits / queue / throughput / reception probability in every
slot.  No tolerance anywhere: both sides use include/rs_detmath.h and IEEE arithmetic in the
same order.
"""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from ranslice.config import make_config
from ranslice.sharding import replica_seed, replica_seeds  # noqa: F401
from ranslice.fading import synth_fading

pytestmark = pytest.mark.gpu


def _small_fading(golden_dir):
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    return [g['t0'], g['t1'], g['t2']]


def _churn(cfg):
    cfg.cbr_lambda, cfg.cbr_t_mean = 2.0 / 1.2, 0.6
    cfg.vbr_lambda, cfg.vbr_t_mean = 5.0 / 1.2, 0.6
    cfg.vbr_b_size, cfg.vbr_b_rate = 40, 12
    return cfg


def _actions(rng, n_envs, n_slices, n_prbs, step):
    mode = step % 6
    if mode == 0:
        a = rng.multinomial(n_prbs, [1.0 / n_slices] * n_slices, size=n_envs)
    elif mode == 2:
        a = rng.multinomial(n_prbs // 2, [1.0 / n_slices] * n_slices, size=n_envs)
        a[np.arange(n_envs), rng.integers(n_slices, size=n_envs)] = 0
    elif mode == 4:
        a = rng.integers(0, 4, size=(n_envs, n_slices))
    else:
        a = rng.multinomial(n_prbs, [1.0 / (n_slices + 1)] * (n_slices + 1), size=n_envs)[:, :n_slices]
    return a.astype(np.int32)


def _wide_actions(rng, n_envs, n_slices, n_prbs, step):
    """agent-like allocations: most of the carrier on one slice (long contested PF allocations over 60-85 RB pairs),
    a few RBs on the others, sometimes a second wide slice"""
    a = rng.integers(0, 6, size=(n_envs, n_slices))
    big = rng.integers(0, n_slices, size=n_envs)
    a[np.arange(n_envs), big] = rng.integers(n_prbs // 2, n_prbs - 5 * n_slices - 10, size=n_envs)
    if step % 3 == 2:  # two wide slices
        a[np.arange(n_envs), big] //= 2
        a[np.arange(n_envs), (big + 1) % n_slices] = a[np.arange(n_envs), big] - (step % 2)
    assert (a.sum(axis=1) <= n_prbs).all()
    return a.astype(np.int32)


def _mid_actions(rng, n_envs, n_slices, n_prbs, step):
    """slices around the width at which the BLOCK instances start taking block rounds (RS_BLOCK_PAIRS = 8 RB pairs): 13 to 40
    RBs each, odd widths (a short last pair) included, one slice of a replica sometimes a single RB pair"""
    a = rng.integers(13, 41, size=(n_envs, n_slices))
    a[np.arange(n_envs), rng.integers(0, n_slices, size=n_envs)] = 15 + (step % 4)  # 7 pairs + 1 RB, 8, 8 + 1 RB, 9
    if step % 5 == 4:
        a[np.arange(n_envs), rng.integers(0, n_slices, size=n_envs)] = 2
    assert (a.sum(axis=1) <= n_prbs).all()
    return a.astype(np.int32)


def _compare(scenario, n_envs, steps, fading, churn, seed0, check_trace=True, sample=None, group=None, hint=None,
             actions=None, tweak=None):
    from ranslice.vec_env import VecRanSlice
    cfg = make_config(scenario, n_envs=n_envs)
    if churn:
        _churn(cfg)
    if tweak:
        tweak(cfg)
    env = VecRanSlice(n_envs=n_envs, cfg=cfg, fading=fading, seed=seed0)
    if group is not None:
        env.set_group_size(group)
    if hint is not None:
        env.set_schedule_hint(hint)
    if check_trace:
        env.set_alloc_trace(True)
    obs0 = env.reset()
    assert not obs0.any()
    reps = list(range(n_envs)) if sample is None else list(sample)
    ocfg = make_config(scenario, n_envs=1)
    if churn:
        _churn(ocfg)
    if tweak:
        tweak(ocfg)
    oracles = []
    for r in reps:
        o = po.OracleEnv(ocfg, fading)
        o.set_seed(replica_seed(seed0, r))
        o.reset()
        oracles.append(o)
    rng = np.random.default_rng(99 + scenario)
    n_slices = cfg.n_embb + cfg.n_mmtc
    for i in range(steps):
        acts = (actions or _actions)(rng, n_envs, n_slices, cfg.n_prbs, i)
        obs, rew, done, info = env.step(acts)
        l1 = env.l1_info()
        tr = env.alloc_trace() if check_trace else None
        for k, r in enumerate(reps):
            out = oracles[k].step(acts[r], trace=check_trace)
            assert obs[r].tobytes() == out['obs'].tobytes(), 'obs bits differ: step %d replica %d' % (i, r)
            assert rew[r] == out['reward']
            assert (info['SLA_labels'][r] == out['labels']).all()
            assert (info['violations'][r] == out['violations']).all()
            assert l1[r].tobytes() == out['info'].tobytes(), 'info differs: step %d replica %d' % (i, r)
            if check_trace and cfg.n_embb:
                a, b = tr[r], out['trace']
                for f in ('serial', 'type', 'e_snr', 'prbs', 'bits'):
                    assert (a[f] == b[f]).all(), '%s differs: step %d replica %d' % (f, i, r)
                for f in ('queue', 'th', 'p'):
                    assert a[f].tobytes() == b[f].tobytes(), '%s bits differ: step %d replica %d' % (f, i, r)
    c = env.counters()
    if sample is None:
        tot = np.sum([o.counters() for o in oracles], axis=0)
        assert c[0] == tot[0] and c[2] == tot[2] and c[3] == tot[3], (c, tot)
    env.close()


def test_scenario0_small_trace(golden_dir):
    _compare(0, n_envs=24, steps=8, fading=_small_fading(golden_dir), churn=False, seed0=1)


@pytest.mark.parametrize('group', [8, 16, 32])
def test_block_round_allocations(golden_dir, group):
    """Wide contested slices (agent-like allocations, high-churn traffic): the PF allocation of the BLOCK instances
    (block rounds: every contender steps ahead, the largest B-th key is the target, everybody takes the pairs above
    it) against the reference loop of the oracle -- per slot and per UE (RBs, bits, queue, throughput average,
    reception probability) on every replica, then the step outputs of the production (non-tracing) BLOCK instance
    and of the plain trip-loop instance."""
    _compare(0, n_envs=40, steps=14, fading=_small_fading(golden_dir), churn=True, seed0=4100 + group, group=group,
             hint=1, actions=_wide_actions)
    for hint in (1, 0):
        _compare(0, n_envs=40, steps=14, fading=_small_fading(golden_dir), churn=True, seed0=4200 + group, group=group,
                 hint=hint, actions=_wide_actions, check_trace=False)


@pytest.mark.parametrize('group', [16, 32])
def test_block_rounds_at_their_threshold(golden_dir, group):
    """Contested slices of 7, 8 and 9 RB pairs (and up to 20) on the BLOCK instances -- below, at and above the width from
    which a slice may take block rounds -- per slot and per UE against the oracle's reference loop, then the production
    (non-tracing) instance's step outputs."""
    _compare(0, n_envs=48, steps=16, fading=_small_fading(golden_dir), churn=True, seed0=4300 + group, group=group,
             hint=1, actions=_mid_actions)
    _compare(0, n_envs=48, steps=16, fading=_small_fading(golden_dir), churn=True, seed0=4400 + group, group=group,
             hint=1, actions=_mid_actions, check_trace=False)


@pytest.mark.parametrize('group', [8, 16, 32])
def test_scenario0_churn(golden_dir, group):
    """arrivals, admission control, departures, compaction, VBR bursts all fire within a few steps;
    with 8 or 16 lanes per task some slices outgrow the fast instance and are replayed by the 32-lane
    one -- results must not depend on the group size"""
    _compare(0, n_envs=40, steps=30, fading=_small_fading(golden_dir), churn=True, seed0=1000, group=group)


@pytest.mark.parametrize('hint', [0, 1])
@pytest.mark.parametrize('scenario,n_envs,steps', [(0, 160, 30), (2, 96, 20), (3, 96, 20)])
def test_production_instance_soak(golden_dir, scenario, n_envs, steps, hint):
    """The instances that serve step() in production (16 lanes per task, no allocation trace, 5 waves per SIMD --
    the only ones that spill registers; hint 0 = the plain one of the headline batch, hint 1 = the one whose
    heaviest waves schedule one RB pair per trip) against the oracle on every replica: observations, rewards,
    labels, violations and the ten info sums bit for bit, through arrivals, departures, bursts, empty slices."""
    _compare(scenario, n_envs=n_envs, steps=steps, fading=_small_fading(golden_dir), churn=True, seed0=500,
             check_trace=False, group=16, hint=hint)


def test_exact_divide_fallback(golden_dir, monkeypatch):
    """rs_create replaces (pf_b * bits) / slot_length by a reciprocal + two fmas after checking every reachable
    `bits`; RANSLICE_EXACT_DIV forces the IEEE divide instead.  Both must match the oracle."""
    monkeypatch.setenv('RANSLICE_DEV_BUILD', '1')   # knobs are read by the test build only (ranslice._lib)
    monkeypatch.setenv('RANSLICE_EXACT_DIV', '1')
    _compare(0, n_envs=48, steps=12, fading=_small_fading(golden_dir), churn=True, seed0=77, check_trace=False)


def test_crowded_slices_replay(golden_dir):
    """heavy arrival rate: most slices hold more than 8 UEs, so the replay path carries the batch"""
    from ranslice.vec_env import VecRanSlice
    fading = _small_fading(golden_dir)

    def cfgf(n):
        c = _churn(make_config(0, n_envs=n))
        c.cbr_lambda, c.vbr_lambda, c.cbr_t_mean, c.vbr_t_mean = 8.0, 18.0, 0.8, 0.8
        return c
    envs = {}
    for g in (8, 32):
        e = VecRanSlice(n_envs=12, cfg=cfgf(12), fading=fading, seed=3)
        e.set_group_size(g)
        e.reset()
        envs[g] = e
    o = po.OracleEnv(cfgf(1), fading)
    o.set_seed(replica_seed(3, 5))
    o.reset()
    rng = np.random.default_rng(1)
    peak = 0
    for i in range(20):
        acts = _actions(rng, 12, 5, 200, i)
        a = envs[8].step(acts)
        b = envs[32].step(acts)
        assert a[0].tobytes() == b[0].tobytes() and (a[1] == b[1]).all()
        assert envs[8].l1_info().tobytes() == envs[32].l1_info().tobytes()
        out = o.step(acts[5])
        assert a[0][5].tobytes() == out['obs'].tobytes()
        peak = max(peak, envs[8].counters()[3])
    assert envs[8].counters() == envs[32].counters()
    assert envs[8].counters()[3] / (20 * 50 * 12 * 5) > 8.0, 'test should crowd the slices beyond 8 UEs'
    for e in envs.values():
        e.close()


@pytest.mark.parametrize('scenario', [1, 2, 3])
def test_mixed_scenarios(golden_dir, scenario):
    _compare(scenario, n_envs=16, steps=25, fading=_small_fading(golden_dir), churn=True, seed0=5)


def test_ragged_batch(golden_dir):
    """n_envs not a multiple of the 8 tasks per block, single replica"""
    _compare(0, n_envs=1, steps=4, fading=_small_fading(golden_dir), churn=True, seed0=77)
    _compare(1, n_envs=3, steps=4, fading=_small_fading(golden_dir), churn=True, seed0=78)


def test_full_size_sampled():
    """BASELINE config 2 shape: 4096 replicas, 10,000-sample traces; oracle follows a sample."""
    fading = [synth_fading(t, 10000) for t in range(3)]
    sample = sorted(set([0, 1, 7, 8, 63, 64, 1000, 2047, 2048, 4095] + list(range(3, 4096, 32))))   # 138 replicas, in process;
    # EVERY replica of this shape is followed by tests/test_gpu_fullsize.py::test_every_replica_at_baseline_size_vs_oracle
    _compare(0, n_envs=4096, steps=12, fading=fading, churn=False, seed0=0, check_trace=False, sample=sample)


@pytest.mark.parametrize('scenario', [0, 2])
def test_full_size_invariants(scenario):
    """size-independent properties on every one of 4096 replicas at the bench's table size: the reward formula
    (ran_slice.py:45-52), labels in {-1, +1} and consistent with the violation counts (slice_l1.py:160-171), the
    observation being the float32 of info / normalisation constant (slice_ran.py:321-325), and a second handle with a different batch split giving the same
    replicas the same trajectories."""
    from ranslice.vec_env import VecRanSlice
    fading = [synth_fading(t, 10000) for t in range(3)]
    N = 4096
    cfg = make_config(scenario, n_envs=N)
    env = VecRanSlice(n_envs=N, cfg=cfg, fading=fading, seed=123)
    half = VecRanSlice(n_envs=N // 8, cfg=make_config(scenario, n_envs=N // 8), fading=fading)
    env.reset()
    half.reset(seeds=replica_seeds(123, 1024, N // 8))   # the same global replicas 1024.. in a smaller batch
    rng = np.random.default_rng(8)
    S = cfg.n_embb + cfg.n_mmtc
    for i in range(30):
        acts = _actions(rng, N, S, cfg.n_prbs, i)
        obs, rew, done, info = env.step(acts)
        o2, r2, _, i2 = half.step(acts[1024:1024 + N // 8])
        viol, lab = info['violations'], info['SLA_labels']
        tv = viol.sum(axis=1)
        expect = np.where(tv > 0, -cfg.penalty * tv, np.maximum(0, cfg.n_prbs - acts.sum(axis=1)))
        assert (rew == expect).all()
        assert np.isin(lab, (-1, 1)).all() and ((lab == 1) == (viol == 0)).all()
        assert not done.any()
        l1 = env.l1_info()
        # eMBB: obs[k] = f32(info[k] / norm[k]) for the ten accumulators of every slice
        for s in range(cfg.n_embb):
            want = (l1[:, s, :] / np.asarray(list(cfg.norm_embb))).astype(np.float32)
            assert obs[:, s * 10:(s + 1) * 10].tobytes() == want.tobytes()
        # same replicas (global ids 1024..1535) in a smaller batch: identical trajectories
        assert o2.tobytes() == obs[1024:1024 + N // 8].tobytes()
        assert (r2 == rew[1024:1024 + N // 8]).all()


def test_determinism_and_independence(golden_dir):
    """same seeds -> identical results; a replica's trajectory does not depend on its batch"""
    from ranslice.vec_env import VecRanSlice
    fading = _small_fading(golden_dir)
    rng = np.random.default_rng(3)
    acts = [_actions(rng, 64, 5, 200, i) for i in range(6)]

    def run(n_envs, seeds, rows):
        env = VecRanSlice(n_envs=n_envs, cfg=_churn(make_config(0, n_envs=n_envs)), fading=fading)
        env.reset(seeds=seeds)
        outs = []
        for a in acts:
            o, r, _, inf = env.step(a[rows])
            outs.append((o.copy(), r.copy(), inf['violations'].copy()))
        env.close()
        return outs
    seeds = np.arange(64, dtype=np.uint64) + 10
    a = run(64, seeds, np.arange(64))
    b = run(64, seeds, np.arange(64))
    rows = np.array([5, 17, 63])
    c = run(3, seeds[rows], rows)
    for (o1, r1, v1), (o2, r2, v2), (o3, r3, v3) in zip(a, b, c):
        assert o1.tobytes() == o2.tobytes() and (r1 == r2).all() and (v1 == v2).all()
        assert o1[rows].tobytes() == o3.tobytes() and (r1[rows] == r3).all() and (v1[rows] == v3).all()


def test_action_validation(golden_dir):
    from ranslice import _lib
    from ranslice.vec_env import VecRanSlice
    env = VecRanSlice(n_envs=2, scenario=0, fading=_small_fading(golden_dir))
    env.reset()
    bad = np.full((2, 5), 50, dtype=np.int32)  # sums to 250 > 200 (reference Q9)
    with pytest.raises(_lib.RanSliceError):
        env.step(bad)
    neg = np.zeros((2, 5), dtype=np.int32)
    neg[0, 0] = -1
    with pytest.raises(_lib.RanSliceError):
        env.step(neg)
    env.close()


@pytest.mark.parametrize('gran,window', [(1, 50), (3, 50), (4, 7), (2, 3)])
def test_pf_granularity_and_window_variants(golden_dir, gran, window):
    """The PF allocation's shortcuts (closed forms, the run test without the divide, block rounds with their "full
    pairs" bookkeeping and the verified reciprocal form of the throughput update) do not lean on the reference's
    defaults: RB granularities 1, 3 and 4 (odd slices leave a short last "pair"), averaging windows 50, 7 and 3, narrow
    and wide contested slices -- per-slot allocations of the tracing BLOCK instance and step outputs of the plain one
    against the oracle's reference loop."""
    def tweak(cfg):
        cfg.pf_granularity, cfg.pf_window = gran, window
    for acts in (None, _wide_actions):
        _compare(0, n_envs=32, steps=10, fading=_small_fading(golden_dir), churn=True, seed0=6000 + 10 * gran + window,
                 group=16, hint=1, actions=acts, tweak=tweak)
        _compare(0, n_envs=32, steps=10, fading=_small_fading(golden_dir), churn=True, seed0=6100 + 10 * gran + window,
                 group=16, hint=0, actions=acts, tweak=tweak, check_trace=False)

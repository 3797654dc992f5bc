"""Edge cases of the HIP path against the oracle: empty / starved / saturated slices, extreme
allocations, capacity errors (the reference's behaviour for each is noted)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from ranslice import _lib
from ranslice.config import make_config
from ranslice.sharding import replica_seed, replica_seeds  # noqa: F401

pytestmark = pytest.mark.gpu


def _fading(golden_dir):
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    return [g['t0'], g['t1'], g['t2']]


def _churn(cfg):
    cfg.cbr_lambda, cfg.cbr_t_mean = 2.0 / 1.2, 0.6
    cfg.vbr_lambda, cfg.vbr_t_mean = 5.0 / 1.2, 0.6
    cfg.vbr_b_size, cfg.vbr_b_rate = 40, 12
    return cfg


def _run(cfg_fn, n_envs, action_rows, fading, seed=11, group=None):
    from ranslice.vec_env import VecRanSlice
    env = VecRanSlice(n_envs=n_envs, cfg=cfg_fn(n_envs), fading=fading, seed=seed)
    if group:
        env.set_group_size(group)
    env.reset()
    ors = []
    for r in range(n_envs):
        o = po.OracleEnv(cfg_fn(1), fading)
        o.set_seed(replica_seed(seed, r))
        o.reset()
        ors.append(o)
    for i, acts in enumerate(action_rows):
        acts = np.ascontiguousarray(np.broadcast_to(acts, (n_envs, len(acts))), dtype=np.int32)
        obs, rew, _, info = env.step(acts)
        l1 = env.l1_info()
        for r, o in enumerate(ors):
            out = o.step(acts[r])
            assert obs[r].tobytes() == out['obs'].tobytes(), (i, r)
            assert rew[r] == out['reward'] and (info['violations'][r] == out['violations']).all()
            assert l1[r].tobytes() == out['info'].tobytes(), (i, r)
    env.close()


def test_starvation_then_recovery(golden_dir):
    """all slices get 0 PRBs for a while (reference Q2/Q3: stale e_snr/bits, walker frozen), then everything"""
    rows = [[0, 0, 0, 0, 0]] * 6 + [[40, 40, 40, 40, 40]] * 3 + [[200, 0, 0, 0, 0], [0, 0, 0, 0, 200], [1, 1, 1, 1, 1]] + \
           [[0, 0, 0, 0, 0]] * 2 + [[13, 7, 1, 0, 179]] * 3
    _run(lambda n: _churn(make_config(0, n_envs=n)), 6, rows, _fading(golden_dir))


@pytest.mark.parametrize('group', [8, 16, 32])
def test_one_slice_takes_the_whole_carrier(golden_dir, group):
    """a 200-PRB slice: numpy's pairwise split above 128 elements, the 8-lane instance's LDS slice (128 RBs)
    is too small -> replay"""
    rows = [[200, 0, 0, 0, 0], [0, 199, 1, 0, 0], [129, 0, 71, 0, 0], [128, 72, 0, 0, 0]] * 3
    _run(lambda n: _churn(make_config(0, n_envs=n)), 5, rows, _fading(golden_dir), group=group)


def test_mmtc_only_and_max_prbs(golden_dir):
    """no eMBB slice at all (no fading needed), and a 256-PRB carrier (the build's maximum)"""
    rows = [[3, 5], [0, 9], [100, 100], [1, 0]] * 5
    _run(lambda n: make_config(None, n_envs=n, n_prbs=256, n_embb=0, n_mmtc=2), 4, rows, None)
    # eMBB carriers wider than 2x the trace's 100 rows cannot be built (the reference's row wrap,
    # channel_models.py:144-148, only extends to 200): 256 PRBs need traces with >= 128 rows
    wide = [np.vstack([t, t[:28]]) for t in _fading(golden_dir)]
    rows = [[256, 0, 0], [100, 100, 56], [3, 250, 3]] * 3
    _run(lambda n: _churn(make_config(None, n_envs=n, n_prbs=256, n_embb=2, n_mmtc=1)), 3, rows, wide)


def test_mmtc_backlog_saturation(golden_dir):
    """mMTC slices starved for a long time: the FIFO grows, delays rise, SLA is violated; capacity 1024 is
    eventually exceeded and reported (the reference's numpy arrays would simply keep growing)"""
    from ranslice.vec_env import VecRanSlice
    fading = _fading(golden_dir)
    cfgf = lambda n: make_config(2, n_envs=n)
    rows = [[20, 0, 0, 1, 0]] * 60
    _run(cfgf, 2, rows, fading)
    env = VecRanSlice(n_envs=2, cfg=cfgf(2), fading=fading, seed=1)
    env.reset()
    acts = np.array([[20, 0, 0, 0, 0]] * 2, dtype=np.int32)
    with pytest.raises(_lib.RanSliceError) as ei:
        for _ in range(400):
            env.step(acts)
    assert ei.value.code == _lib.RS_EOVERFLOW
    env.close()


def test_ue_capacity_overflow_is_reported(golden_dir):
    """arrival rate far beyond the 32-UE capacity -> RS_EOVERFLOW (never a silent drop)"""
    from ranslice.vec_env import VecRanSlice
    cfg = make_config(0, n_envs=4)
    cfg.cbr_lambda, cfg.vbr_lambda, cfg.cbr_t_mean, cfg.vbr_t_mean = 10.0, 200.0, 5.0, 5.0
    env = VecRanSlice(n_envs=4, cfg=cfg, fading=_fading(golden_dir))
    env.reset()
    with pytest.raises(_lib.RanSliceError) as ei:
        for _ in range(40):
            env.step(np.full((4, 5), 30, dtype=np.int32))
    assert ei.value.code == _lib.RS_EOVERFLOW
    env.close()


def test_kbrl_full_dictionary_saturates_without_error():
    """a full dictionary projects further samples instead of growing (build-defined; the reference's dictionary is
    unbounded): learning goes on, nothing raises, and the size report shows the saturation"""
    from ranslice.kbrl_dev import VecKBRL
    ag = VecKBRL(1, [3], 100, capacity=4)
    ag.reset([[5]], [[2]])
    rng = np.random.default_rng(0)
    for i in range(200):
        x = rng.random(4) * 8.0  # far-apart points: every mistake would grow the dictionary
        y = 1 if i % 2 else -1
        ag.predict(0, 0, x)
        ag.update(0, 0, x, y)
    assert ag.dictionary_sizes().tolist() == [[4]]
    ag.synchronize()
    ag.close()


def test_call_order_errors(golden_dir):
    from ranslice.vec_env import VecRanSlice
    import ctypes as C
    L = _lib.load()
    cfg = make_config(0, n_envs=1)
    h = C.c_void_p()
    assert L.rs_create(C.byref(cfg), 0, C.byref(h)) == 0
    seeds = np.zeros(1, dtype=np.uint64)
    assert L.rs_reset(h, seeds.ctypes.data_as(C.POINTER(C.c_uint64)), None) == _lib.RS_ESTATE  # fading not loaded
    assert b'fading' in L.rs_last_error(h)
    L.rs_destroy(h)
    env = VecRanSlice(n_envs=1, cfg=make_config(0, n_envs=1), fading=_fading(golden_dir))
    with pytest.raises(_lib.RanSliceError) as ei:
        env.step(np.zeros((1, 5), dtype=np.int32))  # step before reset
    assert ei.value.code == _lib.RS_ESTATE
    bad = make_config(0, n_envs=1)
    bad.n_prbs = 300
    h2 = C.c_void_p()
    assert L.rs_create(C.byref(bad), 0, C.byref(h2)) == _lib.RS_EINVAL
    L.rs_destroy(h2)
    assert L.rs_create(C.byref(cfg), 99, C.byref(h2)) == _lib.RS_EHIP  # no such device
    L.rs_destroy(h2)
    env.close()


@pytest.mark.parametrize('scenario', [0, 2])
def test_graph_captured_loop_equals_step_by_step(golden_dir, scenario):
    """rs_run_random: the hipGraph replay (device-side slot clock and script index), the plain one-call loop and
    the call-per-step loop leave identical observations, labels, rewards and info behind; odd step counts,
    re-entry at the other counter parity and a reset in between included."""
    from ranslice.vec_env import VecRanSlice
    fading = _fading(golden_dir)
    N = 96

    def make():
        e = VecRanSlice(n_envs=N, cfg=_churn(make_config(scenario, n_envs=N)), fading=fading, seed=5)
        e.reset()
        return e

    ref, loop, graph = make(), make(), make()
    plan = [(0, 7), (7, 2), (9, 1), (10, 12)]
    for step0, n in plan:
        for i in range(n):
            ref.random_actions(77, step0 + i)
            ref.step_resident()
        loop.run_random(77, step0, n, graph=False)
        graph.run_random(77, step0, n, graph=True)
        a, b, c = ref.fetch(), loop.fetch(), graph.fetch()
        for k in ('actions', 'obs', 'reward', 'labels', 'violations'):
            assert np.array_equal(a[k], b[k]), (k, step0)
            assert np.array_equal(a[k], c[k]), (k, step0)
        assert np.array_equal(ref.l1_info(), graph.l1_info())
    # reset keeps the captured graph valid (the clock restarts on the device)
    ref.reset()
    graph.reset()
    for i in range(6):
        ref.random_actions(3, i)
        ref.step_resident()
    graph.run_random(3, 0, 6, graph=True)
    a, c = ref.fetch(), graph.fetch()
    for k in ('actions', 'obs', 'reward', 'labels', 'violations'):
        assert np.array_equal(a[k], c[k]), k


@pytest.mark.parametrize('group', [16, 32])
def test_schedule_hint_does_not_change_results(golden_dir, group):
    """agent-like allocations (most of the carrier on one or two slices, so that spans wider than the per-wave LDS
    buffer and long contested PF allocations occur): the plain instance (trip loop), the BLOCK instance (block rounds
    on wide slices) and the automatic choice (rs_step looks at the allocations it is handed) leave identical
    observations, rewards, labels and info sums behind"""
    from ranslice.vec_env import VecRanSlice
    fading = _fading(golden_dir)
    N = 640

    def make(hint):
        e = VecRanSlice(n_envs=N, cfg=_churn(make_config(0, n_envs=N)), fading=fading, seed=21)
        e.set_group_size(group)
        e.set_schedule_hint(hint)
        e.reset()
        return e

    envs = [make(0), make(1), make(-1)]
    rng = np.random.default_rng(5)
    for i in range(25):
        a = rng.integers(0, 6, size=(N, 5))
        big = rng.integers(0, 5, size=N)
        a[np.arange(N), big] = rng.integers(115, 171, size=N)  # sum <= 170 + 4 * 5 + 9 < 200
        second = (big + 1 + rng.integers(0, 4, size=N)) % 5
        a[np.arange(N), second] += rng.integers(0, 2, size=N) * rng.integers(0, 10, size=N)
        a = a.astype(np.int32)
        outs = [e.step(a) for e in envs]
        infos = [e.l1_info() for e in envs]
        for o, inf in zip(outs[1:], infos[1:]):
            assert o[0].tobytes() == outs[0][0].tobytes(), 'obs differ at step %d' % i
            assert (o[1] == outs[0][1]).all()
            assert (o[3]['SLA_labels'] == outs[0][3]['SLA_labels']).all()
            assert (o[3]['violations'] == outs[0][3]['violations']).all()
            assert inf.tobytes() == infos[0].tobytes()


def _mux_compare(golden_dir, scenario, n_envs, steps, churn, seed0, trace, **shape):
    from oracle import pyoracle as po
    from ranslice.vec_env import VecRanSlice
    fading = _fading(golden_dir)

    def cfgf(n):
        c = make_config(scenario, n_envs=n, L1_level=False, **shape)
        return _churn(c) if churn else c
    env = VecRanSlice(n_envs=n_envs, cfg=cfgf(n_envs), fading=fading, seed=seed0)
    n_l1 = (env.cfg.n_embb > 0) + (env.cfg.n_mmtc > 0)
    assert env.multiplexed and env.n_slices == n_l1
    if trace:
        env.set_alloc_trace(True)
    env.reset()
    oracles = []
    for r in range(n_envs):
        o = po.OracleEnv(cfgf(1), fading)
        o.set_seed(replica_seed(seed0, r))
        o.reset()
        oracles.append(o)
    rng = np.random.default_rng(17 + (scenario or 0))
    n_prbs = env.n_prbs
    for i in range(steps):
        if i % 4 == 0:
            acts = np.stack([rng.multinomial(n_prbs, [1.0 / n_l1] * n_l1) for _ in range(n_envs)])
        elif i % 4 == 2:
            acts = rng.integers(0, 6, size=(n_envs, n_l1))
        else:
            acts = np.stack([rng.multinomial(n_prbs, [1.0 / (n_l1 + 1)] * (n_l1 + 1))[:n_l1] for _ in range(n_envs)])
        acts = acts.astype(np.int32)
        obs, rew, _, info = env.step(acts)
        l1 = env.l1_info()
        tr = env.alloc_trace() if trace and env.cfg.n_embb else None
        for r, o in enumerate(oracles):
            out = o.step(acts[r], trace=tr is not None)
            assert obs[r].tobytes() == out['obs'].tobytes(), ('obs', i, r)
            assert rew[r] == out['reward']
            assert (info['SLA_labels'][r] == out['labels']).all() and (info['violations'][r] == out['violations']).all()
            assert l1[r].tobytes() == out['info'].tobytes(), ('info', i, r)
            if tr is not None:
                a, b = tr[r], out['trace'][0]
                for f in ('serial', 'type', 'e_snr', 'prbs', 'bits'):
                    assert (a[f] == b[f]).all(), (f, i, r)
                for f in ('queue', 'th', 'p'):
                    assert a[f].tobytes() == b[f].tobytes(), (f, i, r)
    c = env.counters()
    tot = np.sum([o.counters() for o in oracles], axis=0)
    assert c[0] == tot[0] and c[2] == tot[2] and c[3] == tot[3], (c, tot)
    env.close()


def test_task_order_and_priority_do_not_change_results(golden_dir, monkeypatch):
    """The cost-ranked task order (rs_order.hip), the share of heavy-led waves and -- implicitly, it is timing-driven --
    the dynamic issue priority decide which lanes simulate which task and who issues first, never what is computed:
    2048 replicas (10240 tasks: the 16-lane instance with the order kernels active) stepped with task-index order, sorted
    order, half and full pairing leave identical observations, rewards, labels, violations and info sums."""
    import hashlib
    from ranslice.vec_env import VecRanSlice
    fading = _fading(golden_dir)
    N = 2048
    digests = []
    monkeypatch.setenv('RANSLICE_DEV_BUILD', '1')   # knobs are read by the test build only (ranslice._lib)
    for order, pair in (('0', None), ('3', None), ('6', '128'), ('6', '256'), ('6', '0')):
        monkeypatch.setenv('RANSLICE_ORDER', order)
        if pair is None:
            monkeypatch.delenv('RANSLICE_PAIR', raising=False)
        else:
            monkeypatch.setenv('RANSLICE_PAIR', pair)
        env = VecRanSlice(n_envs=N, cfg=_churn(make_config(0, n_envs=N)), fading=fading, seed=77)
        env.reset()
        h = hashlib.sha256()
        for i in range(30):
            env.random_actions(2024, i)
            env.step_resident()
            f = env.fetch()
            for key in ('obs', 'reward', 'labels', 'violations'):
                h.update(f[key].tobytes())
            h.update(env.l1_info().tobytes())
        digests.append(h.hexdigest())
        env.close()
    assert len(set(digests)) == 1, digests


@pytest.mark.parametrize('scenario', [0, 1, 2])
def test_multiplexed_l1_matches_oracle(golden_dir, scenario):
    """create_env(..., L1_level=False) (scenario_creator.py:168-177; the oracle's multiplexed mode is pinned to the
    reference by fixtures G13): one UE list / PF scheduler for all eMBB RAN slices, one FIFO for all mMTC ones.
    Bit-exact: observations, rewards, labels / violations per L1 slice, info rows per RAN slice, and every UE of the
    shared list in every slot (RAN slice, e_snr, RBs, bits, queue, throughput, reception probability)."""
    _mux_compare(golden_dir, scenario, n_envs=12, steps=14, churn=True, seed0=640, trace=True)
    _mux_compare(golden_dir, scenario, n_envs=20, steps=10, churn=False, seed0=7, trace=False)


def test_multiplexed_drop_in_env(golden_dir):
    """the drop-in surface: create_env(rng, n, L1_level=False) -> gym env with one action entry per L1 slice and the
    reference's info layout (l1_info[l1][ran index])"""
    import scenario_creator as sc
    sc.set_fading(_fading(golden_dir))
    env = sc.create_env(np.random.default_rng(3), 1, L1_level=False)
    assert env.n_slices == 2 and env.n_variables == 3 * 10 + 2 * 3
    st = env.reset()
    assert st.shape == (36,)
    st, r, done, info = env.step(np.array([100, 20]))
    assert len(info['l1_info']) == 2 and sorted(info['l1_info'][0]) == [0, 1, 2] and sorted(info['l1_info'][1]) == [0, 1]
    assert set(info['l1_info'][0][0]) == set(sc.state_variables_embb) and 'delay' in info['l1_info'][1][1]
    assert info['SLA_labels'].shape == (2,) and isinstance(r, float)
    sc.set_fading(None)


def test_multiplexed_l1_many_mmtc_slices(golden_dir):
    """L1_level=False with six and eight mMTC RAN slices behind the one FIFO: mtc_mux_step_kernel then needs 72 / 96 KB of
    dynamic LDS (FIFO + device tables of every slice), above the 64 KB a launch gets by default -- rs_create asks for it
    (ADVICE r2).  Bit-exact against the oracle; a mode without any eMBB RAN slice is refused at create."""
    from ranslice import _lib
    from ranslice.vec_env import VecRanSlice
    for n_mmtc in (6, 8):
        _mux_compare(golden_dir, None, n_envs=6, steps=8, churn=False, seed0=90 + n_mmtc, trace=False, n_prbs=100, n_embb=1,
                     n_mmtc=n_mmtc)
    with pytest.raises(_lib.RanSliceError) as e:
        VecRanSlice(n_envs=2, cfg=make_config(None, n_envs=2, L1_level=False, n_prbs=100, n_embb=0, n_mmtc=2),
                    fading=_fading(golden_dir))
    assert e.value.code == _lib.RS_EINVAL and 'eMBB' in str(e.value)


def _fuzz_case(k):
    """one random configuration: carrier, slice mix, slots per step, propagation, PF parameters, traffic"""
    rng = np.random.default_rng(7000 + k)
    n_embb = int(rng.integers(0, 6))
    n_mmtc = int(rng.integers(0 if n_embb else 1, 4))
    kw = dict(n_prbs=int(rng.integers(4, 201)), n_embb=n_embb, n_mmtc=n_mmtc,   # (a 100-row trace extends to 200 PRBs, channel_models.py:144-148)
              slots_per_step=int(rng.choice([1, 3, 12, 13, 25, 50, 63])),
              propagation_type=str(rng.choice(['macro_cell_urban_2GHz', 'macro_cell_urban_900MHz', 'macro_cell_rural'])),
              penalty=int(rng.choice([1, 100, 1000])))
    gran, window = int(rng.choice([1, 2, 2, 3, 4])), int(rng.choice([5, 50, 200]))
    churn = bool(rng.integers(0, 2))
    n_envs = int(rng.choice([1, 2, 5, 9, 17, 33]))
    return kw, gran, window, churn, n_envs


@pytest.mark.parametrize('k', range(16))
def test_random_configurations_vs_oracle(golden_dir, k):
    """Configurations no scenario of the reference uses -- carriers of 4 to 200 PRBs, 0-5 eMBB and 0-3 mMTC slices, 1 to 63 slots per
    step, every propagation model, PF granularity 1-4 and window 5-200, quiet and high-churn traffic, ragged batch sizes -- through
    the production instance against the oracle: observations, rewards, labels, violations, info sums and (eMBB, small batches)
    every UE's allocation in every slot, bit for bit."""
    from ranslice.vec_env import VecRanSlice
    from test_gpu_parity import _actions, _churn, _small_fading
    kw, gran, window, churn, n = _fuzz_case(k)
    fading = _small_fading(golden_dir)

    def cfg_for(n_envs):
        c = make_config(None, n_envs=n_envs, **kw)
        c.pf_granularity, c.pf_window = gran, window
        return _churn(c) if churn else c
    cfg = cfg_for(n)
    env = VecRanSlice(n_envs=n, cfg=cfg, fading=fading, seed=4000 + k)
    trace = cfg.n_embb > 0 and n <= 9
    if trace:
        env.set_alloc_trace(True)
    env.reset()
    oracles = []
    for r in range(n):
        o = po.OracleEnv(cfg_for(1), fading)
        o.set_seed(replica_seed(4000 + k, r))
        o.reset()
        oracles.append(o)
    rng = np.random.default_rng(100 + k)
    S = cfg.n_embb + cfg.n_mmtc
    for i in range(10):
        acts = _actions(rng, n, S, cfg.n_prbs, i)
        over = acts.sum(axis=1) > cfg.n_prbs          # (the small-integer script on a carrier narrower than its slices)
        acts[over] = np.minimum(acts[over], cfg.n_prbs // S)
        obs, rew, done, info = env.step(acts)
        l1 = env.l1_info()
        tr = env.alloc_trace() if trace else None
        for r, o in enumerate(oracles):
            out = o.step(acts[r], trace=trace)
            assert obs[r].tobytes() == out['obs'].tobytes(), (kw, 'obs', i, r)
            assert rew[r] == out['reward'] and (info['SLA_labels'][r] == out['labels']).all(), (kw, 'reward/labels', i, r)
            assert (info['violations'][r] == out['violations']).all() and l1[r].tobytes() == out['info'].tobytes(), (kw, 'info', i, r)
            if trace:
                a, b = tr[r], out['trace']
                for f in ('serial', 'type', 'e_snr', 'prbs', 'bits'):
                    assert (a[f] == b[f]).all(), (kw, f, i, r)
                for f in ('queue', 'th', 'p'):
                    assert a[f].tobytes() == b[f].tobytes(), (kw, f, i, r)
    env.close()

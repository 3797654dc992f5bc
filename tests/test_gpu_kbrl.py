"""GPU parity for hot path B: the HIP KBRL agent (kb_* C ABI) against the CPU oracle and the
reference-recorded golden sequences.  Kernel values carry a stated tolerance (the GPU sums k.coeff
tile-wise on MFMA/VALU, numpy through BLAS, the oracle sequentially): |df| <= 1e-9 (1 + sum|k c|),
delta/coeff/Kinv 1e-8 relative; every decision (sign, update branch, action, hit, margin, security
factor, dictionary size) must agree exactly.
"""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from ranslice.config import make_config
from ranslice.sharding import replica_seed, replica_seeds  # noqa: F401

pytestmark = pytest.mark.gpu
# oracle workers are SPAWNED: forking a process whose HIP runtime is already initialised is not safe
import multiprocessing as _mp  # noqa: E402
_SPAWN = _mp.get_context('spawn')
TOL = 1e-9


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


@pytest.mark.parametrize('tag,d', [('d11', 11), ('d4', 4)])
def test_projectron_teacher_forced(golden_dir, tag, d):
    """Projectron.predict/update through kb_predict/kb_update vs the reference's recorded sequence: ALL samples
    (1,500 / 1,200), then the final dictionary: landmarks exact, coeff and Kinv within 1e-8 relative"""
    from ranslice.kbrl_dev import VecKBRL
    g = _load(golden_dir, 'g9_projectron')
    ag = VecKBRL(1, [d - 1], 200, capacity=1024)
    ag.reset([[10]], [[3]])
    xs, ys = g[tag + '_x'], g[tag + '_y']
    n = len(xs)
    for i in range(n):
        yp, f = ag.predict(0, 0, xs[i])
        fr = g[tag + '_f'][i]
        assert f == pytest.approx(fr, rel=1e-8, abs=TOL), i
        if abs(fr) > TOL:
            assert yp == g[tag + '_ypred'][i], i
        br, dl = ag.update(0, 0, xs[i], int(ys[i]))
        assert br == g[tag + '_branch'][i], i
        if br:
            assert dl == pytest.approx(g[tag + '_delta'][i], rel=1e-7, abs=1e-9)
        assert ag.learner(0, 0)['m'] == g[tag + '_m'][i]
    L = ag.learner(0, 0, with_kinv=True)
    m = L['m']
    assert m == len(np.atleast_2d(g[tag + '_landmarks']))
    np.testing.assert_array_equal(L['landmarks'], np.atleast_2d(g[tag + '_landmarks']))
    np.testing.assert_allclose(L['coeff'], g[tag + '_coeff'], rtol=1e-8, atol=1e-9)
    kinv = g[tag + '_kinv']
    np.testing.assert_allclose(L['kinv'], kinv, rtol=1e-8, atol=1e-8 * np.abs(kinv).max())
    ag.close()


def _dims(scenario):
    cfg = make_config(scenario)
    return [10] * cfg.n_embb + [3] * cfg.n_mmtc, cfg.n_prbs


@pytest.mark.parametrize('scenario', [0, 2])
def test_kbrl_control_teacher_forced_golden(golden_dir, scenario):
    """KBRL_Control.update_control/select_action on the reference's recorded (state, action, labels)"""
    from ranslice.kbrl_dev import VecKBRL
    g = _load(golden_dir, 'g10_kbrl_s%d' % scenario)
    dims, n_prbs = _dims(scenario)
    ag = VecKBRL(1, dims, n_prbs, accuracy_range=tuple(g['a_range']))
    ag.reset(g['init_action'][None], g['init_sec'][None])
    steps = len(g['state'])
    acc_i = 0
    for i in range(steps):
        hits = ag.update_control(g['state'][i][None], g['action_in'][i][None], g['labels'][i][None])
        assert (hits[0] == g['hits'][i]).all(), i
        nxt = g['state'][i + 1] if i + 1 < steps else g['final_state']
        act, adj = ag.select_action(nxt[None])
        assert (act[0] == g['action_out'][i]).all(), i
        assert adj[0] == g['adjusted'][i]
        c = ag.control()
        assert (c['margins'][0] == g['margins'][i]).all()
        assert (c['security_factors'][0] == g['security'][i]).all()
        if i % 10 == 9 or i == steps - 1:
            np.testing.assert_allclose(c['accuracies'][0], g['acc'][acc_i], rtol=1e-12, atol=0)
            acc_i += 1
    for s in range(len(dims)):
        L = ag.learner(0, s)
        np.testing.assert_array_equal(L['landmarks'], np.atleast_2d(g['landmarks%d' % s]))
        np.testing.assert_allclose(L['coeff'], g['coeff%d' % s], rtol=1e-7, atol=1e-9)
    ag.close()


def test_batched_agents_vs_oracle(golden_dir):
    """16 agents, closed loop with the HIP simulator, against oracle env + oracle agent per replica
    on the same Philox streams: actions, hits, margins, security factors and dictionary sizes must
    agree step for step (scenario_1: eMBB + mMTC learners of different dimension)."""
    from ranslice.kbrl_dev import VecKBRL
    from ranslice.vec_env import VecRanSlice
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    fading = [g['t0'], g['t1'], g['t2']]
    scenario, N, steps = 1, 16, 40
    dims, n_prbs = _dims(scenario)
    rng = np.random.default_rng(4)
    ia = np.stack([np.concatenate([rng.integers(4, 20, 3), rng.integers(2, 10, 2)]) for _ in range(N)]).astype(np.int32)
    sf = np.stack([np.concatenate([rng.integers(2, 8, 3), rng.integers(1, 4, 2)]) for _ in range(N)]).astype(np.int32)
    env = VecRanSlice(n_envs=N, cfg=make_config(scenario, n_envs=N), fading=fading, seed=50)
    ag = VecKBRL(N, dims, n_prbs, capacity=256)
    ag.reset(ia, sf, seeds=np.arange(N, dtype=np.uint64) + 7)
    oe, oa = [], []
    for r in range(N):
        e = po.OracleEnv(make_config(scenario), fading)
        e.set_seed(replica_seed(50, r))
        e.reset()
        a = po.OracleKBRL(dims, n_prbs, ia[r], sf[r], capacity=256)
        a.set_seed(7 + r)
        oe.append(e)
        oa.append(a)
    state = env.reset()
    action = ia.copy()
    ostate = [np.zeros(env.n_variables, dtype=np.float32) for _ in range(N)]
    oaction = [ia[r].copy() for r in range(N)]
    for i in range(steps):
        obs, rew, _, info = env.step(action)
        hits = ag.update_control(state, action, info['SLA_labels'])
        new_action, adj = ag.select_action(obs)
        c = ag.control()
        for r in range(N):
            out = oe[r].step(oaction[r])
            assert obs[r].tobytes() == out['obs'].tobytes(), (i, r)
            oh = oa[r].update_control(ostate[r], oaction[r], out['labels'])
            na, oadj = oa[r].select_action(out['obs'])
            oa[r].adjusted = oadj
            assert (hits[r] == oh).all(), (i, r)
            assert (new_action[r] == na).all() and adj[r] == oadj, (i, r)
            assert (c['margins'][r] == oa[r].margins).all() and (c['security_factors'][r] == oa[r].security_factors).all()
            ostate[r], oaction[r] = out['obs'], na
        state, action = obs, new_action
    for r in (0, N - 1):
        for s in range(len(dims)):
            L = ag.learner(r, s)
            assert L['m'] == oa[r].m(s)
            np.testing.assert_allclose(L['coeff'], oa[r].coeff(s), rtol=1e-7, atol=1e-9)
    st = ag.stats()
    tot = np.sum([a.stats() for a in oa], axis=0)
    assert st[0] == tot[0] and st[1] == tot[1], (st, tot)
    env.close()
    ag.close()


def test_resident_closed_loop_matches_host_loop(golden_dir):
    """kb_step_resident (everything on the device) == the same loop driven through host buffers"""
    from ranslice.kbrl_dev import VecKBRL
    from ranslice.vec_env import VecRanSlice
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    fading = [g['t0'], g['t1'], g['t2']]
    N, steps = 32, 25
    dims, n_prbs = _dims(0)
    ia = np.full((N, 5), 12, dtype=np.int32)
    sf = np.full((N, 5), 4, dtype=np.int32)

    def make():
        env = VecRanSlice(n_envs=N, cfg=make_config(0, n_envs=N), fading=fading, seed=9)
        ag = VecKBRL(N, dims, n_prbs, capacity=128)
        ag.reset(ia, sf)
        return env, ag
    env, ag = make()
    state = env.reset()
    action = ia.copy()
    host_actions = []
    for i in range(steps):
        obs, rew, _, info = env.step(action)
        ag.update_control(state, action, info['SLA_labels'])
        action, adj = ag.select_action(obs)
        state = obs
        host_actions.append(action.copy())
    env.close(); ag.close()
    env, ag = make()
    env.reset()
    first = ia.copy()
    obs, rew, _, info = env.step(first)   # seeds the device action buffer and obs
    # redo from scratch on a fresh pair: resident loop needs the first action on the device
    env.close(); ag.close()
    env, ag = make()
    env.reset()
    import ctypes as C
    a0 = np.ascontiguousarray(first)
    env._check(env.L.rs_step(env.h, a0.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
    for i in range(steps):
        ag.step_resident(env)
        got = env.fetch()['actions']
        assert (got == host_actions[i]).all(), i
        if i + 1 < steps:
            env.step_resident()
    env.close(); ag.close()


def test_drop_in_experiment_plumbing(golden_dir):
    """create_env / create_kbrl_agent / KBRL_Control.run as experiments_kbrl.Evaluator.evaluate uses
    them (BASELINE config 1 plumbing): result dict schema identical to the reference's (G11)."""
    import scenario_creator as sc
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    sc.set_fading([g['t0'], g['t1'], g['t2']])
    ref = _load(golden_dir, 'g11_results_schema')
    rng = np.random.default_rng(0)
    env = sc.create_env(rng, 0)
    assert env.n_prbs == 200 and env.n_slices == 5 and env.n_variables == 50
    agent = sc.create_kbrl_agent(rng, 0, accuracy_range=[0.97, 0.99])
    res = agent.run(env, 30)
    assert sorted(res) == sorted(k[4:] for k in ref.files)
    for k, v in res.items():
        assert v.dtype == ref['key_' + k].dtype and v.ndim == ref['key_' + k].ndim
        assert v.shape[-1] == 30
    assert agent.learners[0].algorithm.get_set_size() >= 1
    st, r, done, info = env.step(np.array([10, 10, 10, 10, 10], dtype=np.int16))
    assert st.dtype == np.float32 and isinstance(r, float) and done is False
    assert set(info) == {'l1_info', 'SLA_labels', 'violations', 'n_prbs', 'total_violations'}
    assert set(info['l1_info'][0][0]) == set(sc.state_variables_embb)


_ORACLE_FADING = {}


def _oracle_closed_loop(args):
    """oracle env + oracle agent of one replica, closed loop (KBRL_Control.run body, kbrl_control.py:128-141)"""
    scenario, env_seed, ag_seed, ia, sf, steps, cols, capacity = args
    from ranslice.fading import synth_fading
    dims, n_prbs = _dims(scenario)
    if cols not in _ORACLE_FADING:     # (a pool worker follows many replicas: the tables are made once per process)
        _ORACLE_FADING[cols] = [synth_fading(t, cols) for t in range(3)]
    e = po.OracleEnv(make_config(scenario), _ORACLE_FADING[cols])
    e.set_seed(env_seed)
    e.reset()
    a = po.OracleKBRL(dims, n_prbs, ia, sf, capacity=capacity)
    a.set_seed(ag_seed)
    state = np.zeros(e.n_vars, dtype=np.float32)
    action = np.asarray(ia, dtype=np.int32).copy()
    out = []
    for i in range(steps):
        r = e.step(action)
        hits = a.update_control(state, action, r['labels'])
        na, adj = a.select_action(r['obs'])
        a.adjusted = adj
        out.append((action.copy(), r['obs'].copy(), r['labels'].copy(), np.asarray(hits).copy(), na.copy(), int(adj)))
        state, action = r['obs'], na
    return out, [a.m(s) for s in range(len(dims))]


def _full_size_closed_loop(sample, steps, chunksize):
    import ctypes as C
    from concurrent.futures import ProcessPoolExecutor
    from ranslice.fading import synth_fading
    from ranslice.kbrl_dev import VecKBRL
    from ranslice.vec_env import VecRanSlice
    N, cols, cap = 4096, 10000, 256
    scenario = 0
    dims, n_prbs = _dims(scenario)
    rng = np.random.default_rng(11)
    ia = rng.integers(10, 35, size=(N, 5)).astype(np.int32)
    sf = rng.integers(2, 8, size=(N, 5)).astype(np.int32)
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1), mp_context=_SPAWN) as ex:
        fut = ex.map(_oracle_closed_loop, [(scenario, replica_seed(300, r), 7 + r, ia[r], sf[r], steps, cols, cap) for r in sample],
                     chunksize=chunksize)
        env = VecRanSlice(n_envs=N, cfg=make_config(scenario, n_envs=N), fading=[synth_fading(t, cols) for t in range(3)],
                          seed=300)
        ag = VecKBRL(N, dims, n_prbs, capacity=cap)
        ag.reset(ia, sf, seeds=np.arange(N, dtype=np.uint64) + 7)
        env.reset()
        a0 = np.ascontiguousarray(ia)
        env._check(env.L.rs_step(env.h, a0.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
        hip = []
        executed = ia.copy()
        for i in range(steps):
            f = env.fetch()     # results of the step that executed `executed`
            ag.step_resident(env)
            nxt = env.fetch()['actions']
            hip.append((executed[sample].copy(), f['obs'][sample].copy(), f['labels'][sample].copy(), nxt[sample].copy()))
            executed = nxt
            if i + 1 < steps:
                env.step_resident()
        ag.synchronize()
        all_sizes = ag.dictionary_sizes()
        sizes = [[int(all_sizes[r, s]) for s in range(len(dims))] for r in sample]
        env.close()
        ag.close()
        ref = list(fut)
    for k, r in enumerate(sample):
        steps_ref, m_ref = ref[k]
        for i in range(steps):
            act, obs, lab, hits, na, adj = steps_ref[i]
            h = hip[i]
            assert (h[0][k] == act).all(), ('executed action', r, i)
            assert h[1][k].tobytes() == obs.tobytes(), ('obs', r, i)
            assert (h[2][k] == lab).all(), ('labels', r, i)
            assert (h[3][k] == na).all(), ('selected action', r, i, h[3][k], na)
        assert sizes[k] == m_ref, (r, sizes[k], m_ref)


def test_full_size_closed_loop_vs_oracle():
    """BASELINE config 3's loop at its size: 4096 replicas of scenario_0 with one KBRL agent each, closed on the
    device (kb_step_resident: update_control + select_action write the next action into the simulator's buffer),
    10,000-column traces; 272 replicas against oracle env + oracle agent on the same streams: executed
    actions, observations (bits), labels and the selected actions, every step; dictionary sizes at the end."""
    N = 4096
    # 272 replicas (at least 256): the edges of blocks, waves and the batch, and every 16th replica
    sample = sorted(set([0, 1, 2, 3, 15, 16, 63, 64, 100, 255, 256, 777, 1023, 1024, 2000, 2047, 2048, 3000, 3333, 4000, 4093,
                         4094, 4095, 1234] + list(range(5, N, 16))))
    assert len(sample) >= 256
    _full_size_closed_loop(sample, 120, 4)


def test_every_replica_closed_loop_vs_oracle():
    """The agent checked on EVERY replica once (the simulator already is, tests/test_gpu_fullsize.py): all 4096
    replicas of config 3 against oracle env + oracle agent (kbrl_control.py:128-141) for 40 closed-loop steps -- executed and
    selected actions, observations (bits), labels, every step; every dictionary's size at the end."""
    _full_size_closed_loop(list(range(4096)), 40, 32)


def test_kernel_row_and_stale_cache_guard(golden_dir):
    """GaussianKernel.k / predict()[2] (kernel.py:13-28) are served from the row the device learner caches, and
    Projectron.update refuses to use that cache once the dictionary has changed under it (reference Q12: numpy would
    raise on the length mismatch)"""
    import scenario_creator as sc
    from ranslice import _lib
    rng = np.random.default_rng(3)
    agent = sc.create_kbrl_agent(rng, 0, capacity=64)
    alg = agent.learners[0].algorithm
    xs = rng.random((30, 11))
    for i in range(30):
        y = 1 if xs[i].sum() > 5.5 else -1
        alg.predict(xs[i])
        alg.update(xs[i], y)
    x = rng.random(11)
    yp, f, krow = alg.kernel.predict(x)
    L, co = alg.sv.landmarks, alg.sv.coeff
    want = np.exp(-1.0 * ((np.atleast_2d(L) - x) ** 2).sum(axis=1))
    np.testing.assert_allclose(krow, want, rtol=1e-12)
    assert f == pytest.approx(float(want @ co), rel=1e-9, abs=1e-12) and yp == (1 if f > 0 else -1)
    np.testing.assert_allclose(alg.kernel.k(x), want, rtol=1e-12)
    assert alg.kernel.k_eval(x, x) == 1.0
    # update_control grows learner 0's dictionary between a predict and its update: the update must refuse
    alg.predict(x)
    m0 = alg.counter
    state = np.zeros(50, dtype=np.float32)
    state[:10] = 0.9
    for _ in range(3):
        agent.update_control(state, np.full(5, 20, dtype=np.int16), -np.ones(5, dtype=np.int32))
        state[:10] -= 0.2
    assert alg.counter > m0
    with pytest.raises(_lib.RanSliceError) as e:
        alg.update(x, -1 if f > 0 else 1)
    assert e.value.code == _lib.RS_ESTATE


def test_saturated_dictionary_projects(golden_dir):
    """a dictionary at its capacity keeps learning by projection (build-defined; the reference grows without bound):
    no error, sizes stay at the capacity, and the device agent still agrees with the oracle agent of the same
    capacity, decision for decision"""
    from ranslice.kbrl_dev import VecKBRL
    g = _load(golden_dir, 'g10_kbrl_s0')
    dims, n_prbs = _dims(0)
    cap = 6
    ag = VecKBRL(1, dims, n_prbs, accuracy_range=tuple(g['a_range']), capacity=cap)
    ag.reset(g['init_action'][None], g['init_sec'][None])
    oa = po.OracleKBRL(dims, n_prbs, g['init_action'], g['init_sec'], accuracy_range=tuple(g['a_range']), capacity=cap)
    oa.set_seed(0)
    # (a 6-landmark classifier is badly under-fitted: its scores hover around zero, where the 1e-9 differences between
    # the two summation orders eventually flip a sign -- the first 100 recorded steps stay clear of that)
    steps = 100
    for i in range(steps):
        hits = ag.update_control(g['state'][i][None], g['action_in'][i][None], g['labels'][i][None])
        oh = oa.update_control(g['state'][i], g['action_in'][i], g['labels'][i])
        assert (hits[0] == oh).all(), i
        nxt = g['state'][i + 1]
        act, adj = ag.select_action(nxt[None])
        oact, oadj = oa.select_action(nxt)
        oa.adjusted = oadj
        ag.set_adjusted([oadj])
        assert (act[0] == oact).all() and adj[0] == oadj, i
    sizes = ag.dictionary_sizes()
    assert sizes.max() == cap and (sizes <= cap).all()
    assert [oa.m(s) for s in range(len(dims))] == sizes[0].tolist()
    ag.synchronize()   # saturation is not an error
    for s in range(len(dims)):
        np.testing.assert_allclose(ag.learner(0, s)['coeff'], oa.coeff(s), rtol=1e-7, atol=1e-9)
    ag.close()


def test_batched_evaluator_equals_run_by_run(golden_dir, tmp_path):
    """experiments_kbrl.BatchedEvaluator (all runs = replicas of one device-resident loop, histories recorded on the
    device) writes, run for run, the results_K.npz the reference-shaped Evaluator.evaluate(K) writes through the N=1
    drop-in classes: same keys, dtypes, shapes (G11) and the same numbers."""
    import experiments_kbrl as ek
    import scenario_creator as sc
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    sc.set_fading([g['t0'], g['t1'], g['t2']])
    ref = _load(golden_dir, 'g11_results_schema')
    steps, runs = 40, [0, 1, 2, 5]
    for scenario in (0, 2):
        serial = ek.Evaluator(scenario, [0.97, 0.99], steps=steps, out_dir=str(tmp_path / 'serial'))
        batched = ek.BatchedEvaluator(scenario, [0.97, 0.99], steps=steps, out_dir=str(tmp_path / 'batched'))
        files_b = batched.evaluate_all(runs, verbose=False)
        for k, i in enumerate(runs):
            a = np.load(serial.evaluate(i))
            b = np.load(files_b[k])
            assert sorted(a.files) == sorted(b.files) == sorted(x[4:] for x in ref.files)
            for key in a.files:
                assert a[key].dtype == b[key].dtype == ref['key_' + key].dtype, key
                assert a[key].shape == b[key].shape, key
                assert (a[key] == b[key]).all(), (scenario, i, key)
    sc.set_fading(None)


def test_graph_replayed_closed_loop_equals_stepwise(golden_dir):
    """kb_run_resident: the closed loop (agent step + simulator step) enqueued n steps at a time, two captured steps replayed as
    a hipGraph, == the same loop one kb_step_resident / rs_step_resident pair at a time: histories, final observations and
    actions, dictionaries (bit for bit).  Odd and even step counts, two calls in a row (the second finds the graph and has to
    realign with the parity of the alternating buffers), scenario_1 (eMBB + mMTC learners, the side stream inside the capture)."""
    import ctypes as C
    from ranslice.kbrl_dev import VecKBRL
    from ranslice.vec_env import VecRanSlice
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    fading = [g['t0'], g['t1'], g['t2']]
    scenario, N = 1, 24
    dims, n_prbs = _dims(scenario)
    rng = np.random.default_rng(31)
    ia = np.stack([np.concatenate([rng.integers(4, 20, 3), rng.integers(2, 10, 2)]) for _ in range(N)]).astype(np.int32)
    sf = np.stack([np.concatenate([rng.integers(2, 8, 3), rng.integers(1, 4, 2)]) for _ in range(N)]).astype(np.int32)
    total = 61 + 30 + 2
    out = []
    for mode in ('graph', 'stepwise', 'calls without graph'):
        env = VecRanSlice(n_envs=N, cfg=make_config(scenario, n_envs=N), fading=fading, seed=19)
        ag = VecKBRL(N, dims, n_prbs, capacity=512, pool_bytes=256 << 20)
        env.reset()
        ag.reset(ia, sf, seeds=np.arange(N, dtype=np.uint64) + 4)
        ag.history_begin(total)
        env._check(env.L.rs_step(env.h, ia.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
        if mode == 'stepwise':
            for _ in range(total):
                ag.step_resident(env)
                env.step_resident()
        else:
            for k in (61, 30, 2):
                ag.run_resident(env, k, graph=mode == 'graph')
        f = env.fetch()
        h = ag.history_fetch()
        d = [ag.learner(r, s, with_kinv=True) for r in (0, N - 1) for s in range(len(dims))]
        out.append((f, h, d, ag.dictionary_sizes().copy()))
        env.close()
        ag.close()
    ref = out[1]
    assert ref[1]['recorded'] == total and ref[3].max() > 8
    for o in (out[0], out[2]):
        for key in ('obs', 'actions', 'reward', 'labels'):
            assert o[0][key].tobytes() == ref[0][key].tobytes(), key
        for key in ('reward', 'resources', 'hits', 'adjusted', 'SLA', 'violation', 'recorded'):
            assert np.array_equal(o[1][key], ref[1][key]), key
        assert (o[3] == ref[3]).all()
        for a, b in zip(o[2], ref[2]):
            assert a['m'] == b['m'] and all(a[k].tobytes() == b[k].tobytes() for k in ('landmarks', 'coeff', 'kinv'))


def test_agent_side_calls_see_the_end_of_a_graph_loop(golden_dir):
    """kb_run_resident replays its graph on the SIMULATOR's stream; the agent-side calls (kb_get_stats, kb_save_state,
    kb_synchronize) wait for the agent's stream only.  The loop now joins the agent's stream to its last graph launch: a call
    that ENDS on a graph launch (7 = one plain step + three graph launches of two) followed at once by agent.stats() /
    agent.save_state(), with no env.synchronize() in between, sees what a fully synchronised stepwise loop leaves."""
    import ctypes as C
    from ranslice.kbrl_dev import VecKBRL
    from ranslice.vec_env import VecRanSlice
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    fading = [g['t0'], g['t1'], g['t2']]
    scenario, N = 0, 256
    dims, n_prbs = _dims(scenario)
    rng = np.random.default_rng(5)
    ia = rng.integers(4, 20, size=(N, 5)).astype(np.int32)
    sf = rng.integers(2, 8, size=(N, 5)).astype(np.int32)
    got = []
    for mode in ('graph', 'stepwise'):
        env = VecRanSlice(n_envs=N, cfg=make_config(scenario, n_envs=N), fading=fading, seed=23)
        ag = VecKBRL(N, dims, n_prbs, capacity=512, pool_bytes=512 << 20)
        env.reset()
        ag.reset(ia, sf, seeds=np.arange(N, dtype=np.uint64) + 11)
        env._check(env.L.rs_step(env.h, ia.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
        for k in (7, 7, 9):
            if mode == 'graph':
                ag.run_resident(env, k, graph=True)
                stats = ag.stats()                 # agent's stream only
                blob = ag.save_state()
                env.synchronize()                  # everything at rest: the checkpoint taken above must already be this one
                ag.synchronize()
                assert ag.save_state().tobytes() == blob.tobytes(), 'checkpoint torn after a %d-step graph call' % k
                assert list(ag.stats()) == list(stats)
            else:
                for _ in range(k):
                    ag.step_resident(env)
                    env.step_resident()
                env.synchronize()
                ag.synchronize()
                stats = ag.stats()
                blob = ag.save_state()
            # (the blobs of the two modes are not compared: scratch rows of the repair rounds, which the captured sequence always
            # carries, travel in them; the dictionaries and the control state are)
            d = [ag.learner(r, s_, with_kinv=True) for r in (0, N - 1) for s_ in range(5)]
            got.append((mode, k, list(stats), d, ag.dictionary_sizes().copy()))
        env.close()
        ag.close()
    for a, b in zip(got[:3], got[3:]):
        assert a[2] == b[2], (a[1], a[2], b[2])
        assert (a[4] == b[4]).all()
        for x, y in zip(a[3], b[3]):
            assert x['m'] == y['m'] and all(x[k].tobytes() == y[k].tobytes() for k in ('landmarks', 'coeff', 'kinv'))


def test_load_state_refuses_torn_blobs_and_accepts_another_pool_size(golden_dir):
    """kb_load_state recomputes the size its header implies -- a truncated blob or one whose header lies is refused
    before anything is read past its end -- and the pool's own size is not part of the configuration: a blob loads into a
    handle with a LARGER (or smaller but sufficient) pool, and is refused with both sizes named when it does not fit."""
    from ranslice import _lib
    from ranslice.kbrl_dev import VecKBRL
    g = _load(golden_dir, 'g9_projectron')
    dims, n_prbs = _dims(0)
    N = 4
    ag = VecKBRL(N, dims, n_prbs, capacity=512, pool_bytes=64 << 20)
    ag.reset(np.full((N, 5), 10, np.int32), np.full((N, 5), 3, np.int32))
    x, y = g['d11_x'], g['d11_y']
    for i in range(700):     # learner (0, 0) grows past its first shell of 64 landmarks; the others take a few samples each
        e, sl = (0, 0) if i < 600 else (i % N, i % 5)      # (every one of the 20 dictionaries takes its first shell)
        yp, f = ag.predict(e, sl, x[i])
        ag.update(e, sl, x[i], int(y[i]))
    assert ag.learner(0, 0)['m'] > 64
    blob = ag.save_state()
    want = [ag.learner(r, s, with_kinv=True) for r in range(N) for s in range(5)]
    ag.close()
    for pool in (256 << 20, 32 << 20):
        other = VecKBRL(N, dims, n_prbs, capacity=512, pool_bytes=pool)
        other.reset(np.full((N, 5), 1, np.int32), np.full((N, 5), 1, np.int32))
        other.load_state(blob)
        have = [other.learner(r, s, with_kinv=True) for r in range(N) for s in range(5)]
        for a, b in zip(want, have):
            assert a['m'] == b['m'] and all(a[k].tobytes() == b[k].tobytes() for k in ('landmarks', 'coeff', 'kinv'))
        with pytest.raises(_lib.RanSliceError) as e:
            other.load_state(blob[:-4096])                   # truncated
        assert 'truncated or corrupt' in str(e.value)
        lied = blob.copy()
        lied[40:48] = np.frombuffer(np.uint64(blob.size - 4096).tobytes(), dtype=np.uint8)   # kb_state_header.total_bytes
        with pytest.raises(_lib.RanSliceError) as e:
            other.load_state(lied)
        assert 'truncated or corrupt' in str(e.value)
        other.close()
    tiny = VecKBRL(N, dims, n_prbs, capacity=512, pool_bytes=1 << 20)
    tiny.reset(np.full((N, 5), 1, np.int32), np.full((N, 5), 1, np.int32))
    with pytest.raises(_lib.RanSliceError) as e:
        tiny.load_state(blob)
    assert 'bytes of pool' in str(e.value)
    tiny.close()


def test_checkpoint_and_resume(golden_dir, tmp_path):
    """rs_save_state / kb_save_state: (1) a closed loop cut after 25 steps, restored into FRESH handles and continued, makes
    the steps the uncut loop makes (observations, actions, dictionaries bit for bit); a blob of another configuration is
    refused; (2) BatchedEvaluator.evaluate_all interrupted at step 35 and resumed from its checkpoint writes the result
    files of the uninterrupted evaluation."""
    import ctypes as C
    import experiments_kbrl as ek
    import scenario_creator as sc
    from ranslice import _lib
    from ranslice.kbrl_dev import VecKBRL
    from ranslice.vec_env import VecRanSlice
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    fading = [g['t0'], g['t1'], g['t2']]
    scenario, N = 1, 12
    dims, n_prbs = _dims(scenario)
    rng = np.random.default_rng(3)
    ia = np.stack([np.concatenate([rng.integers(4, 20, 3), rng.integers(2, 10, 2)]) for _ in range(N)]).astype(np.int32)
    sf = np.stack([np.concatenate([rng.integers(2, 8, 3), rng.integers(1, 4, 2)]) for _ in range(N)]).astype(np.int32)

    def make(n=N):
        env = VecRanSlice(n_envs=n, cfg=make_config(scenario, n_envs=n), fading=fading, seed=9)
        ag = VecKBRL(n, dims, n_prbs, capacity=512, pool_bytes=256 << 20)
        return env, ag

    def loop(env, ag, k, out):
        for _ in range(k):
            ag.step_resident(env)
            env.step_resident()
            f = env.fetch()
            out.append((f['obs'].copy(), f['actions'].copy(), f['reward'].copy()))
    env, ag = make()
    env.reset()
    ag.reset(ia, sf, seeds=np.arange(N, dtype=np.uint64) + 2)
    ag.history_begin(60)
    env._check(env.L.rs_step(env.h, ia.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
    whole = []
    loop(env, ag, 25, whole)
    blob_e, blob_a = env.save_state(), ag.save_state()
    loop(env, ag, 35, whole)
    hist_w = ag.history_fetch()
    dict_w = [ag.learner(r, s, with_kinv=True) for r in (0, N - 1) for s in range(len(dims))]
    env.close()
    ag.close()
    env2, ag2 = make()                       # fresh handles: nothing but the blobs carries over
    env2.reset()
    ag2.reset(ia * 0 + 1, sf * 0 + 1)
    env2.load_state(blob_e)
    ag2.load_state(blob_a)
    cont = []
    loop(env2, ag2, 35, cont)
    for i in range(35):
        for a, b in zip(whole[25 + i], cont[i]):
            assert a.tobytes() == b.tobytes(), i
    hist_c = ag2.history_fetch()
    assert hist_c['recorded'] == hist_w['recorded'] == 60
    for key in ('reward', 'resources', 'hits', 'adjusted', 'SLA', 'violation'):
        assert (hist_c[key] == hist_w[key]).all(), key
    dict_c = [ag2.learner(r, s, with_kinv=True) for r in (0, N - 1) for s in range(len(dims))]
    for a, b in zip(dict_w, dict_c):
        assert a['m'] == b['m'] and all(a[k].tobytes() == b[k].tobytes() for k in ('landmarks', 'coeff', 'kinv'))
    env3, ag3 = make(n=N + 1)
    with pytest.raises(_lib.RanSliceError):
        env3.load_state(blob_e)
    with pytest.raises(_lib.RanSliceError):
        ag3.load_state(blob_a)
    # a blob of the format before round 6 ("KBSLICE4": its configuration hash covered the pool's size) is refused BY NAME, not as
    # "another configuration" (ADVICE r5); a blob loaded into a roomier pool keeps its dictionaries and drops "pool exhausted"
    old = blob_a.copy()
    old[:8] = np.frombuffer((0x4b42534c49434534).to_bytes(8, 'little'), dtype=np.uint8)
    with pytest.raises(_lib.RanSliceError) as e:
        ag2.load_state(old)
    assert 'older checkpoint format' in str(e.value)
    big = VecKBRL(N, dims, n_prbs, capacity=512, pool_bytes=512 << 20)
    big.reset(ia * 0 + 1, sf * 0 + 1)
    big.load_state(blob_a)          # (the pool's size is not part of the configuration)
    assert big.pool()['pool_full'] == 0
    big.close()
    for h in (env2, ag2, env3, ag3):
        h.close()
    # ---- the evaluator: interrupted and resumed == uninterrupted
    sc.set_fading(fading)
    steps, runs = 60, [0, 1, 2, 3]
    one = ek.BatchedEvaluator(0, [0.97, 0.99], steps=steps, out_dir=str(tmp_path / 'whole'))
    files_w = one.evaluate_all(runs, verbose=False, pool_bytes=1 << 30)
    two = ek.BatchedEvaluator(0, [0.97, 0.99], steps=steps, out_dir=str(tmp_path / 'cut'))
    ck = str(tmp_path / 'cell.npz')
    assert two.evaluate_all(runs, verbose=False, pool_bytes=1 << 30, checkpoint=ck, stop_after=35) is None and os.path.exists(ck)
    files_c = ek.BatchedEvaluator(0, [0.97, 0.99], steps=steps, out_dir=str(tmp_path / 'cut')).evaluate_all(
        runs, verbose=False, pool_bytes=1 << 30, checkpoint=ck)
    for fa, fb in zip(files_w, files_c):
        a, b = np.load(fa), np.load(fb)
        for key in a.files:
            assert (a[key] == b[key]).all(), key
    sc.set_fading(None)


def test_grid_of_cells_equals_cell_by_cell(golden_dir, tmp_path):
    """experiments_kbrl.evaluate_grid runs several (scenario, accuracy range) cells of the reference's experiment grid
    (experiments_kbrl.py:57-70) as ONE job -- every cell's environment and agents on streams of their own, advanced in the same
    host loop -- and writes, file for file, what BatchedEvaluator.evaluate_all writes one cell after the other."""
    import experiments_kbrl as ek
    import scenario_creator as sc
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    sc.set_fading([g['t0'], g['t1'], g['t2']])
    steps, runs = 60, [0, 1, 2, 3, 4, 5]
    cells = [(0, [0.97, 0.99]), (0, [0.99, 0.999]), (2, [0.97, 0.99]), (1, [0.99, 0.999])]
    together = ek.evaluate_grid(cells, runs, steps=steps, out_dir=str(tmp_path / 'grid'), pool_bytes=1 << 30)
    for scenario, a_range in cells:
        one = ek.BatchedEvaluator(scenario, a_range, steps=steps, out_dir=str(tmp_path / 'cell'))
        files = one.evaluate_all(runs, verbose=False, pool_bytes=1 << 30)
        for fa, fb in zip(together[(scenario, a_range[0])], files):
            assert os.path.basename(fa) == os.path.basename(fb)
            a, b = np.load(fa), np.load(fb)
            assert sorted(a.files) == sorted(b.files)
            for key in a.files:
                assert a[key].dtype == b[key].dtype and (a[key] == b[key]).all(), (scenario, a_range, key)
    sc.set_fading(None)


# ------------------------------------------------------------------------------------------------------------------
# round 3: dictionaries of real size -- reference-recorded long sequences, saturation at capacity 1024,
# the pooled storage

def _kinv_digest_check(kinv, g, rtol):
    scale = np.abs(g['kinv_rows']).max()
    np.testing.assert_allclose(kinv[::16], g['kinv_rows'], rtol=rtol, atol=rtol * scale)
    np.testing.assert_allclose(np.diag(kinv), g['kinv_diag'], rtol=rtol, atol=rtol * scale)
    np.testing.assert_allclose(kinv @ g['kinv_probes'], g['kinv_kp'], rtol=rtol, atol=rtol * np.abs(g['kinv_kp']).max())


def test_projectron_long_golden_to_790_landmarks(golden_dir):
    """G14: the reference's Projectron driven to 790 landmarks (7,000 samples, projections still happening above 600)
    through kb_predict / kb_update: f within 1e-8 relative, every predicted sign, branch and dictionary size exact,
    delta within 1e-7; final landmarks exact, coeff 1e-7, Kinv (every 16th row, diagonal, four probe products) 1e-6 --
    its conditioning degrades with 1/delta, and the device sums d* = Kinv K_f column-wise with fused multiply-adds"""
    from ranslice.kbrl_dev import VecKBRL
    g = _load(golden_dir, 'g14_projectron_long')
    ag = VecKBRL(1, [10], 200, capacity=4096)
    ag.reset([[10]], [[3]])
    xs, ys = g['x'], g['y']
    for i in range(len(xs)):
        yp, f = ag.predict(0, 0, xs[i])
        fr = g['f'][i]
        assert f == pytest.approx(fr, rel=1e-8, abs=TOL), i
        if abs(fr) > TOL:
            assert yp == g['ypred'][i], i
        br, dl = ag.update(0, 0, xs[i], int(ys[i]))
        assert br == g['branch'][i], i
        if br:
            assert dl == pytest.approx(g['delta'][i], rel=1e-7, abs=1e-9), i
        if i % 64 == 0 or i > len(xs) - 8:
            assert ag.dictionary_sizes()[0, 0] == g['m'][i], i
    L = ag.learner(0, 0, with_kinv=True)
    assert L['m'] == len(g['landmarks']) == 790
    np.testing.assert_array_equal(L['landmarks'], g['landmarks'])
    np.testing.assert_allclose(L['coeff'], g['coeff'], rtol=1e-7, atol=1e-8)
    _kinv_digest_check(L['kinv'], g, 1e-6)
    assert np.array_equal(L['kinv'], L['kinv'].T)      # bit-symmetric: the column-walk mat-vec relies on it
    p = ag.pool()
    assert p['saturated'] == 0 and p['pool_full'] == 0 and 0 < p['used_bytes'] <= p['total_bytes']
    ag.close()


def test_projectron_reference_golden_to_3000_landmarks(golden_dir):
    """G17: the REFERENCE's Projectron driven to 3,000+ landmarks (12,400 samples) through kb_predict / kb_update -- the device
    against the reference itself where long runs live, not only against the oracle: f within 1e-8 relative, every predicted
    sign, branch and dictionary size exact, delta 1e-6; final landmarks exact, coefficients 1e-6, Kinv (diagonal, every 256th
    row, four probe products) 1e-6 of its scale, symmetric bit for bit."""
    from ranslice.kbrl_dev import VecKBRL
    g = _load(golden_dir, 'g17_projectron_3000')
    xs = np.concatenate([g['state'].astype(np.float64), (g['a'].astype(np.float64) / 200)[:, None]], axis=1)
    ys = g['y']
    ag = VecKBRL(1, [10], 200, capacity=4096)
    ag.reset([[10]], [[3]])
    for i in range(len(xs)):
        yp, f = ag.predict(0, 0, xs[i])
        fr = g['f'][i]
        assert f == pytest.approx(fr, rel=1e-8, abs=TOL), i
        if abs(fr) > 1e-7:
            assert yp == g['ypred'][i], i
        br, dl = ag.update(0, 0, xs[i], int(ys[i]))
        assert br == g['branch'][i], (i, dl, g['delta'][i])
        if br:
            assert dl == pytest.approx(g['delta'][i], rel=1e-6, abs=1e-9), i
        if i % 256 == 0 or i > len(xs) - 8:
            assert ag.dictionary_sizes()[0, 0] == g['m'][i], i
    L = ag.learner(0, 0, with_kinv=True)
    assert L['m'] == g['m'][-1] >= 3000
    np.testing.assert_array_equal(L['landmarks'], xs[g['branch'] == 2])
    np.testing.assert_allclose(L['coeff'], g['coeff'], rtol=1e-6, atol=1e-8)
    scale = np.abs(g['kinv_diag']).max()
    np.testing.assert_allclose(L['kinv'][::256], g['kinv_rows'], rtol=1e-6, atol=1e-6 * scale)
    np.testing.assert_allclose(np.diag(L['kinv']), g['kinv_diag'], rtol=1e-6, atol=1e-6 * scale)
    np.testing.assert_allclose(L['kinv'] @ g['kinv_probes'], g['kinv_kp'], rtol=1e-6, atol=1e-6 * np.abs(g['kinv_kp']).max())
    assert np.array_equal(L['kinv'], L['kinv'].T)
    ag.close()


@pytest.mark.parametrize('name,min_m,heavy_m,rounds', [('g15_kbrl_long_s0', 200, None, None), ('g15_kbrl_long_s0', 200, None, 3),
                                                       ('g16_kbrl_long_tdl_s0', 0, None, 2), ('g15_kbrl_long_s0', 200, 100, 1),
                                                       ('g16_kbrl_long_tdl_s0', 0, 1000000, None)])
def test_kbrl_control_long_golden(golden_dir, monkeypatch, name, min_m, heavy_m, rounds):
    """G15 / G16: KBRL_Control teacher-forced over the reference's 2,200 recorded steps of scenario_0 (G15: dictionaries
    of several hundred landmarks): every hit, selected action, adjusted flag, margin, security factor and dictionary
    size; final landmarks exact, coefficients 1e-6.  Learners with a mistake to repair are queued: small dictionaries for
    update_small_kernel; large ones (192 landmarks and more) for the chip-wide repair rounds, which a single agent never
    triggers on its own (they start once a step has queued eight large learners) -- KBRL_ROUNDS = r forces r rounds per
    step, whatever is left goes to the per-learner clean-up, which does all of it in the first case.  KBRL_HEAVY_M = h
    keeps learners below h landmarks in the one-wave kernel (a mix of paths, and the one-wave path alone)."""
    from ranslice.kbrl_dev import VecKBRL
    if heavy_m is not None or rounds is not None:
        monkeypatch.setenv('RANSLICE_DEV_BUILD', '1')   # knobs are read by the test build only (ranslice._lib)
    if heavy_m is not None:
        monkeypatch.setenv('KBRL_HEAVY_M', str(heavy_m))
    if rounds is not None:
        monkeypatch.setenv('KBRL_ROUNDS', str(rounds))
    g = _load(golden_dir, name)
    dims, n_prbs = _dims(0)
    ag = VecKBRL(1, dims, n_prbs, accuracy_range=tuple(g['a_range']), capacity=4096)
    ag.reset(g['init_action'][None].astype(np.int32), g['init_sec'][None].astype(np.int32))
    steps = len(g['state'])
    for i in range(steps):
        hits = ag.update_control(g['state'][i][None], g['action_in'][i][None], g['labels'][i][None])
        assert (hits[0] == g['hits'][i]).all(), i
        nxt = g['state'][i + 1] if i + 1 < steps else g['final_state']
        act, adj = ag.select_action(nxt[None])
        assert (act[0] == g['action_out'][i]).all(), i
        assert adj[0] == g['adjusted'][i]
        if i % 20 == 0 or i == steps - 1:
            c = ag.control(with_accuracies=False)
            assert (c['margins'][0] == g['margins'][i]).all() and (c['security_factors'][0] == g['security'][i]).all(), i
            sizes = ag.dictionary_sizes()[0]
            want = np.array([0 if z == 0 else z for z in g['set_size'][i]])
            # get_set_size() of a single-landmark dictionary is len(x) (projectron.py:62-64), not 1
            assert all(sz == w or (sz == 1 and w == dims[s] + 1) for s, (sz, w) in enumerate(zip(sizes, want))), i
    np.testing.assert_allclose(ag.control()['accuracies'][0], g['acc'][-1], rtol=1e-12, atol=0)
    for s in range(len(dims)):
        L = ag.learner(0, s)
        np.testing.assert_array_equal(L['landmarks'], np.atleast_2d(g['landmarks%d' % s]))
        np.testing.assert_allclose(L['coeff'], g['coeff%d' % s], rtol=1e-6, atol=1e-8)
    assert ag.dictionary_sizes().max() >= min_m
    ag.close()


def _fast_growing_samples(n, seed=141, spread=1.6, revisit=0.5):
    """a stream on which the dictionary grows by about one landmark every five samples (states spread over [0, spread]^10,
    noisy labels); every second sample revisits an earlier state with a new action, as sample augmentation does"""
    rng = np.random.default_rng(seed)
    xs, ys, states = [], [], []
    for i in range(n):
        if states and rng.random() < revisit:
            s = states[rng.integers(len(states))]
        else:
            s = (rng.random(10) * spread).astype(np.float32)
        states.append(s)
        x = np.append(s, rng.integers(0, 201) / 200)
        score = x[:-1].mean() * 0.5 * 1.6 / spread + 0.35 - x[-1]
        ys.append(-1 if score + rng.normal(0, 0.15) > 0 else 1)
        xs.append(x)
    return np.asarray(xs), np.asarray(ys)


def test_projectron_through_saturation_at_capacity_1024():
    """device vs oracle, Projectron.predict / update on one stream that fills a dictionary of capacity 1,024 and then
    keeps projecting onto it (the fixed-budget behaviour both share; the reference's SVvariable would keep growing):
    f within 1e-8, predicted sign / branch / size exact at every sample, delta 1e-6; final coefficients 1e-6"""
    from ranslice.kbrl_dev import VecKBRL
    xs, ys = _fast_growing_samples(9000)
    cap = 1024
    ag = VecKBRL(1, [10], 200, capacity=cap)
    ag.reset([[10]], [[3]])
    oa = po.OracleKBRL([10], 200, [10], [3], capacity=cap)
    oa.set_seed(0)
    n_sat = 0
    for i in range(len(xs)):
        yp, f = ag.predict(0, 0, xs[i])
        oyp, of = oa.predict(0, xs[i])
        assert f == pytest.approx(of, rel=1e-8, abs=TOL), i
        if abs(of) > 1e-7:
            assert yp == oyp, i
        br, dl = ag.update(0, 0, xs[i], int(ys[i]))
        obr, odl = oa.update(0, xs[i], int(ys[i]))
        assert br == obr, (i, br, obr, dl, odl)
        if br:
            assert dl == pytest.approx(odl, rel=1e-6, abs=1e-9), i
            n_sat += int(oa.m(0) == cap and odl > 0.1)
    assert oa.m(0) == cap and ag.dictionary_sizes()[0, 0] == cap
    assert n_sat > 100, 'the stream should keep hitting the full dictionary'
    np.testing.assert_allclose(ag.learner(0, 0)['coeff'], oa.coeff(0), rtol=1e-6, atol=1e-8)
    p = ag.pool()
    assert p['saturated'] == 1 and p['pool_full'] == 0
    ag.synchronize()          # a dictionary at its capacity is reported, not an error
    ag.close()


def test_projectron_to_3000_landmarks_vs_oracle():
    """device vs oracle where long runs live.  One stream takes Projectron.predict / update to more than
    3,000 landmarks at capacity 4,096 (shell indices up to 47, a Kinv of 9.4 M entries in 1,176 tiles of the lower block
    triangle, 1 / delta conditioning of a dictionary that size): f within 1e-8, predicted sign / branch / dictionary size
    exact at every sample, delta within 1e-6 relative (it is 1 - k.Kinv k: a difference of two numbers near one whose
    Kinv entries have been through thousands of rank-1 updates in two different summation orders); final coefficients
    1e-6, Kinv probes 1e-6 of its scale"""
    from ranslice.kbrl_dev import VecKBRL
    xs, ys = _fast_growing_samples(12400, spread=3.0, revisit=0.35)
    cap = 4096
    ag = VecKBRL(1, [10], 200, capacity=cap)
    ag.reset([[10]], [[3]])
    oa = po.OracleKBRL([10], 200, [10], [3], capacity=cap)
    oa.set_seed(0)
    n_proj_large = 0
    for i in range(len(xs)):
        yp, f = ag.predict(0, 0, xs[i])
        oyp, of = oa.predict(0, xs[i])
        assert f == pytest.approx(of, rel=1e-8, abs=TOL), i
        if abs(of) > 1e-7:
            assert yp == oyp, i
        br, dl = ag.update(0, 0, xs[i], int(ys[i]))
        obr, odl = oa.update(0, xs[i], int(ys[i]))
        assert br == obr, (i, br, obr, dl, odl)
        if br:
            assert dl == pytest.approx(odl, rel=1e-6, abs=1e-9), i
            n_proj_large += int(br == 1 and oa.m(0) > 2000)
    m = oa.m(0)
    assert m >= 3000 and ag.dictionary_sizes()[0, 0] == m
    assert n_proj_large > 20, 'projections onto a dictionary of more than 2,000 landmarks should be part of the stream'
    L = ag.learner(0, 0, with_kinv=True)
    np.testing.assert_array_equal(L['landmarks'], oa.landmarks(0))
    np.testing.assert_allclose(L['coeff'], oa.coeff(0), rtol=1e-6, atol=1e-8)
    ko = oa.kinv(0)
    assert np.array_equal(L['kinv'], L['kinv'].T), 'Kinv is symmetric bit for bit (only its lower block triangle is stored)'
    scale = np.abs(ko).max()
    np.testing.assert_allclose(L['kinv'], ko, rtol=1e-6, atol=1e-6 * scale)
    p = ag.pool()
    # the triangle: shells 0 .. 47 hold 48 pages and 48 * 49 / 2 tiles
    nb = (m + 63) // 64
    assert p['used_bytes'] == 64 * 8 + (nb * 30 * 64 + nb * (nb + 1) // 2 * (4096 + 128)) * 8, p
    assert p['saturated'] == 0 and p['pool_full'] == 0
    ag.close()


@pytest.mark.parametrize('rounds', [None, 2])
def test_control_with_dictionaries_above_1024_vs_oracle(golden_dir, monkeypatch, rounds):
    """KBRL_Control on dictionaries of 1,300 to 1,900 landmarks: every learner of one agent (device and oracle alike) is first
    grown through Projectron.predict / update on its own stream, then both take 120 control steps of G15's recorded
    (state, action, labels): hits, selected actions, adjusted, margins, security factors and dictionary sizes agree at every
    step.  rounds = 2 sends the repairs through the chip-wide rounds (heavy_matvec / heavy_finish / heavy_rank1 over the
    triangle's tiles); None leaves them to the per-learner clean-up workgroups."""
    from ranslice.kbrl_dev import VecKBRL
    if rounds is not None:
        monkeypatch.setenv('RANSLICE_DEV_BUILD', '1')   # knobs are read by the test build only (ranslice._lib)
        monkeypatch.setenv('KBRL_ROUNDS', str(rounds))
    g = _load(golden_dir, 'g15_kbrl_long_s0')
    dims, n_prbs = _dims(0)
    cap = 4096
    ag = VecKBRL(1, dims, n_prbs, accuracy_range=tuple(g['a_range']), capacity=cap)
    ag.reset(g['init_action'][None].astype(np.int32), g['init_sec'][None].astype(np.int32), seeds=np.zeros(1, dtype=np.uint64))
    oa = po.OracleKBRL(dims, n_prbs, g['init_action'], g['init_sec'], accuracy_range=tuple(g['a_range']), capacity=cap)
    oa.set_seed(0)
    for s in range(len(dims)):
        xs, ys = _fast_growing_samples(8000 + 800 * s, seed=300 + s, revisit=0.35)
        for i in range(len(xs)):
            ag.predict(0, s, xs[i])
            ag.update(0, s, xs[i], int(ys[i]))
            oa.predict(s, xs[i])
            oa.update(s, xs[i], int(ys[i]))
        assert ag.dictionary_sizes()[0, s] == oa.m(s) >= 1300, (s, oa.m(s))
    steps, grown = 120, 0
    m0 = [oa.m(s) for s in range(len(dims))]
    for i in range(steps):
        hits = ag.update_control(g['state'][i][None], g['action_in'][i][None], g['labels'][i][None])
        oh = oa.update_control(g['state'][i], g['action_in'][i], g['labels'][i])
        assert (hits[0] == oh).all(), i
        act, adj = ag.select_action(g['state'][i + 1][None])
        oact, oadj = oa.select_action(g['state'][i + 1])
        oa.adjusted = oadj
        assert (act[0] == oact).all() and adj[0] == oadj, i
        c = ag.control(with_accuracies=False)
        assert (c['margins'][0] == oa.margins).all() and (c['security_factors'][0] == oa.security_factors).all(), i
        assert (ag.dictionary_sizes()[0] == [oa.m(s) for s in range(len(dims))]).all(), i
    grown = sum(oa.m(s) - m0[s] for s in range(len(dims)))
    assert grown > 20, 'the control steps should keep inserting into the large dictionaries'
    for s in range(len(dims)):
        np.testing.assert_allclose(ag.learner(0, s)['coeff'], oa.coeff(s), rtol=1e-6, atol=1e-8)
    ag.close()


def test_control_through_saturation_on_the_long_golden(golden_dir):
    """KBRL_Control on G15's recorded sequence with dictionaries capped at 96 landmarks (the reference's reach 200+): device
    and oracle agree on every hit, action and dictionary size while learners saturate and stop augmenting"""
    from ranslice.kbrl_dev import VecKBRL
    g = _load(golden_dir, 'g15_kbrl_long_s0')
    dims, n_prbs = _dims(0)
    # (three of the five learners are full from step 180 on; a saturated, under-fitted classifier's scores hover around
    # zero, where the 1e-9 differences between the two summation orders eventually flip a sign: 260 steps stay clear)
    cap, steps = 96, 260
    ag = VecKBRL(1, dims, n_prbs, accuracy_range=tuple(g['a_range']), capacity=cap)
    ag.reset(g['init_action'][None].astype(np.int32), g['init_sec'][None].astype(np.int32), seeds=np.zeros(1, dtype=np.uint64))
    oa = po.OracleKBRL(dims, n_prbs, g['init_action'], g['init_sec'], accuracy_range=tuple(g['a_range']), capacity=cap)
    oa.set_seed(0)
    for i in range(steps):
        st, ac, lb = g['state'][i], g['action_in'][i].astype(np.int32), g['labels'][i].astype(np.int32)
        hits = ag.update_control(st[None], ac[None], lb[None])
        oh = oa.update_control(st, ac, lb)
        assert (hits[0] == oh).all(), i
        act, adj = ag.select_action(g['state'][i + 1][None])
        oact, oadj = oa.select_action(g['state'][i + 1])
        oa.adjusted = oadj
        ag.set_adjusted([oadj])
        assert (act[0] == oact).all() and adj[0] == oadj, i
        if i % 25 == 0:
            assert ag.dictionary_sizes()[0].tolist() == [oa.m(s) for s in range(len(dims))], i
    assert ag.dictionary_sizes().max() == cap and ag.pool()['saturated'] == 1
    ag.close()


def test_off_grid_landmarks_mix_with_the_control_loop(golden_dir):
    """landmarks inserted through Projectron.update with an arbitrary last coordinate (not a multiple of 1/n_prbs) are
    scored by the direct exponential inside the streaming pass: a dictionary holding both kinds, then update_control /
    select_action, device vs oracle"""
    from ranslice.kbrl_dev import VecKBRL
    g = _load(golden_dir, 'g10_kbrl_s0')
    dims, n_prbs = _dims(0)
    ag = VecKBRL(1, dims, n_prbs, accuracy_range=tuple(g['a_range']), capacity=256)
    ag.reset(g['init_action'][None], g['init_sec'][None])
    oa = po.OracleKBRL(dims, n_prbs, g['init_action'], g['init_sec'], accuracy_range=tuple(g['a_range']), capacity=256)
    oa.set_seed(0)
    rng = np.random.default_rng(5)
    for i in range(60):
        for s in (0, 3):
            x = np.append(g['state'][i][10 * s:10 * s + 10].astype(np.float64) + rng.normal(0, 0.05, 10), rng.random())
            y = 1 if x[-1] > 0.2 else -1
            yp, f = ag.predict(0, s, x)
            oyp, of = oa.predict(s, x)
            assert f == pytest.approx(of, rel=1e-8, abs=TOL)
            assert ag.update(0, s, x, y)[0] == oa.update(s, x, y)[0]
        hits = ag.update_control(g['state'][i][None], g['action_in'][i][None], g['labels'][i][None])
        oh = oa.update_control(g['state'][i], g['action_in'][i], g['labels'][i])
        assert (hits[0] == oh).all(), i
        act, adj = ag.select_action(g['state'][i + 1][None])
        oact, oadj = oa.select_action(g['state'][i + 1])
        oa.adjusted = oadj
        ag.set_adjusted([oadj])
        assert (act[0] == oact).all() and adj[0] == oadj, i
    assert ag.dictionary_sizes()[0].tolist() == [oa.m(s) for s in range(len(dims))]
    for s in range(len(dims)):
        np.testing.assert_allclose(ag.learner(0, s)['coeff'], oa.coeff(s), rtol=1e-7, atol=1e-9)
    ag.close()


def test_pool_grows_on_demand_and_reports_exhaustion():
    """the pool: kb_create accepts capacities far above round 2's 1,024 without reserving them (a 4096 x 5 handle at
    capacity 4,096 would have been 2.7 TB of dense Kinv); usage follows the dictionaries; a pool too small to let a
    dictionary take its next shell makes it project instead (flagged, not an error)"""
    from ranslice import _lib
    from ranslice.kbrl_dev import VecKBRL
    xs, ys = _fast_growing_samples(1200)
    big = VecKBRL(4096, [10] * 5, 200, capacity=4096, pool_bytes=4 << 30)
    assert big.pool()['total_bytes'] <= 4 << 30
    big.close()
    def shell(b):   # bytes of shell b: the vector page (30 rows x 64) + the tiles (b, 0) .. (b, b) of Kinv's lower triangle with
        return (30 * 64 + (b + 1) * (4096 + 128)) * 8   # their 128 partial sums each
    # room for shells 0, 1, 2 of one dictionary and 0, 1 of the other; shell 3 (147 KB) is larger than what two shells leave
    ag = VecKBRL(2, [10], 200, capacity=65536, pool_bytes=64 * 8 + 2 * shell(0) + 2 * shell(1) + shell(2) + 4096)
    ag.reset([[10], [10]], [[3], [3]])
    assert ag.pool()['used_bytes'] == 64 * 8
    for i in range(len(xs)):
        ag.predict(0, 0, xs[i])
        ag.update(0, 0, xs[i], int(ys[i]))
    p = ag.pool()
    m = ag.dictionary_sizes()[0, 0]
    assert m == 192 and p['pool_full'] == 1 and p['saturated'] == 1, (m, p)   # shells 0, 1, 2 fit, shell 3 does not
    # the request that did not fit left the pool as it was (the top only moves when a shell is granted): the
    # other dictionary still finds room for its shells 0 and 1
    assert p['used_bytes'] == 64 * 8 + shell(0) + shell(1) + shell(2), p
    for i in range(len(xs)):
        ag.predict(1, 0, xs[i])
        ag.update(1, 0, xs[i], int(ys[i]))
    p = ag.pool()
    assert ag.dictionary_sizes()[1, 0] == 128 and p['used_bytes'] == 64 * 8 + 2 * shell(0) + 2 * shell(1) + shell(2), p
    assert p['pool_full'] == 2
    ag.synchronize()
    with pytest.raises(_lib.RanSliceError):
        VecKBRL(64, [10] * 5, 200, capacity=1024, pool_bytes=1 << 20)     # not even one shell per dictionary
    ag.close()


def test_long_closed_loop_with_repair_rounds_vs_oracle():
    """BASELINE config 3's loop well past the start of learning: 4096 replicas of scenario_0 with one KBRL agent each, closed
    on the device for 1,200 steps -- long enough for hundreds of dictionaries to pass 192 landmarks, so that the chip-wide
    repair rounds (heavy_matvec / heavy_finish / heavy_rank1), the small-dictionary repair kernel and the per-learner
    clean-up all take part, with the launcher switching the rounds on by itself.  16 sampled replicas against oracle env +
    oracle agent on the same streams: executed actions, observations (bits), labels and selected actions at EVERY step,
    dictionary sizes at the end.  (Kernel values differ by ulps between the two; a sign decision would have to sit within
    1e-15 of zero to tell.)"""
    import ctypes as C
    from concurrent.futures import ProcessPoolExecutor
    from ranslice.fading import synth_fading
    from ranslice.kbrl_dev import VecKBRL
    from ranslice.vec_env import VecRanSlice
    N, steps, cols, cap = 4096, 1200, 10000, 2048
    scenario = 0
    dims, n_prbs = _dims(scenario)
    rng = np.random.default_rng(12)
    ia = rng.integers(10, 35, size=(N, 5)).astype(np.int32)
    sf = rng.integers(2, 8, size=(N, 5)).astype(np.int32)
    sample = [0, 1, 63, 64, 255, 777, 1023, 1024, 2047, 2048, 3000, 3333, 4000, 4093, 4094, 4095]
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1), mp_context=_SPAWN) as ex:
        fut = ex.map(_oracle_closed_loop, [(scenario, replica_seed(301, r), 9 + r, ia[r], sf[r], steps, cols, cap) for r in sample],
                     chunksize=1)
        env = VecRanSlice(n_envs=N, cfg=make_config(scenario, n_envs=N), fading=[synth_fading(t, cols) for t in range(3)],
                          seed=301)
        ag = VecKBRL(N, dims, n_prbs, capacity=cap, pool_bytes=48 << 30)
        ag.reset(ia, sf, seeds=np.arange(N, dtype=np.uint64) + 9)
        env.reset()
        a0 = np.ascontiguousarray(ia)
        env._check(env.L.rs_step(env.h, a0.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
        hip = []
        executed = ia.copy()
        for i in range(steps):
            f = env.fetch()     # results of the step that executed `executed`
            ag.step_resident(env)
            nxt = env.fetch()['actions']
            hip.append((executed[sample].copy(), f['obs'][sample].copy(), f['labels'][sample].copy(), nxt[sample].copy()))
            executed = nxt
            if i + 1 < steps:
                env.step_resident()
        ag.synchronize()
        all_sizes = ag.dictionary_sizes()
        sizes = [all_sizes[r].tolist() for r in sample]
        pool = ag.pool()
        env.close()
        ag.close()
        ref = list(fut)
    assert (all_sizes >= 192).sum() >= 200, 'the run should have grown hundreds of large dictionaries: %d' % (all_sizes >= 192).sum()
    assert pool['saturated'] == 0 and pool['pool_full'] == 0
    for k, r in enumerate(sample):
        steps_ref, m_ref = ref[k]
        for i in range(steps):
            act, obs, lab, hits, na, adj = steps_ref[i]
            h = hip[i]
            assert (h[0][k] == act).all(), ('executed action', r, i)
            assert h[1][k].tobytes() == obs.tobytes(), ('obs', r, i)
            assert (h[2][k] == lab).all(), ('labels', r, i)
            assert (h[3][k] == na).all(), ('selected action', r, i, h[3][k], na)
        assert sizes[k] == m_ref, (r, sizes[k], m_ref)


def test_pool_exhaustion_at_batch_size_vs_oracle():
    """the pool-full path at batch size.  512 replicas of scenario_0 with agents on a pool that holds every
    dictionary's first shell and only a hundred and fifty second ones: within a few hundred steps dictionaries that reach 64
    landmarks find the pool exhausted, project instead of growing (size stays at 64) and flag their replica; the run goes
    on without an error.  WHICH dictionaries get the last shells depends on the order their workgroups reach the allocator
    (atomics), so the oracle cannot be told in advance where growth stops; replicas are independent, though, so
      * a replica that never met the exhausted pool agrees with oracle env + oracle agent at every step, and
      * one that did agrees up to the last poll (every 20 steps) at which its flag was still clear."""
    import ctypes as C
    from concurrent.futures import ProcessPoolExecutor
    from ranslice.fading import synth_fading
    from ranslice.kbrl_dev import VecKBRL
    from ranslice.vec_env import VecRanSlice
    N, steps, cols, cap, poll = 512, 420, 10000, 4096, 20
    scenario = 0
    dims, n_prbs = _dims(scenario)

    def shell(b):
        return (30 * 64 + (b + 1) * (4096 + 128)) * 8
    pool_bytes = 64 * 8 + N * len(dims) * shell(0) + 150 * shell(1)
    rng = np.random.default_rng(21)
    ia = rng.integers(10, 35, size=(N, 5)).astype(np.int32)
    sf = rng.integers(2, 8, size=(N, 5)).astype(np.int32)
    env = VecRanSlice(n_envs=N, cfg=make_config(scenario, n_envs=N), fading=[synth_fading(t, cols) for t in range(3)], seed=77)
    ag = VecKBRL(N, dims, n_prbs, capacity=cap, pool_bytes=pool_bytes)
    ag.reset(ia, sf, seeds=np.arange(N, dtype=np.uint64) + 5)
    env.reset()
    a0 = np.ascontiguousarray(ia)
    env._check(env.L.rs_step(env.h, a0.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
    hip = []
    clear_until = np.full(N, steps, dtype=np.int64)   # steps [0, clear_until) ran with the replica's flag clear
    executed = ia.copy()
    for i in range(steps):
        f = env.fetch()
        ag.step_resident(env)
        nxt = env.fetch()['actions']
        hip.append((executed.copy(), f['obs'].copy(), f['labels'].copy(), nxt.copy()))
        executed = nxt
        if i + 1 < steps:
            env.step_resident()
        if i % poll == poll - 1:
            for r in ag.flagged_replicas()['pool_full']:
                clear_until[r] = min(clear_until[r], i + 1 - poll)
    ag.synchronize()      # a pool that ran out is not an error
    p = ag.pool()
    sizes = ag.dictionary_sizes()
    flagged = set(ag.flagged_replicas()['pool_full'])
    for r in flagged:
        clear_until[r] = min(clear_until[r], steps - steps % poll - poll if steps % poll else steps - poll)
    env.close()
    ag.close()
    assert p['pool_full'] == len(flagged) >= 20, p
    assert p['used_bytes'] <= p['total_bytes'] and p['total_bytes'] - p['used_bytes'] < shell(1), p
    assert (sizes > 64).sum() <= 150 and (sizes == 64).sum() >= 20, ((sizes > 64).sum(), (sizes == 64).sum())
    never = [r for r in range(N) if r not in flagged]
    met = sorted(flagged, key=lambda r: -clear_until[r])
    sample = never[:6] + [r for r in met if clear_until[r] >= 100][:6]
    assert len(never) >= 6 and len(sample) >= 8
    with ProcessPoolExecutor(max_workers=min(12, os.cpu_count() or 1), mp_context=_SPAWN) as ex:
        ref = list(ex.map(_oracle_closed_loop, [(scenario, replica_seed(77, r), 5 + r, ia[r], sf[r], int(clear_until[r]), cols, cap)
                                                for r in sample], chunksize=1))
    for k, r in enumerate(sample):
        steps_ref, m_ref = ref[k]
        for i in range(int(clear_until[r])):
            act, obs, lab, hits, na, adj = steps_ref[i]
            h = hip[i]
            assert (h[0][r] == act).all(), ('executed action', r, i)
            assert h[1][r].tobytes() == obs.tobytes(), ('obs', r, i)
            assert (h[2][r] == lab).all(), ('labels', r, i)
            assert (h[3][r] == na).all(), ('selected action', r, i)
        if r not in flagged:
            assert sizes[r].tolist() == m_ref, (r, sizes[r], m_ref)


def test_long_closed_loop_run_to_run_and_reset():
    """Two runs of 4096 replicas x 900 closed-loop steps from the same seeds on ONE pair of handles (the second after
    env.reset / kb_reset): identical selected actions at every 25th step, identical dictionary sizes and coefficients at
    the end.  Between the runs the order in which learners take shells from the pool differs (atomics), and so does the step
    at which the launcher starts enqueueing the repair rounds (it reads a counter from pinned memory without waiting for
    the device) -- neither may show.  kb_reset hands the whole pool back."""
    import ctypes as C
    import hashlib
    from ranslice.fading import synth_fading
    from ranslice.kbrl_dev import VecKBRL
    from ranslice.vec_env import VecRanSlice
    N, steps, cols = 4096, 900, 10000
    dims, n_prbs = _dims(0)
    rng = np.random.default_rng(13)
    ia = rng.integers(10, 35, size=(N, 5)).astype(np.int32)
    sf = rng.integers(2, 8, size=(N, 5)).astype(np.int32)
    env = VecRanSlice(n_envs=N, cfg=make_config(0, n_envs=N), fading=[synth_fading(t, cols) for t in range(3)], seed=302)
    ag = VecKBRL(N, dims, n_prbs, capacity=2048, pool_bytes=32 << 30)
    results = []
    for rep in range(2):
        env.reset()
        ag.reset(ia, sf, seeds=np.arange(N, dtype=np.uint64) + 11)
        assert ag.pool()['used_bytes'] == 64 * 8
        a0 = np.ascontiguousarray(ia)
        env._check(env.L.rs_step(env.h, a0.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
        h = hashlib.sha256()
        for i in range(steps):
            ag.step_resident(env)
            if i % 25 == 24:
                h.update(env.fetch()['actions'].tobytes())
            if i + 1 < steps:
                env.step_resident()
        ag.synchronize()
        sizes = ag.dictionary_sizes()
        co = b''.join(ag.learner(r, s)['coeff'].tobytes() for r in (0, 1000, 4095) for s in range(5))
        results.append((h.hexdigest(), sizes.tobytes(), co, int((sizes >= 192).sum()), ag.pool()['used_bytes']))
    assert results[0][3] >= 50, 'large dictionaries should have appeared: %d' % results[0][3]
    assert results[0][:3] == results[1][:3]
    assert results[0][4] == results[1][4]      # the same shells were taken, in whatever order
    env.close()
    ag.close()

"""Pin the KBRL oracle (oracle/kb_oracle.c) against teacher-forced sequences recorded from the
reference (fixtures G9, G10).  numpy evaluates k@coeff and Kinv@k through BLAS, whose summation
order is not reproducible, so kernel values are compared with a tolerance (1e-9 relative to the
scale of the terms, as SURVEY.md §8c G9 states) and every DECISION (predicted sign, branch of
Projectron.update, selected action, hit, security factor) must agree exactly -- the fixtures
contain no decision closer than that tolerance to its threshold, which the tests assert.
"""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from ranslice.config import make_config

TOL = 1e-9
ETA = 0.1


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


@pytest.mark.parametrize('tag,d', [('d11', 11), ('d4', 4)])
def test_g9_projectron_teacher_forced(golden_dir, tag, d):
    g = _load(golden_dir, 'g9_projectron')
    ag = po.OracleKBRL([d - 1], 200, [10], [3], capacity=1024)
    ag.set_tape(g[tag + '_ties'])
    xs, ys = g[tag + '_x'], g[tag + '_y']
    n_tie = 0
    for i in range(len(xs)):
        yp, f = ag.predict(0, xs[i])
        fr = g[tag + '_f'][i]
        assert f == pytest.approx(fr, rel=TOL, abs=TOL)
        if fr == 0.0 and g[tag + '_m'][i - 1 if i else 0] > 0 and i > 0:
            n_tie += 1
        if abs(fr) > TOL or fr == 0.0:
            assert yp == g[tag + '_ypred'][i], i
        br, dl = ag.update(0, xs[i], int(ys[i]))
        assert br == g[tag + '_branch'][i], i
        if br:
            assert dl == pytest.approx(g[tag + '_delta'][i], rel=TOL, abs=TOL)
            assert abs(g[tag + '_delta'][i] - ETA) > TOL  # the fixture has no borderline projection test
        assert ag.m(0) == g[tag + '_m'][i]
    assert n_tie >= 1, 'fixture should exercise the f == 0 tie-break (Q11)'
    assert ag.set_size(0) == int(g[tag + '_set_size'])
    np.testing.assert_array_equal(ag.landmarks(0), g[tag + '_landmarks'])
    np.testing.assert_allclose(ag.coeff(0), g[tag + '_coeff'], rtol=1e-8, atol=1e-9)
    kinv = g[tag + '_kinv']
    np.testing.assert_allclose(ag.kinv(0), kinv, rtol=1e-7, atol=1e-7 * np.abs(kinv).max())
    assert ag.error() == 0


def _dims(scenario):
    cfg = make_config(scenario)
    return [10] * cfg.n_embb + [3] * cfg.n_mmtc, cfg.n_prbs


@pytest.mark.parametrize('scenario', [0, 2])
def test_g10_kbrl_control_teacher_forced(golden_dir, scenario):
    g = _load(golden_dir, 'g10_kbrl_s%d' % scenario)
    dims, n_prbs = _dims(scenario)
    ag = po.OracleKBRL(dims, n_prbs, g['init_action'], g['init_sec'], accuracy_range=tuple(g['a_range']))
    ties = g['tape_val'][g['tape_kind'] == 6]
    ag.set_tape(ties)
    steps = len(g['state'])
    acc_i = 0
    for i in range(steps):
        hits = ag.update_control(g['state'][i], g['action_in'][i], g['labels'][i])
        assert (hits == g['hits'][i]).all(), i
        nxt = g['state'][i + 1] if i + 1 < steps else g['final_state']
        act, adj = ag.select_action(nxt)
        ag.adjusted = adj
        assert (act == g['action_out'][i]).all(), i
        assert adj == g['adjusted'][i]
        assert (ag.margins == g['margins'][i]).all()
        assert (ag.security_factors == g['security'][i]).all()
        sizes = [ag.set_size(s) if ag.m(s) else 0 for s in range(len(dims))]
        assert sizes == list(g['set_size'][i]), i
        if i % 10 == 9 or i == steps - 1:
            np.testing.assert_allclose(ag.accuracies, g['acc'][acc_i], rtol=0, atol=0)
            acc_i += 1
    for s in range(len(dims)):
        np.testing.assert_array_equal(ag.landmarks(s), g['landmarks%d' % s])
        np.testing.assert_allclose(ag.coeff(s), g['coeff%d' % s], rtol=1e-8, atol=1e-9)
    n_pred, n_mist = ag.stats()
    assert n_pred > 100 * steps // 10 and ag.error() == 0


@pytest.mark.parametrize('scenario', [0, 2])
def test_g10_closed_loop_env_plus_agent(golden_dir, scenario):
    """BASELINE config 1 (plumbing): oracle env on the reference's tape driven by the oracle agent
    reproduces the reference's closed-loop run (actions, rewards, violations) step for step."""
    g = _load(golden_dir, 'g10_kbrl_s%d' % scenario)
    fad = _load(golden_dir, 'fading_small')
    dims, n_prbs = _dims(scenario)
    kind, val = g['tape_kind'], g['tape_val']
    env = po.OracleEnv(make_config(scenario), [fad['t0'], fad['t1'], fad['t2']])
    env.set_tape(kind[kind != 6], val[kind != 6])
    ag = po.OracleKBRL(dims, n_prbs, g['init_action'], g['init_sec'], accuracy_range=tuple(g['a_range']))
    ag.set_tape(val[kind == 6])
    state = env.reset()
    action = g['init_action'].copy()
    for i in range(len(g['state'])):
        out = env.step(action)
        assert out['reward'] == g['reward'][i] and int(out['violations'].sum()) == g['violation'][i], i
        ag.update_control(state, action, out['labels'])
        action, ag.adjusted = ag.select_action(out['obs'])
        state = out['obs']
        assert (action == g['action_out'][i]).all(), i
    assert env.tape_pos() == int((kind != 6).sum())


def test_g11_results_schema(golden_dir):
    """keys / dtypes / shapes of KBRL_Control.run's result dict (kbrl_control.py:148-155)"""
    g = _load(golden_dir, 'g11_results_schema')
    want = {'reward': ('float64', 1), 'resources': ('int16', 1), 'hits': ('int16', 2), 'adjusted': ('int16', 1),
            'SLA': ('int16', 1), 'violation': ('int16', 1)}
    assert sorted(k[4:] for k in g.files) == sorted(want)
    for k, (dt, nd) in want.items():
        assert str(g['key_' + k].dtype) == dt and g['key_' + k].ndim == nd


def check_kinv_digest(kinv, g, rtol):
    """compare a Kinv with what the long fixtures keep of the reference's (every 16th row, the diagonal, Kinv @ four
    probe vectors: tools/gen_golden_kbrl.py:_kinv_digest)"""
    scale = np.abs(g['kinv_rows']).max()
    np.testing.assert_allclose(kinv[::16], g['kinv_rows'], rtol=rtol, atol=rtol * scale)
    np.testing.assert_allclose(np.diag(kinv), g['kinv_diag'], rtol=rtol, atol=rtol * scale)
    kp = kinv @ g['kinv_probes']
    np.testing.assert_allclose(kp, g['kinv_kp'], rtol=rtol, atol=rtol * np.abs(g['kinv_kp']).max())


def test_g14_projectron_long(golden_dir):
    """Projectron far past G9's sizes: 7,000 recorded samples that take the reference's dictionary to 790 landmarks,
    with projections still happening above 600.  Tolerances as G9; the conditioning of Kinv degrades
    with 1/delta, hence 1e-6 on its entries."""
    g = _load(golden_dir, 'g14_projectron_long')
    ag = po.OracleKBRL([10], 200, [10], [3], capacity=1024)
    ag.set_tape(g['ties'])
    xs, ys = g['x'], g['y']
    worst_f = 0.0
    for i in range(len(xs)):
        yp, f = ag.predict(0, xs[i])
        fr = g['f'][i]
        assert f == pytest.approx(fr, rel=1e-8, abs=TOL), i
        worst_f = max(worst_f, abs(f - fr))
        if abs(fr) > TOL:
            assert yp == g['ypred'][i], i
        br, dl = ag.update(0, xs[i], int(ys[i]))
        assert br == g['branch'][i], i
        if br:
            assert dl == pytest.approx(g['delta'][i], rel=1e-7, abs=1e-9), i
            assert abs(g['delta'][i] - ETA) > 1e-6
        assert ag.m(0) == g['m'][i], i
    m = ag.m(0)
    assert m == len(g['landmarks']) and m > 600
    assert ((g['branch'] == 1) & (g['m'] > 600)).sum() >= 20   # projections against a large dictionary are exercised
    np.testing.assert_array_equal(ag.landmarks(0), g['landmarks'])
    np.testing.assert_allclose(ag.coeff(0), g['coeff'], rtol=1e-7, atol=1e-8)
    check_kinv_digest(ag.kinv(0), g, 1e-6)
    assert ag.error() == 0


def _g17_samples(g):
    return np.concatenate([g['state'].astype(np.float64), (g['a'].astype(np.float64) / 200)[:, None]], axis=1), g['y']


def test_g17_projectron_to_3000_landmarks(golden_dir):
    """G17 (round 4): the REFERENCE's Projectron teacher-forced to 3,000+ landmarks -- the sizes its 50,400-step runs end at --
    replayed by the oracle: f within 1e-8, predicted sign / branch / dictionary size exact at every one of 12,400 samples,
    delta 1e-6 relative (1 - k.Kinv k against a Kinv that has been through three thousand rank-1 updates in numpy's BLAS
    order there and in index order here), final coefficients 1e-6, Kinv (diagonal, every 256th row, four probe products)
    1e-6 of its scale."""
    g = _load(golden_dir, 'g17_projectron_3000')
    ag = po.OracleKBRL([10], 200, [10], [3], capacity=4096)
    ag.set_tape(g['ties'])
    xs, ys = _g17_samples(g)
    for i in range(len(xs)):
        yp, f = ag.predict(0, xs[i])
        fr = g['f'][i]
        assert f == pytest.approx(fr, rel=1e-8, abs=TOL), i
        if abs(fr) > 1e-7:
            assert yp == g['ypred'][i], i
        br, dl = ag.update(0, xs[i], int(ys[i]))
        assert br == g['branch'][i], (i, dl, g['delta'][i])
        if br:
            assert dl == pytest.approx(g['delta'][i], rel=1e-6, abs=1e-9), i
        assert ag.m(0) == g['m'][i], i
    m = ag.m(0)
    assert m == g['m'][-1] >= 3000
    assert ((g['branch'] == 1) & (g['m'] > 2000)).sum() >= 20   # projections onto a dictionary of thousands are exercised
    np.testing.assert_array_equal(ag.landmarks(0), xs[g['branch'] == 2])
    np.testing.assert_allclose(ag.coeff(0), g['coeff'], rtol=1e-6, atol=1e-8)
    kinv = ag.kinv(0)
    scale = np.abs(g['kinv_diag']).max()
    np.testing.assert_allclose(kinv[::256], g['kinv_rows'], rtol=1e-6, atol=1e-6 * scale)
    np.testing.assert_allclose(np.diag(kinv), g['kinv_diag'], rtol=1e-6, atol=1e-6 * scale)
    np.testing.assert_allclose(kinv @ g['kinv_probes'], g['kinv_kp'], rtol=1e-6, atol=1e-6 * np.abs(g['kinv_kp']).max())
    assert ag.error() == 0


@pytest.mark.parametrize('name,min_m', [('g15_kbrl_long_s0', 200), ('g16_kbrl_long_tdl_s0', 0)])
def test_g15_g16_kbrl_control_long(golden_dir, name, min_m):
    """KBRL_Control teacher-forced over 2,200 recorded steps of scenario_0 -- G15 on the first trace profile
    (dictionaries of several hundred landmarks), G16 on the tapped-delay-line traces: every hit, action, adjusted
    flag, margin, security factor and dictionary size of the reference"""
    g = _load(golden_dir, name)
    dims, n_prbs = _dims(0)
    ag = po.OracleKBRL(dims, n_prbs, g['init_action'], g['init_sec'], accuracy_range=tuple(g['a_range']), capacity=2048)
    ag.set_tape(g['ties'])
    steps = len(g['state'])
    assert steps >= 2000
    for i in range(steps):
        hits = ag.update_control(g['state'][i], g['action_in'][i], g['labels'][i])
        assert (hits == g['hits'][i]).all(), i
        nxt = g['state'][i + 1] if i + 1 < steps else g['final_state']
        act, adj = ag.select_action(nxt)
        ag.adjusted = adj
        assert (act == g['action_out'][i]).all(), i
        assert adj == g['adjusted'][i]
        assert (ag.margins == g['margins'][i]).all() and (ag.security_factors == g['security'][i]).all(), i
        sizes = [ag.set_size(s) if ag.m(s) else 0 for s in range(len(dims))]
        assert sizes == list(g['set_size'][i]), i
    np.testing.assert_allclose(ag.accuracies, g['acc'][-1], rtol=1e-12, atol=0)
    for s in range(len(dims)):
        np.testing.assert_array_equal(ag.landmarks(s), g['landmarks%d' % s])
        np.testing.assert_allclose(ag.coeff(s), g['coeff%d' % s], rtol=1e-6, atol=1e-8)
    assert max(ag.m(s) for s in range(len(dims))) >= min_m and ag.error() == 0

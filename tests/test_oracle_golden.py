"""Pin the CPU oracle (oracle/rs_oracle.c) against golden vectors produced by the REFERENCE
implementation (tools/gen_golden.py imports /root/reference in the build container and records
inputs, the full random tape and outputs).  Integers must match exactly; probabilities within
1e-12 relative (numpy's SIMD exp/log vs the oracle's deterministic ones).
"""
import glob
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from ranslice.config import make_config

PROB_RTOL = 1e-12


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def test_detmath_accuracy():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-40, 40, 4000), rng.uniform(-700, 700, 500), [0.0, -0.0, 1e-300, 709.7, -745.0]])
    got = np.array([po.lib().rso_exp(float(v)) for v in x])
    ref = np.exp(x)
    ok = ref > 1e-300
    assert np.max(np.abs(got[ok] / ref[ok] - 1.0)) < 4e-16
    y = np.concatenate([10.0 ** rng.uniform(-300, 300, 3000), rng.uniform(0.5, 2.0, 3000), [1.0, 5e-324, 2.0]])
    got = np.array([po.lib().rso_log(float(v)) for v in y])
    ref = np.log(y)
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)) < 6e-16 or np.max(np.abs(got - ref)) < 2e-16
    z = np.concatenate([rng.uniform(-1, 1, 4000), [1.0, -1.0, 0.0, 0.5, -0.5, 0.999999999, -0.999999999]])
    got = np.array([po.lib().rso_acos(float(v)) for v in z])
    assert np.max(np.abs(got - np.arccos(z))) < 1e-15
    assert po.lib().rso_exp(1000.0) == np.inf and po.lib().rso_exp(-1000.0) == 0.0
    assert po.lib().rso_log(0.0) == -np.inf and np.isnan(po.lib().rso_log(-1.0))


def test_pairwise_matches_numpy():
    rng = np.random.default_rng(5)
    for _ in range(600):
        n = int(rng.integers(1, 420))
        a = rng.normal(0, 30, size=n)
        assert po.pairwise_sum(a) == np.sum(a)
        assert po.pairwise_sum(a) / n == np.mean(a)


def test_g1_mcs_lookup(golden_dir):
    g = _load(golden_dir, 'g1_mcs')
    cfg = make_config(0)
    A, B = po.mcs_factors()
    assert abs(A - float(g['A'])) < 1e-13 and abs(B - float(g['B'])) < 1e-14
    for e, m, r in zip(g['e_snr'], g['mcs'], g['rate']):
        assert po.mcs_lookup(cfg, int(e)) == (int(m), int(r))
    # known answers quoted in SURVEY.md §8a A.2
    assert po.mcs_lookup(cfg, -3) == (0, 63) and po.mcs_lookup(cfg, 0) == (2, 126)
    assert po.mcs_lookup(cfg, 10) == (13, 474) and po.mcs_lookup(cfg, 25) == (25, 853)


def test_g2_response(golden_dir):
    g = _load(golden_dir, 'g2_response')
    cfg = make_config(0)
    off = 0
    for n, m, p in zip(g['length'], g['mcs'], g['p']):
        snr = g['snr'][off:off + n]
        off += n
        got = po.response(cfg, int(m), snr)
        assert got == pytest.approx(p, rel=PROB_RTOL, abs=1e-300)
    assert po.response(cfg, 10, [8.0, 8, 8, 8]) == pytest.approx(0.9998334419352227, rel=1e-12)
    assert po.response(cfg, 10, [5.0, 7, 9, 11]) == pytest.approx(0.99913825526293, rel=1e-12)


def test_g3_pf_allocate(golden_dir):
    g = _load(golden_dir, 'g3_pf')
    cfg = make_config(0)
    for k in range(int(g['n_cases'])):
        c = {key: g['c%d_%s' % (k, key)] for key in ('th', 'queue', 'e_snr', 'snr', 'prbs', 'bits', 'p')}
        prbs, bits, p = po.pf_allocate(cfg, c['th'], c['queue'], c['e_snr'], c['snr'])
        assert (prbs == c['prbs']).all(), k
        assert (bits == c['bits']).all(), k
        np.testing.assert_allclose(p, c['p'], rtol=PROB_RTOL, atol=0)


def test_g4_vbr_source(golden_dir):
    g = _load(golden_dir, 'g4_vbr')
    cfg = make_config(0)
    for r in (0, 1):
        bits, used = po.vbr_source(cfg, g['r%d_gexp' % r], len(g['r%d_bits' % r]))
        assert used == len(g['r%d_gexp' % r])
        assert (bits == g['r%d_bits' % r]).all()
    assert g['r1_bits'][-1] > 0  # the forced-zero run really has immortal bursts (Q5)


def test_g6_macro_cell(golden_dir):
    g = _load(golden_dir, 'g6_macro_cell')
    cfgs = [make_config(0, propagation_type='macro_cell_urban_2GHz'), make_config(0, propagation_type='macro_cell_rural')]
    for uv, used, normal, sinr, model in zip(g['uv'], g['used'], g['normal'], g['sinr'], g['model']):
        got, n = po.macro_cell(cfgs[int(model)], uv, normal)
        assert n == used
        assert got == pytest.approx(sinr, rel=1e-13, abs=1e-12)


def _g7_files(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, 'g7_*.npz')))


def _cfg_for(g):
    kw = {}
    cfg = make_config(int(g['scenario']), **kw)
    if int(g['churn']):
        cfg.cbr_lambda, cfg.cbr_t_mean = 2.0 / 1.2, 0.6
        cfg.vbr_lambda, cfg.vbr_t_mean = 5.0 / 1.2, 0.6
        cfg.vbr_b_size, cfg.vbr_b_rate = 40, 12
    return cfg


@pytest.mark.parametrize('path', _g7_files(os.path.join(os.path.dirname(__file__), 'golden')),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_g7_full_step(path, golden_dir):
    """RanSlice.step replayed on the reference's own random tape: obs bits, reward, SLA labels,
    violations, info accumulators and every per-slot per-UE allocation must agree."""
    g = np.load(path)
    fad = _load(golden_dir, 'fading_small')
    cfg = _cfg_for(g)
    env = po.OracleEnv(cfg, [fad['t0'], fad['t1'], fad['t2']])
    env.set_tape(g['tape_kind'], g['tape_val'])
    obs0 = env.reset()
    assert (obs0 == g['obs0']).all()
    slot_i = 0
    ue_off = 0
    n_embb = cfg.n_embb
    arrivals, departures = [], []        # G5: (slot since reset, slice, type, serial) / (slot, slice, serial)
    present = [dict() for _ in range(n_embb)]
    q_i = q_off = 0                       # G8 cursors
    for i, act in enumerate(g['actions']):
        out = env.step(act, trace=True)
        assert out['obs'].tobytes() == g['obs'][i].tobytes(), 'obs bits differ at step %d' % i
        assert out['reward'] == g['reward'][i]
        assert (out['labels'] == g['labels'][i]).all()
        assert (out['violations'] == g['violations'][i]).all()
        assert (out['info'] == g['info'][i]).all(), 'info accumulators differ at step %d' % i
        tr = out['trace']
        # reference hook order: slot-major, eMBB slices in order
        for t in range(cfg.slots_per_step):
            for s in range(n_embb):
                n = int(g['slot_n_ue'][slot_i])
                slot_i += 1
                rec = tr[s, t]
                assert int((rec['serial'] > 0).sum()) == n
                gi = g['slot_ue_int'][ue_off:ue_off + n]
                gf = g['slot_ue_f64'][ue_off:ue_off + n]
                ue_off += n
                r = rec[:n]
                assert (r['type'] == gi[:, 0]).all()
                assert (r['e_snr'] == gi[:, 1]).all()
                assert (r['prbs'] == gi[:, 2]).all()
                assert (r['bits'] == gi[:, 3]).all()
                assert (r['queue'] == gf[:, 0]).all()
                assert (r['th'] == gf[:, 1]).all()
                # the trace reports p of THIS slot's allocation (0 when the slice was not scheduled);
                # the reference attribute is sticky, so compare where an allocation happened
                hit = r['p'] != 0
                np.testing.assert_allclose(r['p'][hit], gf[hit, 2], rtol=PROB_RTOL, atol=0)
                # G5: arrivals = serials first seen in this slot, departures = serials gone since the last slot
                now = {int(x['serial']): int(x['type']) for x in r}
                gslot = i * cfg.slots_per_step + t
                for ser in sorted(set(now) - set(present[s])):
                    arrivals.append((gslot, s, now[ser], ser))
                for ser in present[s]:           # dict order = UE list order, the order departures() walks
                    if ser not in now:
                        departures.append((gslot, s, ser))
                present[s] = now
        # G8: SliceL1mMTC FIFO (time, remaining repetitions, arrival times) after the step
        for s in range(cfg.n_mmtc):
            tm, rep, start = env.mtc_queue(s)
            n = int(g['g8_n_users'][q_i])
            assert tm == int(g['g8_time'][q_i]) and len(rep) == n, 'mMTC queue length differs at step %d' % i
            assert (rep == g['g8_repetitions'][q_off:q_off + n]).all()
            assert (start == g['g8_t_start'][q_off:q_off + n]).all()
            q_i += 1
            q_off += n
    assert env.tape_pos() == len(g['tape_kind']), 'oracle consumed a different number of draws'
    assert slot_i == len(g['slot_n_ue'])
    assert q_i == len(g['g8_n_users']) and q_off == len(g['g8_repetitions'])
    # G5: SliceRANeMBB.slot's arrivals (with their CAC decisions) and departures, slot by slot
    ga = [tuple(int(v) for v in row) for row in g['g5_arrivals']]
    gd = [tuple(int(v) for v in row) for row in g['g5_departures']]
    assert sorted(arrivals) == sorted(ga), 'arrival slots / types / serials differ'
    assert sorted(departures) == sorted(gd), 'departure slots / serials differ'
    assert len(ga) > 0


def _g13_files(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, 'g13_mux_*.npz')))


@pytest.mark.parametrize('path', _g13_files(os.path.join(os.path.dirname(__file__), 'golden')),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_g13_multiplexed_l1(path, golden_dir):
    """create_env(..., L1_level=False) (scenario_creator.py:168-177) replayed on the reference's own tape: all eMBB
    RAN slices under ONE L1 slice / PF scheduler, all mMTC RAN slices in ONE FIFO.  One action entry per L1 slice;
    observation and info per RAN slice; per-slot per-UE records of the shared UE list incl. the UE's RAN slice."""
    g = np.load(path)
    assert int(g['l1_level']) == 0
    fad = _load(golden_dir, 'fading_small')
    cfg = _cfg_for(g)
    cfg.l1_multiplex = 1
    env = po.OracleEnv(cfg, [fad['t0'], fad['t1'], fad['t2']])
    assert env.n_slices == (cfg.n_embb > 0) + (cfg.n_mmtc > 0) == g['actions'].shape[1]
    env.set_tape(g['tape_kind'], g['tape_val'])
    obs0 = env.reset()
    assert (obs0 == g['obs0']).all()
    slot_i = ue_off = 0
    arrivals, departures = [], []
    present = {}
    q_i = q_off = 0
    max_ue_seen = 0
    for i, act in enumerate(g['actions']):
        out = env.step(act, trace=True)
        assert out['obs'].tobytes() == g['obs'][i].tobytes(), 'obs bits differ at step %d' % i
        assert out['reward'] == g['reward'][i]
        assert (out['labels'] == g['labels'][i]).all()
        assert (out['violations'] == g['violations'][i]).all()
        assert (out['info'] == g['info'][i]).all(), 'info accumulators differ at step %d' % i
        tr = out['trace']
        for t in range(cfg.slots_per_step if cfg.n_embb else 0):
            n = int(g['slot_n_ue'][slot_i])
            slot_i += 1
            max_ue_seen = max(max_ue_seen, n)
            rec = tr[0, t]
            assert int((rec['serial'] > 0).sum()) == n
            gi = g['slot_ue_int'][ue_off:ue_off + n]
            gf = g['slot_ue_f64'][ue_off:ue_off + n]
            gr = g['slot_ue_ran'][ue_off:ue_off + n]
            ue_off += n
            r = rec[:n]
            assert ((r['type'] & 0xff) == gi[:, 0]).all() and ((r['type'] >> 8) == gr).all()
            assert (r['e_snr'] == gi[:, 1]).all()
            assert (r['prbs'] == gi[:, 2]).all()
            assert (r['bits'] == gi[:, 3]).all()
            assert (r['queue'] == gf[:, 0]).all()
            assert (r['th'] == gf[:, 1]).all()
            hit = r['p'] != 0
            np.testing.assert_allclose(r['p'][hit], gf[hit, 2], rtol=PROB_RTOL, atol=0)
            now = {(int(x['type']) >> 8, int(x['serial'])): int(x['type']) & 0xff for x in r}
            gslot = i * cfg.slots_per_step + t
            for key in sorted(set(now) - set(present)):
                arrivals.append((gslot, key[0], now[key], key[1]))
            for key in present:
                if key not in now:
                    departures.append((gslot, key[0], key[1]))
            present = now
        if cfg.n_mmtc:
            tm, rep, start = env.mtc_queue(0)
            n = int(g['g8_n_users'][q_i])
            assert tm == int(g['g8_time'][q_i]) and len(rep) == n, 'mMTC queue length differs at step %d' % i
            assert (rep == g['g8_repetitions'][q_off:q_off + n]).all()
            assert (start == g['g8_t_start'][q_off:q_off + n]).all()
            q_i += 1
            q_off += n
    assert env.tape_pos() == len(g['tape_kind']), 'oracle consumed a different number of draws'
    assert slot_i == len(g['slot_n_ue']) and q_i == len(g['g8_n_users'])
    ga = [tuple(int(v) for v in row) for row in g['g5_arrivals']]
    gd = [tuple(int(v) for v in row) for row in g['g5_departures']]
    assert sorted(arrivals) == sorted(ga) and sorted(departures) == sorted(gd)
    if cfg.n_embb > 1:
        assert len(set(g['slot_ue_ran'].tolist())) > 1, 'fixture should mix RAN slices in one UE list'

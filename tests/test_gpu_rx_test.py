"""The reception test by guard band (rs_embb.hip: fast_sigmoid, rs_api.hip: rx_fast_setup).

`rng.random() < mcs_codeset.response(mcs, snr)` (reference slice_l1.py:219-224, channel_models.py:297-313) is decided by a
float32 evaluation of both sides of the equivalent comparison S > S*(u) unless they lie within a guard band of each other;
inside the band -- and in the tracing instances, which report the probability -- the f64 probability is formed as before.  The
decision must be the exact comparison's in every case: the production path (short test), a run that evaluates every UE exactly
and runs with the band widened 2,000 and 50,000 times (a third / nearly all of the UEs inside it) must produce identical outputs.

The channel estimates round(mean(snr)) (slice_ran.py:43-45, channel_models.py:171-191) are decided the same way from per-column
prefix sums (two loads per estimate) unless the mean lies within 1e-9 of a half-integer; the same test runs them all through
numpy's pairwise sum (RANSLICE_EST_EXACT) and with a band of 0.2 (two estimates in five by the pairwise sum).
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from ranslice.config import make_config
from ranslice.sharding import replica_seed

pytestmark = pytest.mark.gpu


def _fading(golden_dir):
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    return [g['t0'], g['t1'], g['t2']]


def _churn(cfg):
    cfg.cbr_lambda, cfg.cbr_t_mean = 2.0 / 1.2, 0.6
    cfg.vbr_lambda, cfg.vbr_t_mean = 5.0 / 1.2, 0.6
    cfg.vbr_b_size, cfg.vbr_b_rate = 40, 12
    return cfg


def _digest(env, steps, wide):
    from test_gpu_parity import _wide_actions
    h = hashlib.sha256()
    rng = np.random.default_rng(5)
    for i in range(steps):
        if wide and i % 2:
            acts = _wide_actions(rng, env.n_envs, 5, 200, i)
            env.step(acts)
        else:
            env.random_actions(2024, i)
            env.step_resident()
        f = env.fetch()
        for key in ('obs', 'reward', 'labels', 'violations'):
            h.update(f[key].tobytes())
        h.update(env.l1_info().tobytes())
    return h.hexdigest()


@pytest.mark.parametrize('wide', [False, True])
def test_short_reception_test_equals_exact_probability(golden_dir, monkeypatch, wide):
    from ranslice.vec_env import VecRanSlice
    fading = _fading(golden_dir)
    N = 1536
    monkeypatch.setenv('RANSLICE_DEV_BUILD', '1')   # the knobs are read by the test build only (ranslice._lib)
    digests, stats = {}, {}
    for name, var, val in (('short', None, None), ('exact', 'RANSLICE_RX_EXACT', '1'), ('band x 2e3', 'RANSLICE_RX_BAND_SCALE', '2000'),
                           ('band x 5e4', 'RANSLICE_RX_BAND_SCALE', '50000'), ('estimates exact', 'RANSLICE_EST_EXACT', '1'),
                           ('estimates band 0.2', 'RANSLICE_EST_BAND', '0.2')):
        for v in ('RANSLICE_RX_EXACT', 'RANSLICE_RX_BAND_SCALE', 'RANSLICE_EST_EXACT', 'RANSLICE_EST_BAND'):
            monkeypatch.delenv(v, raising=False)
        if var:
            monkeypatch.setenv(var, val)
        env = VecRanSlice(n_envs=N, cfg=_churn(make_config(0, n_envs=N)), fading=fading, seed=314)
        env.reset()
        digests[name] = _digest(env, 24, wide)
        stats[name] = env.rx_stats()
        env.close()
    assert len(set(digests.values())) == 1, digests
    tests, exact, avail = stats['short']
    assert avail and tests > 100000, stats
    assert exact < 2e-3 * tests, stats          # expected: a few 1e-4 (band 6e-6 per RB against a threshold spread over ~0.05)
    assert stats['exact'][1] == stats['exact'][0] == tests and not stats['exact'][2], stats
    assert stats['band x 2e3'][1] > 0.05 * tests and stats['band x 5e4'][1] > 0.5 * tests, stats


def test_short_reception_test_is_refused_for_slopes_outside_its_proof(golden_dir):
    """mcsA / k below 1 is outside the range the band was derived for (rs_api.hip: rx_fast_setup): the handle evaluates every UE
    exactly and says so; results still equal the oracle's."""
    from ranslice.vec_env import VecRanSlice
    fading = _fading(golden_dir)

    def cfg_for(n):
        c = _churn(make_config(0, n_envs=n))
        c.mi_k[1] = 9.5   # steeper than mcsA = 8
        return c
    n = 24
    env = VecRanSlice(n_envs=n, cfg=cfg_for(n), fading=fading, seed=99)
    env.reset()
    oracles = []
    for r in range(n):
        o = po.OracleEnv(cfg_for(1), fading)
        o.set_seed(replica_seed(99, r))
        o.reset()
        oracles.append(o)
    rng = np.random.default_rng(1)
    for i in range(8):
        acts = rng.multinomial(200, [1 / 5.0] * 5, size=n).astype(np.int32)
        obs, rew, _, info = env.step(acts)
        for r, o in enumerate(oracles):
            out = o.step(acts[r])
            assert obs[r].tobytes() == out['obs'].tobytes() and rew[r] == out['reward'], (i, r)
            assert (info['violations'][r] == out['violations']).all(), (i, r)
    tests, exact, avail = env.rx_stats()
    assert not avail and exact == tests > 0
    env.close()


def test_short_paths_step_aside_for_samples_beyond_their_bounds(golden_dir):
    """The bands of both short paths were derived for samples of at most 1e3 in magnitude (rs_api.hip: upload_fading).  Traces scaled
    beyond that load as before, the handle reports that the short reception test is off, and the results still equal the oracle's
    (every estimate by the pairwise sum, every reception by the f64 probability)."""
    from ranslice.vec_env import VecRanSlice
    fading = [np.asarray(t) * 40.0 for t in _fading(golden_dir)]     # |samples| up to ~1,600
    assert max(float(np.nanmax(np.abs(t))) for t in fading) > 1.0e3
    n = 16
    env = VecRanSlice(n_envs=n, cfg=_churn(make_config(0, n_envs=n)), fading=fading, seed=5)
    env.reset()
    oracles = []
    for r in range(n):
        o = po.OracleEnv(_churn(make_config(0)), fading)
        o.set_seed(replica_seed(5, r))
        o.reset()
        oracles.append(o)
    rng = np.random.default_rng(2)
    for i in range(6):
        acts = rng.multinomial(200, [1 / 5.0] * 5, size=n).astype(np.int32)
        obs, rew, _, info = env.step(acts)
        for r, o in enumerate(oracles):
            out = o.step(acts[r])
            assert obs[r].tobytes() == out['obs'].tobytes() and rew[r] == out['reward'], (i, r)
    tests, exact, avail = env.rx_stats()
    assert not avail and exact == tests
    env.close()

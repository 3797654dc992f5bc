"""Real-trace ingestion: the reference's headerless CSV layout and ns-3's .fad layout, NaN fields
included, round-trip through the loaders; on the GPU the loaded tables drive the simulator exactly like in-memory
ones (NaN columns are skipped as channel_models.py:175-189 does)."""
import os

import numpy as np
import pytest

from ranslice.fading import load_csv, load_fad, load_traces, synth_fading


def _tables():
    return [synth_fading(t, 96, seed=5, nan_cols=(7, 40) if t == 1 else ()) for t in range(3)]


def _write(tmp_path, tabs):
    paths = []
    for t, tab in enumerate(tabs):
        if t == 2:   # ns-3 .fad: RB-major, whitespace separated
            p = tmp_path / ('fading_trace_%d.fad' % t)
            with open(p, 'w') as f:
                for row in tab:
                    f.write(' '.join('nan' if v != v else repr(float(v)) for v in row) + ' \n')
        else:        # the reference's CSV: pd.read_csv(header=None) layout; NaN as an empty field or 'nan'
            p = tmp_path / ('fading_trace_%d.csv' % t)
            with open(p, 'w') as f:
                for r, row in enumerate(tab):
                    f.write(','.join(('' if r % 2 else 'nan') if v != v else repr(float(v)) for v in row) + '\n')
        paths.append(str(p))
    return paths


def test_csv_and_fad_round_trip(tmp_path):
    tabs = _tables()
    paths = _write(tmp_path, tabs)
    got = load_traces(paths)
    for a, b in zip(tabs, got):
        assert a.shape == b.shape == (100, 96)
        assert (np.isnan(a) == np.isnan(b)).all()
        assert np.nan_to_num(a).tobytes() == np.nan_to_num(b).tobytes()
    assert np.isnan(got[1]).sum() == 2
    assert load_csv(paths[0]).dtype == np.float64 and load_fad(paths[2]).flags['C_CONTIGUOUS']
    with pytest.raises(ValueError):
        bad = tmp_path / 'bad.fad'
        bad.write_text('1.0 2.0 3.0')
        load_fad(str(bad))


@pytest.mark.gpu
def test_loaded_traces_drive_the_simulator(tmp_path):
    """files -> loaders -> rs_load_fading: bit-exact against the oracle fed the in-memory tables, through NaN columns"""
    from oracle import pyoracle as po
    from ranslice.config import make_config
    from ranslice.sharding import replica_seed, replica_seeds  # noqa: F401
    from ranslice.vec_env import VecRanSlice
    tabs = _tables()
    loaded = load_traces(_write(tmp_path, tabs))
    n = 12
    cfg = make_config(0, n_envs=n)
    cfg.cbr_lambda, cfg.cbr_t_mean, cfg.vbr_lambda, cfg.vbr_t_mean = 2.0 / 1.2, 0.6, 5.0 / 1.2, 0.6
    env = VecRanSlice(n_envs=n, cfg=cfg, fading=loaded, seed=31)
    env.reset()
    ocfg = make_config(0, n_envs=1)
    ocfg.cbr_lambda, ocfg.cbr_t_mean, ocfg.vbr_lambda, ocfg.vbr_t_mean = 2.0 / 1.2, 0.6, 5.0 / 1.2, 0.6
    oracles = []
    for r in range(n):
        o = po.OracleEnv(ocfg, tabs)
        o.set_seed(replica_seed(31, r))
        o.reset()
        oracles.append(o)
    rng = np.random.default_rng(2)
    for i in range(12):
        acts = rng.multinomial(200, [1 / 6.0] * 6, size=n)[:, :5].astype(np.int32)
        obs, rew, _, info = env.step(acts)
        l1 = env.l1_info()
        for r, o in enumerate(oracles):
            out = o.step(acts[r])
            assert obs[r].tobytes() == out['obs'].tobytes(), (i, r)
            assert l1[r].tobytes() == out['info'].tobytes()
    env.close()

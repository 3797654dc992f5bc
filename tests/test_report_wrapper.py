"""ReportWrapper-style normalised interface against vectors recorded through the
reference's wrapper.ReportWrapper (fixture G12)."""
import os

import numpy as np

from ranslice.report import normalise_obs, simplex_to_prbs


def test_g12_action_and_obs_mapping(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g12_report_wrapper.npz'))
    prbs = simplex_to_prbs(g['action'], 200, 5)
    assert (prbs == g['prbs']).all()
    assert (prbs.sum(axis=1) <= 200).all()
    out = normalise_obs(g['obs_in'])
    assert out.tobytes() == g['obs_out'].astype(out.dtype).tobytes()
    ints = np.array([[10, 20, 30, 40, 50]])
    assert (simplex_to_prbs(ints, 200, 5) == ints).all()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_vec_report_wrapper_on_the_batched_env(golden_dir, tmp_path):
    """VecReportWrapper (wrapper.py:71-134) driving the batched simulator: float action simplex in, normalised
    observations out, the reference's history npz (keys violation / reward / resources, int16 / float64 / int16,
    wrapper.py:120-123) written every control_steps, set_evaluation extending the arrays; the numbers are the
    simulator's (checked against the oracle replica by replica)."""
    from oracle import pyoracle as po
    from ranslice.config import make_config
    from ranslice.sharding import replica_seed
    from ranslice.report import VecReportWrapper, normalise_obs, simplex_to_prbs
    from ranslice.vec_env import VecRanSlice
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    fading = [g['t0'], g['t1'], g['t2']]
    n, steps = 8, 10
    env = VecRanSlice(n_envs=n, cfg=make_config(0, n_envs=n), fading=fading, seed=77)
    w = VecReportWrapper(env, steps=steps, control_steps=5, env_id=3, path=str(tmp_path) + '/')
    obs0 = w.reset()
    assert obs0.shape == (n, 50) and not obs0.any()
    oracles = []
    for r in range(n):
        o = po.OracleEnv(make_config(0, n_envs=1), fading)
        o.set_seed(replica_seed(77, r))
        o.reset()
        oracles.append(o)
    rng = np.random.default_rng(9)
    rewards = np.zeros((n, steps + 4))
    viol = np.zeros((n, steps + 4), dtype=np.int64)
    res = np.zeros((n, steps + 4), dtype=np.int64)

    def one(i):
        a = rng.random((n, 6)) - 0.2          # negative entries: abs() is taken (wrapper.py:78)
        obs, rew, done, info = w.step(a)
        prbs = simplex_to_prbs(a, 200, 5)
        assert info == {0: 0} and not done.any()
        assert obs.min() >= -1.0 and obs.max() <= 1.0
        for r, o in enumerate(oracles):
            out = o.step(prbs[r])
            assert obs[r].tobytes() == normalise_obs(out['obs']).tobytes(), (i, r)
            assert rew[r] == out['reward']
            rewards[r, i], viol[r, i], res[r, i] = out['reward'], out['violations'].sum(), prbs[r].sum()
    for i in range(steps):
        one(i)
    f = np.load(str(tmp_path) + '/history_3.npz')
    assert sorted(f.files) == ['resources', 'reward', 'violation']
    assert f['violation'].dtype == np.int16 and f['reward'].dtype == np.float64 and f['resources'].dtype == np.int16
    assert f['violation'].shape == (n, steps)
    assert (f['reward'] == rewards[:, :steps]).all() and (f['violation'] == viol[:, :steps]).all()
    assert (f['resources'] == res[:, :steps]).all()
    # evaluation phase: arrays grow, file name changes (wrapper.py:125-134)
    w.set_evaluation(4, change_name=True)
    assert w.violation_history.shape == (n, steps + 4) and w.step_counter == steps
    for i in range(steps, steps + 4):
        one(i)
    w.save_results()
    e = np.load(str(tmp_path) + '/evaluation_3.npz')
    assert e['reward'].shape == (n, steps + 4) and (e['reward'] == rewards).all() and (e['resources'] == res).all()
    env.close()

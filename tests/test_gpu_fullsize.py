"""GPU parity at BASELINE size: the path bench.py times (device-generated action script + resident step, 4096
replicas, 10,000-sample traces, default task order and kernel instance) against the CPU oracle, and the large
checks that used to be developer scripts (soak, run-to-run determinism).  Bit-exact, no tolerance.
"""
import hashlib
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pytest

from oracle import pyoracle as po
from ranslice.config import make_config
from ranslice.sharding import replica_seed, replica_seeds  # noqa: F401
from ranslice.fading import synth_fading

pytestmark = pytest.mark.gpu
# oracle workers are SPAWNED: forking a process whose HIP runtime is already initialised is not safe
import multiprocessing as _mp  # noqa: E402
_SPAWN = _mp.get_context('spawn')

ACTION_SEED = 2024  # bench.py's script seed
N_FULL = 4096
COLS = 10000

_FADING = {}


def _fading(cols=COLS):
    if cols not in _FADING:
        _FADING[cols] = [synth_fading(t, cols) for t in range(3)]
    return _FADING[cols]


@pytest.mark.parametrize('scenario', [0, 2])
def test_random_actions_script_matches_oracle(scenario):
    """rs_random_actions == rso_random_actions for every replica (bench.py's cpu_baseline claims the same script):
    the multinomial over S slices + "unused", sum <= n_prbs."""
    from ranslice.vec_env import VecRanSlice
    cfg = make_config(scenario, n_envs=N_FULL)
    env = VecRanSlice(n_envs=N_FULL, cfg=cfg, fading=_fading(512))
    env.reset()
    ocfg = make_config(scenario, n_envs=1)
    for step in (0, 1, 2, 1000, 123456789, 2 ** 33 + 5):
        env.random_actions(ACTION_SEED, step)
        got = env.fetch()['actions']
        assert (got >= 0).all() and (got.sum(axis=1) <= cfg.n_prbs).all()
        for r in range(N_FULL):
            want = po.random_actions(ocfg, ACTION_SEED, step, r)
            assert (got[r] == want).all(), (step, r, got[r], want)
    # a different seed gives a different script
    env.random_actions(ACTION_SEED + 1, 0)
    other = env.fetch()['actions']
    env.random_actions(ACTION_SEED, 0)
    assert (other != env.fetch()['actions']).any()
    env.close()


def _oracle_script_run(args):
    """one oracle replica driven by the bench action script; returns per-step outputs"""
    scenario, seed, replica, steps, cols = args
    cfg = make_config(scenario, n_envs=1)
    o = po.OracleEnv(cfg, [synth_fading(t, cols) for t in range(3)])
    o.set_seed(seed)
    o.reset()
    out = []
    for i in range(steps):
        a = po.random_actions(cfg, ACTION_SEED, i, replica)
        r = o.step(a)
        out.append((a, r['obs'].copy(), r['reward'], r['labels'].copy(), r['violations'].copy(), r['info'].copy()))
    return out


SAMPLE = [0, 1, 2, 3, 4, 5, 6, 7, 8, 15, 16, 17, 63, 64, 65, 255, 256, 511, 512, 1000, 1023, 1024, 1500, 2047, 2048,
          2049, 3000, 3071, 3072, 4000, 4094, 4095]


def test_bench_path_steady_state_vs_oracle():
    """The timed path of bench.py, oracle-checked: random_actions + step_resident on 4096 replicas of scenario_0 with
    the 10,000-column traces, default order and instance, 500 device steps (25 s of simulated time: arrivals,
    departures, VBR bursts, walker wraps, steady-state population); 32 sampled replicas are followed by oracle
    replicas stepped by the same script and compared at EVERY step: actions, observations (f32 bits), rewards,
    labels, violations and the ten info sums per slice (f64 bits)."""
    from ranslice.vec_env import VecRanSlice
    steps = 500
    cfg = make_config(0, n_envs=N_FULL)
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1), mp_context=_SPAWN) as ex:
        fut = ex.map(_oracle_script_run, [(0, replica_seed(0, r), r, steps, COLS) for r in SAMPLE], chunksize=1)
        env = VecRanSlice(n_envs=N_FULL, cfg=cfg, fading=_fading())
        env.reset()
        hip = []
        for i in range(steps):
            env.random_actions(ACTION_SEED, i)
            env.step_resident()
            f = env.fetch()
            l1 = env.l1_info()
            hip.append((f['actions'][SAMPLE].copy(), f['obs'][SAMPLE].copy(), f['reward'][SAMPLE].copy(),
                        f['labels'][SAMPLE].copy(), f['violations'][SAMPLE].copy(), l1[SAMPLE].copy()))
        c = env.counters()
        env.close()
        ref = list(fut)
    mean_ue = c[3] / (steps * cfg.slots_per_step * N_FULL * cfg.n_embb)
    assert mean_ue > 1.5, 'population should have grown past the post-reset state (%.2f UEs/slice)' % mean_ue
    for k, r in enumerate(SAMPLE):
        for i in range(steps):
            a, obs, rew, lab, viol, info = ref[k][i]
            h = hip[i]
            assert (h[0][k] == a).all(), ('actions', r, i)
            assert h[1][k].tobytes() == obs.tobytes(), ('obs', r, i)
            assert h[2][k] == rew, ('reward', r, i)
            assert (h[3][k] == lab).all() and (h[4][k] == viol).all(), ('labels', r, i)
            assert h[5][k].tobytes() == info.tobytes(), ('info', r, i)


def _oracle_script_chunk(args):
    """oracle replicas [lo, hi) driven by the bench action script: `skip` steps inside the C loop (rso_bench_run returns the
    running sum of the rewards, added in step order), then `tail` steps one at a time with every output"""
    scenario, lo, hi, skip, tail, cols = args
    cfg = make_config(scenario, n_envs=1)
    fading = [synth_fading(t, cols) for t in range(3)]
    out = []
    for r in range(lo, hi):
        o = po.OracleEnv(cfg, fading)
        o.set_seed(replica_seed(0, r))
        o.reset()
        acc = o.bench_run(ACTION_SEED, r, 0, skip) if skip else 0.0
        rows = []
        for i in range(skip, skip + tail):
            a = po.random_actions(cfg, ACTION_SEED, i, r)
            q = o.step(a)
            rows.append((a, q['obs'].copy(), q['reward'], q['labels'].copy(), q['violations'].copy(), q['info'].copy()))
        out.append((acc, rows))
    return out


# RANSLICE_SOAK_STEPS=k adds one more case: scenario 0 with k steps inside the oracle's loop (k = 3000: the stationary population of
# the bench, ~6 minutes of oracle time on 16 cores; profiles/r06_ac_soak_every_replica.txt is such a run)
_SOAK = [(0, int(os.environ['RANSLICE_SOAK_STEPS']), 4)] if os.environ.get('RANSLICE_SOAK_STEPS') else []


@pytest.mark.parametrize('scenario,skip,tail', [(0, 0, 12), (0, 150, 4), (2, 60, 4)] + _SOAK)
def test_every_replica_at_baseline_size_vs_oracle(scenario, skip, tail):
    """the bench path (device action script + resident step, 10,000-column traces, default order and instance)
    with EVERY one of the 4096 replicas followed by an oracle replica.  (0, 12): the first twelve steps, every output of every
    step.  (skip, 4): the oracle runs `skip` steps inside its C loop and hands back the sum of the rewards, which must equal,
    bit for bit, the sum of the device's per-step rewards added in the same order; then four steps with every output
    (actions, observations as f32 bits, rewards, labels, violations, the ten info sums per slice as f64 bits) -- the state after
    `skip` steps of arrivals, departures and bursts has to be right in every replica for those to match."""
    from ranslice.vec_env import VecRanSlice
    cfg = make_config(scenario, n_envs=N_FULL)
    workers = min(16, os.cpu_count() or 1)
    per = 64
    jobs = [(scenario, lo, min(lo + per, N_FULL), skip, tail, COLS) for lo in range(0, N_FULL, per)]
    with ProcessPoolExecutor(max_workers=workers, mp_context=_SPAWN) as ex:
        fut = ex.map(_oracle_script_chunk, jobs, chunksize=1)
        env = VecRanSlice(n_envs=N_FULL, cfg=cfg, fading=_fading())
        env.reset()
        acc = np.zeros(N_FULL)
        hip = []
        for i in range(skip + tail):
            env.random_actions(ACTION_SEED, i)
            env.step_resident()
            f = env.fetch()
            if i < skip:
                acc += f['reward']
            else:
                hip.append((f['actions'].copy(), f['obs'].copy(), f['reward'].copy(), f['labels'].copy(), f['violations'].copy(),
                            env.l1_info().copy()))
        env.close()
        bad = []
        r = 0
        for chunk in fut:
            for o_acc, rows in chunk:
                ok = o_acc == acc[r]
                for i, (a, obs, rew, lab, viol, info) in enumerate(rows):
                    h = hip[i]
                    ok = ok and (h[0][r] == a).all() and h[1][r].tobytes() == obs.tobytes() and h[2][r] == rew
                    ok = ok and (h[3][r] == lab).all() and (h[4][r] == viol).all() and h[5][r].tobytes() == info.tobytes()
                if not ok:
                    bad.append(r)
                r += 1
    assert r == N_FULL
    assert not bad, '%d of %d replicas differ from the oracle, first %s' % (len(bad), N_FULL, bad[:10])


def _oracle_acts_run(args):
    scenario, seed, acts, churn = args
    cfg = make_config(scenario, n_envs=1)
    if churn:
        from test_gpu_parity import _churn
        _churn(cfg)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fading_small.npz'))
    o = po.OracleEnv(cfg, [g['t0'], g['t1'], g['t2']])
    o.set_seed(seed)
    o.reset()
    out = []
    for a in acts:
        r = o.step(a)
        out.append((r['obs'].copy(), r['reward'], r['labels'].copy(), r['violations'].copy(), r['info'].copy()))
    return out


@pytest.mark.parametrize('scenario,n,steps', [(0, 1024, 40), (1, 512, 30)])
def test_soak_every_replica(golden_dir, scenario, n, steps):
    """(was tests/soak_check.py) production instance, high-churn traffic, NaN-column trace: EVERY replica of a
    batch large enough for the 16-lane instance and the cost-ranked task order, bit for bit against the oracle"""
    from ranslice.vec_env import VecRanSlice
    from test_gpu_parity import _actions, _churn
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    cfg = _churn(make_config(scenario, n_envs=n))
    seed0 = 9000 + scenario
    rng = np.random.default_rng(seed0)
    ns = cfg.n_embb + cfg.n_mmtc
    acts = [_actions(rng, n, ns, cfg.n_prbs, i) for i in range(steps)]
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1), mp_context=_SPAWN) as ex:
        fut = ex.map(_oracle_acts_run, [(scenario, replica_seed(seed0, r), [a[r] for a in acts], True) for r in range(n)],
                     chunksize=8)
        env = VecRanSlice(n_envs=n, cfg=cfg, fading=[g['t0'], g['t1'], g['t2']], seed=seed0)
        env.set_group_size(16)
        env.set_schedule_hint(0)
        env.reset()
        hip = []
        for a in acts:
            obs, rew, done, info = env.step(a)
            hip.append((obs, rew, info['SLA_labels'], info['violations'], env.l1_info()))
        env.close()
        bad = []
        for r, ref in enumerate(fut):
            for i in range(steps):
                o, w, lab, v, inf = ref[i]
                h = hip[i]
                if (h[0][r].tobytes() != o.tobytes() or h[1][r] != w or (h[2][r] != lab).any() or
                        (h[3][r] != v).any() or h[4][r].tobytes() != inf.tobytes()):
                    bad.append((r, i))
                    break
    assert not bad, 'mismatching (replica, first step): %s' % bad[:10]


@pytest.mark.parametrize('scenario', [0, 2])
def test_run_to_run_determinism_full_size(scenario):
    """(was tests/determinism_check.py) the resident loop at 4096 replicas twice from the same seeds: identical
    observations, rewards, labels, violations and info sums (a hash over 150 steps)."""
    from ranslice.vec_env import VecRanSlice
    hs = []
    for rep in range(2):
        env = VecRanSlice(n_envs=N_FULL, cfg=make_config(scenario, n_envs=N_FULL), fading=_fading())
        env.reset()
        h = hashlib.sha256()
        for i in range(150):
            env.random_actions(ACTION_SEED, i)
            env.step_resident()
            if i % 10 == 9:
                f = env.fetch()
                for k in ('obs', 'reward', 'labels', 'violations'):
                    h.update(f[k].tobytes())
                h.update(env.l1_info().tobytes())
        hs.append(h.hexdigest())
        env.close()
    assert hs[0] == hs[1]


@pytest.mark.parametrize('knobs', [{'RANSLICE_SNAKE': '0'}, {'RANSLICE_SNAKE_MASK': '0x16', 'RANSLICE_SNAKE_ROT': '0x08'},
                                   {'RANSLICE_KEY_W': '16,8,4,32'}, {'RANSLICE_ORDER': '0'}, {'RANSLICE_PAIR': '128'},
                                   {'RANSLICE_MIXED': '16'}, {'RANSLICE_MIXED': '4', 'RANSLICE_MIXED_UE': '4'},
                                   {'RANSLICE_MIXED_LIGHT': '96'}, {'RANSLICE_MIXED': '32', 'RANSLICE_MIXED_LIGHT': '128'}])
def test_results_do_not_depend_on_the_task_order(monkeypatch, knobs):
    """The launch order of the step tasks (cost key, heavy-led waves, serpentine dealing of the rounds: csrc/rs_order.hip) only
    decides which lanes simulate which (replica, slice): the same 4096-replica run with the order's knobs set differently
    gives the same observations, rewards, labels, violations and info sums, bit for bit."""
    from ranslice.vec_env import VecRanSlice
    hs = []
    for setting in ({}, knobs):
        for k in ('RANSLICE_SNAKE', 'RANSLICE_SNAKE_MASK', 'RANSLICE_SNAKE_ROT', 'RANSLICE_KEY_W', 'RANSLICE_ORDER', 'RANSLICE_PAIR',
                  'RANSLICE_MIXED', 'RANSLICE_MIXED_UE', 'RANSLICE_MIXED_LIGHT'):
            monkeypatch.delenv(k, raising=False)
        # the first run is the production library, the second the test build (the only one that reads the knobs)
        monkeypatch.setenv('RANSLICE_DEV_BUILD', '1' if setting else '0')
        for k, v in setting.items():
            monkeypatch.setenv(k, v)  # (read by rs_create)
        env = VecRanSlice(n_envs=N_FULL, cfg=make_config(0, n_envs=N_FULL), fading=_fading())
        env.reset()
        h = hashlib.sha256()
        for i in range(60):
            env.random_actions(ACTION_SEED, i)
            env.step_resident()
            if i % 10 == 9:
                f = env.fetch()
                for k in ('obs', 'reward', 'labels', 'violations'):
                    h.update(f[k].tobytes())
                h.update(env.l1_info().tobytes())
        hs.append(h.hexdigest())
        env.close()
    assert hs[0] == hs[1]


def test_graph_replay_equals_stepwise_full_size():
    """hipGraph replay of the scripted loop (BASELINE config 5's loop form) at 4096 replicas == step by step"""
    from ranslice.vec_env import VecRanSlice
    outs = []
    for graph in (False, True):
        env = VecRanSlice(n_envs=N_FULL, cfg=make_config(0, n_envs=N_FULL), fading=_fading())
        env.reset()
        env.run_random(ACTION_SEED, 0, 41, graph=graph)
        f = env.fetch()
        outs.append((f['obs'].tobytes(), f['reward'].tobytes(), f['violations'].tobytes(), env.l1_info().tobytes()))
        env.close()
    assert outs[0] == outs[1]


def test_config5_per_gpu_shard_graph_loop_vs_oracle():
    """BASELINE config 5's per-GPU shard (65,536 replicas / 8 GPUs = 8,192 per GPU) in its loop form: the scripted loop
    replayed from a captured hipGraph (rs_run_random(graph=True)), checked against the oracle on sampled replicas --
    among them the shard's last ones and replicas on both sides of the 4,096 boundary -- after 60, 61 and 121 steps:
    observations (f32 bits), rewards, labels, violations and the info sums (f64 bits).  The replica ids are those of
    rank 3 of the 8-GPU run (global ids 24,576..32,767), so the seeds are the ones that rank would use."""
    from ranslice.vec_env import VecRanSlice
    n, rank = 8192, 3
    first = rank * n
    sample = [0, 1, 17, 63, 64, 1000, 4095, 4096, 4097, 6000, 8190, 8191]
    marks = (60, 61, 121)
    cfg = make_config(0, n_envs=n)
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1), mp_context=_SPAWN) as ex:
        fut = ex.map(_oracle_script_run_ids, [(0, replica_seed(0, first + r), r, marks[-1], COLS, ACTION_SEED + rank)
                                              for r in sample], chunksize=1)
        env = VecRanSlice(n_envs=n, cfg=cfg, fading=_fading())
        env.reset(seeds=replica_seeds(0, first, n))
        hip = {}
        done = 0
        for mk in marks:
            env.run_random(ACTION_SEED + rank, done, mk - done, graph=True)   # bench.py's per-rank script seed
            done = mk
            f = env.fetch()
            hip[mk] = (f['actions'][sample].copy(), f['obs'][sample].copy(), f['reward'][sample].copy(),
                       f['labels'][sample].copy(), f['violations'][sample].copy(), env.l1_info()[sample].copy())
        env.close()
        ref = list(fut)
    for k, r in enumerate(sample):
        for mk in marks:
            a, obs, rew, lab, viol, info = ref[k][mk - 1]
            h = hip[mk]
            assert (h[0][k] == a).all(), ('actions', r, mk)
            assert h[1][k].tobytes() == obs.tobytes(), ('obs', r, mk)
            assert h[2][k] == rew and (h[3][k] == lab).all() and (h[4][k] == viol).all(), ('reward/labels', r, mk)
            assert h[5][k].tobytes() == info.tobytes(), ('info', r, mk)


def _oracle_churn_script_run(args):
    """one oracle replica of the high-churn configuration driven by the bench action script; outputs of the steps in `marks`"""
    seed, replica, marks = args
    from test_gpu_parity import _churn
    cfg = _churn(make_config(0, n_envs=1))
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fading_small.npz'))
    o = po.OracleEnv(cfg, [g['t0'], g['t1'], g['t2']])
    o.set_seed(seed)
    o.reset()
    out = {}
    for i in range(marks[-1]):
        a = po.random_actions(cfg, ACTION_SEED, i, replica)
        r = o.step(a)
        if i + 1 in marks:
            out[i + 1] = (a, r['obs'].copy(), r['reward'], r['labels'].copy(), r['violations'].copy(), r['info'].copy())
    return out


def test_split_step_at_its_batch_size_vs_oracle(golden_dir):
    """From 49,152 tasks (9,831 replicas of five slices) on the production library deals the cost ranking out to TWO launches side by side: the
    lightest 7/8 of the tasks eight to a wave on the 8-lane instance, the rest (every task with eight UEs or more among them) on
    the 16-lane one; a task that outgrows eight lanes inside a step is replayed by the 32-lane instance (rs_api.hip, round 5).
    16,384 replicas of the high-churn configuration (slices of 0 to 15 UEs, NaN-column trace), the scripted loop replayed from a
    captured hipGraph and stepped one by one: 96 sampled replicas against the oracle after 8, 9, 30 and 61 steps, bit for bit."""
    from ranslice.vec_env import VecRanSlice
    from test_gpu_parity import _churn
    n = 16384
    g = np.load(os.path.join(golden_dir, 'fading_small.npz'))
    sample = sorted(set([0, 1, 7, 8, 63, 64, 4095, 4096, 8191, 8192, 12287, 12288, 16382, 16383] + list(range(11, n, 199))))
    marks = (8, 9, 30, 61)
    cfg = _churn(make_config(0, n_envs=n))
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1), mp_context=_SPAWN) as ex:
        fut = ex.map(_oracle_churn_script_run, [(replica_seed(0, r), r, marks) for r in sample], chunksize=4)
        env = VecRanSlice(n_envs=n, cfg=cfg, fading=[g['t0'], g['t1'], g['t2']])
        env.reset()
        hip, done = {}, 0
        for mk in marks:
            if mk - done >= 3:
                env.run_random(ACTION_SEED, done, mk - done, graph=True)
            else:
                for i in range(done, mk):
                    env.random_actions(ACTION_SEED, i)
                    env.step_resident()
            done = mk
            f = env.fetch()
            hip[mk] = (f['actions'][sample].copy(), f['obs'][sample].copy(), f['reward'][sample].copy(),
                       f['labels'][sample].copy(), f['violations'][sample].copy(), env.l1_info()[sample].copy())
        c = env.counters()
        env.close()
        ref = list(fut)
    mean_ue = c[3] / (marks[-1] * cfg.slots_per_step * n * cfg.n_embb)
    assert mean_ue > 1.0, mean_ue
    bad = []
    for k, r in enumerate(sample):
        for mk in marks:
            a, obs, rew, lab, viol, info = ref[k][mk]
            h = hip[mk]
            if not ((h[0][k] == a).all() and h[1][k].tobytes() == obs.tobytes() and h[2][k] == rew and (h[3][k] == lab).all()
                    and (h[4][k] == viol).all() and h[5][k].tobytes() == info.tobytes()):
                bad.append((r, mk))
                break
    assert not bad, bad[:10]


def _oracle_script_run_ids(args):
    """as _oracle_script_run with the script seed given (a rank's own seed) and the replica's LOCAL index in it"""
    scenario, seed, replica, steps, cols, action_seed = args
    cfg = make_config(scenario, n_envs=1)
    o = po.OracleEnv(cfg, [synth_fading(t, cols) for t in range(3)])
    o.set_seed(seed)
    o.reset()
    out = []
    for i in range(steps):
        a = po.random_actions(cfg, action_seed, i, replica)
        r = o.step(a)
        out.append((a, r['obs'].copy(), r['reward'], r['labels'].copy(), r['violations'].copy(), r['info'].copy()))
    return out

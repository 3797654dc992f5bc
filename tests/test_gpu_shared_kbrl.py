"""Shared-dictionary KBRL (build-defined extension: no counterpart in the reference, parity unpinned).
Pinned by self-consistency, as SURVEY.md §8e asks: (1) with one replica it is the reference's sequential
algorithm (equals the per-replica agent, which is pinned to the reference's goldens); (2) the learned
dictionaries do not depend on how replicas are sharded over ranks."""
import os

import numpy as np
import pytest

from ranslice.config import make_config

pytestmark = pytest.mark.gpu


def _dims(scenario):
    cfg = make_config(scenario)
    return [10] * cfg.n_embb + [3] * cfg.n_mmtc, cfg.n_prbs


@pytest.mark.parametrize('scenario', [0, 2])
def test_single_replica_equals_reference_algorithm(golden_dir, scenario):
    from ranslice.kbrl_dev import SharedVecKBRL, VecKBRL
    g = np.load(os.path.join(golden_dir, 'g10_kbrl_s%d.npz' % scenario))
    dims, n_prbs = _dims(scenario)
    a = VecKBRL(1, dims, n_prbs, accuracy_range=tuple(g['a_range']), capacity=256)
    b = SharedVecKBRL(1, dims, n_prbs, accuracy_range=tuple(g['a_range']), capacity=256, budget=8, max_rounds=300)
    a.reset(g['init_action'][None], g['init_sec'][None])
    b.reset(g['init_action'][None], g['init_sec'][None])
    steps = 70
    for i in range(steps):
        ha = a.update_control(g['state'][i][None], g['action_in'][i][None], g['labels'][i][None])
        hb = b.update_control(g['state'][i][None], g['action_in'][i][None], g['labels'][i][None])
        assert (ha == hb).all() and (ha[0] == g['hits'][i]).all(), i
        nxt = g['state'][i + 1]
        aa, ja = a.select_action(nxt[None])
        ab, jb = b.select_action(nxt[None])
        assert (aa == ab).all() and ja[0] == jb[0] and (aa[0] == g['action_out'][i]).all(), i
    for s in range(len(dims)):
        la, lb = a.learner(0, s, with_kinv=True), b.learner(0, s, with_kinv=True)
        assert la['m'] == lb['m']
        assert la['landmarks'].tobytes() == lb['landmarks'].tobytes()
        assert la['coeff'].tobytes() == lb['coeff'].tobytes() and la['kinv'].tobytes() == lb['kinv'].tobytes()
    a.close(); b.close()


def test_dictionaries_do_not_depend_on_sharding():
    """24 replicas on one handle == 3 handles x 8 replicas exchanging proposals (what 3 ranks do over RCCL)"""
    from ranslice.kbrl_dev import SharedVecKBRL
    dims, n_prbs = [10] * 5, 200
    N, W = 24, 3
    rng = np.random.default_rng(8)
    ia = rng.integers(4, 20, size=(N, 5)).astype(np.int32)
    sf = rng.integers(2, 8, size=(N, 5)).astype(np.int32)
    whole = SharedVecKBRL(N, dims, n_prbs, capacity=256, budget=16, max_rounds=3)
    whole.reset(ia, sf)
    parts = []
    box = {}

    def make_exchange(w):
        def ex(counts, props):
            return np.stack(box['counts']), np.stack(box['props']), w
        return ex
    for w in range(W):
        p = SharedVecKBRL(N // W, dims, n_prbs, capacity=256, budget=16, max_rounds=3, first_env=w * (N // W),
                          exchange=make_exchange(w))
        p.reset(ia[w * 8:(w + 1) * 8], sf[w * 8:(w + 1) * 8], seeds=np.arange(w * 8, (w + 1) * 8, dtype=np.uint64))
        parts.append(p)

    def parts_update(state, action, labels):
        """drive the three 'ranks' in lockstep, round by round, as an all_gather would"""
        import ctypes as C
        from ranslice.kbrl_dev import PROP_W, merge_proposals
        hits = [np.zeros((8, 5), dtype=np.int32) for _ in range(W)]
        ip, fp, dp = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)
        for rnd in range(3):
            cs, ps = [], []
            for w, p in enumerate(parts):
                st = np.ascontiguousarray(state[w * 8:(w + 1) * 8], dtype=np.float32)
                ac = np.ascontiguousarray(action[w * 8:(w + 1) * 8], dtype=np.int32)
                lb = np.ascontiguousarray(labels[w * 8:(w + 1) * 8], dtype=np.int32)
                counts = np.zeros(5, dtype=np.int32)
                props = np.zeros((5, 16, PROP_W))
                p._check(p.L.kb_shared_scan(p.h, st.ctypes.data_as(fp) if rnd == 0 else None,
                                            ac.ctypes.data_as(ip) if rnd == 0 else None,
                                            lb.ctypes.data_as(ip) if rnd == 0 else None, rnd, 16,
                                            hits[w].ctypes.data_as(ip), counts.ctypes.data_as(ip), props.ctypes.data_as(dp)))
                cs.append(counts); ps.append(props)
            if int(np.sum(cs)) == 0:
                break
            mc, mp, taken = merge_proposals(np.stack(cs), np.stack(ps), 16)
            mc = np.ascontiguousarray(mc, dtype=np.int32); mp = np.ascontiguousarray(mp)
            for w, p in enumerate(parts):
                p._check(p.L.kb_shared_apply(p.h, mc.ctypes.data_as(ip), mp.ctypes.data_as(dp), 16))
                acc = np.ascontiguousarray(taken[w], dtype=np.int32)
                p._check(p.L.kb_shared_commit(p.h, acc.ctypes.data_as(ip)))
        return np.concatenate(hits)
    state = rng.random((N, 50)).astype(np.float32) * 0.5
    for i in range(25):
        action = rng.integers(5, 60, size=(N, 5)).astype(np.int32)
        # a smooth ground truth: a slice is satisfied when its allocation exceeds a state-dependent demand
        demand = (state.reshape(N, 5, 10)[:, :, [0, 5]].sum(axis=2) * 60).astype(np.int32) + 8
        labels = np.where(action >= demand, 1, -1).astype(np.int32)
        hw = whole.update_control(state, action, labels)
        hp = parts_update(state, action, labels)
        assert (hw == hp).all(), i
        state = rng.random((N, 50)).astype(np.float32) * 0.5
        aw, jw = whole.select_action(state)
        ap = np.concatenate([p.select_action(state[w * 8:(w + 1) * 8])[0] for w, p in enumerate(parts)])
        assert (aw == ap).all(), i
    sizes = []
    for s in range(5):
        lw = whole.learner(0, s, with_kinv=True)
        sizes.append(lw['m'])
        for p in parts:
            lp = p.learner(0, s, with_kinv=True)
            assert lw['m'] == lp['m'] and lw['coeff'].tobytes() == lp['coeff'].tobytes()
            assert lw['landmarks'].tobytes() == lp['landmarks'].tobytes() and lw['kinv'].tobytes() == lp['kinv'].tobytes()
    assert min(sizes) >= 3, sizes
    whole.close()
    for p in parts:
        p.close()


def test_shared_loop_run_to_run_determinism():
    """(was tests/shared_determinism_check.py) two identical shared-dictionary closed loops side by side (two
    environments, two agents, one process) at 2048 replicas of scenario_2 (BASELINE config 4's per-GPU workload
    shape): observations, hits, dictionary sizes, coefficients and actions identical at every step.  Guards the
    per-handle memory layout: a reset that wrote one dictionary counter per learner into the per-slice array used to
    corrupt neighbouring allocations (commit 9726634), which showed up as run-to-run differences."""
    from ranslice.config import EMBB_A, EMBB_SEC, MMTC_A, MMTC_SEC
    from ranslice.fading import synth_fading
    from ranslice.kbrl_dev import SharedVecKBRL
    from ranslice.vec_env import VecRanSlice
    N, steps = 2048, 25
    cfg = make_config(2, n_envs=N)
    fading = [synth_fading(t, 10000) for t in range(3)]
    dims = [10] * cfg.n_embb + [3] * cfg.n_mmtc

    def make():
        env = VecRanSlice(n_envs=N, cfg=cfg, fading=fading)
        agent = SharedVecKBRL(N, dims, cfg.n_prbs, budget=64, max_rounds=4, capacity=256)
        rng = np.random.default_rng(1000)
        ia = np.concatenate([rng.integers(EMBB_A[0], EMBB_A[1], size=(N, cfg.n_embb)),
                             rng.integers(MMTC_A[0], MMTC_A[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
        sf = np.concatenate([rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(N, cfg.n_embb)),
                             rng.integers(MMTC_SEC[0], MMTC_SEC[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
        state = env.reset()
        agent.reset(ia, sf)
        return dict(env=env, agent=agent, state=state, action=ia.copy())
    A, B = make(), make()
    for i in range(steps):
        res = []
        for R in (A, B):
            obs, rew, _, info = R['env'].step(R['action'])
            hits = R['agent'].update_control(R['state'], R['action'], info['SLA_labels'])
            learners = [R['agent'].learner(0, s) for s in range(len(dims))]
            action, adj = R['agent'].select_action(obs)
            R['state'], R['action'] = obs, action
            res.append((obs, hits, [l['m'] for l in learners], [l['coeff'].tobytes() for l in learners], action))
        a, b = res
        assert a[0].tobytes() == b[0].tobytes(), ('obs', i)
        assert (a[1] == b[1]).all(), ('hits', i)
        assert a[2] == b[2], ('sizes', i, a[2], b[2])
        assert a[3] == b[3], ('coeff', i)
        assert (a[4] == b[4]).all(), ('action', i)
    assert max(a[2]) > 1, 'the loop should have learned something'
    for R in (A, B):
        R['env'].close()
        R['agent'].close()


def test_batched_apply_of_full_dictionaries_equals_serial(monkeypatch):
    """A shared dictionary at capacity only projects, so the apply kernel computes the kernel columns and d* = Kinv k_f
    of a whole proposal list at once and runs only predict + coefficient update in order (one wave, no block
    barriers).  With a capacity small enough to fill within a few steps (16 landmarks), that path against the
    one-proposal-at-a-time path (KBRL_SERIAL_APPLY): same hits, sizes, coefficients (bits) and actions at every step."""
    from ranslice.config import EMBB_A, EMBB_SEC, MMTC_A, MMTC_SEC
    from ranslice.fading import synth_fading
    from ranslice.kbrl_dev import SharedVecKBRL
    from ranslice.vec_env import VecRanSlice
    N, steps = 512, 30
    cfg = make_config(2, n_envs=N)
    fading = [synth_fading(t, 4000) for t in range(3)]
    dims = [10] * cfg.n_embb + [3] * cfg.n_mmtc

    def make(serial):
        if serial:
            monkeypatch.setenv('KBRL_SERIAL_APPLY', '1')
        else:
            monkeypatch.delenv('KBRL_SERIAL_APPLY', raising=False)
        env = VecRanSlice(n_envs=N, cfg=cfg, fading=fading)
        agent = SharedVecKBRL(N, dims, cfg.n_prbs, budget=64, max_rounds=4, capacity=16)
        monkeypatch.delenv('KBRL_SERIAL_APPLY', raising=False)
        rng = np.random.default_rng(77)
        ia = np.concatenate([rng.integers(EMBB_A[0], EMBB_A[1], size=(N, cfg.n_embb)),
                             rng.integers(MMTC_A[0], MMTC_A[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
        sf = np.concatenate([rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(N, cfg.n_embb)),
                             rng.integers(MMTC_SEC[0], MMTC_SEC[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
        state = env.reset()
        agent.reset(ia, sf)
        return dict(env=env, agent=agent, state=state, action=ia.copy())
    A, B = make(False), make(True)
    full_seen = False
    for i in range(steps):
        res = []
        for R in (A, B):
            obs, rew, _, info = R['env'].step(R['action'])
            hits = R['agent'].update_control(R['state'], R['action'], info['SLA_labels'])
            learners = [R['agent'].learner(0, s) for s in range(len(dims))]
            action, adj = R['agent'].select_action(obs)
            R['state'], R['action'] = obs, action
            res.append((hits, [l['m'] for l in learners], [l['coeff'].tobytes() for l in learners], action))
        a, b = res
        assert (a[0] == b[0]).all(), ('hits', i)
        assert a[1] == b[1], ('sizes', i, a[1], b[1])
        assert a[2] == b[2], ('coeff', i)
        assert (a[3] == b[3]).all(), ('action', i)
        full_seen = full_seen or max(a[1]) >= 16
    assert full_seen, 'no dictionary reached its capacity: the batched path was not exercised'
    for R in (A, B):
        R['env'].close()
        R['agent'].close()


def _run_ranks(world, n_per_rank, steps, tmp_path, extra_env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    id_file = str(tmp_path / ('uid_%d' % world))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'shared_rccl_worker.py'), str(r), str(world),
                               id_file, str(n_per_rank), str(steps)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    return outs


def _result(o):
    line = [x for x in o.splitlines() if x.startswith('RESULT')][-1].split()
    return line[2]


def test_device_exchange_world1_rccl_equals_plain(tmp_path):
    """kb_shared_step through a real RCCL communicator of one rank (ncclGetUniqueId / ncclCommInitRank /
    ncclAllGather bound by libranslice.so at run time) == the same step without a communicator"""
    plain = _run_ranks(1, 16, 6, tmp_path)
    assert plain[0][0] == 0, plain[0][2][-2000:]
    rccl = _run_ranks(1, 16, 6, tmp_path, {'RCCL_WORLD1': '1'})
    assert rccl[0][0] == 0, rccl[0][2][-2000:]
    assert _result(plain[0][1]) == _result(rccl[0][1])


def test_device_exchange_two_ranks_over_rccl(tmp_path):
    """two processes, 8 replicas each, exchanging proposals with ncclAllGather on the device: both ranks end with
    bitwise-identical dictionaries, equal to those of ONE handle holding all 16 replicas (the learned dictionaries do
    not depend on the sharding).  One GPU serves both ranks here; if RCCL refuses two ranks on one device the
    2-rank half is skipped (the driver's multi-GPU run exercises it) and the world-1 test above stands."""
    whole = _run_ranks(1, 16, 6, tmp_path)
    assert whole[0][0] == 0, whole[0][2][-2000:]
    two = _run_ranks(2, 8, 6, tmp_path)
    if any(rc != 0 for rc, _, _ in two):
        err = ' '.join(e[-400:] for _, _, e in two)
        if 'ncclCommInitRank' in err or 'Duplicate' in err or 'invalid usage' in err:
            pytest.skip('RCCL does not form a 2-rank communicator on a single device: %s' % err[-300:])
        assert False, err[-3000:]
    assert _result(two[0][1]) == _result(two[1][1]) == _result(whole[0][1])

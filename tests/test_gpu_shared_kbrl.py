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


def _sharding_independence(N, W, dims, n_prbs, budget, rounds, steps, capacity=256, whole_on_device=False):
    """N replicas on one handle == W handles x N/W replicas exchanging proposals (what W ranks do over RCCL).  The
    W handles are driven round by round through kb_shared_scan / host merge / kb_shared_apply / kb_shared_commit; the
    single handle through the same calls, or (whole_on_device) through kb_shared_step: scan, collect, device merge,
    apply and commit without leaving the GPU."""
    import ctypes as C
    from ranslice.kbrl_dev import PROP_W, SharedVecKBRL, merge_proposals
    S, nv, n = len(dims), int(sum(dims)), N // W
    rng = np.random.default_rng(8)
    ia = rng.integers(4, 20, size=(N, S)).astype(np.int32)
    sf = rng.integers(2, 8, size=(N, S)).astype(np.int32)
    box = {}
    whole = SharedVecKBRL(N, dims, n_prbs, capacity=capacity, budget=budget, max_rounds=rounds,
                          exchange=None if whole_on_device else (lambda c, p: (c[None], p[None], 0)))
    whole.reset(ia, sf)
    parts = []
    for w in range(W):
        p = SharedVecKBRL(n, dims, n_prbs, capacity=capacity, budget=budget, max_rounds=rounds, first_env=w * n)
        p.reset(ia[w * n:(w + 1) * n], sf[w * n:(w + 1) * n], seeds=np.arange(w * n, (w + 1) * n, dtype=np.uint64))
        parts.append(p)
    ip, fp, dp = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)

    def parts_update(state, action, labels):
        """drive the W 'ranks' in lockstep, round by round, as an all_gather would"""
        hits = [np.zeros((n, S), dtype=np.int32) for _ in range(W)]
        for rnd in range(rounds):
            cs, ps = [], []
            for w, p in enumerate(parts):
                st = np.ascontiguousarray(state[w * n:(w + 1) * n], dtype=np.float32)
                ac = np.ascontiguousarray(action[w * n:(w + 1) * n], dtype=np.int32)
                lb = np.ascontiguousarray(labels[w * n:(w + 1) * n], dtype=np.int32)
                counts = np.zeros(S, dtype=np.int32)
                props = np.zeros((S, budget, PROP_W))
                p._check(p.L.kb_shared_scan(p.h, st.ctypes.data_as(fp) if rnd == 0 else None,
                                            ac.ctypes.data_as(ip) if rnd == 0 else None,
                                            lb.ctypes.data_as(ip) if rnd == 0 else None, rnd, budget,
                                            hits[w].ctypes.data_as(ip), counts.ctypes.data_as(ip), props.ctypes.data_as(dp)))
                cs.append(counts); ps.append(props)
            if int(np.sum(cs)) == 0:
                break
            mc, mp, taken = merge_proposals(np.stack(cs), np.stack(ps), budget)
            mc = np.ascontiguousarray(mc, dtype=np.int32); mp = np.ascontiguousarray(mp)
            for w, p in enumerate(parts):
                p._check(p.L.kb_shared_apply(p.h, mc.ctypes.data_as(ip), mp.ctypes.data_as(dp), budget))
                acc = np.ascontiguousarray(taken[w], dtype=np.int32)
                p._check(p.L.kb_shared_commit(p.h, acc.ctypes.data_as(ip)))
        return np.concatenate(hits)
    lead = np.cumsum([0] + list(dims))[:-1]          # a smooth ground truth on each learner's first state variable
    state = rng.random((N, nv)).astype(np.float32) * 0.5
    for i in range(steps):
        action = rng.integers(5, min(60, n_prbs), size=(N, S)).astype(np.int32)
        # a slice is satisfied when its allocation exceeds a state-dependent demand
        demand = (state[:, lead] * 2 * min(60, n_prbs)).astype(np.int32) + 8
        labels = np.where(action >= demand, 1, -1).astype(np.int32)
        hw = whole.update_control(state, action, labels)
        hp = parts_update(state, action, labels)
        assert (hw == hp).all(), i
        state = rng.random((N, nv)).astype(np.float32) * 0.5
        aw, jw = whole.select_action(state)
        ap = np.concatenate([p.select_action(state[w * n:(w + 1) * n])[0] for w, p in enumerate(parts)])
        assert (aw == ap).all(), i
    sizes = []
    for s in range(S):
        lw = whole.learner(0, s, with_kinv=True)
        sizes.append(lw['m'])
        for p in parts:
            lp = p.learner(0, s, with_kinv=True)
            assert lw['m'] == lp['m'] and lw['coeff'].tobytes() == lp['coeff'].tobytes()
            assert lw['landmarks'].tobytes() == lp['landmarks'].tobytes() and lw['kinv'].tobytes() == lp['kinv'].tobytes()
    whole.close()
    for p in parts:
        p.close()
    return sizes


def test_dictionaries_do_not_depend_on_sharding():
    """24 replicas on one handle == 3 handles x 8 replicas exchanging proposals (what 3 ranks do over RCCL)"""
    sizes = _sharding_independence(24, 3, [10] * 5, 200, budget=16, rounds=3, steps=25)
    assert min(sizes) >= 3, sizes


def test_config4_per_gpu_shard_one_handle_equals_split_handles():
    """BASELINE config 4's per-GPU shard (32,768 replicas / 8 GPUs = 4,096 per GPU; scenario_2: one eMBB and four mMTC
    learners, 100 PRBs) with the shared agent: one handle learning through kb_shared_step (scan, collect, device merge,
    apply, commit -- the resident exchange path, its own world) == two handles of 2,048 replicas exchanging their
    proposal lists round by round through the host merge rule; hits and selected actions at every step, the five
    dictionaries bit for bit at the end"""
    dims, n_prbs = _dims(2)
    sizes = _sharding_independence(4096, 2, dims, n_prbs, budget=64, rounds=4, steps=6, whole_on_device=True)
    assert max(sizes) >= 16, sizes


def test_device_merge_kernel_equals_host_rule():
    """shared_merge_kernel on gathered buffers of 2, 3 and 4 ranks (what ncclAllGather delivers to kb_shared_step) against
    ranslice.kbrl_dev.merge_proposals, the rule pinned by the gloo test: merged lists, counts, every rank's `taken` and the
    all-rank proposer total -- with ranks that propose nothing, more proposers than the budget, and interleaved ids"""
    import ctypes as C
    from ranslice.kbrl_dev import PROP_W, SharedVecKBRL, merge_proposals
    S, budget = 5, 16
    ag = SharedVecKBRL(8, [10] * S, 200, capacity=64, budget=budget)
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    rng = np.random.default_rng(31)
    for W in (2, 3, 4):
        for trial in range(4):
            counts = rng.integers(0, 2 * budget, size=(W, S)).astype(np.int32)   # proposers per rank, may exceed the budget
            counts[rng.integers(W), :] = 0 if trial == 1 else counts[0]
            props = np.zeros((W, S, budget, PROP_W))
            for s in range(S):
                # contiguous replica shards: rank w's global ids lie in [1000 w, 1000 w + 999], ascending in its list
                for w in range(W):
                    k = min(int(counts[w, s]), budget)
                    ids = np.sort(rng.choice(1000, size=k, replace=False)) + 1000 * (w if trial != 2 else (W - 1 - w))
                    props[w, s, :k, 0] = ids
                    props[w, s, :k, 1:] = rng.random((k, PROP_W - 1))
            mc, mp, taken = merge_proposals(counts, props, budget)
            gathered = np.concatenate([np.concatenate([counts[w].astype(np.float64), props[w].ravel()]) for w in range(W)])
            for me in range(W):
                merged = np.zeros((S, budget, PROP_W))
                dc, dt, tot = np.zeros(S, dtype=np.int32), np.zeros(S, dtype=np.int32), C.c_int32()
                ag._check(ag.L.kb_shared_merge(ag.h, gathered.ctypes.data_as(dp), W, me, budget, merged.ctypes.data_as(dp),
                                               dc.ctypes.data_as(ip), dt.ctypes.data_as(ip), C.byref(tot)))
                assert (dc == mc).all() and (dt == taken[me]).all() and tot.value == int(counts.sum()), (W, trial, me)
                for s in range(S):
                    assert merged[s, :mc[s]].tobytes() == mp[s, :mc[s]].tobytes(), (W, trial, me, s)
    ag.close()


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


@pytest.mark.parametrize('capacity', [16, 64, 192])
def test_batched_apply_of_full_dictionaries_equals_serial(monkeypatch, capacity):
    """A shared dictionary at capacity only projects, so the apply kernel computes the kernel columns and d* = Kinv k_f
    of a whole proposal list at once and runs only predict + coefficient update in order (one wave, no block
    barriers).  With a capacity small enough to fill within a few steps, that path against the
    one-proposal-at-a-time path (KBRL_SERIAL_APPLY): same hits, sizes, coefficients (bits) and actions at every step.
    Capacity 16: d* by the column walk (shared_matvec_kernel); 64 and 192: d* = KF Kinv on the matrix cores, eight
    accumulator tiles per wave standing for the column walk's eight partial sums (shared_matvec_mfma_kernel)."""
    from ranslice.config import EMBB_A, EMBB_SEC, MMTC_A, MMTC_SEC
    from ranslice.fading import synth_fading
    from ranslice.kbrl_dev import SharedVecKBRL
    from ranslice.vec_env import VecRanSlice
    N, steps = 512, 30
    cfg = make_config(2, n_envs=N)
    fading = [synth_fading(t, 4000) for t in range(3)]
    dims = [10] * cfg.n_embb + [3] * cfg.n_mmtc

    def make(serial):
        # the serial apply is a knob of the test build; the batched one runs on the production library
        monkeypatch.setenv('RANSLICE_DEV_BUILD', '1' if serial else '0')
        if serial:
            monkeypatch.setenv('KBRL_SERIAL_APPLY', '1')
        else:
            monkeypatch.delenv('KBRL_SERIAL_APPLY', raising=False)
        env = VecRanSlice(n_envs=N, cfg=cfg, fading=fading)
        agent = SharedVecKBRL(N, dims, cfg.n_prbs, budget=64, max_rounds=4, capacity=capacity)
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
        full_seen = full_seen or max(a[1]) >= capacity
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
    not depend on the sharding).  Each rank takes device rank % rs_device_count(): on a box with two GPUs the exchange
    crosses xGMI; on a single-GPU box RCCL refuses two ranks on one device and the 2-rank half is skipped -- NOT
    verified there (the merge kernel itself is covered for 2-4 ranks by test_device_merge_kernel_equals_host_rule and the
    scan / apply / commit path by the split-handle tests)."""
    whole = _run_ranks(1, 16, 6, tmp_path)
    assert whole[0][0] == 0, whole[0][2][-2000:]
    two = _run_ranks(2, 8, 6, tmp_path)
    if any(rc != 0 for rc, _, _ in two):
        err = ' '.join(e[-400:] for _, _, e in two)
        if 'ncclCommInitRank' in err or 'Duplicate' in err or 'invalid usage' in err:
            pytest.skip('RCCL does not form a 2-rank communicator on a single device: %s' % err[-300:])
        assert False, err[-3000:]
    assert _result(two[0][1]) == _result(two[1][1]) == _result(whole[0][1])


def test_failure_in_a_round_leaves_the_step_through_the_exchange(monkeypatch, tmp_path):
    """kb_shared_step's abort path (VERDICT r3 #6a): a rank whose round fails locally still takes part in the all-gather, with
    a mark in the place of its proposer counts; the merge kernel shows the mark to every rank and all of them return RS_EHIP
    after that same round -- nobody stays behind in the collective.  Here: one handle (its own world) with the failure
    injected in round 1 of a step, in process and through a real one-rank RCCL communicator; the handle goes on working
    afterwards.  The two-rank case runs where two GPUs are visible (below)."""
    from ranslice import _lib
    from ranslice.kbrl_dev import SharedVecKBRL
    sys_path_worker = os.path.join(os.path.dirname(os.path.abspath(__file__)))
    import importlib.util
    spec = importlib.util.spec_from_file_location('rccl_worker', os.path.join(sys_path_worker, 'shared_rccl_worker.py'))
    wk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wk)
    ia, sf, seq = wk.batch(16, 5)
    monkeypatch.setenv('RANSLICE_DEV_BUILD', '1')   # the fault injector is read by the test build only (ranslice._lib)
    ag = SharedVecKBRL(16, [10] * 5, 200, capacity=256, budget=16, max_rounds=3)
    ag.reset(ia, sf)
    for i, (state, action, labels, nxt) in enumerate(seq):
        if i == 2:
            monkeypatch.setenv('KBRL_INJECT_FAIL_ROUND', '1')
            with pytest.raises(_lib.RanSliceError) as e:
                ag.update_control(state, action, labels)
            assert 'failed in round 1' in str(e.value) and 'leave the step' in str(e.value), str(e.value)
            monkeypatch.delenv('KBRL_INJECT_FAIL_ROUND')
            continue
        ag.update_control(state, action, labels)      # before and after the failed step: business as usual
        ag.select_action(nxt)
    assert ag.dictionary_sizes().sum() > 0
    ag.close()
    out = _run_ranks(1, 16, 6, tmp_path, {'RCCL_WORLD1': '1', 'FAIL_RANK': '0', 'FAIL_STEP': '3'})
    assert out[0][0] == 3 and 'FAILED 0 step 3' in out[0][1], (out[0][1][-500:], out[0][2][-1500:])


def test_failure_on_one_of_two_ranks_ends_both(tmp_path):
    """two ranks over RCCL, rank 1's round fails at step 3: BOTH ranks leave kb_shared_step with RS_EHIP at step 3 (skipped
    -- not verified -- where RCCL cannot form a two-rank communicator: one GPU)"""
    two = _run_ranks(2, 8, 6, tmp_path, {'FAIL_RANK': '1', 'FAIL_STEP': '3', 'KBRL_COLLECTIVE_TIMEOUT_S': '60'})
    err = ' '.join(e[-400:] for _, _, e in two)
    if any('ncclCommInitRank' in e or 'Duplicate' in e or 'invalid usage' in e for _, _, e in two):
        pytest.skip('RCCL does not form a 2-rank communicator on a single device: %s' % err[-300:])
    for r, (rc, o, e) in enumerate(two):
        assert rc == 3 and ('FAILED %d step 3' % r) in o, (r, rc, o[-500:], e[-1500:])
    assert 'this rank failed' in two[1][1] and 'another rank' in two[0][1]


def test_aborted_communicator_refuses_until_rejoined(tmp_path):
    """after ncclCommAbort (here: the bounded wait's timeout, injected in the test build, on a real one-rank RCCL
    communicator) the handle is NOT a one-rank world: kb_shared_step and kb_comm_info return RS_ESTATE until kb_comm_init joins a
    new communicator, after which the steps go on.  kb_comm_info reports what the communicator itself says."""
    out = _run_ranks(1, 16, 6, tmp_path, {'RCCL_WORLD1': '1', 'ABORT_STEP': '2'})
    rc, o, e = out[0]
    assert rc == 0, (o[-800:], e[-1500:])
    assert 'ABORTED 2: code -3' in o and 'communicator aborted' in o
    assert 'REFUSED step: code -4' in o and 'REFUSED info: code -4' in o and 'was aborted' in o
    assert 'REJOINED (0, 1)' in o and 'RESULT 0' in o


def test_silent_peer_ends_in_a_timeout_not_a_hang(tmp_path):
    """one of two ranks stops calling kb_shared_step (a hung peer).  The other rank's per-round flag comes back
    through pinned memory, so its bounded wait runs: it leaves with RS_EHIP after KBRL_COLLECTIVE_TIMEOUT_S instead of sitting
    in a blocking copy behind the all-gather.  Needs a two-rank communicator (two GPUs): skipped -- not verified -- elsewhere."""
    import time
    t0 = time.time()
    two = _run_ranks(2, 8, 6, tmp_path, {'SILENT_RANK': '1', 'SILENT_STEP': '3', 'SILENT_SECONDS': '40', 'KBRL_COLLECTIVE_TIMEOUT_S': '5'})
    err = ' '.join(e[-400:] for _, _, e in two)
    if any('ncclCommInitRank' in e or 'Duplicate' in e or 'invalid usage' in e for _, _, e in two):
        pytest.skip('RCCL does not form a 2-rank communicator on a single device: %s' % err[-300:])
    rc, o, e = two[0]
    assert rc == 3 and 'FAILED 0 step 3' in o and 'no answer from the other ranks' in o, (rc, o[-500:], e[-1500:])
    assert 'SILENT 1 from step 3' in two[1][1]
    assert time.time() - t0 < 120


def test_resident_loop_through_rccl(tmp_path):
    """kb_shared_step_resident through a communicator.  One rank: the closed loop on the device through a real one-rank RCCL
    communicator == the loop without one (dictionaries, last actions).  Two ranks (two GPUs; skipped -- not verified -- elsewhere):
    rank 1's last round fails locally at step 3; both ranks leave kb_shared_step_resident at step 3, told by the merged mark.
    (Round 5 tried reading that mark a fixed number of calls later instead of waiting for it once per step: 3.3-3.4 ms per step
    against 2.9 on one GPU -- slower, for reasons in the streams' interplay that were not pursued -- and went back.)"""
    plain = _run_ranks(1, 16, 8, tmp_path, {'RESIDENT': '1'})
    assert plain[0][0] == 0, plain[0][2][-2000:]
    rccl = _run_ranks(1, 16, 8, tmp_path, {'RESIDENT': '1', 'RCCL_WORLD1': '1'})
    assert rccl[0][0] == 0, rccl[0][2][-2000:]
    assert plain[0][1].split()[2:5] == rccl[0][1].split()[2:5]
    two = _run_ranks(2, 8, 8, tmp_path, {'RESIDENT': '1', 'FAIL_RANK': '1', 'FAIL_STEP': '3', 'KBRL_COLLECTIVE_TIMEOUT_S': '60'})
    err = ' '.join(e[-400:] for _, _, e in two)
    if any('ncclCommInitRank' in e or 'Duplicate' in e or 'invalid usage' in e for _, _, e in two):
        pytest.skip('one-rank part verified; RCCL does not form a 2-rank communicator on a single device: %s' % err[-200:])
    for r in (0, 1):
        assert two[r][0] == 3 and ('FAILED %d step 3' % r) in two[r][1], two[r][1][-500:]


def test_shared_resident_loop_equals_host_loop():
    """kb_shared_step_resident (the shared learning step and select_action on the simulator's own device buffers) == the
    same closed loop driven through host buffers (env.step / update_control / select_action): selected actions at every
    step, dictionaries bit for bit at the end (scenario_2, 512 replicas)"""
    import ctypes as C
    from ranslice.config import EMBB_A, EMBB_SEC, MMTC_A, MMTC_SEC
    from ranslice.fading import synth_fading
    from ranslice.kbrl_dev import SharedVecKBRL
    from ranslice.vec_env import VecRanSlice
    N, steps = 512, 14
    cfg = make_config(2, n_envs=N)
    fading = [synth_fading(t, 4000) for t in range(3)]
    dims = [10] * cfg.n_embb + [3] * cfg.n_mmtc
    rng = np.random.default_rng(5)
    ia = np.concatenate([rng.integers(EMBB_A[0], EMBB_A[1], size=(N, cfg.n_embb)),
                         rng.integers(MMTC_A[0], MMTC_A[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
    sf = np.concatenate([rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(N, cfg.n_embb)),
                         rng.integers(MMTC_SEC[0], MMTC_SEC[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)

    def make():
        env = VecRanSlice(n_envs=N, cfg=cfg, fading=fading)
        agent = SharedVecKBRL(N, dims, cfg.n_prbs, budget=64, max_rounds=4, capacity=256)
        state = env.reset()
        agent.reset(ia, sf)
        return env, agent, state
    env, agent, state = make()
    action = ia.copy()
    host_actions = []
    for i in range(steps):
        obs, rew, _, info = env.step(action)
        agent.update_control(state, action, info['SLA_labels'])
        action, adj = agent.select_action(obs)
        state = obs
        host_actions.append(action.copy())
    host_dicts = [agent.learner(0, s, with_kinv=True) for s in range(len(dims))]
    env.close(); agent.close()
    env, agent, state = make()
    a0 = np.ascontiguousarray(ia)
    env._check(env.L.rs_step(env.h, a0.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
    for i in range(steps):
        agent.step_resident(env)
        assert (env.fetch()['actions'] == host_actions[i]).all(), i
        if i + 1 < steps:
            env.step_resident()
    for s in range(len(dims)):
        L = agent.learner(0, s, with_kinv=True)
        assert L['m'] == host_dicts[s]['m'] and L['coeff'].tobytes() == host_dicts[s]['coeff'].tobytes()
        assert L['kinv'].tobytes() == host_dicts[s]['kinv'].tobytes()
    assert max(L['m'] for L in host_dicts) > 1
    env.close(); agent.close()

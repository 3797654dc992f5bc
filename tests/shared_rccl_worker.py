"""One rank of the RCCL test of the shared-dictionary exchange (tests/test_gpu_shared_kbrl.py starts `world` of these
on the same box).  usage: shared_rccl_worker.py <rank> <world> <id_file> <n_per_rank> <steps>
Rank 0 writes the ncclUniqueId to <id_file>; every rank learns from its contiguous shard of one synthetic batch and
prints a hash of the (replicated) dictionaries."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
from ranslice import _lib  # noqa: E402
from ranslice.kbrl_dev import SharedVecKBRL  # noqa: E402


def batch(n_total, steps, seed=8):
    """the synthetic (state, action, labels, next state) sequence every layout of the test learns from"""
    rng = np.random.default_rng(seed)
    ia = rng.integers(4, 20, size=(n_total, 5)).astype(np.int32)
    sf = rng.integers(2, 8, size=(n_total, 5)).astype(np.int32)
    seq = []
    state = rng.random((n_total, 50)).astype(np.float32) * 0.5
    for _ in range(steps):
        action = rng.integers(5, 60, size=(n_total, 5)).astype(np.int32)
        demand = (state.reshape(n_total, 5, 10)[:, :, [0, 5]].sum(axis=2) * 60).astype(np.int32) + 8
        labels = np.where(action >= demand, 1, -1).astype(np.int32)
        nxt = rng.random((n_total, 50)).astype(np.float32) * 0.5
        seq.append((state, action, labels, nxt))
        state = nxt
    return ia, sf, seq


def digest(agent):
    h = hashlib.sha256()
    sizes = []
    for s in range(5):
        L = agent.learner(0, s, with_kinv=True)
        sizes.append(L['m'])
        for k in ('landmarks', 'coeff', 'kinv'):
            h.update(np.ascontiguousarray(L[k]).tobytes())
    return h.hexdigest(), sizes


def resident_loop(agent, rank, world, n, steps, device):
    """the closed loop on the device (kb_shared_step_resident, max_rounds rounds, no round count asked for): the last round's failure
    mark comes back once per step.  Environments: scenario_0 replicas
    [rank * n, (rank + 1) * n) of one batch.  FAIL_RANK / FAIL_STEP: that rank's round fails locally (it reports at once); the other
    ranks must leave at the same step."""
    import ctypes as C
    from ranslice.config import make_config
    from ranslice.fading import synth_fading
    from ranslice.sharding import replica_seeds
    from ranslice.vec_env import VecRanSlice
    env = VecRanSlice(n_envs=n, cfg=make_config(0, n_envs=n), fading=[synth_fading(t, 2000) for t in range(3)], device=device)
    env.reset(seeds=replica_seeds(0, rank * n, n))
    a0 = np.full((n, 5), 20, np.int32)
    env._check(env.L.rs_step(env.h, a0.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
    fail_rank, fail_step = int(os.environ.get('FAIL_RANK', '-1')), int(os.environ.get('FAIL_STEP', '-1'))
    for i in range(steps):
        if rank == fail_rank and i == fail_step:
            os.environ['KBRL_INJECT_FAIL_ROUND'] = str(agent.max_rounds - 1)   # the LAST permitted round of this step
        try:
            agent.step_resident(env)
        except _lib.RanSliceError as e:
            print('FAILED %d step %d: %s' % (rank, i, e), flush=True)
            agent.close()
            sys.exit(3)
        env.step_resident()
    env.synchronize()
    agent.synchronize()
    d, sizes = digest(agent)
    acts = env.fetch()['actions']
    print('RESULT %d %s %s %s device=%d' % (rank, d, sizes, hashlib.sha256(acts.tobytes()).hexdigest(), device), flush=True)
    agent.close()
    env.close()


def main():
    rank, world, id_file, n, steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    ia, sf, seq = batch(n * world, steps)
    lo, hi = rank * n, (rank + 1) * n
    from ranslice import _lib
    if int(os.environ.get('FAIL_RANK', '-1')) >= 0 or int(os.environ.get('ABORT_STEP', '-1')) >= 0:
        os.environ['RANSLICE_DEV_BUILD'] = '1'    # the fault injectors exist in the test build only
    device = rank % _lib.device_count()       # one GPU per rank where the box has them
    agent = SharedVecKBRL(n, [10] * 5, 200, capacity=256, budget=16, max_rounds=3, first_env=lo, device=device)
    if world > 1 or os.environ.get('RCCL_WORLD1'):
        if rank == 0:
            uid = SharedVecKBRL.unique_id()
            with open(id_file + '.tmp', 'wb') as f:
                f.write(uid)
            os.replace(id_file + '.tmp', id_file)
        else:
            t0 = time.time()
            while not os.path.exists(id_file):
                if time.time() - t0 > 120:
                    raise SystemExit('no unique id')
                time.sleep(0.05)
            with open(id_file, 'rb') as f:
                uid = f.read()
        agent.comm_init(uid, rank, world)
    agent.reset(ia[lo:hi], sf[lo:hi])
    if os.environ.get('RESIDENT'):
        resident_loop(agent, rank, world, n, steps, device)
        return
    acts = []
    fail_rank, fail_step = int(os.environ.get('FAIL_RANK', '-1')), int(os.environ.get('FAIL_STEP', '-1'))
    abort_step = int(os.environ.get('ABORT_STEP', '-1'))
    silent_rank, silent_step = int(os.environ.get('SILENT_RANK', '-1')), int(os.environ.get('SILENT_STEP', '-1'))
    for i, (state, action, labels, nxt) in enumerate(seq):
        if rank == silent_rank and i == silent_step:
            # this rank simply stops taking part (a hung or dead peer): the others must leave kb_shared_step on their own
            print('SILENT %d from step %d' % (rank, i), flush=True)
            time.sleep(float(os.environ.get('SILENT_SECONDS', '30')))
            os._exit(0)
        if i == abort_step:
            # the bounded wait gives up (injected): RS_EHIP and the communicator is aborted; the handle must then REFUSE shared
            # steps (not carry on as a world of its own) until a new communicator is joined
            os.environ['KBRL_INJECT_TIMEOUT'] = '1'
            try:
                agent.update_control(state[lo:hi], action[lo:hi], labels[lo:hi])
                print('NO ERROR at the injected timeout', flush=True)
                sys.exit(4)
            except _lib.RanSliceError as e:
                print('ABORTED %d: code %d %s' % (i, e.code, e), flush=True)
            del os.environ['KBRL_INJECT_TIMEOUT']
            for what, call in (('step', lambda: agent.update_control(state[lo:hi], action[lo:hi], labels[lo:hi])),
                               ('info', agent.comm_info)):
                try:
                    call()
                    print('NOT REFUSED %s' % what, flush=True)
                    sys.exit(4)
                except _lib.RanSliceError as e:
                    print('REFUSED %s: code %d %s' % (what, e.code, e), flush=True)
            agent.comm_init(SharedVecKBRL.unique_id(), 0, 1)      # (the one-rank form of the test)
            print('REJOINED %s' % (agent.comm_info(),), flush=True)
        if rank == fail_rank and i == fail_step:
            os.environ['KBRL_INJECT_FAIL_ROUND'] = '0'   # this rank's round 0 "fails" (kb_api.hip: shared_step_core)
        try:
            agent.update_control(state[lo:hi], action[lo:hi], labels[lo:hi])
        except _lib.RanSliceError as e:
            # every rank of the group must get here at the same step, told by the exchange itself
            print('FAILED %d step %d: %s' % (rank, i, e), flush=True)
            agent.close()
            sys.exit(3)
        a, _ = agent.select_action(nxt[lo:hi])
        acts.append(a)
    d, sizes = digest(agent)
    print('RESULT %d %s %s %s device=%d' % (rank, d, sizes, hashlib.sha256(np.stack(acts).tobytes()).hexdigest(), device), flush=True)
    agent.close()


if __name__ == '__main__':
    main()

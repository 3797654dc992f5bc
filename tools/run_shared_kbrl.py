#!/usr/bin/env python3
"""BASELINE config 4: scenario_2 (1 eMBB + 4 mMTC slices), env replicas sharded over the GPUs of one node,
ONE shared KBRL dictionary per slice learned from all replicas; the only collective is the per-round
all_gather of proposed landmarks over RCCL (xGMI).  Launch:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         tools/run_shared_kbrl.py --envs-per-gpu 4096 --steps 200

Prints one JSON line on rank 0 (env-steps/s with the shared agent in the loop, dictionary sizes, rounds).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'network-slicing_amd'))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenario', type=int, default=2)
    ap.add_argument('--envs-per-gpu', type=int, default=4096)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--budget', type=int, default=256)
    ap.add_argument('--rounds', type=int, default=1)
    ap.add_argument('--count-rounds', action='store_true',
                    help='resident loop: wait for the per-step count of exchange rounds (one host round trip more)')
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--host-loop', action='store_true', help='drive the loop through host buffers (env.step / update_control)')
    ap.add_argument('--capacity', type=int, default=1024,
                    help='landmarks per shared dictionary; a full dictionary projects instead of growing, and one '
                         'projection reads capacity^2 doubles of Kinv from a single workgroup')
    args = ap.parse_args()
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (('RANK', 0), ('LOCAL_RANK', 0), ('WORLD_SIZE', 1)))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    use_dist = 'RANK' in os.environ
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
    from ranslice.config import make_config, EMBB_A, EMBB_SEC, MMTC_A, MMTC_SEC
    from ranslice.fading import synth_fading
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from dist_util import max_over_ranks
    from ranslice.kbrl_dev import SharedVecKBRL
    from ranslice.sharding import shard_range, replica_seeds
    from ranslice.vec_env import VecRanSlice
    N = args.envs_per_gpu
    cfg = make_config(args.scenario, n_envs=N)
    first, count = shard_range(world * N, rank, world)
    env = VecRanSlice(n_envs=N, cfg=cfg, fading=[synth_fading(t, 10000) for t in range(3)], device=local_rank)
    dims = [10] * cfg.n_embb + [3] * cfg.n_mmtc
    agent = SharedVecKBRL(N, dims, cfg.n_prbs, budget=args.budget, max_rounds=args.rounds, capacity=args.capacity,
                          device=local_rank, first_env=first)
    if use_dist:
        # the exchange itself is ncclAllGather inside libranslice.so (kb_shared_step); the launcher's process group
        # only carries the 128-byte communicator id from rank 0 to the others
        box = [SharedVecKBRL.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        agent.comm_init(box[0], rank, world)
    rng = np.random.default_rng(1000 + rank)
    ia = np.concatenate([rng.integers(EMBB_A[0], EMBB_A[1], size=(N, cfg.n_embb)),
                         rng.integers(MMTC_A[0], MMTC_A[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
    sf = np.concatenate([rng.integers(EMBB_SEC[0], EMBB_SEC[1], size=(N, cfg.n_embb)),
                         rng.integers(MMTC_SEC[0], MMTC_SEC[1], size=(N, cfg.n_mmtc))], axis=1).astype(np.int32)
    state = env.reset(seeds=replica_seeds(0, first, count))
    agent.reset(ia, sf, seeds=replica_seeds(7, first, count))
    action = ia.copy()
    rounds = 0
    viol = 0.0
    import ctypes as C
    if args.host_loop:
        # the same loop through host buffers (observations, labels and actions cross PCIe every step)
        def run(k):
            nonlocal state, action, rounds, viol
            for _ in range(k):
                obs, rew, _, info = env.step(action)
                agent.update_control(state, action, info['SLA_labels'])
                rounds += agent.rounds_last
                action, adj = agent.select_action(obs)
                state = obs
                viol += float(info['total_violations'].mean())
    else:
        # device-resident: kb_shared_step_resident after rs_step_resident; only the per-round flag returns to the host
        a0 = np.ascontiguousarray(ia)
        env._check(env.L.rs_step(env.h, a0.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))

        def run(k):
            nonlocal rounds
            for _ in range(k):
                agent.step_resident(env, count_rounds=args.count_rounds)
                rounds += agent.rounds_last or 0
                env.step_resident()
    run(args.warmup)
    rounds, viol = 0, 0.0
    env.synchronize()
    agent.synchronize()
    agent.set_kernel_timing(True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    env.synchronize()
    agent.synchronize()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = max_over_ranks(time.perf_counter() - t0, device='cuda')
    sizes = [agent.learner(0, s)['m'] for s in range(len(dims))]
    ph = agent.phase_times_ms()
    if not args.host_loop:
        f = env.fetch()
        viol = float(f['violations'].sum(axis=1).mean()) * args.steps
        action = f['actions']
    if rank == 0:
        print(json.dumps({'config': 'scenario_%d, %d envs x %d GPUs, shared KBRL dictionary per slice, RCCL all_gather merge'
                                    % (args.scenario, N, world), 'env_steps_per_s': world * N * args.steps / dt,
                          'ms_per_step': 1e3 * dt / args.steps, 'exchange_rounds_per_step': (rounds / args.steps) if (args.host_loop or args.count_rounds) else None,
                          'loop': 'host buffers' if args.host_loop else 'device-resident (kb_shared_step_resident)',
                          'capacity': args.capacity, 'scan_kernel_ms': ph['update_ms'], 'select_kernel_ms': ph['select_ms'],
                          'dictionary_sizes': sizes, 'violations_per_env_step%s' % ('' if args.host_loop else '_last'): viol / args.steps,
                          'mean_prbs_last': float(action.sum(axis=1).mean())}))
    env.close(); agent.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

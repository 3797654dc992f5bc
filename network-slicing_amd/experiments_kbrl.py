#!/usr/bin/env python3
"""Counterpart of the reference's experiments_kbrl.py (reference experiments_kbrl.py:22-70): evaluates
KBRL in the three scenarios and stores results/scenario_N/KBRL_xx/results_K.npz with the same keys
and dtypes (kbrl_control.py:148-155), so the reference's plot_results.py reads them unchanged.

The reference fans its RUNS independent runs over a process pool (experiments_kbrl.py:67-70); that is exactly the
replica axis the simulator is batched over, so here all runs of one (scenario, accuracy range) are the replicas of
ONE VecRanSlice + VecKBRL pair, advanced by the device-resident closed loop with the per-step histories recorded on
the device (BatchedEvaluator).  Run i is seeded exactly as Evaluator.evaluate(i) seeds it, so its results file is
the one the run-by-run path writes (tests/test_gpu_kbrl.py::test_batched_evaluator_equals_run_by_run).

The whole grid (3 scenarios x 2 accuracy ranges x RUNS) is ONE job: evaluate_grid advances all six cells in one host loop,
each on its own streams, so their kernels share the chip.

  python experiments_kbrl.py [--steps 50400] [--runs 30] [--scenarios 0 1 2] [--serial | --cell-by-cell]
"""
import argparse
import os
from itertools import product

from numpy import savez
from numpy.random import default_rng

import numpy as np

import scenario_creator as sc
from scenario_creator import create_env, create_kbrl_agent

STEPS = 50400
RUNS = 30
CHUNK = int(os.environ.get('KBRL_EVAL_CHUNK', '64'))   # closed-loop steps enqueued per call of the batched evaluators (kb_run_resident)
GRAPH = os.environ.get('KBRL_EVAL_GRAPH', '1') != '0'   # ... replayed from a captured hipGraph
scenarios = [0, 1, 2]
accuracy_list = [[0.97, 0.99], [0.99, 0.999]]
name = 'KBRL'


class Evaluator():
    def __init__(self, scenario, a_range, steps=STEPS, out_dir='./results'):
        self.scenario = scenario
        self.a_range = a_range
        self.steps = steps
        a = int(a_range[0] * 100)
        self.path = '{}/scenario_{}/{}_{}/'.format(out_dir, scenario, name, a)
        os.makedirs(self.path, exist_ok=True)

    def evaluate(self, i):
        rng = default_rng(seed=i)
        node_env = create_env(rng, self.scenario)
        kbrl_agent = create_kbrl_agent(rng, self.scenario, accuracy_range=self.a_range)
        results = kbrl_agent.run(node_env, self.steps)
        file_path = '{}results_{}.npz'.format(self.path, i)
        savez(file_path, **results)
        print('run {}: Results saved!'.format(i))
        return file_path


class BatchedEvaluator(Evaluator):
    """All runs of one (scenario, accuracy range) at once: run i = replica i."""

    def _setup(self, runs, device=0, capacity=16384, pool_bytes=32 << 30):
        """environment + agent of this cell with run i as replica i, reset and given the first action; nothing waits"""
        import ctypes as C
        from ranslice import config as _c
        from ranslice.kbrl_dev import VecKBRL
        from ranslice.vec_env import VecRanSlice, default_fading
        runs = list(runs)
        n = len(runs)
        scn = sc.scenarios[self.scenario]
        n_embb, n_mmtc, n_prbs = scn['n_embb'], scn['n_mmtc'], scn['n_prbs']
        # the draws of Evaluator.evaluate(i), in its order: create_env (one seed), create_kbrl_agent (initial action and
        # security factor per learner, eMBB learners first; then the agent's tie-break seed)
        env_seeds = np.zeros(n, dtype=np.uint64)
        ag_seeds = np.zeros(n, dtype=np.uint64)
        ia = np.zeros((n, n_embb + n_mmtc), dtype=np.int32)
        sf = np.zeros((n, n_embb + n_mmtc), dtype=np.int32)
        for k, i in enumerate(runs):
            rng = default_rng(seed=i)
            env_seeds[k] = int(rng.integers(0, 2 ** 63 - 1))
            for s in range(n_embb):
                ia[k, s] = rng.integers(sc.embb_a[0], sc.embb_a[1])
                sf[k, s] = rng.integers(sc.embb_sec[0], sc.embb_sec[1])
            for s in range(n_embb, n_embb + n_mmtc):
                ia[k, s] = rng.integers(sc.mmtc_a[0], sc.mmtc_a[1])
                sf[k, s] = rng.integers(sc.mmtc_sec[0], sc.mmtc_sec[1])
            ag_seeds[k] = int(rng.integers(0, 2 ** 63 - 1))
        fading = sc._FADING if sc._FADING is not None else default_fading()
        cfg = _c.make_config(self.scenario, n_envs=n)
        env = VecRanSlice(n_envs=n, cfg=cfg, fading=fading, device=device)
        dims = [len(sc.state_variables_embb)] * n_embb + [len(sc.state_variables_mmtc)] * n_mmtc
        agent = VecKBRL(n, dims, n_prbs, alfa=sc.alfa, accuracy_range=tuple(self.a_range), capacity=capacity,
                        device=device, pool_bytes=pool_bytes)
        env.reset(seeds=env_seeds)
        agent.reset(ia, sf, seeds=ag_seeds)
        agent.history_begin(self.steps)
        # KBRL_Control.run (kbrl_control.py:126-141): the first action is the learners' initial action
        a0 = np.ascontiguousarray(ia)
        env._check(env.L.rs_step(env.h, a0.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, None))
        self._ctx = dict(runs=runs, env=env, agent=agent, capacity=capacity)

    def _advance(self, i, k=1, graph=True):
        """steps i .. i + k - 1 of KBRL_Control.run's loop, enqueued on this cell's own streams (no host wait).  Every step
        but the run's last is an agent step followed by a simulator step: those go out k at a time (kb_run_resident, two
        captured steps replayed as a hipGraph)."""
        c = self._ctx
        paired = min(k, self.steps - 1 - i)       # steps followed by a simulator step
        if paired > 0:
            c['agent'].run_resident(c['env'], paired, graph=graph)
        if i + k >= self.steps:
            c['agent'].step_resident(c['env'])    # the last step: update_control + select_action + its history column

    def _finish(self, verbose=True):
        c = self._ctx
        env, agent, runs, capacity = c['env'], c['agent'], c['runs'], c['capacity']
        hist = agent.history_fetch()          # waits for the loop; raises on a device-side error
        env.fetch()                           # surfaces simulator capacity errors
        assert hist['recorded'] == self.steps
        sizes = agent.dictionary_sizes()
        pool = agent.pool()
        self.last_run = dict(max_dictionary=int(sizes.max()), mean_dictionary=float(sizes.mean()), pool=pool)
        if pool['saturated'] or pool['pool_full']:
            import warnings
            # the replicas that carry the flags (err bits 8 / 16), not a guess from the sizes: a dictionary that found the pool
            # exhausted is below its capacity
            flagged = agent.flagged_replicas()
            warnings.warn('KBRL dictionaries of runs %s reached their capacity (%d landmarks) and of runs %s found the pool '
                          'exhausted (%.1f of %.1f MB in use): they projected further samples instead of growing -- raise '
                          'capacity / pool_bytes'
                          % (sorted(runs[k] for k in flagged['saturated']), capacity,
                             sorted(runs[k] for k in flagged['pool_full']),
                             pool['used_bytes'] / 2 ** 20, pool['total_bytes'] / 2 ** 20))
        files = []
        for k, i in enumerate(runs):
            results = {'reward': hist['reward'][k], 'resources': hist['resources'][k], 'hits': hist['hits'][k],
                       'adjusted': hist['adjusted'][k], 'SLA': hist['SLA'][k], 'violation': hist['violation'][k]}
            file_path = '{}results_{}.npz'.format(self.path, i)
            savez(file_path, **results)
            files.append(file_path)
            if verbose:
                print('run {}: mean resources = {}, total violations = {}. Results saved!'.format(
                    i, results['resources'].mean(), results['violation'].sum()))
        env.close()
        agent.close()
        self._ctx = None
        return files

    def save_checkpoint(self, path, next_step):
        """the cell's whole state (simulator + agents + the histories recorded so far) and the step to go on from"""
        c = self._ctx
        # both streams at rest before either state is read (the loop's last launches may sit on the other handle's stream)
        c['env'].synchronize()
        c['agent'].synchronize()
        env_blob = c['env'].save_state()
        agent_blob = c['agent'].save_state()
        # written beside the old checkpoint and moved onto it: a crash during the write -- the case this exists for -- leaves
        # the previous checkpoint intact instead of a torn .npz
        tmp = path + '.tmp.npz'
        np.savez(tmp, next_step=np.int64(next_step), steps=np.int64(self.steps), runs=np.asarray(c['runs'], dtype=np.int64),
                 env=env_blob, agent=agent_blob)
        os.replace(tmp, path)

    def load_checkpoint(self, path):
        """-> the step to go on from.  The cell must have been set up for the same runs, capacity and pool."""
        z = np.load(path)
        c = self._ctx
        if int(z['steps']) != self.steps or list(z['runs']) != list(c['runs']):
            raise ValueError('checkpoint %s belongs to another evaluation (steps / runs differ)' % path)
        c['env'].load_state(z['env'])
        c['agent'].load_state(z['agent'])
        return int(z['next_step'])

    def evaluate_all(self, runs, device=0, capacity=16384, pool_bytes=32 << 30, verbose=True, checkpoint=None,
                     checkpoint_every=0, stop_after=None, graph=GRAPH):
        """checkpoint: a .npz path.  If it exists the evaluation resumes from it; with checkpoint_every = k it is rewritten
        every k steps (a 50,400-step evaluation that dies loses at most k steps -- the reference starts over).  stop_after = s
        ends the call after step s - 1 with the checkpoint written and no result files (tests, planned interruptions)."""
        self._setup(runs, device=device, capacity=capacity, pool_bytes=pool_bytes)
        first = 0
        ck = checkpoint if (checkpoint is None or checkpoint.endswith('.npz')) else checkpoint + '.npz'
        if ck and os.path.exists(ck):
            first = self.load_checkpoint(ck)
        i = first
        while i < self.steps:
            # up to CHUNK steps per call, cut at the next checkpoint / stop boundary
            k = min(CHUNK, self.steps - i)
            if stop_after is not None and i < stop_after:
                k = min(k, stop_after - i)
            if ck and checkpoint_every:
                k = min(k, checkpoint_every - i % checkpoint_every)
            self._advance(i, k, graph=graph)
            i += k
            last = stop_after is not None and i == stop_after
            if ck and (last or (checkpoint_every and i % checkpoint_every == 0 and i < self.steps)):
                self.save_checkpoint(ck, i)
            if last:
                c = self._ctx
                c['env'].close()
                c['agent'].close()
                self._ctx = None
                return None
        return self._finish(verbose=verbose)


def evaluate_grid(cells, runs, steps=STEPS, out_dir='./results', device=0, capacity=16384, pool_bytes=16 << 30, verbose=False,
                  graph=False):
    """The reference's whole experiment (experiments_kbrl.py:57-70: every scenario x accuracy range x run) as ONE job on
    one GPU: a BatchedEvaluator per cell, each with its environment and agents on streams of their own, all advanced in
    the same host loop -- a cell of 30 runs leaves most of the chip idle (a handful of waves per kernel), so the cells'
    kernels run beside each other.  Nothing in the loop waits for the device.  (graph=False: graph launches of different
    handles do not overlap on this runtime -- 22.4 s per 6,000 steps of the six cells against 16.1 s with plain launches,
    profiles/r04_w_grid_graph.txt -- while a single cell gains 5 % from the graph.)  Returns {(scenario, a_lo): [files]}, file
    for file what evaluate_all writes cell by cell (tests/test_gpu_kbrl.py::test_grid_of_cells_equals_cell_by_cell)."""
    evs = []
    for scenario, a_range in cells:
        ev = BatchedEvaluator(scenario, a_range, steps=steps, out_dir=out_dir)
        ev._setup(runs, device=device, capacity=capacity, pool_bytes=pool_bytes)
        evs.append(ev)
    for i in range(0, steps, CHUNK):
        for ev in evs:
            ev._advance(i, min(CHUNK, steps - i), graph=graph)
    out = {}
    for (scenario, a_range), ev in zip(cells, evs):
        out[(scenario, a_range[0])] = ev._finish(verbose=verbose)
    evaluate_grid.last_runs = {(c[0], c[1][0]): ev.last_run for c, ev in zip(cells, evs)}
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=STEPS)
    ap.add_argument('--runs', type=int, default=RUNS)
    ap.add_argument('--scenarios', type=int, nargs='*', default=scenarios)
    ap.add_argument('--out', default='./results')
    ap.add_argument('--serial', action='store_true', help='one run at a time through the N=1 drop-in classes')
    ap.add_argument('--cell-by-cell', action='store_true', help='one (scenario, accuracy range) after the other instead of all at once')
    args = ap.parse_args()
    cells = list(product(args.scenarios, accuracy_list))
    if args.serial:
        for scenario, a_range in cells:
            evaluator = Evaluator(scenario, a_range, steps=args.steps, out_dir=args.out)
            for run in range(args.runs):
                evaluator.evaluate(run)
    elif args.cell_by_cell:
        for scenario, a_range in cells:
            BatchedEvaluator(scenario, a_range, steps=args.steps, out_dir=args.out).evaluate_all(range(args.runs))
    else:
        evaluate_grid(cells, range(args.runs), steps=args.steps, out_dir=args.out, verbose=True)

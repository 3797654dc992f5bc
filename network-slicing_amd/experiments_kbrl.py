#!/usr/bin/env python3
"""Counterpart of the reference's experiments_kbrl.py (reference experiments_kbrl.py:22-70): evaluates
KBRL in the three scenarios and stores results/scenario_N/KBRL_xx/results_K.npz with the same keys
and dtypes (kbrl_control.py:148-155), so the reference's plot_results.py reads them unchanged.

  python experiments_kbrl.py [--steps 50400] [--runs 30] [--scenarios 0 1 2]
"""
import argparse
import os
from itertools import product

from numpy import savez
from numpy.random import default_rng

from scenario_creator import create_env, create_kbrl_agent

STEPS = 50400
RUNS = 30
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


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=STEPS)
    ap.add_argument('--runs', type=int, default=RUNS)
    ap.add_argument('--scenarios', type=int, nargs='*', default=scenarios)
    ap.add_argument('--out', default='./results')
    args = ap.parse_args()
    for scenario, a_range in product(args.scenarios, accuracy_list):
        evaluator = Evaluator(scenario, a_range, steps=args.steps, out_dir=args.out)
        for run in range(args.runs):
            evaluator.evaluate(run)

"""KBRL_Control / Learner with the reference's surface (reference kbrl_control.py:12-157).

This is the N=1 host view of ranslice.kbrl_dev.VecKBRL: select_action, update_control and the
Projectron dictionaries all run on the GPU (kb_* C ABI); Python only carries the per-step vectors.
"""
import numpy as np

from ranslice.kbrl_dev import VecKBRL

DEBUG = True


class Learner:
    """record (kbrl_control.py:12-21)"""

    def __init__(self, algorithm, indexes, initial_action, security_factor):
        self.algorithm = algorithm
        self.indexes = indexes
        self.initial_action = initial_action
        self.security_factor = security_factor
        self.step = 1


class KBRL_Control:
    def __init__(self, learners, n_prbs, alfa=0.05, accuracy_range=[0.99, 0.999], capacity=4096, device=0, seed=0):
        self.learners = learners
        self.accuracy_range = accuracy_range
        self.n_slices = len(learners)
        self.n_prbs = n_prbs
        self.alfa = alfa
        self.adjusted = 0
        dims = []
        expect = 0
        for h in learners:
            idx = h.indexes
            if idx.start != expect or (idx.step not in (None, 1)):
                raise ValueError('learner state slices must be contiguous and in order (scenario_creator.py:224-235)')
            dims.append(idx.stop - idx.start)
            expect = idx.stop
        alg0 = learners[0].algorithm
        self._dev = VecKBRL(1, dims, n_prbs, alfa=alfa, accuracy_range=tuple(accuracy_range),
                            gamma=alg0.kernel.gamma, eta=alg0.eta, capacity=capacity, device=device)
        self._dev.reset([[int(h.initial_action) for h in learners]], [[int(h.security_factor) for h in learners]],
                        seeds=np.array([seed], dtype=np.uint64))
        for s, h in enumerate(learners):
            h.algorithm._bind(self._dev, 0, s)
        self.action = np.array([h.initial_action for h in learners], dtype=np.int16)

    # ---- public fields of the reference, read from the device
    @property
    def security_factors(self):
        return self._dev.control(with_accuracies=False)['security_factors'][0].astype(np.int16)

    @property
    def margins(self):
        return self._dev.control(with_accuracies=False)['margins'][0].astype(np.int16)

    @property
    def accuracies(self):
        return self._dev.control()['accuracies'][0]

    def select_action(self, state):
        """kbrl_control.py:41-73 -> (action int16[S], adjusted)"""
        action, adjusted = self._dev.select_action(np.asarray(state, dtype=np.float32)[None, :])
        self.action = action[0].astype(np.int16)
        return self.action, int(adjusted[0])

    def adjust_action(self, action, assigned_prbs, n_prbs):
        """kbrl_control.py:75-78 (kept for API parity; select_action applies it on the device)"""
        relative_p = np.asarray(action) / assigned_prbs
        new_action = np.array([np.floor(n_prbs * p) for p in relative_p], dtype=np.int16)
        return new_action, action - new_action

    def update_control(self, state, action, reward):
        """kbrl_control.py:80-114 -> hits int16[S]; `reward` is the SLA label vector"""
        self._dev.set_adjusted([int(self.adjusted)])
        hits = self._dev.update_control(np.asarray(state, dtype=np.float32)[None, :],
                                        np.asarray(action, dtype=np.int32)[None, :],
                                        np.asarray(reward, dtype=np.int32)[None, :])
        return hits[0].astype(np.int16)

    # result keys of run() and their dtypes: the npz schema plot_results.py reads (kbrl_control.py:119-124,148-155)
    _RUN_SCHEMA = (('reward', np.float64), ('resources', np.int16), ('hits', np.int16), ('adjusted', np.int16),
                   ('SLA', np.int16), ('violation', np.int16))

    def run(self, system, steps, learning_time=-1):
        """The control loop of kbrl_control.py:116-157 for one environment: act, observe the SLA labels, learn from them
        (while learning_time < steps, as the reference tests it), choose the next allocation.  Returns the reference's dict:
        one column per step of reward f64, resources / adjusted / SLA / violation int16 and hits int16[S, steps]."""
        n_learners = len(self.action)
        out = {key: np.zeros((n_learners, steps) if key == 'hits' else steps, dtype=dt) for key, dt in self._RUN_SCHEMA}
        alloc = self.action
        obs = system.reset()
        last_hits = np.zeros(n_learners, dtype=np.int16)
        learning = learning_time < steps
        for t in range(steps):
            nxt, rew, _, info = system.step(alloc)
            labels = info['SLA_labels']
            if learning:
                last_hits = self.update_control(obs, alloc, labels)
            alloc, self.adjusted = self.select_action(nxt)
            obs = nxt
            out['reward'][t] = rew
            out['SLA'][t] = labels.sum()
            out['violation'][t] = info['total_violations']
            out['resources'][t] = alloc.sum()       # the allocation chosen FOR the next step, as the reference records it
            out['adjusted'][t] = self.adjusted
            out['hits'][:, t] = last_hits
        pool = self._dev.pool()
        if pool['saturated'] or pool['pool_full']:
            import warnings
            sizes = self._dev.dictionary_sizes()
            warnings.warn('KBRL dictionaries (sizes %s) could not grow any further -- capacity %d landmarks, dictionary pool '
                          '%.1f of %.1f MB in use -- and projected the samples they would have added onto their span (the '
                          "reference's SVvariable grows without bound): pass a larger capacity / pool_bytes to "
                          'KBRL_Control / create_kbrl_agent (capacity up to 65536; the pool is bounded by device memory)'
                          % (sizes[0].tolist(), self._dev.capacity, pool['used_bytes'] / 2 ** 20, pool['total_bytes'] / 2 ** 20))
        # the summary the reference's run prints (kbrl_control.py:143-146): same four `label = value` lines on stdout, so that log
        # parsers written against it keep working
        for label, value in (('mean resources', out['resources'].mean()), ('total violations', out['violation'].sum()),
                             ('mean adjusted', out['adjusted'].mean()), ('mean accuracy', out['hits'].mean(axis=1))):
            print('%s = %s' % (label, value))
        return out

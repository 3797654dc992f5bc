"""NodeB: the N=1 host view of the batched simulator, with the reference's method surface
(reference node_b.py:8-91).  All slots run in the HIP kernels; this object only forwards one
action vector and unpacks the result."""
import numpy as np

from ranslice.config import STATE_VARIABLES_EMBB
from ranslice.sharding import replica_seed


class NodeB:
    def __init__(self, vec_env, env_index=0):
        if vec_env.n_envs != 1 or env_index != 0:
            raise ValueError('NodeB is the N=1 view of the batched simulator; use VecRanSlice for batches')
        self.vec = vec_env
        self.k = env_index
        self.n_slices_l1 = vec_env.n_slices
        self.slots_per_step = vec_env.cfg.slots_per_step
        self.n_prbs = vec_env.n_prbs
        self.slot_length = vec_env.cfg.slot_length
        self.steps = 0
        self._seed = None
        self._resets = 0

    def seed(self, seed):
        self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._resets = 0

    def get_n_variables(self):
        return self.vec.n_variables

    def reset(self):
        """node_b.py:17-22.  The reference keeps drawing from one generator across resets, so every episode differs;
        here the first reset uses the seed it was given and every later one a fresh stream derived from it."""
        self.steps = 0
        seeds = None
        if self._seed is not None:
            s = self._seed if self._resets == 0 else replica_seed(self._seed, self._resets)
            seeds = np.array([s], dtype=np.uint64)
        elif self._resets:
            seeds = np.array([replica_seed(self.vec.base_seed, 1 << 32 | self._resets)], dtype=np.uint64)
        self._resets += 1
        return self.vec.reset(seeds=seeds)[self.k]

    def get_info(self, violations=0, SLA_labels=0, l1=None, action=None):
        """node_b.py:46-49: l1_info is a list (per L1 slice) of {slice_ran_index: info dict}"""
        l1_info = []
        cfg = self.vec.cfg

        def ran_info(s):
            row = l1[s] if l1 is not None else np.zeros(10)
            if s < cfg.n_embb:
                return {k: row[i] for i, k in enumerate(STATE_VARIABLES_EMBB)}
            return {'delay': row[0], 'avg_rep': row[1], 'devices': row[2]}
        if self.vec.multiplexed:   # one L1 slice holds several RAN slices: {index in the L1 slice: info}
            if cfg.n_embb:
                l1_info.append({i: ran_info(i) for i in range(cfg.n_embb)})
            if cfg.n_mmtc:
                l1_info.append({i: ran_info(cfg.n_embb + i) for i in range(cfg.n_mmtc)})
        else:
            for s in range(self.n_slices_l1):
                l1_info.append({0: ran_info(s)})
        return {'l1_info': l1_info, 'SLA_labels': SLA_labels, 'violations': violations,
                'n_prbs': list(action) if action is not None else [0] * self.n_slices_l1}

    def step(self, action):
        """node_b.py:59-91; returns (state, info)"""
        action = np.asarray(action)
        if len(action) != self.n_slices_l1:
            raise ValueError('The action must contain as many elements as slices!')
        obs, reward, _, info = self.vec.step(action.reshape(1, -1))
        self.steps += 1
        self._last_reward = float(reward[self.k])
        out = self.get_info(violations=info['violations'][self.k].astype(np.int64),
                            SLA_labels=info['SLA_labels'][self.k].astype(np.int64),
                            l1=self.vec.l1_info()[self.k], action=action)
        return obs[self.k], out

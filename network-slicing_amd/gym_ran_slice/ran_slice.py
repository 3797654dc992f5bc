"""gym_ran_slice.RanSlice with the reference's surface (reference
gym-ran_slice/gym_ran_slice/ran_slice.py:15-54): old-gym reset() -> state and
step(action) -> (state, float reward, False, info)."""
import numpy as np

from ranslice.gymshim import Env, spaces


class RanSlice(Env):
    def __init__(self, node_b=None, penalty=100):
        self.node_b = node_b
        self.penalty = penalty
        self.n_prbs = node_b.n_prbs
        self.n_slices = node_b.n_slices_l1
        self.n_variables = node_b.get_n_variables()
        self.action_space = spaces.Box(low=0, high=self.n_prbs, shape=(self.n_slices,), dtype=np.int64)
        self.observation_space = spaces.Box(low=-float('inf'), high=+float('inf'), shape=(self.n_variables,),
                                            dtype=np.float64)
        if float(penalty) != float(node_b.vec.penalty):
            raise ValueError('penalty must match the one the simulator was created with')

    def reset(self):
        return self.node_b.reset()

    def step(self, action):
        state, info = self.node_b.step(action)
        total_violations = info['violations'].sum()
        info['total_violations'] = total_violations
        # reward is evaluated on the device with ran_slice.py:45-52's rule
        return state, float(self.node_b._last_reward), False, info

    def render(self):
        pass

"""SVvariable / Projectron with the reference's surface (reference algorithms/projectron.py:3-64).

The dictionary (landmarks, coefficients, K^-1) lives on the GPU inside a kb_* agent; predict(x) and
update(x, y) are kb_predict / kb_update calls.  A Projectron created on its own owns a private
1-learner agent; one handed to kbrl_control.KBRL_Control is re-bound to that agent's learner."""
import numpy as np

from ranslice.kbrl_dev import VecKBRL


class SVvariable:
    """growable landmark/coefficient store (projectron.py:3-21); a view of the device dictionary"""

    def __init__(self):
        self._owner = None

    @property
    def counter(self):
        return self._owner._get()['m'] if self._owner and self._owner._bound() else 0

    @property
    def landmarks(self):
        d = self._owner._get()
        return d['landmarks'][0] if d['m'] == 1 else d['landmarks']  # 1-D while single (projectron.py:16-17)

    @property
    def coeff(self):
        return self._owner._get()['coeff']


class Projectron:
    def __init__(self, kernel, eta=0.1, capacity=4096):
        self.kernel = kernel
        self.sv = kernel.sv
        self.eta = eta
        self.capacity = capacity
        self.f = 0.0
        self._agent = None
        self._e = 0
        self._s = 0
        kernel._owner = self
        self.sv._owner = self

    # ---- binding to a device learner
    def _bound(self):
        return self._agent is not None

    def _bind(self, agent, e, s):
        self._agent, self._e, self._s = agent, e, s

    def _ensure(self, x):
        if self._agent is None:
            self._agent = VecKBRL(1, [len(x) - 1], 200, gamma=self.kernel.gamma, eta=self.eta,
                                  capacity=self.capacity)
            self._agent.reset([[0]], [[0]])
            self._e = self._s = 0

    def _get(self, with_kinv=False):
        return self._agent.learner(self._e, self._s, with_kinv=with_kinv)

    def _kernel_row(self):
        """K_f of the last predict (projectron.py:34)"""
        return self._agent.kernel_row(self._e, self._s)

    @property
    def counter(self):
        return self._get()['m'] if self._bound() else 0

    @property
    def Kinv(self):
        return self._get(with_kinv=True)['kinv']

    # ---- reference surface
    def predict(self, x):
        """projectron.py:32-37"""
        x = np.asarray(x, dtype=np.float64)
        self._ensure(x)
        y, self.f = self._agent.predict(self._e, self._s, x)
        return y

    def update(self, x, y):
        """projectron.py:39-60; uses the f / k cached by the preceding predict(x) (reference Q12)"""
        x = np.asarray(x, dtype=np.float64)
        self._ensure(x)
        self._agent.update(self._e, self._s, x, int(y))

    def get_set_size(self):
        """projectron.py:62-64: landmarks.shape[0] (the vector length while a single landmark is held)"""
        d = self._get()
        if d['m'] == 0:
            raise AttributeError("'SVvariable' object has no attribute 'landmarks'")
        return d['landmarks'].shape[1] if d['m'] == 1 else d['m']

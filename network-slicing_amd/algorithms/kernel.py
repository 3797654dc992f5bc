"""GaussianKernel with the reference's surface (reference algorithms/kernel.py:3-34).  The kernel
row k(x) = exp(-gamma ||l_j - x||^2) and f = k . coeff are evaluated on the GPU (kb_predict); this
object is the handle the reference's call sites expect."""
import numpy as np


class GaussianKernel:
    def __init__(self, sv, gamma=1.0):
        self.sv = sv
        self.gamma = gamma
        self._owner = None  # Projectron that binds this kernel to a device learner

    def k_eval(self, x1, x2):
        """scalar kernel (kernel.py:8-11); k(x, x) = 1"""
        dist = (np.asarray(x1, dtype=np.float64) - np.asarray(x2, dtype=np.float64)) ** 2
        return float(np.exp(-self.gamma * dist.sum()))

    def predict(self, x):
        """-> (y, f, k) as kernel.py:22-28; evaluated by the device learner"""
        if self._owner is None:
            raise RuntimeError('GaussianKernel must be wrapped in a Projectron (device learner)')
        y = self._owner.predict(x)
        return y, self._owner.f, self._owner._kernel_row()

    def k(self, x):
        """affinity row k_j = exp(-gamma ||l_j - x||^2) (kernel.py:13-20): computed by the device learner's predict
        (the row is what it caches for the following update), fetched from there"""
        if self._owner is None:
            raise RuntimeError('GaussianKernel must be wrapped in a Projectron (device learner)')
        self._owner.predict(x)
        return self._owner._kernel_row()


class SV:
    """fixed-budget store (kernel.py:36-50).  Never instantiated by any scenario (create_kbrl_agent uses SVvariable,
    scenario_creator.py:217) and its own demo is broken in the reference (kernel.py:58 unpacks two of predict's
    three values): out of scope, kept as a name that says so."""

    def __init__(self, dimension, budget):
        raise NotImplementedError('SV (fixed budget) is out of scope: create_kbrl_agent uses SVvariable')

"""Import and instrument the reference implementation IN THIS CONTAINER ONLY.

Used by tools/gen_golden.py to produce tests/golden/*.npz.  Applies the three shims of
SURVEY.md §8c (numpy alias, gym stub, cwd with datasets/) and wraps every source of
randomness with a recording proxy so that the full "random tape" of a run is captured
in call order.  Nothing here is shipped as product and nothing here is copied from the
reference: it only *calls* it.
"""
import os
import sys
import shutil
import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))

# tape kinds (must match oracle/rs_oracle.h RS_TAPE_*)
K_RANDOM, K_EXP, K_INT, K_CHOICE, K_NORMAL, K_GEXP, K_GCHOICE = range(7)


class Tape:
    def __init__(self):
        self.kind = []
        self.val = []
        self.on = True

    def add(self, kind, v):
        if self.on:
            self.kind.append(kind)
            self.val.append(float(v))

    def clear(self):
        self.kind.clear()
        self.val.clear()

    def arrays(self):
        return np.asarray(self.kind, dtype=np.uint8), np.asarray(self.val, dtype=np.float64)


class TapeRNG:
    """Duck-types the subset of numpy Generator the reference uses."""

    def __init__(self, rng, tape):
        self._rng = rng
        self.tape = tape

    def random(self, size=None):
        v = self._rng.random(size)
        if size is None:
            self.tape.add(K_RANDOM, v)
        else:
            for x in np.ravel(v):
                self.tape.add(K_RANDOM, x)
        return v

    def exponential(self, scale=1.0):
        v = self._rng.exponential(scale)
        self.tape.add(K_EXP, v)
        return v

    def integers(self, *a, **k):
        v = self._rng.integers(*a, **k)
        self.tape.add(K_INT, v)
        return v

    def choice(self, a):
        v = self._rng.choice(a)
        self.tape.add(K_CHOICE, v)
        return v

    def normal(self, loc=0.0, scale=1.0):
        v = self._rng.normal(loc, scale)
        self.tape.add(K_NORMAL, v)
        return v


class _GlobalRandomShim:
    def __init__(self, tape):
        self.tape = tape

    def exponential(self, scale=1.0):
        v = np.random.exponential(scale)
        self.tape.add(K_GEXP, v)
        return v

    def choice(self, a):
        v = np.random.choice(a)
        self.tape.add(K_GCHOICE, v)
        return v


class _NumpyShim:
    """Module-like object: `.random` records, everything else is numpy."""

    def __init__(self, tape):
        self.random = _GlobalRandomShim(tape)

    def __getattr__(self, name):
        return getattr(np, name)


def setup(workdir, fading_tables):
    """fading_tables: list of 3 float64 arrays [100][T] (may contain NaN)."""
    if not hasattr(np, 'int'):
        np.int = int
    if not hasattr(np, 'float'):
        np.float = float
    os.makedirs(os.path.join(workdir, 'datasets'), exist_ok=True)
    shutil.copy(os.path.join(REF, 'datasets', 'mcs_codeset.csv'), os.path.join(workdir, 'datasets'))
    names = ['EPA_3kmph', 'ETU_3kmph', 'EVA_60kmph']
    for n, tab in zip(names, fading_tables):
        np.savetxt(os.path.join(workdir, 'datasets', 'fading_trace_%s.csv' % n), tab,
                   delimiter=',', fmt='%.17g')
    os.chdir(workdir)
    for p in (os.path.join(HERE, 'gym_stub'), REF, os.path.join(REF, 'gym-ran_slice')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import matplotlib
    matplotlib.use('Agg')


def install_tape(tape):
    import traffic_generators
    import algorithms.kernel as akernel
    traffic_generators.np = _NumpyShim(tape)
    akernel.np = _NumpyShim(tape)

"""create_env / create_kbrl_agent with the reference's signatures (reference
scenario_creator.py:100-183, 197-238), wiring the MI355X-native simulator and agent instead of the
Python object graph.  Constants live in ranslice/config.py and are re-exported here under the
reference's names."""
import numpy as np

from ranslice import config as _c
from ranslice.gymshim import make
from ranslice.vec_env import VecRanSlice, default_fading
from node_b import NodeB
from kbrl_control import KBRL_Control, Learner
from algorithms.kernel import GaussianKernel
from algorithms.projectron import SVvariable, Projectron
import gym_ran_slice  # noqa: F401  (registers RanSlice-v1)

scenarios = [dict(s) for s in _c.SCENARIOS]
CBR_description = _c.CBR_DESCRIPTION
VBR_description = _c.VBR_DESCRIPTION
SLA_embb = _c.SLA_EMBB
state_variables_embb = _c.STATE_VARIABLES_EMBB
MTC_description = _c.MTC_DESCRIPTION
state_variables_mmtc = _c.STATE_VARIABLES_MMTC
SLA_mmtc = _c.SLA_MMTC
alfa = _c.KBRL_ALFA
embb_sec, embb_a, mmtc_sec, mmtc_a = _c.EMBB_SEC, _c.EMBB_A, _c.MMTC_SEC, _c.MMTC_A

_FADING = None


def set_fading(tables):
    """Install the three fading traces (reference layout [PRB][time]) used by create_env; by default
    the build's seeded synthetic traces stand in for the absent ns-3 files."""
    global _FADING
    _FADING = tables


def create_env(rng, n, slots_per_step=50, propagation_type='macro_cell_urban_2GHz', L1_level=True, penalty=100,
               device=0):
    """scenario_creator.py:100-183.  `rng` seeds the replica's counter-based streams (one draw).
    L1_level=False multiplexes the slices in the L1 (scenario_creator.py:168-177): the env then has one action entry
    per L1 slice -- one for all eMBB RAN slices, one for all mMTC ones."""
    global _FADING
    if _FADING is None:
        _FADING = default_fading()
    cfg = _c.make_config(n, n_envs=1, slots_per_step=slots_per_step, propagation_type=propagation_type,
                         penalty=penalty, L1_level=L1_level)
    vec = VecRanSlice(n_envs=1, cfg=cfg, fading=_FADING, device=device)
    node = NodeB(vec)
    node.seed(int(rng.integers(0, 2 ** 63 - 1)))
    return make('gym_ran_slice:RanSlice-v1', node_b=node, penalty=penalty)


def create_kbrl_agent(rng, n, accuracy_range=[0.99, 0.999], device=0, capacity=4096):
    """scenario_creator.py:197-238: one Projectron learner per slice, random initial action/offset.
    capacity: most landmarks a learner's dictionary may hold -- a limit, not a reservation: dictionaries take their
    storage from a pool 64 landmarks at a time (the authors' 50,400-step runs end at 45-305 landmarks on average, 1,025 at
    most).  A dictionary that reaches it projects instead of growing and KBRL_Control.run warns."""
    sc = scenarios[n]
    n_prbs, n_embb, n_mmtc = sc['n_prbs'], sc['n_embb'], sc['n_mmtc']
    embb_dim, mmtc_dim = len(state_variables_embb), len(state_variables_mmtc)
    learners = []
    i = 0
    for _ in range(n_embb):
        algorithm = Projectron(GaussianKernel(SVvariable(), 1))
        initial_action = rng.integers(embb_a[0], embb_a[1])
        sec = rng.integers(embb_sec[0], embb_sec[1])
        learners.append(Learner(algorithm, slice(i, i + embb_dim), initial_action, sec))
        i += embb_dim
    for _ in range(n_mmtc):
        algorithm = Projectron(GaussianKernel(SVvariable(), 1))
        initial_action = rng.integers(mmtc_a[0], mmtc_a[1])
        sec = rng.integers(mmtc_sec[0], mmtc_sec[1])
        learners.append(Learner(algorithm, slice(i, i + mmtc_dim), initial_action, sec))
        i += mmtc_dim
    seed = int(rng.integers(0, 2 ** 63 - 1))
    return KBRL_Control(learners, n_prbs, alfa=alfa, accuracy_range=accuracy_range, device=device, seed=seed,
                        capacity=capacity)

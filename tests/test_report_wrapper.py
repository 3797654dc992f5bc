"""ReportWrapper-style normalised interface against vectors recorded through the
reference's wrapper.ReportWrapper (fixture G12)."""
import os

import numpy as np

from ranslice.report import normalise_obs, simplex_to_prbs


def test_g12_action_and_obs_mapping(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g12_report_wrapper.npz'))
    prbs = simplex_to_prbs(g['action'], 200, 5)
    assert (prbs == g['prbs']).all()
    assert (prbs.sum(axis=1) <= 200).all()
    out = normalise_obs(g['obs_in'])
    assert out.tobytes() == g['obs_out'].astype(out.dtype).tobytes()
    ints = np.array([[10, 20, 30, 40, 50]])
    assert (simplex_to_prbs(ints, 200, 5) == ints).all()

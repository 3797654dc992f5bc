"""experiments_kbrl.BatchedEvaluator / evaluate_grid: how the steps of KBRL_Control.run's loop (kbrl_control.py:126-141) are
dealt out in chunks, checkpoints and cells -- on CPU, with the device classes replaced by counters (the numbers themselves are
the GPU tests' business: tests/test_gpu_kbrl.py::test_grid_of_cells_equals_cell_by_cell, test_checkpoint_and_resume)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'network-slicing_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

LOG = []


class _Lib:
    def rs_step(self, *a):
        return 0


class FakeEnv:
    def __init__(self, n_envs, cfg, fading=None, device=0, **kw):
        self.n, self.h, self.L = n_envs, object(), _Lib()
        self.env_steps = 1 - 1          # simulator steps after the first action
        self.closed = False

    def reset(self, seeds=None):
        pass

    def _check(self, rc):
        assert rc == 0

    def step_resident(self):
        self.env_steps += 1

    def fetch(self):
        return {}

    def synchronize(self):
        self.synced = getattr(self, 'synced', 0) + 1

    def save_state(self):
        return np.array([self.env_steps], dtype=np.int64).view(np.uint8)

    def load_state(self, blob):
        self.env_steps = int(np.asarray(blob, dtype=np.uint8).view(np.int64)[0])

    def close(self):
        self.closed = True


class FakeAgent:
    def __init__(self, n, dims, n_prbs, **kw):
        self.n, self.S = n, len(dims)
        self.agent_steps = 0
        self.calls = []
        self.kw = kw

    def reset(self, ia, sf, seeds=None):
        pass

    def history_begin(self, steps):
        self.hsteps = steps

    def step_resident(self, env):
        self.agent_steps += 1
        self.calls.append(('step', 1))

    def run_resident(self, env, k, graph=True):
        assert k >= 1
        self.agent_steps += k
        env.env_steps += k
        self.calls.append(('run', k, graph))

    def history_fetch(self):
        z = lambda *s: np.zeros(s, dtype=np.int16)
        return dict(reward=np.zeros((self.n, self.hsteps)), resources=z(self.n, self.hsteps), hits=z(self.n, self.S, self.hsteps),
                    adjusted=z(self.n, self.hsteps), SLA=z(self.n, self.hsteps), violation=z(self.n, self.hsteps),
                    recorded=self.agent_steps)

    def dictionary_sizes(self):
        return np.zeros((self.n, self.S), dtype=np.int32)

    def pool(self):
        return dict(used_bytes=0, total_bytes=1, saturated=0, pool_full=0)

    def synchronize(self):
        self.synced = getattr(self, 'synced', 0) + 1

    def save_state(self):
        return np.array([self.agent_steps], dtype=np.int64).view(np.uint8)

    def load_state(self, blob):
        self.agent_steps = int(np.asarray(blob, dtype=np.uint8).view(np.int64)[0])

    def close(self):
        LOG.append(self)


@pytest.fixture
def ek(monkeypatch):
    import experiments_kbrl as ek
    import scenario_creator as sc
    from ranslice import kbrl_dev, vec_env
    monkeypatch.setattr(vec_env, 'VecRanSlice', FakeEnv)
    monkeypatch.setattr(kbrl_dev, 'VecKBRL', FakeAgent)
    monkeypatch.setattr(vec_env, 'default_fading', lambda: [np.zeros((4, 200))] * 3)
    sc.set_fading(None)
    del LOG[:]
    return ek


@pytest.mark.parametrize('steps,chunk', [(1, 64), (2, 64), (65, 64), (130, 64), (50, 7)])
def test_a_cell_makes_steps_agent_steps_and_one_simulator_step_fewer(ek, monkeypatch, tmp_path, steps, chunk):
    monkeypatch.setattr(ek, 'CHUNK', chunk)
    ev = ek.BatchedEvaluator(0, [0.99, 0.999], steps=steps, out_dir=str(tmp_path))
    files = ev.evaluate_all([0, 1, 2], verbose=False)
    ag = LOG[-1]
    assert ag.agent_steps == steps                       # one update_control + select_action per step of the run
    assert sum(c[1] for c in ag.calls if c[0] == 'run') == steps - 1   # every step but the last is followed by a simulator step
    assert ag.calls[-1] == ('step', 1)                   # the last one is not (kbrl_control.py:129-141 ends with the agent)
    assert all(c[1] <= chunk for c in ag.calls)
    assert len(files) == 3 and all(os.path.exists(f) for f in files)
    assert sorted(np.load(files[0]).files) == sorted(['reward', 'resources', 'hits', 'adjusted', 'SLA', 'violation'])


def test_checkpoints_cut_the_chunks_and_a_resumed_run_finishes_the_count(ek, monkeypatch, tmp_path):
    monkeypatch.setattr(ek, 'CHUNK', 64)
    ck = str(tmp_path / 'c.npz')
    ev = ek.BatchedEvaluator(0, [0.97, 0.99], steps=200, out_dir=str(tmp_path))
    assert ev.evaluate_all([0, 1], verbose=False, checkpoint=ck, checkpoint_every=50, stop_after=120) is None
    ag = LOG[-1]
    runs = [c[1] for c in ag.calls if c[0] == 'run']
    assert sum(runs) == 120 and runs == [50, 50, 20]     # cut at the checkpoint boundaries (50, 100) and at the stop (120)
    z = np.load(ck)
    assert int(z['next_step']) == 120 and int(z['steps']) == 200 and list(z['runs']) == [0, 1]
    files = ek.BatchedEvaluator(0, [0.97, 0.99], steps=200, out_dir=str(tmp_path)).evaluate_all([0, 1], verbose=False, checkpoint=ck,
                                                                                          checkpoint_every=50)
    ag2 = LOG[-1]
    assert ag2.agent_steps == 200                        # 120 from the checkpoint + 80 here
    assert [c[1] for c in ag2.calls if c[0] == 'run'] == [30, 49] and ag2.calls[-1] == ('step', 1)
    assert len(files) == 2
    with pytest.raises(ValueError):                      # a checkpoint of another evaluation is refused
        ek.BatchedEvaluator(0, [0.97, 0.99], steps=300, out_dir=str(tmp_path)).evaluate_all([0, 1], verbose=False, checkpoint=ck)


def test_the_grid_advances_every_cell_in_the_same_loop(ek, monkeypatch, tmp_path):
    monkeypatch.setattr(ek, 'CHUNK', 16)
    cells = [(0, [0.97, 0.99]), (1, [0.99, 0.999]), (2, [0.97, 0.99])]
    out = ek.evaluate_grid(cells, [0, 1], steps=40, out_dir=str(tmp_path))
    assert sorted(out) == [(0, 0.97), (1, 0.99), (2, 0.97)]
    assert len(LOG) == 3
    for ag in LOG:
        assert ag.agent_steps == 40 and [c[1] for c in ag.calls if c[0] == 'run'] == [16, 16, 7]
        assert all(c[2] is False for c in ag.calls if c[0] == 'run')   # plain launches: graph launches of different handles serialise
    for (scenario, a0), files in out.items():
        assert all('scenario_%d/KBRL_%d/' % (scenario, int(a0 * 100)) in f for f in files)

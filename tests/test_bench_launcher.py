"""bench.py --gpus N launches ONE rank per GPU through torch.distributed.run with the rendezvous on 127.0.0.1 and hands
the CPU baseline it timed to rank 0 (CPU test: subprocess.call is intercepted, nothing is launched)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('gpus', [2, 8])
def test_gpus_n_builds_the_launcher_command(monkeypatch, gpus):
    bench = _load_bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = list(cmd), dict(env or {})
        return 0
    monkeypatch.setattr(bench.subprocess, 'call', fake_call)
    monkeypatch.setattr(bench, 'cpu_baseline', lambda burn, timed: {'value': 1.0, 'unit': 'env-steps/s', 'cores': 1,
                                                                   'kind': 'port', 'sample': 'stub'})
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', str(gpus), '--steps', '7', '--warmup', '3'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == str(gpus)
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert int(cmd[cmd.index('--master-port') + 1]) > 0
    script = cmd.index(os.path.join(ROOT, 'bench.py'))
    rest = cmd[script + 1:]
    assert rest[:6] == ['--gpus', str(gpus), '--steps', '7', '--warmup', '3']
    assert '--cpu-baseline-json' in rest            # rank 0 reads the baseline this process timed before launching
    assert seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'


def test_rank_count_must_match(monkeypatch):
    bench = _load_bench()
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('WORLD_SIZE', '4')
    monkeypatch.setenv('LOCAL_RANK', '0')
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--no-cpu-baseline'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'WORLD_SIZE' in str(e.value.code)

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


def test_scaling_report_runs_end_to_end_on_a_one_gpu_box(monkeypatch, capsys):
    """python bench.py --scaling 1,2 on a box with ONE device: the N=1 point is a child run of this script (intercepted: its
    line is supplied), N=2 is listed as skipped with the reason, the CPU baseline is in the same line (VERDICT r3 #6c)."""
    import json
    import types
    bench = _load_bench()
    from ranslice import _lib
    monkeypatch.setattr(_lib, 'device_count', lambda: 1)
    monkeypatch.setattr(bench, 'cpu_baseline', lambda burn, timed: {'value': 2.0e4, 'unit': 'env-steps/s', 'cores': 16,
                                                                   'kind': 'port', 'sample': 'stub', 'single_core_value': 1.5e3})
    seen = []

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen.append((list(cmd), dict(env or {})))
        n = int(cmd[cmd.index('--gpus') + 1])
        line = {'value': 3.6e6 * n, 'ms_per_step': 1.13, 'roofline': {'frac': 0.104, 'kernel_ms': 1.1},
                'config': {'global_envs': 4096 * n}}
        return types.SimpleNamespace(returncode=0, stdout='noise\n' + json.dumps(line) + '\n')
    monkeypatch.setattr(bench.subprocess, 'run', fake_run)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--scaling', '1,2', '--steps', '50', '--warmup', '5'])
    bench.main()
    out = [x for x in capsys.readouterr().out.splitlines() if x.startswith('{')]
    rep = json.loads(out[-1])
    assert rep['scaling'] == 'weak' and rep['cpu_baseline']['cores'] == 16 and rep['unit'] == 'env-steps/s'
    assert [c['n_gpus'] for c in rep['curve']] == [1, 2]
    one, two = rep['curve']
    assert one['value'] == 3.6e6 and one['per_gpu_vs_1gpu'] == 1.0 and one['global_envs'] == 4096
    assert 'value' not in two and '1 GPU' in two['skipped']
    assert len(seen) == 1                                   # only N = 1 was launched
    cmd, env = seen[0]
    assert cmd[cmd.index('--gpus') + 1] == '1' and '--no-cpu-baseline' in cmd and '--no-kbrl' in cmd
    assert 'RANK' not in env and 'WORLD_SIZE' not in env

"""bench.py --gpus N starts ONE rank per GPU itself (or runs under torch.distributed.run) with the rendezvous on 127.0.0.1 and hands
the CPU baseline it timed to rank 0; its last line fits the driver's capture window; the ranks meet over a plain socket (no torch).
CPU tests: process launches are intercepted."""
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
def test_gpus_n_spawns_one_rank_per_gpu(monkeypatch, gpus):
    """outside a launcher `bench.py --gpus N` starts N copies of itself with the launcher's environment contract (no torch)"""
    bench = _load_bench()
    seen = []

    class FakeProc:
        def wait(self):
            return 0

    def fake_popen(cmd, env=None):
        seen.append((list(cmd), dict(env or {})))
        return FakeProc()
    monkeypatch.setattr(bench.subprocess, 'Popen', fake_popen)
    monkeypatch.setattr(bench, 'cpu_baseline', lambda burn, timed: {'value': 1.0, 'unit': 'env-steps/s', 'cores': 1,
                                                                   'kind': 'port', 'sample': 'stub'})
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', str(gpus), '--steps', '7', '--warmup', '3'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert len(seen) == gpus
    ports = set()
    for r, (cmd, env) in enumerate(seen):
        assert cmd[0] == sys.executable and cmd[1] == os.path.join(ROOT, 'bench.py')
        assert cmd[2:8] == ['--gpus', str(gpus), '--steps', '7', '--warmup', '3']
        assert '--cpu-baseline-json' in cmd            # rank 0 reads the baseline this process timed before launching
        assert env['RANK'] == str(r) and env['LOCAL_RANK'] == str(r) and env['WORLD_SIZE'] == str(gpus)
        assert env['MASTER_ADDR'] == '127.0.0.1' and int(env['MASTER_PORT']) > 0
        assert env.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'
        ports.add(env['MASTER_PORT'])
    assert len(ports) == 1


def test_bench_imports_no_torch():
    """north_star: no PyTorch -- not in bench.py, not in the rank group, not anywhere in the product package"""
    import re
    for rel in ('bench.py', 'tools/rank_group.py', '__graft_entry__.py'):
        src = open(os.path.join(ROOT, rel)).read()
        assert not re.search(r'^\s*(import|from)\s+torch', src, flags=re.M), rel
    for dp, dn, fn in os.walk(os.path.join(ROOT, 'network-slicing_amd')):
        for f in fn:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(import|from)\s+torch', src, flags=re.M), f


def test_bench_and_product_import_with_torch_masked():
    """the same at run time: with `torch` made unimportable, bench.py, the rank group and every module of the product package
    still import (what `python bench.py --gpus 1` touches before it reaches the GPU)"""
    import subprocess
    code = (
        "import sys, importlib.util\n"
        "sys.modules['torch'] = None\n"
        "root = %r\n"
        "sys.path[:0] = [root, root + '/network-slicing_amd', root + '/tools']\n"
        "spec = importlib.util.spec_from_file_location('bench_masked', root + '/bench.py')\n"
        "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
        "import rank_group, gym_ran_slice, scenario_creator, kbrl_control, node_b, experiments_kbrl\n"
        "from ranslice import _lib, config, fading, sharding, vec_env, kbrl_dev, report, gymshim\n"
        "from algorithms import kernel, projectron\n"
        "g = rank_group.RankGroup(0, 1); assert g.max(2.5) == 2.5 and g.allgather('x') == ['x']\n"
        "assert 'torch' not in [k for k, v in sys.modules.items() if v is not None]\n"
        "print('ok')\n" % ROOT)
    out = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-1500:]


def _canned_full(bench):
    """a full record with every sub-record at the length a real run produces (long strings, nested dicts)"""
    long = 'x' * 900
    point = {'steps': [3000, 3200], 'value': 1.53e6, 'ms_per_step': 2.67, 'embb_kernel_ms': 1.5, 'kb_update_phase_ms': 0.9,
             'kb_select_ms': 0.3, 'select_mfma': {'instructions_per_launch': 1253376, 'note': long},
             'dictionary_size_mean': 229.4, 'dictionary_size_max': 1650, 'pool': {'used_bytes': 10.3e9, 'total_bytes': 64 << 30},
             'pool_horizon': {'exhausted_near_step': 18500.0, 'note': long},
             'select_bin': {'frac': 0.52}, 'kinv_streaming': {'rank1': {'frac': 0.64, 'note': long}, 'matvec': {'frac': 0.46}}}
    return {
        'metric': 'env-steps/sec (batched RanSlice.step, scenario_0)', 'value': 3686543.21, 'unit': 'env-steps/s', 'n_gpus': 1,
        'steps': 2000, 'warmup': 200, 'ms_per_step': 1.1111111, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'scenario_0 (200 PRBs, 5 eMBB slices, 50 slots/step), 4096 env replicas per GPU, step() only, '
                               'random multinomial actions generated on device', 'envs_per_gpu': 4096, 'global_envs': 4096,
                   'burn_in_steps': 2500, 'burn_in': 'until stationary', 'fading': '3 synthetic traces', 'parallelism': 'x1',
                   'loop': 'one launch sequence per step from the host'},
        'roofline': {'bound': 'hbm', 'kernel': 'embb_step_kernel', 'achieved': 847.123456, 'peak': 8000.0, 'unit': 'GB/s',
                     'frac': 0.10589, 'traffic': 7.15e8, 'traffic_source': 'profiles/r05_pmc_hbm.txt',
                     'algorithmic_bytes_per_launch': 915.2e6, 'kernel_ms': 1.08, 'launches_timed': 2000,
                     'bytes_per_env_step': 223437.5, 'mean_ues_per_slice': 3.3, 'pf_iterations_per_env_step': 1234.5,
                     'profiled_traffic': {'source': long},
                     'limiter': {'bound': 'valu_issue', 'frac': 0.78, 'valu_insts_per_launch': 4.49e8, 'source': 'profiles/r05_pmc_sq.txt'}},
        'burn_in_history_mean_ues': [3.1] * 16,
        'cpu_baseline': {'value': 27000.0, 'unit': 'env-steps/s', 'cores': 16, 'kind': 'port', 'single_core_value': 2070.0,
                         'steps': 3000, 'burn_in': 1000, 'sample': long},
        'kbrl': {'workload': long, 'traces': 'tdl', 'early': dict(point, steps=[100, 300], value=2.49e6), 'late': point,
                 'scoring': long, 'profiled_counters': {'kernels': {('k%d' % i): {'a': 1.0, 'b': long} for i in range(12)}}},
        'kbrl_sos': {'workload': long, 'early': point, 'late': point},
        'shared_kbrl': {'workload': long, 'value': 2.19e6, 'unit': 'env-steps/s', 'ms_per_step': 1.87, 'steps': 200, 'n_gpus': 1,
                        'rccl_ranks': 1, 'allgather_bytes_total_per_step': 184360, 'capacity': 1024,
                        'dictionary_sizes': [1024, 300, 280, 290, 310], 'dictionaries_identical_on_all_ranks': True,
                        'collective': long},
    }


def test_compact_line_fits_the_drivers_capture_window():
    """a 20 KB final line left BENCH_r04.parsed = null.  The last stdout line is built by compact_line and stays
    under 4 KB whatever the sub-records hold; it carries every key the rules credit."""
    import json
    bench = _load_bench()
    full = _canned_full(bench)
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 4096, len(text)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line, k
    assert line['config']['workload'].startswith('scenario_0')
    for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes_per_launch', 'kernel_ms', 'limiter'):
        assert k in line['roofline'], k
    assert line['roofline']['limiter']['bound'] == 'valu_issue'
    for k in ('value', 'cores', 'kind', 'single_core_value', 'sample', 'unit'):
        assert k in line['cpu_baseline'], k
    for k in ('early_value', 'late_value', 'late_ms_per_step', 'rank1_frac', 'matvec_frac', 'select_bin_frac', 'mfma_instructions',
              'pool_exhausted_near_step'):
        assert line['kbrl'][k] is not None, k
    assert line['shared_kbrl']['rccl_ranks'] == 1 and line['shared_kbrl']['value'] == 2.19e6
    assert abs(line['value'] - full['value']) < 1e-6 * full['value']
    # failed sub-records stay small and visible
    full['kbrl'] = {'error': 'RuntimeError(...)'}
    full['shared_kbrl'] = {'error': 'rc 1: ncclCommInitRank: invalid usage'}
    line = bench.compact_line(full)
    assert line['kbrl']['error'] and line['shared_kbrl']['error'] and len(json.dumps(line)) < 4096


def _group_worker(rank, world, path, q):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from rank_group import RankGroup
    g = RankGroup(rank, world, rdzv_file=path, timeout=30.0)
    g.barrier()
    got = [g.max(float(rank) * 1.5), g.sum([1.0, rank]), g.allgather({'r': rank}),
           g.bcast_bytes(bytes(range(128)) if rank == 0 else None).hex()]
    g.barrier()
    g.close()
    q.put((rank, got))


@pytest.mark.parametrize('world', [2, 4])
def test_rank_group_over_a_plain_socket(tmp_path, world):
    """tools/rank_group.py: barrier, MAX, SUM, all-gather and a 128-byte broadcast (the RCCL communicator id) between
    processes over 127.0.0.1, rendezvous through a file -- what bench.py uses in place of a torch process group"""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    path = str(tmp_path / 'rdzv')
    # a stale file of an earlier launch (nobody listens there): the ranks must get past it
    with open(path, 'w') as f:
        f.write('{"port": 1, "token": "stale"}')
    ps = [ctx.Process(target=_group_worker, args=(r, world, path, q)) for r in range(world)]
    for pr in reversed(ps):      # the clients first: they poll until rank 0 has published
        pr.start()
    res = dict(q.get(timeout=60) for _ in ps)
    for pr in ps:
        pr.join(30)
        assert pr.exitcode == 0
    for r in range(world):
        mx, sm, ag, bc = res[r]
        assert mx == 1.5 * (world - 1)
        assert sm == [float(world), float(sum(range(world)))]
        assert ag == [{'r': i} for i in range(world)]
        assert bc == bytes(range(128)).hex()
    assert not os.path.exists(path)          # rank 0 removes the rendezvous file


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


def test_hbm_traffic_summary_feeds_the_roofline():
    """profiles/hbm_traffic.json (tools/make_hbm_traffic.py from a tools/profile_round.sh run) carries what bench.py quotes as
    roofline.traffic and roofline.limiter, taken at the population of the default bench's timed run, and its sources are committed"""
    import json
    t = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))
    for k in ('embb_step_kernel_bytes_per_launch', 'mean_ues_per_slice', 'n_envs', 'valu_issue_frac', 'valu_insts_per_launch', 'file',
              'valu_issue_file', 'algorithmic_bytes_per_launch_same_run', 'kernel_ms_hip_events', 'kernel_ms_rocprofv3_trace'):
        assert t.get(k) is not None, k
    assert t['n_envs'] == 4096 and 3.2 < t['mean_ues_per_slice'] < 3.4          # the stationary population of the default bench
    assert 0.5 < t['embb_step_kernel_bytes_per_launch'] / t['algorithmic_bytes_per_launch_same_run'] < 1.5
    assert 0.5 < t['valu_issue_frac'] < 1.0
    assert abs(t['kernel_ms_hip_events'] - t['kernel_ms_rocprofv3_trace']) < 0.05 * t['kernel_ms_hip_events']   # the two clocks agree
    for f in (t['file'], t['valu_issue_file']):
        assert os.path.exists(os.path.join(ROOT, f)), f


def test_committed_counter_summaries_are_of_the_current_round():
    """profiles/kbrl_mfma_share.json and profiles/hbm_traffic.json are printed beside the timings of the round's bench record
    (`profiled_counters`, `roofline.traffic` / `limiter`): their sources must not be older than the newest committed bench line
    (round-4 counters beside round-5 timings), and the counter corrections must name a committed calibration."""
    import glob
    import json
    import re
    prof = os.path.join(ROOT, 'profiles')
    rounds = [int(m.group(1)) for m in (re.match(r'r(\d+)_.*bench_line\.json$', os.path.basename(f)) for f in glob.glob(os.path.join(prof, 'r*_bench_line.json'))) if m]
    newest = max(rounds)
    k = json.load(open(os.path.join(prof, 'kbrl_mfma_share.json')))
    m = re.search(r'profiles/r(\d+)_', k['source'])
    assert m and int(m.group(1)) >= newest, (k['source'], newest)
    assert os.path.exists(os.path.join(ROOT, re.search(r'(profiles/\S+\.txt)', k['source']).group(1)))
    g = k['kernels']['select_gemm_kernel']
    assert g['SQ_INSTS_MFMA'] > 0 and 0.05 < g['mfma_util'] < 1.0            # MFMA-busy time / kernel time, from the same state
    t = json.load(open(os.path.join(prof, 'hbm_traffic.json')))
    m = re.search(r'profiles/r(\d+)_', t['file'])
    assert m and int(m.group(1)) >= newest, (t['file'], newest)
    assert t['fetch_correction'] == 2.0 and t['write_correction'] == 1.0
    cal = json.load(open(os.path.join(prof, t['calibration_file'].replace('.txt', '.json'))))
    for name in ('read_4B_per_lane', 'read_8B_per_lane', 'read_16B_per_lane', 'read_128B_segments'):
        assert abs(cal[name]['fetch_ratio'] * t['fetch_correction'] - 1.0) < 0.01, name     # every coalesced shape: counter x 2 = true bytes
    for name in ('write_4B_per_lane', 'write_8B_per_lane', 'write_16B_per_lane'):
        assert abs(cal[name]['write_ratio'] * t['write_correction'] - 1.0) < 0.01, name


def test_rendezvous_file_is_private_and_replaces_stale_ones(tmp_path, monkeypatch):
    """the rendezvous file carries the join token -- it lives in a directory only the user can enter, is created 0600 without
    following links, and a stale file of a crashed launch (or a link somebody planted under its name) is removed, not reused."""
    import stat
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import rank_group as rg
    monkeypatch.delenv('RANSLICE_RDZV_FILE', raising=False)
    monkeypatch.delenv('XDG_RUNTIME_DIR', raising=False)
    monkeypatch.setenv('MASTER_PORT', '29777')
    f = rg.default_rdzv_file()
    d = os.path.dirname(f)
    st = os.lstat(d)
    assert d == os.path.join('/tmp', 'ranslice-%d' % os.getuid()) and stat.S_ISDIR(st.st_mode) and (st.st_mode & 0o077) == 0
    assert os.path.basename(f) == 'rdzv_29777_%d' % os.getppid()
    monkeypatch.setenv('XDG_RUNTIME_DIR', str(tmp_path))
    assert os.path.dirname(rg.default_rdzv_file()) == str(tmp_path)
    target = tmp_path / 'somebody_elses_file'
    target.write_text('{"port": 1, "token": "stale"}')
    path = str(tmp_path / 'rdzv_x')
    os.symlink(str(target), path)                        # a planted link under the rendezvous name
    with pytest.raises(OSError):
        rg._read_published(path)                         # never followed
    rg._publish(path, {'port': 4242, 'token': 'fresh'})
    assert not os.path.islink(path) and (os.lstat(path).st_mode & 0o777) == 0o600
    assert rg._read_published(path) == {'port': 4242, 'token': 'fresh'}
    assert target.read_text() == '{"port": 1, "token": "stale"}'     # the link's target was not written through
    rg._publish(path, {'port': 4243, 'token': 'newer'})  # a stale regular file is replaced as well
    assert rg._read_published(path)['port'] == 4243

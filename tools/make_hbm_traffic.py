#!/usr/bin/env python3
"""profiles/hbm_traffic.json from one tools/profile_round.sh run: python tools/make_hbm_traffic.py <tag> > profiles/hbm_traffic.json
(reads gpurun_out/<tag>_pmc_hbm.txt, _pmc_sq.txt, _pmc_bench_line.json, _kernel_trace_stats.txt; the "file" fields name the
copies committed under profiles/).  bench.py quotes `embb_step_kernel_bytes_per_launch` as roofline.traffic when its own run has
the same UE population, and `valu_issue_frac` as roofline.limiter."""
import json
import re
import sys

tag = sys.argv[1]
src = 'gpurun_out/%s_' % tag


def counters(path):
    out = {}
    for line in open(path):
        m = re.search(r'\s(FETCH_SIZE|WRITE_SIZE|SQ_\w+|GRBM_\w+)\s+([0-9.]+)\s+\(last', line)
        if m and 'embb_step_kernel<16' in line:
            out[m.group(1)] = float(m.group(2))
    return out


hbm = counters(src + 'pmc_hbm.txt')
sq = counters(src + 'pmc_sq.txt')
line = json.loads(open(src + 'pmc_bench_line.json').read().strip().splitlines()[-1])
roof = line['roofline']
kernel_ms = roof['kernel_ms']                       # HIP events, the same command without the profiler
m = re.search(r'last 300 launches of .*?: mean (\d+) ns', open(src + 'kernel_trace_stats.txt').read())
trace_ms = int(m.group(1)) / 1e6 if m else None
simds, ghz = 1024, 2.4
# Counter corrections calibrated on this box against known byte counts (tools/ubench/fetch_calib.hip, tools/calibrate_fetch.sh,
# profiles/r06_fetch_calibration.txt): FETCH_SIZE reads exactly 1/2 of the bytes of every coalesced shape tried (4, 8, 16 B per
# lane, scattered 128-byte segments) -- 128-byte requests tallied at 64 B, the guide's gfx950 note --, WRITE_SIZE reads 1.000.
FETCH_CORR, WRITE_CORR = 2.0, 1.0
res = {
    'embb_step_kernel_bytes_per_launch': 1024.0 * (FETCH_CORR * hbm.get('FETCH_SIZE', 0.0) + WRITE_CORR * hbm.get('WRITE_SIZE', 0.0)),
    'fetch_bytes_per_launch': 1024.0 * FETCH_CORR * hbm.get('FETCH_SIZE', 0.0), 'write_bytes_per_launch': 1024.0 * WRITE_CORR * hbm.get('WRITE_SIZE', 0.0),
    'fetch_counter_kb': hbm.get('FETCH_SIZE', 0.0), 'write_counter_kb': hbm.get('WRITE_SIZE', 0.0),
    'fetch_correction': FETCH_CORR, 'write_correction': WRITE_CORR, 'calibration_file': 'r06_fetch_calibration.txt',
    'mean_ues_per_slice': roof['mean_ues_per_slice'], 'n_envs': line['config']['envs_per_gpu'],
    'algorithmic_bytes_per_launch_same_run': roof['algorithmic_bytes_per_launch'],
    'file': 'profiles/%s_pmc_hbm.txt' % tag,
    'source': 'rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes, KB -> bytes, mean of the last 20 launches; environments '
              'restored at the stationary population of the default bench (bench.py --state-file)',
    'note': 'fabric-side requests in 128-byte lines: sub-line accesses (16 B of prefix sums per channel estimate, unaligned 80-250-byte '
            'float spans) count whole lines',
    'valu_insts_per_launch': sq.get('SQ_INSTS_VALU'), 'salu_insts_per_launch': sq.get('SQ_INSTS_SALU'),
    'kernel_ms_hip_events': kernel_ms, 'kernel_ms_rocprofv3_trace': trace_ms,
    # a VALU wave-instruction occupies its SIMD's issue port for 4 cycles; SQ_ACTIVE_INST_VALU counts those quad-cycles directly
    'valu_issue_frac': (sq['SQ_ACTIVE_INST_VALU'] * 4.0 / (simds * kernel_ms * 1e-3 * ghz * 1e9)) if sq.get('SQ_ACTIVE_INST_VALU') and kernel_ms else None,
    'valu_issue_file': 'profiles/%s_pmc_sq.txt' % tag,
    'valu_issue_source': 'SQ_ACTIVE_INST_VALU quad-cycles x 4 over 1024 SIMDs x kernel_ms (HIP events, same command) x 2.4 GHz',
    'wait_share_of_wave_cycles': (sq['SQ_WAIT_ANY'] / sq['SQ_WAVE_CYCLES']) if sq.get('SQ_WAIT_ANY') and sq.get('SQ_WAVE_CYCLES') else None,
}
print(json.dumps(res, indent=1))

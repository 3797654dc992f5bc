#!/usr/bin/env python3
"""profiles/kbrl_mfma_share.json from a tools/profile_kbrl.sh run (PMC summary + kernel trace of the same checkpointed state):
   python tools/make_mfma_share.py gpurun_out/r06_k_kbrl_late_pmc.txt profiles/r06_k_kbrl_late_pmc.txt \
          gpurun_out/r06_k_kbrl_late_kernel_trace.txt > profiles/kbrl_mfma_share.json
The second argument is the committed copy the "source" field names; the third (optional) the kernel trace whose durations turn
the MFMA-busy cycles into a utilisation (mfma_util = busy time / kernel time)."""
import json
import re
import sys

src, named = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
trace = sys.argv[3] if len(sys.argv) > 3 else None
avg_ns = {}
if trace:
    for line in open(trace):
        m = re.match(r'(?:void )?kb::(\w+)\S*.*?\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+[0-9.]+%', line)
        if m:
            avg_ns[m.group(1)] = float(m.group(4))
per = {}
for line in open(src):
    m = re.match(r'\s+(?:void )?kb::(\w+)\S*.*?\s(SQ_\w+)\s+([0-9.]+)\s+\(last', line)
    if m:
        per.setdefault(m.group(1), {})[m.group(2)] = float(m.group(3))
rnd = re.search(r'r(\d+)_', named)
out = {'source': '%s (rocprofv3 --pmc, one counter group per run; means per launch over the 12 steps after step 3000 of learning, '
                 '4096 replicas x 5 learners, state restored from a checkpoint: tools/profile_kbrl.sh)' % named,
       'round': int(rnd.group(1)) if rnd else None, 'kernels': {}}
tot_mfma = 0.0
for k, c in per.items():
    if 'SQ_INSTS_VALU' not in c:
        continue
    e = {x: c[x] for x in ('SQ_INSTS_MFMA', 'SQ_INSTS_VALU_MFMA_MOPS_F64', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU',
                           'SQ_INSTS_LDS', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_BUSY_CYCLES') if x in c}
    if c.get('SQ_WAVE_CYCLES'):
        e['wait_share_of_wave_cycles'] = c.get('SQ_WAIT_ANY', 0.0) / c['SQ_WAVE_CYCLES']
    if c.get('SQ_INSTS_MFMA'):
        # v_mfma_f64_16x16x4: 64 cycles of a SIMD's matrix pipe each; 1024 SIMDs; the kernel's own duration is in the kernel trace
        e['mfma_busy_cycles_per_instruction'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_INSTS_MFMA']
        e['mfma_pipe_time_us_if_spread_over_1024_simds_at_2.4GHz'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / 2400.0
        e['mfma_share_of_valu_instructions'] = c['SQ_INSTS_MFMA'] / (c['SQ_INSTS_MFMA'] + c['SQ_INSTS_VALU'])
        if k in avg_ns:  # busy time of the matrix pipes (spread over the chip) over the kernel's mean duration in the kernel trace
            e['kernel_us_kernel_trace'] = avg_ns[k] / 1e3
            e['mfma_util'] = e['mfma_pipe_time_us_if_spread_over_1024_simds_at_2.4GHz'] / (avg_ns[k] / 1e3)
            e['tflops_f64'] = c['SQ_INSTS_MFMA'] * 2048.0 / (avg_ns[k] * 1e-9) / 1e12
            e['frac_of_78.6_tflops_f64'] = e['tflops_f64'] / 78.6
    tot_mfma += c.get('SQ_INSTS_MFMA', 0.0)
    out['kernels'][k] = e
out['mfma_instructions'] = tot_mfma
out['what'] = ('select_action scores every candidate of 16 learners at a time as F = T W^T on v_mfma_f64_16x16x4 '
               '(select_gemm_kernel: 16 candidate tiles x 51 instructions x 1,536 workgroups per launch); update_control starts '
               'from those scores, so its kernels issue none')
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Register/scratch budget of the step-kernel instances, from the remarks of the last build
(network-slicing_amd/csrc/build/resources.log).  The two 16-lane production instances are built for 5 waves per
SIMD (96 VGPRs) and spill; instances that spilled more than ~256 B per lane have twice been seen to compute wrong
values on this toolchain, so the build fails if a change pushes them past 240 B."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, 'network-slicing_amd', 'csrc', 'build', 'resources.log')
LIMIT = 240
KB_SCRATCH = {'update_small_kernel': 64, 'update_heavy_kernel': 16, 'update_control_kernelILb0E': 16, 'heavy_finish_kernel': 0,
              'heavy_matvec_kernel': 0, 'heavy_rank1_kernel': 0, 'select_bin_kernel': 16, 'select_bin_big_kernel': 16, 'select_gemm_kernel': 0}
PRODUCTION = ["embb_step_kernelILi16ELb0ELb0ELb1E", "embb_step_kernelILi16ELb0ELb1ELb1E"]   # <16, false, plain | BLOCK, FDIV>


def parse(path=LOG):
    out, cur = {}, None
    for line in open(path, errors='replace'):
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]+\])?: (\d+)', line)
        if m and cur:
            out[cur][m.group(1).strip()] = int(m.group(2))
    return out


def check(path=LOG):
    res = parse(path)
    bad = []
    for key in PRODUCTION:
        hit = [k for k in res if key in k]
        if not hit:
            bad.append('%s: not found in %s' % (key, path))
            continue
        r = res[hit[0]]
        print('%s: VGPRs %s, scratch %s B/lane, occupancy %s' % (key, r.get('VGPRs'), r.get('ScratchSize'), r.get('Occupancy')))
        if r.get('ScratchSize', 0) > LIMIT:
            bad.append('%s spills %d B/lane (> %d)' % (key, r['ScratchSize'], LIMIT))
        if r.get('Occupancy', 0) < 5:
            bad.append('%s: occupancy %s < 5 waves/SIMD' % (key, r.get('Occupancy')))
    # The agent's per-learner kernels sit at their register limit too: a refactoring of a helper they inline (round 5: lambdas in the
    # triangle mat-vec) put 608 B per lane of scratch into update_small_kernel / update_heavy_kernel and doubled their time unnoticed.
    for key, limit in KB_SCRATCH.items():
        hit = [k for k in res if key in k]
        if not hit:
            bad.append('%s: not found in %s' % (key, path))
        elif res[hit[0]].get('ScratchSize', 0) > limit:
            bad.append('%s spills %d B/lane (> %d)' % (key, res[hit[0]]['ScratchSize'], limit))
    return bad


if __name__ == '__main__':
    problems = check()
    for p in problems:
        print('RESOURCE CHECK FAILED:', p)
    sys.exit(1 if problems else 0)

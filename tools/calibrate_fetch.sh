#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts (tools/ubench/fetch_calib.hip), one --pmc pass per counter.
# usage (on the GPU box): bash tools/calibrate_fetch.sh <tag>   -> gpurun_out/<tag>_fetch_calibration.txt (+ .json)
TAG=${1:-r06}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
BIN=$PWD/tools/ubench/fetch_calib
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN tools/ubench/fetch_calib.hip || exit 1
$BIN > /tmp/calib_true.txt || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/cal_$C && timeout 600 rocprofv3 --pmc $C -d /tmp/cal_$C -o c -- $BIN > /tmp/cal_$C.log 2>&1; echo "$C rc=$?" )
  DB=$(find /tmp/cal_$C -name '*.db' | head -1)
  python tools/rocpd_summary.py $DB > /tmp/cal_$C.txt
done
python - "$TAG" <<'PY'
import json, re, sys
tag = sys.argv[1]
true = {}
for l in open('/tmp/calib_true.txt'):
    m = re.match(r'(\S+)\s+true bytes per launch (\d+)', l)
    if m:
        true[m.group(1)] = float(m.group(2))
res = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    for l in open('/tmp/cal_%s.txt' % C):
        m = re.match(r'\s+(\S+)\(.*?\s(%s)\s+([0-9.]+)\s' % C, l) or re.match(r'\s+(\S+?)[\s(].*?(%s)\s+([0-9.]+)\s' % C, l)
        if m and m.group(1) in true:
            res.setdefault(m.group(1), {})[C] = 1024.0 * float(m.group(3))
lines = ['# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB x 1024) against the bytes tools/ubench/fetch_calib.hip really moves',
         '# (2 GiB buffer, every byte touched once per launch; mean of 3 launches).  ratio = counter / true; correction = 1 / ratio.',
         '%-44s %16s %16s %8s %16s %8s' % ('kernel', 'true bytes', 'FETCH_SIZE B', 'ratio', 'WRITE_SIZE B', 'ratio')]
out = {}
for k, t in true.items():
    f, w = res.get(k, {}).get('FETCH_SIZE', 0.0), res.get(k, {}).get('WRITE_SIZE', 0.0)
    lines.append('%-44s %16.0f %16.0f %8.3f %16.0f %8.3f' % (k, t, f, f / t, w, w / t))
    out[k] = {'true_bytes': t, 'fetch_bytes': f, 'fetch_ratio': f / t, 'write_bytes': w, 'write_ratio': w / t}
open('gpurun_out/%s_fetch_calibration.txt' % tag, 'w').write('\n'.join(lines) + '\n')
json.dump(out, open('gpurun_out/%s_fetch_calibration.json' % tag, 'w'), indent=1)
print('\n'.join(lines))
PY

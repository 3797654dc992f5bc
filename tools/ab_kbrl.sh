#!/bin/bash
# A/B of config 3 (agents in the loop, tdl traces) at the late point between library builds: bash tools/ab_kbrl.sh <tag> <lib> [<lib> ...]
TAG=$1; shift
OUT=gpurun_out/${TAG}_ab_kbrl.txt
: > $OUT
ST=/tmp/late_ab
python tools/bench_kbrl.py --profile tdl --warmup 3000 --save-state $ST > /dev/null || exit 1
for rep in 1 2; do
  for L in "$@"; do
    RANSLICE_LIB=$PWD/network-slicing_amd/csrc/build/$L timeout 600 python tools/bench_kbrl.py --profile tdl --load-state $ST --steps 300 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$L: env_steps_per_s %.5g ms_per_step %.4f embb_kernel_ms %.4f update %.4f select %.4f' % (d['env_steps_per_s'], d['ms_per_step'], d['embb_kernel_ms'], d['kb_update_phase_ms'], d['kb_select_ms']))" | tee -a $OUT
  done
done
rm -rf $ST

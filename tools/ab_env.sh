#!/bin/bash
# A/B of config 3 at the late point with a test-build knob set / unset: bash tools/ab_env.sh <VAR>   (the test build: libranslice_dev.so)
VAR=$1
ST=/tmp/late_ab
python tools/bench_kbrl.py --profile tdl --warmup 3000 --save-state $ST > /dev/null || exit 1
for rep in 1 2; do
 for V in 1 0; do
  if [ $V = 1 ]; then export $VAR=1; else unset $VAR; fi
  RANSLICE_LIB=$PWD/network-slicing_amd/csrc/build/libranslice_dev.so python tools/bench_kbrl.py --profile tdl --load-state $ST --steps 300 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['per_step_ms']
print('$VAR=$V: env_steps_per_s %.5g ms_per_step %.4f embb %.4f update %.4f select %.4f | matvec %.4f rank1 %.4f finish %.4f bin %.4f' % (d['env_steps_per_s'], d['ms_per_step'], d['embb_kernel_ms'], d['kb_update_phase_ms'], d['kb_select_ms'], p['matvec'], p['rank1'], p['finish'], p['select_bin']))"
 done
done
rm -rf $ST

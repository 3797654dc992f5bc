#!/bin/bash
# the split step under the AGENTS' allocations (config 3, state restored at step 3000): step-kernel time per variant of the test build
TAG=${1:-r05_x}
OUT=gpurun_out/${TAG}_split_late.txt
ST=/tmp/late_tdl
python tools/bench_kbrl.py --profile tdl --warmup 3000 --save-state $ST > /dev/null || exit 1
echo "# $(date): config 3 from the checkpoint of step 3000 (tdl), 200 steps; step kernel = HIP events around the step launches" > $OUT
export RANSLICE_DEV_BUILD=1
run() {
  local label=$1; shift
  env "$@" python tools/bench_kbrl.py --profile tdl --load-state $ST --steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s ms/step %.3f step_kernel %.3f update %.3f select %.3f' % ('$label', d['ms_per_step'], d['embb_kernel_ms'], d['kb_update_phase_ms'], d['kb_select_ms']))" | tee -a $OUT
}
run base A=0
run head16 RANSLICE_MIXED=16
run head32_light128 RANSLICE_MIXED=32 RANSLICE_MIXED_LIGHT=128
run light128 RANSLICE_MIXED_LIGHT=128
run light192 RANSLICE_MIXED_LIGHT=192
run light224 RANSLICE_MIXED_LIGHT=224
run head64_light192 RANSLICE_MIXED=64 RANSLICE_MIXED_LIGHT=192
run head16_light0 RANSLICE_MIXED=16 RANSLICE_MIXED_LIGHT=0
run base A=0
rm -rf $ST

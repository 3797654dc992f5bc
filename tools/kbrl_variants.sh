#!/bin/bash
# experiment builds of the KBRL kernels at the late point of learning, same box, same restored state:
#   make -C network-slicing_amd/csrc variant VNAME=x VFLAGS="..."   (here, before the gpurun call), then
#   bash tools/kbrl_variants.sh <tag> base x y ...
TAG=$1; shift
OUT=gpurun_out/${TAG}_kbrl_variants.txt
mkdir -p gpurun_out
ST=/tmp/late_tdl
python tools/bench_kbrl.py --profile tdl --warmup 3000 --save-state $ST > /dev/null || exit 1
echo "# $(date): per-step ms of the KBRL kernels (HIP events around every launch), 200 steps from the checkpoint of step 3000, tdl traces" > $OUT
for v in "$@" "$1"; do
  RANSLICE_LIB=network-slicing_amd/csrc/build/libranslice_$v.so python tools/bench_kbrl.py --profile tdl --load-state $ST --steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d['per_step_ms']
print('%-8s ms/step %.3f step_kernel %.3f update %.3f select %.3f | ' % ('$v', d['ms_per_step'], d['embb_kernel_ms'], d['kb_update_phase_ms'], d['kb_select_ms']) + ' '.join('%s %.4f' % (k, p[k]) for k in sorted(p)) + ' | select_bin %.0f GB/s' % (d['select_bin_GBs'] or 0))
" | tee -a $OUT
done
rm -rf $ST

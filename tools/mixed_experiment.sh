#!/bin/bash
# the split step against the 16-lane step, same command, test build.  The cost ranking is dealt out to up to
# three launches side by side: its head one task per 16-lane wave (RANSLICE_MIXED = n: the first 1/n), its light end eight tasks per
# 8-lane wave (RANSLICE_MIXED_LIGHT = l: the last l/256), the middle four per 16-lane wave as without a split.
# usage: bash tools/mixed_experiment.sh <tag>
TAG=${1:-r05_mix}
OUT=gpurun_out/${TAG}_mixed.txt
mkdir -p gpurun_out
export RANSLICE_DEV_BUILD=1
run() {  # label, envs-per-gpu, env assignments...
  local label=$1 n=$2; shift 2
  local line=$(env "$@" timeout 600 python bench.py --no-cpu-baseline --no-kbrl --no-shared --steps 600 --warmup 100 --burn-in 2500 --envs-per-gpu $n 2>/dev/null | tail -1)
  echo "$label n=$n $* :: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("value %.4g ms_per_step %.4f kernel_ms %.4f ues %.3f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["mean_ues_per_slice"]))' 2>&1)" | tee -a $OUT
}
echo "# $(date) split-step experiment (kernel_ms = HIP events around the step launches of one step)" > $OUT
python -m pytest tests/test_gpu_fullsize.py -q -x -k "task_order" 2>&1 | tail -2 | tee -a $OUT
run base 4096 A=0
for l in 32 64 96 128 160; do
  run light$l 4096 RANSLICE_MIXED_LIGHT=$l
done
run head64_light64 4096 RANSLICE_MIXED=64 RANSLICE_MIXED_LIGHT=64
run head32_light128 4096 RANSLICE_MIXED=32 RANSLICE_MIXED_LIGHT=128
run base 8192 A=0
run light128 8192 RANSLICE_MIXED_LIGHT=128
run light192 8192 RANSLICE_MIXED_LIGHT=192
run head32 8192 RANSLICE_MIXED=32
run head32_light192 8192 RANSLICE_MIXED=32 RANSLICE_MIXED_LIGHT=192
run base 16384 A=0
run head32 16384 RANSLICE_MIXED=32
run head24 16384 RANSLICE_MIXED=24
run head32_light192 16384 RANSLICE_MIXED=32 RANSLICE_MIXED_LIGHT=192
run light224 16384 RANSLICE_MIXED_LIGHT=224

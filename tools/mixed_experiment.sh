#!/bin/bash
# the split step (8-lane instance for the tasks that fit it, the head of the cost ranking on the 16-lane instance one
# task per wave beside it) against the 16-lane step, same command, test build.  usage: bash tools/mixed_experiment.sh <tag>
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
for n in 4096 16384; do
  run base $n A=0
  run mixed16 $n RANSLICE_MIXED=16
  run mixed32 $n RANSLICE_MIXED=32
  run mixed8 $n RANSLICE_MIXED=8
  run mixed16_ue7 $n RANSLICE_MIXED=16 RANSLICE_MIXED_UE=7
  run mixed64 $n RANSLICE_MIXED=64
done
run base 65536 A=0
run mixed16 65536 RANSLICE_MIXED=16
run mixed32 65536 RANSLICE_MIXED=32

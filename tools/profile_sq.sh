#!/bin/bash
# SQ issue/wait counters of the step kernel, a few counters per --pmc pass (run on the GPU box).  usage: bash tools/profile_sq.sh <tag>
TAG=${1:-r01_x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 20 --warmup 5 --burn-in 400 --no-cpu-baseline"
echo "# rocprofv3 --pmc passes (one counter group per run) of: $CMD ; means per launch of embb_step_kernel<16,false>" > $OUT/${TAG}_pmc_sq.txt
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C -d /tmp/sq_${TAG}_$i -o p -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > /tmp/sq_$i.log 2>&1; echo "pass $i rc=$?" )
  DB=$(find /tmp/sq_${TAG}_$i -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB | grep -E "embb_step_kernel<16" | grep -E "SQ_|GRBM" >> $OUT/${TAG}_pmc_sq.txt
done
cat $OUT/${TAG}_pmc_sq.txt

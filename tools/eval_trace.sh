#!/bin/bash
# kernel trace of a 30-replica evaluation (experiments_kbrl.BatchedEvaluator), per-kernel account of its last STEPS/5 steps
TAG=${1:-r04_x}; STEPS=${STEPS:-20000}; PROFILE=${PROFILE:-tdl}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 1500 rocprofv3 --kernel-trace -d /tmp/et_$TAG -o t -- bash -c "cd $GRAFT_REPO_ROOT && STEPS=$STEPS PROFILE=$PROFILE bash tools/eval_scale.sh > /tmp/et_$TAG.txt" > /tmp/et_$TAG.log 2>&1
DB=$(find /tmp/et_$TAG -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace of: STEPS=$STEPS PROFILE=$PROFILE bash tools/eval_scale.sh"; cat /tmp/et_$TAG.txt
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB --steps $((STEPS / 5)) --anchor kb::history_advance_kernel | grep -A40 "the last"; } > $OUT/${TAG}_eval_trace.txt
cat $OUT/${TAG}_eval_trace.txt | head -40

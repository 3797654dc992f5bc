#!/bin/bash
# HBM traffic counters of the step kernel, one --pmc pass per counter (run on the GPU box).  usage: bash tools/profile_pmc.sh <tag>
TAG=${1:-r01_x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 20 --warmup 5 --burn-in 400 --no-cpu-baseline"
echo "# rocprofv3 --pmc <counter> (one pass each) of: $CMD" > $OUT/${TAG}_pmc_hbm.txt
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $C -d /tmp/pmc_${TAG}_$C -o p -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > /tmp/pmc_$C.log 2>&1; echo "$C rc=$?" )
  DB=$(find /tmp/pmc_${TAG}_$C -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB | grep -E "embb_step_kernel<16.*$C" >> $OUT/${TAG}_pmc_hbm.txt
done
cat $OUT/${TAG}_pmc_hbm.txt

#!/bin/bash
# Round-end measurement set (run on the GPU box): default bench line, rocprofv3 kernel trace of a short bench run,
# and the two HBM traffic counters in separate --pmc passes.  usage: bash tools/profile_round.sh <tag>
TAG=${1:-r01_x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
tail -c 2500 $OUT/bench_$TAG.json
CMD="python bench.py --steps 300 --warmup 50 --burn-in 1000 --no-cpu-baseline"
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o t -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > /tmp/prof_$TAG.log 2>&1 )
DB=$(find /tmp/prof_$TAG -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats of: $CMD"; python tools/rocpd_summary.py $DB; } > $OUT/${TAG}_kernel_trace_stats.txt
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $C -d /tmp/pmc_${TAG}_$C -o p -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 30 --warmup 5 --burn-in 1000 --no-cpu-baseline" > /tmp/pmc_$C.log 2>&1 )
  DB=$(find /tmp/pmc_${TAG}_$C -name '*.db' | head -1)
  python tools/rocpd_summary.py $DB | grep -E "embb_step_kernel.*$C" >> $OUT/${TAG}_pmc_hbm.txt
done
cat $OUT/${TAG}_pmc_hbm.txt
head -12 $OUT/${TAG}_kernel_trace_stats.txt

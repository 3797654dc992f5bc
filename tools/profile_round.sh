#!/bin/bash
# Round-end measurement set (run on the GPU box): default bench line, rocprofv3 kernel trace of a short bench run,
# and the two HBM traffic counters in separate --pmc passes.  usage: bash tools/profile_round.sh <tag>
TAG=${1:-r01_x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
tail -c 2500 $OUT/bench_$TAG.json
CMD="python bench.py --steps 300 --warmup 50 --burn-in 1000 --no-cpu-baseline"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o t -- bash -c "cd $GRAFT_REPO_ROOT && $CMD > /tmp/prof_$TAG.json" > /tmp/prof_$TAG.log 2>&1 )
DB=$(find /tmp/prof_$TAG -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats of: $CMD"
  python -c "import json; l = json.loads(open('/tmp/prof_$TAG.json').read().strip().splitlines()[-1]); print('# the same run\'s bench line: %.0f env-steps/s, ms_per_step %.3f, roofline.kernel_ms %.4f (HIP events, %d timed launches)' % (l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['roofline']['launches_timed']))"
  python tools/rocpd_summary.py $DB --last 300; } > $OUT/${TAG}_kernel_trace_stats.txt
head -16 $OUT/${TAG}_kernel_trace_stats.txt

#!/bin/bash
# Round-end measurement set of the step kernel (run on the GPU box): rocprofv3 kernel trace of a bench run (HIP-event kernel time
# of the same run beside it), the two HBM traffic counters in separate --pmc passes, the SQ issue / wait counters -- ALL at the UE
# population of bench.py's timed run: the environments are burnt in once (until stationary, as the default bench does) and every
# pass restores that state (bench.py --state-file).  Then profiles/hbm_traffic.json (what bench.py's roofline.traffic and
# roofline.limiter quote) is rebuilt from the passes.
# usage: bash tools/profile_round.sh <tag>      (writes gpurun_out/<tag>_*; copy what is to be judged into profiles/)
TAG=${1:-r05_x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
SF=/tmp/bench_state_$TAG
rm -f $SF.npz
COMMON="--state-file $SF --no-cpu-baseline --no-kbrl --no-shared"
python bench.py $COMMON --steps 1 --warmup 0 > /dev/null || exit 1
CMD="python bench.py $COMMON --steps 300 --warmup 50"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o t -- bash -c "cd $GRAFT_REPO_ROOT && $CMD > /tmp/prof_$TAG.json" > /tmp/prof_$TAG.log 2>&1 )
DB=$(find /tmp/prof_$TAG -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats of: $CMD   (environments restored at the stationary population of the default bench)"
  python -c "import json; l = json.loads(open('/tmp/prof_$TAG.json').read().strip().splitlines()[-1]); print('# the same run\'s bench line: %.0f env-steps/s, ms_per_step %.3f, roofline.kernel_ms %.4f (HIP events, %d timed launches), %.2f UEs/slice, algorithmic bytes per launch %.0f' % (l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['roofline']['launches_timed'], l['roofline']['mean_ues_per_slice'], l['roofline']['algorithmic_bytes_per_launch']))"
  python tools/rocpd_summary.py $DB --last 300; } > $OUT/${TAG}_kernel_trace_stats.txt
head -14 $OUT/${TAG}_kernel_trace_stats.txt
PCMD="python bench.py $COMMON --steps 20 --warmup 5"
( cd $GRAFT_REPO_ROOT && $PCMD | tail -1 > $OUT/${TAG}_pmc_bench_line.json )
echo "# rocprofv3 --pmc <counter> (one pass each) of: $PCMD ; per launch of embb_step_kernel<16,false>, mean of the last 20 launches; FETCH_SIZE / WRITE_SIZE in KB" > $OUT/${TAG}_pmc_hbm.txt
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $C -d /tmp/pmc_${TAG}_$C -o p -- bash -c "cd $GRAFT_REPO_ROOT && $PCMD" > /tmp/pmc_$C.log 2>&1; echo "$C rc=$?" )
  DB=$(find /tmp/pmc_${TAG}_$C -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB --last 20 | grep -E "embb_step_kernel<16.*$C" | grep "last 20" >> $OUT/${TAG}_pmc_hbm.txt
done
cat $OUT/${TAG}_pmc_hbm.txt
echo "# rocprofv3 --pmc passes (one counter group per run) of: $PCMD ; means per launch of the step kernel (last 20 launches)" > $OUT/${TAG}_pmc_sq.txt
i=0
for GRP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  ( cd /tmp && timeout 400 rocprofv3 --pmc $GRP -d /tmp/sq_${TAG}_$i -o p -- bash -c "cd $GRAFT_REPO_ROOT && $PCMD" > /tmp/sq_$i.log 2>&1; echo "group $i rc=$?" )
  DB=$(find /tmp/sq_${TAG}_$i -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB --last 20 | grep -E "embb_step_kernel<16" | grep "last 20" >> $OUT/${TAG}_pmc_sq.txt
done
cat $OUT/${TAG}_pmc_sq.txt
python tools/make_hbm_traffic.py $TAG > $OUT/${TAG}_hbm_traffic.json && cat $OUT/${TAG}_hbm_traffic.json
rm -f $SF.npz

#!/bin/bash
# KBRL kernels (BASELINE config 3) under rocprofv3: kernel trace of the closed loop at two points of learning, then PMC
# passes (one counter group per run, no tracing flags beside --pmc) of a shorter run.  Run on the GPU box:
#   bash tools/profile_kbrl.sh <tag>      -> gpurun_out/<tag>_kbrl_*  (copy what is to be judged into profiles/)
TAG=${1:-r03_x}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for W in 100 3000; do
  CMD="python tools/bench_kbrl.py --warmup $W --steps 200"
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_${TAG}_$W -o t -- bash -c "cd $GRAFT_REPO_ROOT && $CMD > /tmp/kt_${TAG}_$W.json" > /tmp/kt_$W.log 2>&1
  DB=$(find /tmp/kt_${TAG}_$W -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats of: $CMD   (the summary covers the warm-up as well; see the last-200 lines)"
    echo "# the run's own line: $(cat /tmp/kt_${TAG}_$W.json)"
    python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB --last 200 --steps 200 --anchor kb::adjust_kernel
    for KN in 'void kb::update_control_kernel<false>' 'kb::update_small_kernel' 'kb::heavy_matvec_kernel' 'kb::heavy_finish_kernel' 'kb::heavy_rank1_kernel' 'kb::update_heavy_kernel' 'kb::select_bin_kernel' 'kb::select_gemm_kernel'; do
      python - "$DB" "$KN" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
d = [r[0] for r in c.execute("select end - start from kernels where name like ? order by start desc limit ?", (sys.argv[2] + '%', 600 if 'heavy_' in sys.argv[2] and 'update' not in sys.argv[2] else 200))]
if d:
    print('# last %d launches of %s (= the last 200 steps): mean %.0f ns (min %d, max %d)' % (len(d), sys.argv[2], sum(d) / len(d), min(d), max(d)))
PY
    done; } > $OUT/${TAG}_kbrl_w${W}_kernel_trace.txt
  grep -A30 'the last 200 steps' $OUT/${TAG}_kbrl_w${W}_kernel_trace.txt | head -34
done
[ -n "$KONLY" ] && exit 0   # KONLY=1: the kernel traces only
PCMD="python tools/bench_kbrl.py --warmup 150 --steps 20"
echo "# rocprofv3 --pmc passes (one counter group per run) of: $PCMD ; means per launch over the last 20 launches of each kb:: kernel (steps 150-170 of learning: dictionaries of ~40 landmarks on average; counter collection costs ~20 ms per dispatch, so the late phase is not replayed under PMC)" > $OUT/${TAG}_kbrl_pmc.txt
i=0
for GRP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $GRP -d /tmp/kpmc_${TAG}_$i -o p -- bash -c "cd $GRAFT_REPO_ROOT && $PCMD" > /tmp/kpmc_$i.log 2>&1; echo "group $i ($GRP) rc=$?"
  DB=$(find /tmp/kpmc_${TAG}_$i -name '*.db' | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB --last 20 | grep -E "kb::" | grep "last 20" >> $OUT/${TAG}_kbrl_pmc.txt
done
cat $OUT/${TAG}_kbrl_pmc.txt

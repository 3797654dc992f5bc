#!/bin/bash
# KBRL kernels (BASELINE config 3) under rocprofv3 at the LATE point of learning.  The closed loop runs unprofiled to step
# $LATE (graph-replayed), its state is checkpointed (rs_save_state / kb_save_state), and every profiled pass restores the checkpoint:
# a kernel trace of 200 steps, then PMC passes (one counter group per run, no tracing flags beside --pmc) of 12 steps.
#   bash tools/profile_kbrl.sh <tag> [tdl|sos]      -> gpurun_out/<tag>_kbrl_*  (copy what is to be judged into profiles/)
TAG=${1:-r05_x}
PROF=${2:-tdl}
LATE=${LATE:-3000}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ST=/tmp/late_${PROF}
cd /tmp
( cd $GRAFT_REPO_ROOT && python tools/bench_kbrl.py --profile $PROF --warmup $LATE --save-state $ST ) || exit 1
CMD="python tools/bench_kbrl.py --profile $PROF --load-state $ST --steps 200"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_${TAG} -o t -- bash -c "cd $GRAFT_REPO_ROOT && $CMD > /tmp/kt_${TAG}.json" > /tmp/kt.log 2>&1
DB=$(find /tmp/kt_${TAG} -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats of: $CMD   (state restored from a checkpoint taken at step $LATE of the closed loop, $PROF traces)"
  echo "# the run's own line: $(cat /tmp/kt_${TAG}.json)"
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB --last 200 --steps 200 --anchor kb::adjust_kernel
} > $OUT/${TAG}_kbrl_late_kernel_trace.txt
grep -A24 'the last 200 steps' $OUT/${TAG}_kbrl_late_kernel_trace.txt | head -28
# where the step kernel's cycles go under the agents' allocations (s_memtime marks of the -DRS_SECTION_PROFILE build: make profile)
if [ -f $GRAFT_REPO_ROOT/network-slicing_amd/csrc/build/libranslice_prof.so ]; then
  ( cd $GRAFT_REPO_ROOT && RANSLICE_LIB=network-slicing_amd/csrc/build/libranslice_prof.so timeout 600 python tools/section_profile.py --load-state $ST --traces $PROF > $OUT/${TAG}_late_sections.txt 2>&1 )
  head -24 $OUT/${TAG}_late_sections.txt
fi
[ -n "$KONLY" ] && { rm -rf $ST; exit 0; }   # KONLY=1: the kernel trace and the section profile only
PCMD="python tools/bench_kbrl.py --profile $PROF --load-state $ST --steps 12"
echo "# rocprofv3 --pmc passes (one counter group per run) of: $PCMD ; means per launch over the last launches of each kb:: kernel (steps $LATE.. of learning)" > $OUT/${TAG}_kbrl_late_pmc.txt
i=0
for GRP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $GRP -d /tmp/kpmc_${TAG}_$i -o p -- bash -c "cd $GRAFT_REPO_ROOT && $PCMD" > /tmp/kpmc_$i.log 2>&1; echo "group $i ($GRP) rc=$?"
  DB=$(find /tmp/kpmc_${TAG}_$i -name '*.db' | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB --last 12 | grep -E "kb::|embb_step_kernel<16" | grep "last 12" >> $OUT/${TAG}_kbrl_late_pmc.txt
done
cat $OUT/${TAG}_kbrl_late_pmc.txt
rm -rf $ST

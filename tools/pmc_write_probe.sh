# WRITE_SIZE / FETCH_SIZE per launch of the step kernel for a forced group size (developer probe)
export TMPDIR=/tmp
G=${1:-8}
for C in WRITE_SIZE FETCH_SIZE; do
( cd /tmp && RANSLICE_GROUP=$G timeout 200 rocprofv3 --pmc $C -d /tmp/pw_${G}_$C -o p -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 20 --warmup 5 --burn-in 600 --no-cpu-baseline --no-kbrl" > /tmp/pw.log 2>&1 )
DB=$(find /tmp/pw_${G}_$C -name '*.db' | head -1)
python tools/rocpd_summary.py $DB --last 20 | grep -E "embb_step_kernel<.*$C.*last"
done

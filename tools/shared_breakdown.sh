#!/bin/bash
# steady-state kernel breakdown of the shared-dictionary step (config 4 on one GPU): rocprofv3 kernel trace of
# tools/run_shared_kbrl.py, means over the launches of the timed steps.  usage: bash tools/shared_breakdown.sh [capacity] [out]
CAP=${1:-1024}; OUT=${2:-gpurun_out/shared_breakdown_$CAP.txt}
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/shb && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/shb -o t -- bash -c "cd $GRAFT_REPO_ROOT && python tools/run_shared_kbrl.py --steps 200 --warmup 100 --capacity $CAP > /tmp/shb.json" > /tmp/shb.log 2>&1 )
DB=$(find /tmp/shb -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace of: python tools/run_shared_kbrl.py --steps 200 --warmup 100 --capacity $CAP (scenario_2, 4096 replicas, one GPU, device-resident loop); means over the launches of the 200 timed steps"
  echo "# the run: $(cat /tmp/shb.json)"
  python - $DB <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for (n,) in c.execute("select distinct name from kernels").fetchall():
    if not (n.startswith("kb::") or "embb_step" in n or "mtc_step" in n):
        continue
    tot = c.execute("select count(*) from kernels where name = ?", (n,)).fetchone()[0]
    k = int(tot * 200 / 300)
    d = [r[0] for r in c.execute("select end - start from kernels where name = ? order by start desc limit ?", (n, k))]
    if d:
        print("%-64s %5d launches: mean %9.0f ns  max %9d  per step %8.1f us" % (n[:64], len(d), sum(d) / len(d), max(d), sum(d) / 200 / 1000))
PY
} > $OUT
cut -c1-170 $OUT

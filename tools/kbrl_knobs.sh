# developer sweep: order / pairing knobs on the agent-in-the-loop record
for kv in "RANSLICE_PAIR=256" "RANSLICE_PAIR=0" "RANSLICE_PAIR=128" "RANSLICE_ORDER=3" "RANSLICE_ORDER=4" "RANSLICE_ORDER=5" "RANSLICE_GROUP=32" "RANSLICE_GROUP=8"; do
echo "== $kv"
env $kv python tools/bench_kbrl.py --warmup 100 --steps 200 | python -c "
import json,sys
k=json.loads(sys.stdin.readline())
print("env-steps/s %.0f ms/step %.3f embb %.3f kb %.3f" % (k["env_steps_per_s"], k["ms_per_step"], k["embb_kernel_ms"], k["kb_kernel_ms_mean_of_update_and_select"]))"
done

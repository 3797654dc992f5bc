# developer sweep: task order / pairing of the step kernel under the agents' allocations (config 3, steps W..W+200)
W=${W:-100}
for kv in "RANSLICE_PAIR=256" "RANSLICE_PAIR=192" "RANSLICE_PAIR=128" "RANSLICE_PAIR=64" "RANSLICE_PAIR=0" "RANSLICE_ORDER=4" "RANSLICE_ORDER=5" "RANSLICE_ORDER=3" "RANSLICE_ORDER=1" "RANSLICE_ORDER=0"; do
  env $kv python tools/bench_kbrl.py --warmup $W --steps 200 2>/dev/null | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-20s env-steps/s %.0f ms/step %.3f embb %.3f' % ('$kv', k['env_steps_per_s'], k['ms_per_step'], k['embb_kernel_ms']))"
done

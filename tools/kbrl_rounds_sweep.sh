# developer sweep: chip-wide repair rounds per step (KBRL_ROUNDS) at the late point of config 3
for r in 2 3 4 5; do
  KBRL_ROUNDS=$r python tools/bench_kbrl.py --warmup 3000 --steps 200 2>/dev/null | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('KBRL_ROUNDS=$r  env-steps/s %.0f ms/step %.3f update phase %.3f select %.3f' % (k['env_steps_per_s'], k['ms_per_step'], k['kb_update_phase_ms'], k['kb_select_ms']))"
done

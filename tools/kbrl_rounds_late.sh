for R in 3 2 4 5 3; do
KBRL_ROUNDS=$R timeout 300 python tools/bench_kbrl.py --warmup 3000 --steps 200 --profile tdl 2>/dev/null | tail -1 | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('KBRL_ROUNDS=$R late: env-steps/s %.0f ms/step %.3f embb %.3f update %.3f select %.3f' % (k['env_steps_per_s'], k['ms_per_step'], k['embb_kernel_ms'], k['kb_update_phase_ms'], k['kb_select_ms']))"
done

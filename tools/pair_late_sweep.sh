# share of the waves led by one heavy task (RANSLICE_PAIR / 256) on config 3 late and early
for P in ${PAIRS:-256 224 192 160 128 256}; do for W in ${POINTS:-3000 100}; do
RANSLICE_PAIR=$P timeout 300 python tools/bench_kbrl.py --warmup $W --steps 200 --profile tdl 2>/dev/null | tail -1 | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PAIR=$P agents w$W: env-steps/s %.0f ms/step %.3f embb %.3f' % (k['env_steps_per_s'], k['ms_per_step'], k['embb_kernel_ms']))"
done; done

for CFG in "0x0a 0" "0x0a 0x04" "0x0a 0x14" "0x0a 0" "0x0a 0x04"; do set -- $CFG
for W in 100 3000; do
RANSLICE_SNAKE_MASK=$1 RANSLICE_SNAKE_ROT=$2 timeout 300 python tools/bench_kbrl.py --warmup $W --steps 200 --profile tdl 2>/dev/null | tail -1 | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('MASK=$1 ROT=$2 agents w$W: env-steps/s %.0f ms/step %.3f embb %.3f' % (k['env_steps_per_s'], k['ms_per_step'], k['embb_kernel_ms']))"
done; done

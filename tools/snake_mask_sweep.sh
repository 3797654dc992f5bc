for M in 0x0a 0x06 0x16 0x1a 0x0c 0x12 0x0e 0x0a; do for W in 100 3000; do
RANSLICE_SNAKE_MASK=$M timeout 300 python tools/bench_kbrl.py --warmup $W --steps 200 --profile tdl 2>/dev/null | tail -1 | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('MASK=$M agents w$W: env-steps/s %.0f ms/step %.3f embb %.3f' % (k['env_steps_per_s'], k['ms_per_step'], k['embb_kernel_ms']))"
done; done

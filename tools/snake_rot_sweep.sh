for CFG in "0x0a 0" "0x0a 0x04" "0x02 0x0c" "0 0x0a" "0x08 0x06" "0x0a 0x10" "0x0a 0"; do set -- $CFG
RANSLICE_SNAKE_MASK=$1 RANSLICE_SNAKE_ROT=$2 timeout 300 python bench.py --steps 300 --warmup 30 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('MASK=$1 ROT=$2 plain: env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f' % (l['value'], l['ms_per_step'], r['kernel_ms']))"
RANSLICE_SNAKE_MASK=$1 RANSLICE_SNAKE_ROT=$2 timeout 300 python tools/bench_kbrl.py --warmup 3000 --steps 200 --profile tdl 2>/dev/null | tail -1 | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('MASK=$1 ROT=$2 agents w3000: env-steps/s %.0f ms/step %.3f embb %.3f' % (k['env_steps_per_s'], k['ms_per_step'], k['embb_kernel_ms']))"
done

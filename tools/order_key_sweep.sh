# the cost key of the task order (RANSLICE_ORDER 4..8 = keys 1..5 of rs_order.hip with the heavy + light pairing) under the serpentine
# dealing: config 3 early / late, then the bench's random script
for O in ${ORDERS:-6 4 5 7 8 6}; do for W in 100 3000; do
RANSLICE_ORDER=$O timeout 300 python tools/bench_kbrl.py --warmup $W --steps 200 --profile tdl 2>/dev/null | tail -1 | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ORDER=$O agents w$W: env-steps/s %.0f ms/step %.3f embb %.3f' % (k['env_steps_per_s'], k['ms_per_step'], k['embb_kernel_ms']))"
done
RANSLICE_ORDER=$O timeout 300 python bench.py --steps 300 --warmup 30 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('ORDER=$O plain: env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f' % (l['value'], l['ms_per_step'], r['kernel_ms']))"
done

# the serpentine dealing of the cost-ranked waves (RANSLICE_SNAKE: 0 off, 1 = rounds of one wave per SIMD, > 1 = that many waves per
# round) on config 3's early and late points and on the bench's random script.  usage: SNAKES="1 0 512" bash tools/snake_ab.sh
for S in ${SNAKES:-1 0}; do for W in ${POINTS:-100 3000}; do
RANSLICE_SNAKE=$S timeout 300 python tools/bench_kbrl.py --warmup $W --steps 200 --profile tdl 2>/dev/null | tail -1 | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('SNAKE=$S agents w$W: env-steps/s %.0f ms/step %.3f embb %.3f' % (k['env_steps_per_s'], k['ms_per_step'], k['embb_kernel_ms']))"
done; done
for S in ${SNAKES:-1 0}; do
RANSLICE_SNAKE=$S timeout 300 python bench.py --steps 300 --warmup 30 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('SNAKE=$S plain: env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f' % (l['value'], l['ms_per_step'], r['kernel_ms']))"
done

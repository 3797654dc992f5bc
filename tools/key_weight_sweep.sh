# weights of the task order's cost key (RANSLICE_KEY_W = a,b,c,d in sixteenths: fading samples per slot, last step's PF rounds, UEs,
# backlog drain slots x (UEs + 1)) on the bench's random script and on config 3 late / early
for KW in ${KWS:-16,16,0,0 16,16,0,16 16,16,0,32 16,8,0,32 16,0,0,32 16,16,0,64 16,8,0,64 16,0,0,64 16,16,0,0}; do
RANSLICE_KEY_W=$KW timeout 300 python bench.py --steps 300 --warmup 30 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('KW=$KW plain: env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f' % (l['value'], l['ms_per_step'], r['kernel_ms']))"
for W in ${POINTS:-3000}; do
RANSLICE_KEY_W=$KW timeout 300 python tools/bench_kbrl.py --warmup $W --steps 200 --profile tdl 2>/dev/null | tail -1 | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('KW=$KW agents w$W: env-steps/s %.0f ms/step %.3f embb %.3f' % (k['env_steps_per_s'], k['ms_per_step'], k['embb_kernel_ms']))"
done; done

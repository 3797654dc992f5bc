# developer sweep: task order modes (rs_order.hip) at the stationary population
for o in ${ORDERS:-0 1 2 3 4 5 6}; do
echo "== order $o"
RANSLICE_ORDER=$o timeout 300 python bench.py --envs-per-gpu ${ENVS:-4096} --steps 300 --warmup 30 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f  mean_ue %.2f' % (l['value'], l['ms_per_step'], r['kernel_ms'], r['mean_ues_per_slice']))"
done

for o in ${ORDERS:-6 7 8 0}; do
echo "== order $o"
RANSLICE_ORDER=$o RANSLICE_GROUP=${GROUP:-16} python bench.py --steps 300 --warmup 30 --burn-in 400 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f  mean_ue %.2f' % (l['value'], l['ms_per_step'], r['kernel_ms'], r['mean_ues_per_slice']))"
done

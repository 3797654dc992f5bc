# developer sweep: replicas per GPU x lanes per task (config 5 of BASELINE.json is >= 8192 replicas per GPU)
for n in ${SIZES:-8192 16384}; do for g in ${GROUPS_:-8 16}; do
echo "== envs $n group $g"
RANSLICE_GROUP=$g timeout 400 python bench.py --envs-per-gpu $n --steps 200 --warmup 20 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f  mean_ue %.2f' % (l['value'], l['ms_per_step'], r['kernel_ms'], r['mean_ues_per_slice']))"
done; done

# developer sweep: share of the waves led by one heavy task (RANSLICE_PAIR, /256) x replicas per GPU
for n in ${SIZES:-4096}; do for p in ${PAIRS:-0 16 32 64 96 128 192 256}; do
echo "== envs $n pair $p"
RANSLICE_PAIR=$p timeout 300 python bench.py --envs-per-gpu $n --steps 300 --warmup 30 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f  mean_ue %.2f' % (l['value'], l['ms_per_step'], r['kernel_ms'], r['mean_ues_per_slice']))"
done; done

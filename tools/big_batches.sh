# larger batches per GPU (config 5's shard is 8192 replicas): bench.py's step loop replayed from a captured graph, 1500-step burn-in
for N in ${NS:-8192 16384 65536}; do
timeout 600 python bench.py --envs-per-gpu $N --steps 200 --warmup 20 --burn-in 1500 --graph --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('$N %.0f env-steps/s  %.3f ms/step  step kernel %.3f ms  %.2f UEs/slice  (%s)' % (l['value'], l['ms_per_step'], r['kernel_ms'], r['mean_ues_per_slice'], l['config']['loop']))"
done

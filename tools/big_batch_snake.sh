# larger batches (config 5's per-GPU shard and beyond) with the serpentine dealing on / off
for N in 8192 16384; do for S in 1 0; do
RANSLICE_SNAKE=$S timeout 400 python bench.py --envs-per-gpu $N --steps 150 --warmup 20 --burn-in 800 --graph --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('N=$N SNAKE=$S: env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f' % (l['value'], l['ms_per_step'], r['kernel_ms']))"
done; done

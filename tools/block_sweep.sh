# developer sweep: builds of libranslice (LIBS) on the bench workload and on the agent-in-the-loop record
bash tools/quick_check.sh
for lib in $LIBS; do
echo "== $lib"
RANSLICE_LIB=network-slicing_amd/csrc/build/$lib timeout 300 python bench.py --steps 300 --warmup 30 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f  mean_ue %.2f' % (l['value'], l['ms_per_step'], r['kernel_ms'], r['mean_ues_per_slice']))"
RANSLICE_LIB=network-slicing_amd/csrc/build/$lib bash tools/kbrl_quick.sh | cut -c1-160
done

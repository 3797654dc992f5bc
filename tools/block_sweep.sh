# developer sweep: builds of libranslice (LIBS) on the agent-in-the-loop record and, with the BLOCK instance forced, on the bench workload
for lib in $LIBS; do
echo "== $lib"
RANSLICE_LIB=network-slicing_amd/csrc/build/$lib python tools/bench_kbrl.py --warmup 100 --steps 200 | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('agents: env-steps/s %.0f ms/step %.3f embb %.3f kb %.3f' % (k['env_steps_per_s'], k['ms_per_step'], k['embb_kernel_ms'], k['kb_kernel_ms_mean_of_update_and_select']))"
RANSLICE_HINT=1 RANSLICE_LIB=network-slicing_amd/csrc/build/$lib timeout 300 python bench.py --steps 300 --warmup 30 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('random script, BLOCK instance: env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f' % (l['value'], l['ms_per_step'], r['kernel_ms']))"
done

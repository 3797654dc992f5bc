# developer sweep: builds LIBS of the library (compiler-flag variants, names under csrc/build without .so) on the bench's random
# script (plain instance) and on config 3's early point (BLOCK instance)
for lib in $LIBS; do
RANSLICE_LIB=$PWD/network-slicing_amd/csrc/build/$lib.so timeout 300 python bench.py --steps 300 --warmup 30 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('$lib plain: env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f' % (l['value'], l['ms_per_step'], r['kernel_ms']))"
RANSLICE_LIB=$PWD/network-slicing_amd/csrc/build/$lib.so timeout 300 python tools/bench_kbrl.py --warmup 100 --steps 200 --profile tdl 2>/dev/null | tail -1 | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib agents early: env-steps/s %.0f ms/step %.3f embb %.3f' % (k['env_steps_per_s'], k['ms_per_step'], k['embb_kernel_ms']))"
done

for lib in ${LIBS:-libranslice}; do
RANSLICE_LIB=$PWD/network-slicing_amd/csrc/build/$lib.so timeout 300 python bench.py --steps 300 --warmup 30 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('$lib plain: kernel_ms %.3f' % (r['kernel_ms']))"
for W in 100 3000; do
RANSLICE_LIB=$PWD/network-slicing_amd/csrc/build/$lib.so timeout 300 python tools/bench_kbrl.py --warmup $W --steps 200 --profile tdl 2>/dev/null | tail -1 | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib agents w$W: ms/step %.3f embb %.3f' % (k['ms_per_step'], k['embb_kernel_ms']))"
done; done

# the BLOCK instance forced on the bench's random script (RANSLICE_HINT=1) for builds LIBS; the plain instance first
for lib in $LIBS; do
for H in $HINTS; do
RANSLICE_HINT=$H RANSLICE_LIB=$PWD/network-slicing_amd/csrc/build/$lib.so timeout 300 python bench.py --steps 300 --warmup 30 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('$lib hint=$H: env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f' % (l['value'], l['ms_per_step'], r['kernel_ms']))"
done
done

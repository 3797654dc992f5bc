ST=/tmp/late_tdl
python tools/bench_kbrl.py --profile tdl --warmup 3000 --save-state $ST > /dev/null || exit 1
RANSLICE_LIB=network-slicing_amd/csrc/build/libranslice_stamps.so python tools/bench_kbrl.py --profile tdl --load-state $ST --steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
w = d['stamps_raw_w4_w7']
print('stamps', w, 'per wave: setup %.0f  landmarks %.0f  chunks %.2f  (s_memtime units)' % (w[0]/w[3], w[1]/w[3], w[2]/w[3]), 'select_bin ms/step', d['per_step_ms']['select_bin'])
"
rm -rf $ST

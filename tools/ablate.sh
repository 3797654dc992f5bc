#!/bin/bash
for a in 6 7; do echo "ablation $a (6=no tail, 7=no MI exp)"; RANSLICE_LIB=network-slicing_amd/csrc/build/libranslice_abl$a.so python bench.py --steps 300 --warmup 30 --burn-in 400 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('   kernel_ms %.3f  mean_ue %.2f' % (r['kernel_ms'], r['mean_ues_per_slice']))"; done

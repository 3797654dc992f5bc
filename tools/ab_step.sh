#!/bin/bash
# A/B of the step kernel between library builds on one box: bash tools/ab_step.sh <tag> <lib> [<lib> ...]
# (each library twice, interleaved; bench.py's headline leg only)
TAG=$1; shift
OUT=gpurun_out/${TAG}_ab.txt
: > $OUT
for rep in 1 2; do
  for L in "$@"; do
    RANSLICE_LIB=$PWD/network-slicing_amd/csrc/build/$L timeout 600 python bench.py --no-cpu-baseline --no-kbrl --no-shared --steps 1000 --warmup 100 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$L: value %.5g ms_per_step %.4f kernel_ms %.4f ues %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['mean_ues_per_slice']))" | tee -a $OUT
  done
done

#!/bin/bash
# One GPU call of the round: GPU tests, the default bench (its last line is what the driver records), the N = 2 launch path on
# one device (rank group + guarded config-4 leg), the large batches.  usage: bash tools/gpu_round.sh <tag> [pytest args]
TAG=${1:-r05_x}
shift
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q "$@" > $OUT/${TAG}_gputests.txt 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_gputests.txt )
tail -5 $OUT/${TAG}_gputests.txt
( timeout 900 python bench.py > $OUT/${TAG}_bench_stdout.txt 2> $OUT/${TAG}_bench_stderr.txt; echo "bench rc=$?" )
tail -1 $OUT/${TAG}_bench_stdout.txt > $OUT/${TAG}_bench_line.json
wc -c $OUT/${TAG}_bench_line.json
cat $OUT/${TAG}_bench_line.json
cp profiles/bench_full_last.json $OUT/${TAG}_bench_full.json 2>/dev/null
tail -5 $OUT/${TAG}_bench_stderr.txt
( RANSLICE_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 300 --warmup 50 --burn-in 500 --no-cpu-baseline > $OUT/${TAG}_bench2_stdout.txt 2> $OUT/${TAG}_bench2_stderr.txt; echo "bench2 rc=$?" )
tail -1 $OUT/${TAG}_bench2_stdout.txt | cut -c1-600
for n in 8192 16384 65536; do
  timeout 600 python bench.py --no-cpu-baseline --no-kbrl --no-shared --steps 400 --warmup 100 --burn-in 2500 --envs-per-gpu $n --graph 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('envs_per_gpu $n (hipGraph loop): value %.4g ms_per_step %.4f kernel_ms %.4f ues %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['mean_ues_per_slice']))" | tee -a $OUT/${TAG}_big_batches.txt
done
# the launch path the driver uses at N = 2, and the one-command curve, on one device (a check of the path, not a scaling measurement)
( RANSLICE_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/${TAG}_driver2_stdout.txt 2> $OUT/${TAG}_driver2_stderr.txt; echo "driver-style --gpus 2 rc=$?" )
tail -1 $OUT/${TAG}_driver2_stdout.txt > $OUT/${TAG}_driver_style_two_ranks_one_gpu.json; cut -c1-300 $OUT/${TAG}_driver_style_two_ranks_one_gpu.json
( RANSLICE_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --scaling 1,2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_scaling_1_2_one_gpu.json 2> $OUT/${TAG}_scaling_stderr.txt; echo "scaling rc=$?" )
cut -c1-600 $OUT/${TAG}_scaling_1_2_one_gpu.json

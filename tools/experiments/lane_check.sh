# the lane-per-task engine (RANSLICE_LANE=1) through the parity tests that do not need the allocation trace, then its throughput by batch size
RANSLICE_LANE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py "tests/test_gpu_parity.py::test_lane_engine_matches_oracle" -x -q -m gpu 2>&1 | tail -4
for n in ${SIZES:-4096 8192 16384 65536}; do for l in 0 1; do
echo "== envs $n lane $l"
RANSLICE_LANE=$l timeout 400 python bench.py --envs-per-gpu $n --steps 100 --warmup 10 --burn-in 1500 --no-cpu-baseline --no-kbrl 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('env-steps/s %.0f  ms/step %.3f  kernel_ms %.3f  mean_ue %.2f' % (l['value'], l['ms_per_step'], r['kernel_ms'], r['mean_ues_per_slice']))"
done; done

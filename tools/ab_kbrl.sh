# A/B of two builds (usage: ab_kbrl.sh libA libB ...; names under network-slicing_amd/csrc/build without .so) on config 3's
# early point and a 30-run evaluation of EVAL_STEPS steps
for L in "$@"; do
  export RANSLICE_LIB=$PWD/network-slicing_amd/csrc/build/$L.so
  echo "== $L"
  timeout 300 python tools/bench_kbrl.py --warmup 100 --steps 200 --profile tdl 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('early: %.0f env-steps/s %.3f ms/step embb %.3f ms' % (l['env_steps_per_s'], l['ms_per_step'], l['embb_kernel_ms']))"
  STEPS=${EVAL_STEPS:-20000} RUNS=30 PROFILE=tdl timeout 600 bash tools/eval_scale.sh | head -1 | cut -c1-150
done

#!/bin/bash
# quick look at config 3 after a KBRL change: the closed loop early (steps 100-300) and late (3000-3200) in learning
TAG=${1:-r04_x}
mkdir -p gpurun_out
for W in 100 3000; do
  timeout 600 python tools/bench_kbrl.py --warmup $W --steps 200 > gpurun_out/${TAG}_kbrl_w$W.json 2> gpurun_out/${TAG}_kbrl_w$W.err
  python - gpurun_out/${TAG}_kbrl_w$W.json <<'PY'
import json, sys
try:
    l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%s: %.0f env-steps/s  %.3f ms/step  embb %.3f ms  kb mean %.3f ms  dict mean %.0f max %d  pool %.1f GB' % (
        sys.argv[1], l['env_steps_per_s'], l['ms_per_step'], l['embb_kernel_ms'], l['kb_kernel_ms_mean_of_update_and_select'],
        l['dictionary_size_mean'], l['dictionary_size_max'], l['pool']['used_bytes'] / 2**30))
    print('   direct passes/step %.2f, direct landmarks/step %.1f' % (l['direct_passes_per_step'], l['direct_landmarks_per_step']))
    print('   update phase %.3f ms, select %.3f ms; Kinv streaming: %s' % (l['kb_update_phase_ms'], l['kb_select_ms'], json.dumps(l['kinv_streaming'])))
except Exception as e:
    print('failed', e); print(open(sys.argv[1].replace('.json', '.err')).read()[-2000:])
PY
done

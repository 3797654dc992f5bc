ST=/tmp/late_ab
python tools/bench_kbrl.py --profile tdl --warmup 3000 --save-state $ST > /dev/null || exit 1
for rep in 1 2; do
 for V in 1 0; do
  if [ $V = 1 ]; then export KBRL_BIN_TWO_LAUNCHES=1; else unset KBRL_BIN_TWO_LAUNCHES; fi
  RANSLICE_LIB=$PWD/network-slicing_amd/csrc/build/libranslice_dev.so python tools/bench_kbrl.py --profile tdl --load-state $ST --steps 300 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('two_launches=$V: env_steps_per_s %.5g ms_per_step %.4f embb_kernel_ms %.4f update %.4f select %.4f bin %.4f' % (d['env_steps_per_s'], d['ms_per_step'], d['embb_kernel_ms'], d['kb_update_phase_ms'], d['kb_select_ms'], d['per_step_ms']['select_bin']))"
 done
done

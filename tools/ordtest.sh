for l in 8 4 2; do export RANSLICE_LIB=network-slicing_amd/csrc/build/libranslice_lpu$l.so; echo "LPU $l"; python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
for m in 0 6; do RANSLICE_ORDER=$m python tools/group_sweep.py 4096 2>&1 | grep "group=16"; RANSLICE_ORDER=$m python tools/bench_kbrl.py --envs 4096 --steps 100 --warmup 200 --capacity 256 | python -c "
import json,sys; l=json.loads(sys.stdin.readline()); print('kbrl loop: %.0f env-steps/s, embb %.3f ms' % (l['env_steps_per_s'], l['embb_kernel_ms']))"; done; done

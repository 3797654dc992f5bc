"""Developer aid: cycles inside a contested PF round (tools/pf_micro_patch.py apply; make -C network-slicing_amd/csrc profile;
RANSLICE_LIB=network-slicing_amd/csrc/build/libranslice_prof.so PROFILE_ENVS=30 python tools/pf_micro_profile.py [--random])"""
import ctypes as C, os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT,'network-slicing_amd'))
import numpy as np
from ranslice.config import make_config
from ranslice.fading import synth_fading
from ranslice.vec_env import VecRanSlice
from ranslice.kbrl_dev import VecKBRL
N=int(os.environ.get('PROFILE_ENVS','30'))
env=VecRanSlice(n_envs=N,cfg=make_config(0,n_envs=N),fading=[synth_fading(t,10000) for t in range(3)])
RANDOM='--random' in sys.argv     # the bench's random action script on the plain instance instead of agents on the BLOCK one
env.reset()
step_no=[0]
if RANDOM:
    def adv():
        env.random_actions(2024, step_no[0]); step_no[0]+=1; env.step_resident()
    for i in range(1000): adv()
else:
    env.set_schedule_hint(1)
    agent=VecKBRL(N,[10]*5,200,capacity=512)
    rng=np.random.default_rng(0)
    ia=rng.integers(4,20,size=(N,5)).astype(np.int32)
    agent.reset(ia,rng.integers(2,8,size=(N,5)).astype(np.int32))
    env._check(env.L.rs_step(env.h, ia.ctypes.data_as(C.POINTER(C.c_int32)),None,None,None,None))
    def adv():
        agent.step_resident(env); env.step_resident()
    for i in range(300): adv()
env.synchronize()
a=(C.c_uint64*16)(); env.L.rs_get_section_profile(env.h,a); base=list(a)
K=100
for i in range(K): adv()
env.synchronize(); env.L.rs_get_section_profile(env.h,a)
d=[a[i]-base[i] for i in range(16)]
rounds=d[15]; runs=d[13]
names={0:'loop top (control)',1:'block: contenders, B',2:'block: pass 1 loop + B-th key',3:'rank: limit + commit',10:'rank: set-up + keys',11:'rank: ranking',4:'block: max + u*',5:'block: pass 2',6:'block: tail (sum, metric divide)',7:'trip: reductions',8:'trip: leader run / closed form',9:'trip: take broadcast + update',12:'everything else (the slot outside contested PF rounds)'}
tot=sum(d[i] for i in names)
print('N=%d: PF rounds (wave-level) %d, leader-run iterations (lane-level sum) %d over %d steps' % (N, rounds, runs, K))
for i in names: print('%-40s %6.2f%%  %9.0f cycles per wave-round' % (names[i], 100.0*d[i]/tot, d[i]/max(1,rounds)))

raw=np.zeros(N*5*4+16,dtype=np.uint64)
env.L.rs_get_task_profile(env.h, raw.ctypes.data_as(C.POINTER(C.c_uint64)))
sw=raw[N*5*4:].astype(np.float64)
tots=sum(sw[i] for i in names)
print('slowest wave of the run: %.0f cycles per slot, %.1f PF rounds per slot (%.1f rank rounds), %.2f leader-run iterations per round' % (tots/50, sw[15]/50, sw[14]/50, sw[13]/max(1.0,sw[15])))
for i in names: print('   %-40s %6.2f%%  %9.0f cycles per round  %9.0f per slot' % (names[i], 100.0*sw[i]/tots, sw[i]/max(1.0,sw[15]), sw[i]/50))

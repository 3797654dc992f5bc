# reference-scale evaluation on the device: experiments_kbrl.BatchedEvaluator, RUNS x STEPS of one (scenario, range)
# PROFILE = sos (the fixture profile) | tdl (tapped-delay-line traces); CAP = dictionary capacity (a limit), POOL_GB = pool
STEPS=${STEPS:-50400}; RUNS=${RUNS:-30}; PROFILE=${PROFILE:-sos}; CAP=${CAP:-16384}; POOL_GB=${POOL_GB:-96}
cd network-slicing_amd && python -c "
import experiments_kbrl as ek, scenario_creator as sc, time, numpy as np
from ranslice.fading import synth_traces
sc.set_fading(synth_traces(10000, '$PROFILE'))
t=time.time()
ev=ek.BatchedEvaluator(0,[0.99,0.999],steps=$STEPS,out_dir='/tmp/res_$PROFILE')
f=ev.evaluate_all(range($RUNS), capacity=$CAP, pool_bytes=int($POOL_GB*2**30), verbose=False)
dt=time.time()-t
r=[np.load(x) for x in f]
h=$STEPS//2
print('%d runs x %d steps of scenario_0 on %s traces in %.1f s (%.2f ms/step); violations/step %.4f (second half %.4f), mean PRBs %.1f, adjusted %.3f, hit rate %.3f' % ($RUNS, $STEPS, '$PROFILE', dt, 1e3*dt/$STEPS, np.mean([x['violation'].mean() for x in r]), np.mean([x['violation'][h:].mean() for x in r]), np.mean([x['resources'].mean() for x in r]), np.mean([x['adjusted'].mean() for x in r]), np.mean([x['hits'].mean() for x in r])))
print('dictionaries at the end: max %d, mean %.1f landmarks (capacity %d); pool %s' % (ev.last_run['max_dictionary'], ev.last_run['mean_dictionary'], $CAP, ev.last_run['pool']))
" 2>&1 | tail -4

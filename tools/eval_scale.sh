# reference-scale evaluation on the device: experiments_kbrl.BatchedEvaluator, RUNS x STEPS of one (scenario, range)
STEPS=${STEPS:-50400}; RUNS=${RUNS:-30}
cd network-slicing_amd && python -c "
import experiments_kbrl as ek, time, numpy as np
t=time.time()
f=ek.BatchedEvaluator(0,[0.99,0.999],steps=$STEPS,out_dir='/tmp/res').evaluate_all(range($RUNS), verbose=False)
dt=time.time()-t
r=[np.load(x) for x in f]
print('%d runs x %d steps of scenario_0 in %.1f s (%.2f ms/step); violations/step %.4f, mean PRBs %.1f, adjusted %.3f, hit rate %.3f' % ($RUNS, $STEPS, dt, 1e3*dt/$STEPS, np.mean([x['violation'].mean() for x in r]), np.mean([x['resources'].mean() for x in r]), np.mean([x['adjusted'].mean() for x in r]), np.mean([x['hits'].mean() for x in r])))
" 2>&1 | tail -3

cd network-slicing_amd && (time python -c "
import experiments_kbrl as ek, time, numpy as np
t=time.time()
f=ek.BatchedEvaluator(0,[0.99,0.999],steps=50400,out_dir='/tmp/res').evaluate_all(range(30), verbose=False)
dt=time.time()-t
r=[np.load(x) for x in f]
print('30 runs x 50400 steps of scenario_0 in %.1f s; violations/step %.4f, mean PRBs %.1f, adjusted %.3f, hit rate %.3f' % (dt, np.mean([x['violation'].mean() for x in r]), np.mean([x['resources'].mean() for x in r]), np.mean([x['adjusted'].mean() for x in r]), np.mean([x['hits'].mean() for x in r])))
") 2>&1 | tail -8

# the reference's whole experiment grid on the device as ONE job (experiments_kbrl.evaluate_grid): CELLS x RUNS x STEPS
STEPS=${STEPS:-50400}; RUNS=${RUNS:-30}; PROFILE=${PROFILE:-tdl}; CAP=${CAP:-16384}; POOL_GB=${POOL_GB:-16}; SCN=${SCN:-"0 1 2"}
cd network-slicing_amd && python -c "
import experiments_kbrl as ek, scenario_creator as sc, time, numpy as np
from itertools import product
from ranslice.fading import synth_traces
sc.set_fading(synth_traces(10000, '$PROFILE'))
cells = list(product([int(x) for x in '$SCN'.split()], ek.accuracy_list))
t=time.time()
out=ek.evaluate_grid(cells, range($RUNS), steps=$STEPS, out_dir='/tmp/grid_$PROFILE', capacity=$CAP, pool_bytes=int($POOL_GB*2**30))
dt=time.time()-t
print('%d cells x %d runs x %d steps on %s traces as one job in %.1f s (%.2f ms per step of all cells)' % (len(cells), $RUNS, $STEPS, '$PROFILE', dt, 1e3*dt/$STEPS))
for c in cells:
    r=[np.load(x) for x in out[(c[0], c[1][0])]]
    lr=ek.evaluate_grid.last_runs[(c[0], c[1][0])]
    print('  scenario %d, accuracy range %s: violations/step %.4f, mean PRBs %.1f, adjusted %.3f, hit rate %.3f; dictionaries max %d mean %.0f, pool %.1f GB' % (c[0], c[1], np.mean([x['violation'].mean() for x in r]), np.mean([x['resources'].mean() for x in r]), np.mean([x['adjusted'].mean() for x in r]), np.mean([x['hits'].mean() for x in r]), lr['max_dictionary'], lr['mean_dictionary'], lr['pool']['used_bytes']/2**30))
" 2>&1 | tail -9

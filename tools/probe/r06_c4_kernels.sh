# config 4, lean timing, three processes: ms per step and the in-step kernel table (instrumented pass)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do python $R/bench.py --workload c4 --steps 100 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('c4', round(r['ms_per_step'],4), {k:round(v['avg_ms_timed_region'],4) for k,v in r['kernels'].items()}, 'e_pme', r['config'].get('e_pme'))"; done

# config workloads, lean timing: ms per step of c2 / c3 / c4 (two processes each)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for w in c4 c4 c3 c3 c2 c2; do python $R/bench.py --workload $w --steps 100 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('$w', round(r['ms_per_step'],4), 'instrumented median', round(r['stats']['step_ms_median'],4), r['roofline']['kernel'], round(r['roofline']['launch_ms'],4), round(r['roofline']['frac'],3))"; done

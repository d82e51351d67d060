# D3 energy pass in two launches with a yield point between them (an empty one-wave kernel: the CUs drain for a few microseconds), so that the PME
# mesh solve's plane kernel (one block = a whole CU's LDS) can be placed before the end of the pass.  NVALCHEMIOPS_D3_ENERGY_SPLIT = percent of the
# grid in the first launch, NVALCHEMIOPS_D3_ENERGY_YIELDS = empty launches at the boundary
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "0 1" "40 1" "50 1" "40 2" "0 1" "45 1" "60 1" "50 3"; do
  set -- $cfg
  export NVALCHEMIOPS_D3_ENERGY_SPLIT=$1 NVALCHEMIOPS_D3_ENERGY_YIELDS=$2
  python $R/bench.py --processes 1 --steps 100 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('split $1 yields $2', round(r['ms_per_step'],4), {n:round(k[n]['avg_ms_timed_region'],3) for n in ('d3_energy','d3_chain','pme_solve_fwd','pme_solve_cols','pme_solve_inv','pme_gather_finish') if n in k})"
done

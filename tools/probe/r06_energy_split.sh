# D3 energy pass in two launches (NVALCHEMIOPS_D3_ENERGY_SPLIT = percent of the grid in the first): does the PME mesh solve get its CUs earlier?
R=${GRAFT_REPO_ROOT:-$(pwd)}
for s in 0 30 40 50 0 35 60; do
  export NVALCHEMIOPS_D3_ENERGY_SPLIT=$s
  python $R/bench.py --processes 1 --steps 100 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('split', $s, round(r['ms_per_step'],4), {n:round(k[n]['avg_ms_timed_region'],3) for n in ('d3_energy','d3_chain','pme_solve_fwd','pme_solve_cols','pme_solve_inv','pme_gather_finish') if n in k})"
done

# schedule 5 (three streams: reciprocal half | 9 A list + real-space sum | 40-Bohr list + D3) against the default two-stream schedule, alternating, one process each
R=${GRAFT_REPO_ROOT:-$(pwd)}
for o in 1 5 1 5 1 5; do
  python $R/bench.py --processes 1 --steps 100 --cpu-sample 0 --overlap $o 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('overlap $o', round(r['ms_per_step'],4), {n:round(k[n]['avg_ms_timed_region'],3) for n in ('nl_query_matrix_f32','d3_energy','d3_chain','nl_query_matrix_f64','ewald_real','spline_spread','pme_solve_fwd','pme_solve_cols','pme_solve_inv','pme_gather_finish') if n in k})"
done

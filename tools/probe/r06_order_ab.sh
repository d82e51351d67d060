# schedule 1 (PME branch as particle_mesh_ewald: real space, then reciprocal) vs schedule 4 (reciprocal half first, real-space sum last), same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { env BENCH_CONFIGS=0 BENCH_CALIB=0 python $R/bench.py --processes 1 --steps 100 --cpu-sample 0 "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
g=lambda n: round(k.get(n,{}).get('avg_ms_timed_region',0),3)
print(round(r['ms_per_step'],4), {n:g(n) for n in ('nl_query_matrix_f32','d3_energy','d3_chain','nl_query_matrix_f64','ewald_real','pme_solve_fwd','pme_solve_cols','pme_solve_inv','pme_gather_finish','spline_spread')})"; }
for rep in 1 2 3; do
  echo "overlap 1   $(run --overlap 1)"
  echo "overlap 4   $(run --overlap 4)"
done

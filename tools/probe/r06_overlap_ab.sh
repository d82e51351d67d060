# same-box A/B of stream priorities / persistent grid of the 40-Bohr fill on the round-6 build (the CN pass is gone from the main stream: the
# electrostatics stream's tail now sticks out behind it).  ms per step of bench.py --processes 1 --steps 100 --cpu-sample 0, alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { env "$@" BENCH_CONFIGS=0 BENCH_CALIB=0 python $R/bench.py --processes 1 --steps 100 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
g=lambda n: round(k.get(n,{}).get('avg_ms_timed_region',0),3)
print(round(r['ms_per_step'],4), {n:g(n) for n in ('nl_query_matrix_f32','d3_energy','d3_chain','nl_query_matrix_f64','ewald_real','pme_solve_fwd','pme_solve_cols','pme_solve_inv','pme_gather_finish','spline_spread')})"; }
for rep in 1 2; do
  echo "default              $(run X=1)"
  echo "SIDE_PRIORITY=-1     $(run BENCH_SIDE_PRIORITY=-1)"
  echo "MAIN_PRIORITY=-1     $(run BENCH_MAIN_PRIORITY=-1)"
  echo "grid 1280            $(run NVALCHEMIOPS_NL_TILED_GRID=1280)"
  echo "grid 1024            $(run NVALCHEMIOPS_NL_TILED_GRID=1024)"
  echo "grid 1280 side -1    $(run NVALCHEMIOPS_NL_TILED_GRID=1280 BENCH_SIDE_PRIORITY=-1)"
done

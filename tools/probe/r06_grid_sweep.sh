# persistent-grid sweep of the tiled list kernel on the round-6 build, two-stream step and serial step (isolated fill)
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { env "$@" BENCH_CONFIGS=0 BENCH_CALIB=0 python $R/bench.py --processes 1 --steps 100 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
g=lambda n: round(k.get(n,{}).get('avg_ms_timed_region',0),3)
i=lambda n: round(k.get(n,{}).get('isolated_median_ms',0) or 0,3)
print(round(r['ms_per_step'],4), 'serial', round(r['stats']['step_ms_median_serial_untimed'],3), 'fill iso', i('nl_query_matrix_f32'), 'f64 iso', i('nl_query_matrix_f64'), {n:g(n) for n in ('nl_query_matrix_f32','d3_energy','d3_chain','nl_query_matrix_f64','ewald_real','pme_solve_fwd','pme_solve_cols')})"; }
for rep in 1 2; do
  for g in 1536 1152 1024 896 768 640; do
    echo "grid $g   $(run NVALCHEMIOPS_NL_TILED_GRID=$g)"
  done
done

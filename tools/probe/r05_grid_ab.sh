for g in 1536 1280 1024 768 1536 1024; do
  NVALCHEMIOPS_NL_TILED_GRID=$g timeout 300 python bench.py --processes 1 --cpu-sample 0 --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('grid $g step', round(d['ms_per_step'],3), {n: round(v['avg_ms_timed_region'],3) for n,v in k.items() if n in ('nl_query_matrix_f32','d3_cn','d3_energy','d3_chain','nl_query_matrix_f64','pme_solve_cols','ewald_real')}, 'trial', d['config']['d3_list_buffers'].get('trial_ms'))"
done

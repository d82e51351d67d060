#!/bin/bash
# A/B of the staged row output of the matrix fill (NVALCHEMIOPS_NL_STAGED=0/1), alternating, stand-alone list and full step
export BENCH_CALIB=0
for i in 1 2; do for s in 0 1; do
  echo "== staged=$s"
  NVALCHEMIOPS_NL_STAGED=$s timeout 100 python tools/nl_modes.py matrix 2>&1 | grep -v "amdgpu\|calib"
  NVALCHEMIOPS_NL_STAGED=$s timeout 250 python bench.py --processes 1 --steps 40 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']
t=lambda n: k[n]['median_ms_timed_region']
print('step %.3f ms (serial %.3f) | nl_f32 %.3f iso %.3f | nl_f64 %.3f | cn %.3f energy %.3f chain %.3f' % (d['ms_per_step'], d['stats']['step_ms_median_serial_untimed'], t('nl_query_matrix_f32'), k['nl_query_matrix_f32']['isolated_median_ms'], t('nl_query_matrix_f64'), t('d3_cn'), t('d3_energy'), t('d3_chain')))"
done; done

#!/bin/bash
# Same-box A/B of several builds of the library (box-to-box spread on the pool is +-5-10 %, so only same-box numbers compare):
#   tools/ab_libs.sh <rounds> <name1> <name2> ...     with lib/alt_<name>.so prepared beforehand; the LAST one stays installed
rounds=$1; shift
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
for r in $(seq $rounds); do for v in "$@"; do
  cp $L/alt_$v.so $L/libnvalchemiops_hip.so
  timeout 300 python bench.py --processes 1 --steps 40 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']
g=lambda n: round(k[n]['isolated_median_ms'],4) if n in k and k[n]['isolated_median_ms'] else None
print('%-10s step %.3f (median %.3f) | nl_f32 %s cn %s energy %s chain %s | nl_f64 %s ewald %s spread %s gather %s build %s' % ('$v', d['ms_per_step'], d['stats']['step_ms_median'],
      g('nl_query_matrix_f32'), g('d3_cn'), g('d3_energy'), g('d3_chain'), g('nl_query_matrix_f64'), g('ewald_real'), g('spline_spread'), g('pme_gather_finish'),
      [round(v['isolated_median_ms'],4) for n,v in k.items() if n.startswith('nl_build')]))"
done; done

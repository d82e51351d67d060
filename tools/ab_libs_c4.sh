#!/bin/bash
# Same-box A/B of several builds of the library on config 4 (the electrostatics branch alone):
#   tools/ab_libs_c4.sh <rounds> <name1> <name2> ...     with lib/alt_<name>.so prepared beforehand; the LAST one stays installed
rounds=$1; shift
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
for r in $(seq $rounds); do for v in "$@"; do
  cp $L/alt_$v.so $L/libnvalchemiops_hip.so
  timeout 300 python bench.py --workload c4 --steps 200 --warmup 20 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']
g=lambda n: round(k[n]['median_ms_timed_region'],4) if n in k and k[n].get('median_ms_timed_region') else None
print('%-10s step %.4f | nl_f64 %s ewald %s spread %s solve %s gather %s' % ('$v', d['ms_per_step'], g('nl_query_matrix_f64'), g('ewald_real'), g('spline_spread'),
      [g(n) for n in k if n.startswith('pme_solve')], g('pme_gather_finish')))"
done; done

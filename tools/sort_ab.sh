#!/bin/bash
# A/B of the spatial order inside mi_d3 (NVALCHEMIOPS_D3_SORT = 0 never / 1 always / unset: decided from the measured atom order) and of the
# tile-ordered PME gather, on the headline box with its atoms in lattice order and in random order (BENCH_SHUFFLE_ATOMS=1)
export BENCH_CALIB=0
for sh in 0 1; do for mode in 0 auto 1; do
  if [ $mode = auto ]; then unset NVALCHEMIOPS_D3_SORT; else export NVALCHEMIOPS_D3_SORT=$mode; fi
  BENCH_SHUFFLE_ATOMS=$sh timeout 250 python bench.py --processes 1 --steps 30 --warmup 5 --cpu-sample 0 --overlap 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']
t=lambda n: k[n]['median_ms_timed_region']
print('shuffled=$sh d3_sort=$mode  step %.3f | nl_f32 %.3f | cn %.3f energy %.3f chain %.3f | ewald %.3f gather %.3f spread %.3f | %r' % (d['ms_per_step'], t('nl_query_matrix_f32'), t('d3_cn'), t('d3_energy'), t('d3_chain'), t('ewald_real'), t('pme_gather_finish'), t('spline_spread'), d.get('energies')))"
done; done

#!/bin/bash
# Does the in-run HBM calibration (8 GiB of scratch allocated, swept and freed before the warm-up) change the step that follows it?
for i in 1 2 3; do for c in 0 1; do
BENCH_CALIB=$c timeout 250 python bench.py --processes 1 --steps 40 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']
t=lambda n: k[n]['median_ms_timed_region']
print('BENCH_CALIB=$c step %.3f ms | nl_f32 %.3f iso %.3f | cn %.3f energy %.3f chain %.3f' % (d['ms_per_step'], t('nl_query_matrix_f32'), k['nl_query_matrix_f32']['isolated_median_ms'], t('d3_cn'), t('d3_energy'), t('d3_chain')))"
done; done

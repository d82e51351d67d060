export TMPDIR=/tmp BENCH_CALIB=0
for g in 1536 1280 1024 768 2048; do
NVALCHEMIOPS_NL_TILED_GRID=$g timeout 300 python bench.py --processes 1 --steps 40 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']
g=lambda n: round(k[n]['isolated_median_ms'],4)
t=lambda n: round(k[n]['median_ms_timed_region'],4)
print('grid $g step %.3f (median %.3f) serial %.3f | iso nl_f32 %s nl_f64 %s | timed nl_f32 %s nl_f64 %s cn %s ewald %s' % (d['ms_per_step'], d['stats']['step_ms_median'], d['stats']['step_ms_median_serial_untimed'], g('nl_query_matrix_f32'), g('nl_query_matrix_f64'), t('nl_query_matrix_f32'), t('nl_query_matrix_f64'), t('d3_cn'), t('ewald_real')))"
done

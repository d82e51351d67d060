#!/bin/bash
# The warm-up drift of the 40-Bohr list fill next to what rocm-smi reports (temperatures, clocks, power) between back-to-back bench runs.
export BENCH_CALIB=0
smi() { rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -E "Temperature|sclk|mclk|fclk|socclk|Power" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';' ; echo; }
echo "before: $(smi)"
for i in 1 2 3 4 5 6; do
  timeout 250 python bench.py --processes 1 --steps 40 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']
t=lambda n: k[n]['median_ms_timed_region']
print('run $i step %.3f ms | nl_f32 %.3f | cn %.3f energy %.3f chain %.3f' % (d['ms_per_step'], t('nl_query_matrix_f32'), t('d3_cn'), t('d3_energy'), t('d3_chain')))"
  echo "  smi: $(smi)"
done

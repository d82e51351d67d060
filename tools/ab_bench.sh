#!/bin/bash
# A/B of an environment knob on ONE box (box-to-box spread on the pool is +-4 %): tools/ab_bench.sh VAR "bench args" [reps]
var=$1; args=$2; reps=${3:-2}
for r in $(seq $reps); do for v in 1 0; do
  env $var=$v timeout 300 python bench.py --processes 1 $args 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$var=$v', round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['kernel_ms'].items() if k.startswith('d3') or k.startswith('nl_query_matrix_f32')})"
done; done

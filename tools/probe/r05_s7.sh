#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s7
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.load(open('$1')); print('%-28s' % '$2', round(d['ms_per_step'],4), {k.replace('nl_query_matrix','nlq'): round(v['avg_ms_timed_region'],4) for k,v in d['kernels'].items()})"; }
for rep in 1 2; do
for pol in 0 auto; do
  NVALCHEMIOPS_NL_PACKED=$pol timeout 200 python bench.py --workload c3 --steps 100 --cpu-sample 0 > $OUT/c3_$pol.json 2>> $OUT/err.log; show $OUT/c3_$pol.json c3_packed_$pol
done
done
for pol in 0 auto; do
  NVALCHEMIOPS_NL_PACKED=$pol timeout 300 python bench.py --workload c5 --steps 20 --warmup 3 --cpu-sample 0 > $OUT/c5_$pol.json 2>> $OUT/err.log; show $OUT/c5_$pol.json c5_packed_$pol
done
tail -3 $OUT/err.log

#!/bin/bash
# Round 5, GPU session 3: fixed tests, CU-partitioned streams, kernel timeline of the two-stream step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s3
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_packed_companion_gpu.py tests/test_pme_gpu.py tests/test_autograd_gpu.py -m gpu -q -x > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?"
tail -6 $OUT/pytest_new.log
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --processes 1 --steps 100 --cpu-sample 0 > $OUT/bench_$name.json 2>> $OUT/bench.err
  python -c "
import json; d=json.load(open('$OUT/bench_$name.json')); print('%-22s' % '$name', round(d['ms_per_step'],3), 'serial', round(d['stats']['step_ms_median_serial_untimed'],3), {k.replace('nl_query_matrix','nlq').replace('pme_',''): round(v['avg_ms_timed_region'],3) for k,v in d['kernels'].items() if 'build' not in k})" 2>&1 | tail -1
}
echo "# CU-partitioned streams: side stream = k CUs per XCD, main stream = the other 32 - k (or all 256)" | tee $OUT/cu_mask.log
for rep in 1 2; do
  run base_$rep A=1 | tee -a $OUT/cu_mask.log
  for k in 2 4 6 8; do
    run side${k}_compl_$rep BENCH_CU_SIDE=$k BENCH_CU_MAIN=complement | tee -a $OUT/cu_mask.log
  done
  run side4_mainall_$rep BENCH_CU_SIDE=4 BENCH_CU_MAIN=all | tee -a $OUT/cu_mask.log
  run side8_mainall_$rep BENCH_CU_SIDE=8 BENCH_CU_MAIN=all | tee -a $OUT/cu_mask.log
done
run side4_block BENCH_CU_SIDE=4 BENCH_CU_MAIN=complement BENCH_CU_LAYOUT=block | tee -a $OUT/cu_mask.log
# timeline of the default two-stream step
cd /tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace -d /tmp/prof_tl -- python $R/bench.py --processes 1 --steps 6 --warmup 3 --cpu-sample 0 > /tmp/prof_tl.log 2>&1
python $R/tools/rocpd_timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) nl_setup 120 > $OUT/timeline_default.txt 2>&1
head -70 $OUT/timeline_default.txt

#!/bin/bash
# One GPU-box session: parity suite, default bench, the reference's published configurations, the N > 1 launcher.
# Usage (from the repo root on the GPU box):  bash tools/gpu_session.sh <tag> [tests|bench|ref|all]
TAG=${1:-s1}
WHAT=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
if [[ $WHAT == all || $WHAT == tests ]]; then
  timeout 1800 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  tail -30 $OUT/pytest.log
  cp gpurun_out/d3_error_budget.json $OUT/ 2>/dev/null
fi
if [[ $WHAT == all || $WHAT == bench ]]; then
  timeout 600 python bench.py --processes 1 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
  timeout 300 python bench.py --processes 1 --steps 20 --warmup 5 --cpu-sample 0 > $OUT/bench_k20.json 2>> $OUT/bench_default.err
  timeout 300 python bench.py --processes 1 --overlap 0 --steps 50 --cpu-sample 0 > $OUT/bench_serial.json 2>> $OUT/bench_default.err
  timeout 300 python bench.py --workload c5 --steps 20 --warmup 3 > $OUT/bench_c5.json 2>> $OUT/bench_default.err
  timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --cpu-sample 0 > $OUT/bench_gpus2.json 2> $OUT/bench_gpus2.err; echo "gpus2 rc=$?"
  head -c 1500 $OUT/bench_default.json; echo
fi
if [[ $WHAT == all || $WHAT == ref ]]; then
  for w in ref-nlist ref-d3 ref-pme; do
    timeout 300 python bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w rc=$?"
    head -c 1200 $OUT/bench_$w.json; echo
  done
fi

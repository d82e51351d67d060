#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s17
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_c5_gpu.py tests/test_pme_gpu.py tests/test_distributed_gpu.py tests/test_reference_scenarios_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 400 python bench.py --workload c5 --steps 20 --warmup 3 > $OUT/bench_c5.json 2>> $OUT/err.log; echo "c5 rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench_c5.json')); print(round(d['ms_per_step'],3), {k: round(v['avg_ms_timed_region'],4) for k,v in d['kernels'].items()})"

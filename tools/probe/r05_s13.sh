#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s13
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 300 python bench.py --processes 1 --steps 100 --cpu-sample 0 > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(round(d['ms_per_step'],3), 'serial', round(d['stats']['step_ms_median_serial_untimed'],3), {k: round(v['avg_ms_timed_region'],3) for k,v in d['kernels'].items()})"

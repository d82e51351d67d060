#!/bin/bash
# Round-4 closing run on one box: the GPU suite, then the bench lines and kernel statistics profiles/ holds (reduced: the PMC passes of
# tools/profile_round.sh r04b stay valid for the kernels that did not change).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final_r04
mkdir -p $OUT
cd $R
python -m pytest tests -q -m gpu --tb=line 2>&1 | grep -E "^/root|Error|Warning: hipFFT|passed|failed" | cut -c1-250 | tail -8 | tee $OUT/pytest_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
export BENCH_CALIB=0
BENCH_GRAPH=1 python $R/bench.py --workload c4 > $OUT/bench_c4.json 2>> $OUT/bench.err
rm -rf /tmp/ps4; rocprofv3 --kernel-trace --stats -d /tmp/ps4 -- python $R/bench.py --workload c4 --processes 1 --steps 30 --cpu-sample 0 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/ps4 -name "*.db" | head -1) $OUT/kernel_stats_c4.csv > /dev/null 2>&1
rm -rf /tmp/ps; rocprofv3 --kernel-trace --stats -d /tmp/ps -- python $R/bench.py --processes 1 --cpu-sample 0 --steps 50 > $OUT/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/ps -name "*.db" | head -1) $OUT/kernel_stats.csv > /dev/null 2>&1
python $R/bench.py --workload c5 --cpu-sample 0 --processes 1 > $OUT/bench_c5.json 2>> $OUT/bench.err
python $R/bench.py --workload pme-train > $OUT/bench_pme_train.json 2>> $OUT/bench.err
python $R/bench.py --workload c3 > $OUT/bench_c3.json 2>> $OUT/bench.err
python $R/bench.py --workload c2 > $OUT/bench_c2.json 2>> $OUT/bench.err
python $R/bench.py --processes 1 --overlap 0 --cpu-sample 0 --steps 50 > $OUT/bench_serial.json 2>> $OUT/bench.err
ls $OUT

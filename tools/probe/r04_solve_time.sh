#!/bin/bash
# parity of the fused mesh solve + its per-kernel times in config 4 (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python -m pytest tests/test_pme_gpu.py -q -m gpu -x -k "fused_mesh_solve" 2>&1 | tail -3
BENCH_CALIB=0 timeout 200 python bench.py --workload c4 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', round(d['ms_per_step'],4), {n: round(v['avg_ms_timed_region'],4) for n,v in d['kernels'].items()})"
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ps
BENCH_CALIB=0 rocprofv3 --kernel-trace --stats -d /tmp/ps -- python $R/bench.py --workload c4 --processes 1 --steps 20 --cpu-sample 0 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/ps -name "*.db" | head -1) /tmp/ks.csv >/dev/null 2>&1
grep "pme_solve\|fft_rtc\|pme_convolve" /tmp/ks.csv | sed 's/_ZN12_GLOBAL__N_1//' | cut -c1-40,100-200

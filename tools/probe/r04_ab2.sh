#!/bin/bash
# same-box A/B: density threshold for 2 cells per cutoff (NL_K2_APC) on config 4's 9 A fp64 list and config 2's 5 A fp32 list
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp BENCH_CALIB=0
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
O=gpurun_out/r04_ab2.log; : > $O
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(sys.argv[1], "step %.4f" % d["ms_per_step"], {n: round(v["avg_ms_timed_region"],4) for n,v in k.items()})'
for r in 1 2; do for lib in k64 k40 k24; do
  cp $L/alt_$lib.so $L/libnvalchemiops_hip.so
  for w in c4 c2; do timeout 200 python bench.py --workload $w --steps 100 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$show" "$w $lib" >> $O; done
done; done
cp $L/alt_k24.so $L/libnvalchemiops_hip.so
python -m pytest tests/test_nlist_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $O
cat $O

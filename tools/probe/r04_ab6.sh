#!/bin/bash
# Round-4 A/B: occupancy of the fp64 tiled list kernel (centres per wave NC, LDS tile) on config 4; the last variant stays installed = base
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp BENCH_CALIB=0
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
O=gpurun_out/r04_ab6.log; : > $O
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(sys.argv[1], "step %.4f" % d["ms_per_step"], {n: round(v["avg_ms_timed_region"],4) for n,v in k.items() if n.startswith("nl_")})'
for r in 1 2; do for lib in base nc1 nc1t512 nc1t384 nc2t512; do
  cp $L/alt_$lib.so $L/libnvalchemiops_hip.so
  timeout 200 python bench.py --workload c4 --steps 100 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$show" "c4 $lib" >> $O
done; done
cp $L/alt_base.so $L/libnvalchemiops_hip.so
for w in c3 c5; do timeout 300 python bench.py --workload $w --steps 40 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$show" "$w base(new setup)" >> $O; done
python -m pytest tests/test_nlist_gpu.py tests/test_c5_gpu.py tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $O
cat $O

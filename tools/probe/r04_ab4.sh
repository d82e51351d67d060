#!/bin/bash
# Round-4 A/B: real-space sum and reciprocal front half of one particle_mesh_ewald call on two streams (NVALCHEMIOPS_PME_FORK) -- config 4 and headline
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp BENCH_CALIB=0
O=gpurun_out/r04_ab4.log; : > $O
python -m pytest tests/test_pme_gpu.py tests/test_autograd_gpu.py tests/test_compile_gpu.py tests/test_c5_gpu.py tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -4 >> $O
python -m pytest tests -x -q -m gpu -k "capturable or graph" 2>&1 | tail -2 >> $O
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(sys.argv[1], "step %.4f" % d["ms_per_step"], {n: round(v["avg_ms_timed_region"],4) for n,v in k.items() if not n.startswith("d3") })'
for r in 1 2; do for f in 0 1; do
  NVALCHEMIOPS_PME_FORK=$f timeout 200 python bench.py --workload c4 --steps 100 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$show" "c4 fork=$f" >> $O
  NVALCHEMIOPS_PME_FORK=$f timeout 300 python bench.py --processes 1 --steps 60 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$show" "headline fork=$f" >> $O
done; done
NVALCHEMIOPS_PME_FORK=1 timeout 300 python bench.py --workload c5 --steps 30 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$show" "c5 fork=1" >> $O
NVALCHEMIOPS_PME_FORK=0 timeout 300 python bench.py --workload c5 --steps 30 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$show" "c5 fork=0" >> $O
python bench.py --workload pme-train 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for r in d['rows']: print('train', r['spline_order'], r['loss'], 'infer %.3f fwd %.3f bwd %.3f' % (r['forward_inference_ms'], r['forward_ms'], r['backward_ms']))" >> $O
cat $O

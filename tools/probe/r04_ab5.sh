#!/bin/bash
# Round-4 A/B of the two-stream schedule: 1 (default) vs 4 (reciprocal half of PME behind mi_d3's after-CN event), same box, alternating, fresh processes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp BENCH_CALIB=0
O=gpurun_out/r04_ab5.log; : > $O
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; g=lambda n: round(k[n]["avg_ms_timed_region"],3) if n in k else None; print(sys.argv[1], "step %.3f median %.3f | list40 %s cn %s energy %s chain %s | list9 %s ewald %s spread %s c2r %s gather %s | e_pme %.6f e_d3 %.6f" % (d["ms_per_step"], d["stats"]["step_ms_median"], g("nl_query_matrix_f32"), g("d3_cn"), g("d3_energy"), g("d3_chain"), g("nl_query_matrix_f64"), g("ewald_real"), g("spline_spread"), g("fft_c2r"), g("pme_gather_finish"), d["energies"]["e_pme"], d["energies"]["e_d3_Ha"]))'
for r in 1 2 3; do for o in 1 4; do
  timeout 300 python bench.py --processes 1 --steps 60 --warmup 5 --cpu-sample 0 --overlap $o 2>/dev/null | tail -1 | python -c "$show" "overlap $o" >> $O
done; done
cat $O

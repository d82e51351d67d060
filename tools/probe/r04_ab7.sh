#!/bin/bash
# Round-4 A/B: persistent blocks of the tiled list kernels.  1536 blocks of the 40-Bohr fill hold all the LDS of the chip (5 x 31 KB per CU),
# so the side stream's 9 A list (36 KB of LDS per block) cannot start before the fill's persistent blocks retire (in-step 1.1 ms vs 0.2 isolated).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp BENCH_CALIB=0
O=gpurun_out/r04_ab7.log; : > $O
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; g=lambda n: round(k[n]["avg_ms_timed_region"],3) if n in k else None; print(sys.argv[1], "step %.3f serial %.3f | list40 %s (iso %.3f) cn %s energy %s chain %s | list9 %s ewald %s spread %s c2r %s gather %s" % (d["ms_per_step"], d["stats"]["step_ms_median_serial_untimed"], g("nl_query_matrix_f32"), k["nl_query_matrix_f32"]["isolated_median_ms"], g("d3_cn"), g("d3_energy"), g("d3_chain"), g("nl_query_matrix_f64"), g("ewald_real"), g("spline_spread"), g("fft_c2r"), g("pme_gather_finish")))'
for r in 1 2; do for g in 1536 1280 1024 768; do
  NVALCHEMIOPS_NL_TILED_GRID=$g timeout 300 python bench.py --processes 1 --steps 60 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$show" "grid $g" >> $O
done; done
cat $O

#!/bin/bash
# Round-4 A/B: LDS tile of the fp32 tiled list kernel (1024 / 768 / 640 candidates): a smaller tile leaves LDS for one block of the side stream's
# 9 A list per CU while the 40-Bohr fill's persistent blocks are resident
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp BENCH_CALIB=0
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
O=gpurun_out/r04_ab8.log; : > $O
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; g=lambda n: round(k[n]["avg_ms_timed_region"],3) if n in k else None; print(sys.argv[1], "step %.3f serial %.3f | list40 %s (iso %.3f) cn %s energy %s chain %s | list9 %s ewald %s spread %s c2r %s gather %s" % (d["ms_per_step"], d["stats"]["step_ms_median_serial_untimed"], g("nl_query_matrix_f32"), k["nl_query_matrix_f32"]["isolated_median_ms"], g("d3_cn"), g("d3_energy"), g("d3_chain"), g("nl_query_matrix_f64"), g("ewald_real"), g("spline_spread"), g("fft_c2r"), g("pme_gather_finish")))'
for r in 1 2; do for lib in base t768 t640; do
  cp $L/alt_$lib.so $L/libnvalchemiops_hip.so
  timeout 300 python bench.py --processes 1 --steps 60 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$show" "tile $lib" >> $O
done; done
cp $L/alt_base.so $L/libnvalchemiops_hip.so
cat $O

#!/bin/bash
# Round-4 drift experiment: the 40-Bohr list fill is 0.87 ms in the first processes on a box and 1.0 - 1.17 ms later.
# Use the box (part of the GPU test-suite, as the driver does before its bench), then alternate allocation variants, fresh process each:
#   base      torch caching allocator (two hipMallocs of 1.0 and 3.1 GB)
#   expand    PYTORCH_HIP_ALLOC_CONF=expandable_segments:True (virtual-memory segments of 2 MiB physical chunks)
#   arena16   both row buffers carved 2 MiB-aligned out of one 16 GiB block
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
O=gpurun_out/r04_drift.log; : > $O
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; c=d.get("calibration") or {}; print(sys.argv[1], "step %.3f serial %.3f | nl_f32 iso %.4f in-step %.4f | cn iso %.4f | fill %.0f copy %.0f GB/s" % (d["ms_per_step"], d["stats"]["step_ms_median_serial_untimed"], k["nl_query_matrix_f32"]["isolated_median_ms"], k["nl_query_matrix_f32"]["avg_ms_timed_region"], k["d3_cn"]["isolated_median_ms"], c.get("fill_GBps",0), c.get("copy_GBps",0)))'
run() { tag=$1; shift; env "$@" BENCH_CALIB_GIB=1 timeout 300 python bench.py --processes 1 --steps 40 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$show" "$tag" >> $O; }
run "fresh-1 base" A=1
run "fresh-2 base" A=1
python -m pytest tests/test_d3_gpu.py tests/test_c5_gpu.py -x -q -m gpu 2>&1 | tail -1 >> $O
for r in 1 2 3; do
  run "used-$r base" A=1
  run "used-$r expand" PYTORCH_HIP_ALLOC_CONF=expandable_segments:True
  run "used-$r arena16" BENCH_ARENA=16
done
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used\|total" | head -4 >> $O
cat $O

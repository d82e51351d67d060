#!/bin/bash
# Round-4 same-box A/B of the PME-branch changes (config 4 = 9 A fp64 list + PME): fp32 pre-filter of the fp64 search (alt_pre0 / alt_pre1)
# x FFT path (library plans / torch.fft); then the headline step under the three overlap schedules; then kernel stats of c4.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp BENCH_CALIB=0
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
O=gpurun_out/r04_ab1.log; : > $O
python -m pytest tests/test_nlist_gpu.py tests/test_pme_gpu.py tests/test_coulomb_gpu.py tests/test_autograd_gpu.py tests/test_c5_gpu.py tests/test_sweep_gpu.py tests/test_reference_scenarios_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $O
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(sys.argv[1], "step %.4f" % d["ms_per_step"], {n: round(v["avg_ms_timed_region"],4) for n,v in k.items()})'
for r in 1 2; do for lib in pre0 pre1; do for fft in torch own; do
  cp $L/alt_$lib.so $L/libnvalchemiops_hip.so
  NVALCHEMIOPS_PME_FFT=$fft timeout 200 python bench.py --workload c4 --steps 100 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$show" "c4 $lib fft=$fft" >> $O
done; done; done
cp $L/alt_pre1.so $L/libnvalchemiops_hip.so
for r in 1 2; do for o in 1 2 3; do
timeout 300 python bench.py --processes 1 --steps 40 --warmup 5 --cpu-sample 0 --overlap $o 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('headline overlap $o step %.3f median %.3f serial %.3f  e_pme %.6f' % (d['ms_per_step'], d['stats']['step_ms_median'], d['stats']['step_ms_median_serial_untimed'], d['energies']['e_pme']))" >> $O
done; done
cd /tmp; rm -rf /tmp/prof_c4b
rocprofv3 --kernel-trace --stats -d /tmp/prof_c4b -- python $R/bench.py --workload c4 --steps 50 --cpu-sample 0 > /tmp/prof_c4b.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_c4b -name "*.db" | head -1) $R/gpurun_out/r04_c4_kernel_stats_b.csv 2>&1 | tail -1
cat $R/$O

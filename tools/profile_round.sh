#!/bin/bash
# Collects what profiles/ holds for a round: bench JSON, rocprofv3 kernel-trace stats, PMC traffic + SQ counter passes.
# Run on the GPU box from the repo root:  bash tools/profile_round.sh <tag>     (outputs under gpurun_out/profile_<tag>/)
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
python $R/bench.py --processes 1 --overlap 0 --cpu-sample 0 --steps 50 > $OUT/bench_serial.json 2>> $OUT/bench.err
# kernel-trace statistics of the SAME command as bench.json (default flags) and of the serial variant
rm -rf /tmp/prof_stats /tmp/prof_stats_serial
rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -- python $R/bench.py --processes 1 --cpu-sample 0 --steps 50 > $OUT/bench_under_rocprof.json 2> /tmp/prof_stats.log
python $R/tools/rocpd_stats.py $(find /tmp/prof_stats -name "*.db" | head -1) $OUT/kernel_stats.csv 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d /tmp/prof_stats_serial -- python $R/bench.py --processes 1 --cpu-sample 0 --overlap 0 --steps 50 > /tmp/prof_stats_serial.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_stats_serial -name "*.db" | head -1) $OUT/kernel_stats_serial.csv 2>&1 | tail -2
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace --pmc $set -d /tmp/prof_$tag -- python $R/bench.py --processes 1 --steps 2 --warmup 1 --cpu-sample 0 --overlap 0 > /tmp/prof_$tag.log 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/prof_$tag -name "*.db" | head -1) $OUT/pmc_$tag.csv 2>&1 | tail -2
done
python $R/tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv $OUT/pmc_traffic.json
# config 4 (the electrostatics branch alone): kernel stats + traffic of the solve / spread / gather kernels
rm -rf /tmp/prof_c4
rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -- python $R/bench.py --workload c4 --steps 50 --cpu-sample 0 > $OUT/bench_c4_under_rocprof.json 2> /tmp/prof_c4.log
python $R/tools/rocpd_stats.py $(find /tmp/prof_c4 -name "*.db" | head -1) $OUT/kernel_stats_c4.csv 2>&1 | tail -2
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/prof_c4_$set
  rocprofv3 --kernel-trace --pmc $set -d /tmp/prof_c4_$set -- python $R/bench.py --workload c4 --steps 2 --warmup 1 --cpu-sample 0 > /tmp/prof_c4_$set.log 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/prof_c4_$set -name "*.db" | head -1) $OUT/pmc_c4_$set.csv 2>&1 | tail -2
done
python $R/tools/pmc_traffic.py $OUT/pmc_c4_FETCH_SIZE.csv $OUT/pmc_c4_WRITE_SIZE.csv $OUT/pmc_traffic_c4.json
python $R/tools/pmc_valu.py $OUT/pmc_SQ_INSTS_VALU.csv $OUT/pmc_valu.json
ls -la $OUT

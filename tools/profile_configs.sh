#!/bin/bash
# Companion of profile_round.sh for BASELINE configs 2 and 3: bench line, rocprofv3 kernel stats and PMC traffic (FETCH_SIZE / WRITE_SIZE in
# separate passes) per config.  Run on the GPU box from the repo root:  bash tools/profile_configs.sh <tag>   (outputs under gpurun_out/profile_<tag>/)
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in c2 c3; do
  rm -rf /tmp/prof_$W
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$W -- python $R/bench.py --workload $W --steps 50 --cpu-sample 0 > /tmp/prof_$W.json 2> /tmp/prof_$W.log
  python $R/tools/rocpd_stats.py $(find /tmp/prof_$W -name "*.db" | head -1) $OUT/kernel_stats_$W.csv 2>&1 | tail -2
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/prof_${W}_$set
    rocprofv3 --kernel-trace --pmc $set -d /tmp/prof_${W}_$set -- python $R/bench.py --workload $W --steps 2 --warmup 1 --cpu-sample 0 > /tmp/prof_${W}_$set.log 2>&1
    python $R/tools/rocpd_pmc.py $(find /tmp/prof_${W}_$set -name "*.db" | head -1) $OUT/pmc_${W}_$set.csv 2>&1 | tail -2
  done
  python $R/tools/pmc_traffic.py $OUT/pmc_${W}_FETCH_SIZE.csv $OUT/pmc_${W}_WRITE_SIZE.csv $OUT/pmc_traffic_$W.json
  # the bench line last, with the fresh traffic file in place so that its `traffic` fields are filled
  cp $OUT/pmc_traffic_$W.json $R/profiles/${TAG}_pmc_traffic_$W.json
  python $R/bench.py --workload $W > $OUT/bench_$W.json 2> $OUT/bench_$W.err
done
ls -la $OUT

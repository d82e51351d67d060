#!/bin/bash
# rocprofv3 kernel-trace statistics of an arbitrary command: tools/prof_cmd.sh <tag> <command...>   -> gpurun_out/<tag>_kernel_stats.csv
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -- "$@" > /tmp/prof_$TAG.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_$TAG -name "*.db" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
head -25 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150 | awk -F, '{printf "%-90s %6s %10s %10s\n", substr($1,1,90), $2, $4, $7}'

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s11
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python tools/probe/dft_vs_plan.py > $OUT/dft_vs_plan.log 2>&1; cat $OUT/dft_vs_plan.log | grep -v amdgpu

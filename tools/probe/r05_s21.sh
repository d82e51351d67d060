#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s21
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python tools/probe/hbm_region_probe.py 2 130 > $OUT/region_probe.log 2>&1; grep -v amdgpu $OUT/region_probe.log | tail -140

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s19
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python tools/probe/nl_vmm_probe.py > $OUT/vmm_probe.log 2>&1; grep -v amdgpu $OUT/vmm_probe.log | tail -40

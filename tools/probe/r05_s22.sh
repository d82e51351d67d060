#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s22
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 800 python tools/probe/vmm_stream_probe.py > $OUT/vmm_stream.log 2>&1; grep -v amdgpu $OUT/vmm_stream.log | tail -20

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s20
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python tools/probe/nl_buffer_stream_probe.py 10 > $OUT/stream_probe.log 2>&1; grep -v amdgpu $OUT/stream_probe.log | tail -14

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s15
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_packed_companion_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log

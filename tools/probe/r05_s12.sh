#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s12
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pme_gpu.py -m gpu -q -x -k "dense_dft or failing_fft or bounded or guarded" > $OUT/pytest_dft.log 2>&1; echo "pytest dft rc=$?"; tail -5 $OUT/pytest_dft.log
timeout 600 python tools/probe/dft_vs_plan.py > $OUT/dft_vs_plan.log 2>&1; cat $OUT/dft_vs_plan.log | grep -v amdgpu

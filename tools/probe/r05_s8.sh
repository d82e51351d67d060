#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s8
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pme_gpu.py -m gpu -q -x -k "guarded_fft" > $OUT/pytest_ops.log 2>&1; echo "pytest ops rc=$?"; tail -15 $OUT/pytest_ops.log
timeout 1500 python -m pytest tests/test_autograd_gpu.py tests/test_compile_gpu.py tests/test_reference_scenarios_gpu.py tests/test_pme_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 $OUT/pytest.log

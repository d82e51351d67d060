#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s6
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pme_gpu.py -m gpu -q -x -k "dense_dft or failing_fft or bounded" > $OUT/pytest_dft.log 2>&1; echo "pytest dft rc=$?"; tail -5 $OUT/pytest_dft.log
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -14 $OUT/pytest.log
grep -n "failed its impulse\|could not be self-tested" $OUT/pytest.log | head

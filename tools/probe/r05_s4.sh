#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s4
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python tools/probe/nl_buffer_shop.py 8 > $OUT/buffer_shop_1.log 2>&1; cat $OUT/buffer_shop_1.log | grep round
timeout 300 python tools/probe/nl_buffer_shop.py 8 > $OUT/buffer_shop_2.log 2>&1; cat $OUT/buffer_shop_2.log | grep round
timeout 600 python -m pytest tests/test_pme_gpu.py -m gpu -q -x -k "failing_fft or bounded" > $OUT/pytest_fft.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_fft.log

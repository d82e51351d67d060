#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s24
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
echo "# same-box A/B: shifts rows zero-filled up front by the owner wave, un-shifted hits store no shift (NL_PREZERO_SHIFTS=1) vs the shipped per-hit 12-byte stores; tools/probe/nl_buffer_shop.py 8 (fill with companion, per buffer set)" | tee $OUT/ab_prezero.log
for v in base prezero prezero_plain prezero prezero_plain; do
  cp $L/alt_$v.so $L/libnvalchemiops_hip.so
  echo "== $v" | tee -a $OUT/ab_prezero.log
  timeout 300 python tools/probe/nl_buffer_shop.py 8 2>/dev/null | grep "round 1" | sed 's/matrix 0x[0-9a-f]* shifts 0x[0-9a-f]* companion 0x[0-9a-f]*//' | tee -a $OUT/ab_prezero.log
done
cp $L/alt_prezero_plain.so $L/libnvalchemiops_hip.so
timeout 900 python -m pytest tests/test_nlist_gpu.py tests/test_packed_companion_gpu.py tests/test_reference_scenarios_gpu.py tests/test_sweep_gpu.py -m gpu -q -x > $OUT/pytest_prezero.log 2>&1; echo "pytest prezero rc=$?"; tail -3 $OUT/pytest_prezero.log
cp $L/alt_base.so $L/libnvalchemiops_hip.so

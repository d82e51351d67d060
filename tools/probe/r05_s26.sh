#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s26
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
echo "# same-box A/B: centre atoms per sweep over a cell's candidate tiles (NL_CCHUNK; 64 = shipped, 8 = one pass of 4 waves x 2 centres: every row is written in one go, tiles re-staged per pass); tools/probe/nl_buffer_shop.py 8, round 1 medians" | tee $OUT/ab_cchunk.log
for v in c64 c8 c16 c32 c64 c8; do
  cp $L/alt_$v.so $L/libnvalchemiops_hip.so
  echo "== $v" | tee -a $OUT/ab_cchunk.log
  timeout 300 python tools/probe/nl_buffer_shop.py 8 2>/dev/null | grep "round 1" | sed 's/matrix 0x[0-9a-f]* shifts 0x[0-9a-f]* companion 0x[0-9a-f]*//' | tee -a $OUT/ab_cchunk.log
done
cp $L/alt_c64.so $L/libnvalchemiops_hip.so

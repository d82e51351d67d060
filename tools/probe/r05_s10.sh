#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s10
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
echo "# same-box A/B of the D3 chain pass's pipeline constants (words / records in flight: 2/1 = the shipped form), packed companion as input (tools/probe/packed_ab.py --reps 20; medians, ms)" | tee $OUT/ab_chain_pipeline.log
for rep in 1 2; do
for v in ch21 ch31 ch32 ch42; do
  cp $L/alt_$v.so $L/libnvalchemiops_hip.so
  timeout 300 python tools/probe/packed_ab.py --reps 20 > $OUT/ab_$v.json 2>> $OUT/err.log
  python -c "
import json; d=json.load(open('$OUT/ab_$v.json')); r=d['round1_companion1']; r0=d['round1_companion0']
print('%-8s' % '$v', 'companion on: branch', round(r['branch_ms_wall'],3), {k: round(v['median_ms'],4) for k,v in r.items() if isinstance(v, dict)}, '| off: chain', round(r0['d3_chain']['median_ms'],4), 'identical', d['round1_bit_identical'])" | tee -a $OUT/ab_chain_pipeline.log
done
done
cp $L/alt_ch21.so $L/libnvalchemiops_hip.so
timeout 600 python -m pytest tests/test_d3_gpu.py tests/test_packed_companion_gpu.py -m gpu -q -x > $OUT/pytest_d3.log 2>&1; echo "pytest d3 rc=$?"; tail -3 $OUT/pytest_d3.log

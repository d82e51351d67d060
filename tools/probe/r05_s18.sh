#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s18
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python tools/probe/nl_tlb_probe.py 4 > $OUT/tlb_probe_times.log 2>&1; grep -v amdgpu $OUT/tlb_probe_times.log | tail -20
cd /tmp; rm -rf /tmp/prof_tlb
REPS=1 timeout 600 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum -d /tmp/prof_tlb -- python $R/tools/probe/nl_tlb_probe.py 4 > $OUT/tlb_probe_pmc_run.log 2>&1
python $R/tools/probe/nl_tlb_probe.py --dump $(find /tmp/prof_tlb -name "*.db" | head -1) > $OUT/tlb_probe_pmc.log 2>&1; cat $OUT/tlb_probe_pmc.log | head -60
grep -v amdgpu $OUT/tlb_probe_pmc_run.log | grep round | head -20

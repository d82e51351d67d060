#!/bin/bash
# PMC passes over the three modes of the 40-Bohr list kernel (tools/nl_modes.py): where do the waves wait when the stores are on?
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_nl; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python $R/tools/nl_modes.py 2>&1 | grep -v "^/opt" > $OUT/modes.log
i=0
for set in "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  for mode in count "matrix, indices" "matrix + shifts"; do
    tag=$(echo $mode | tr -d ' ,+' )
    rm -rf /tmp/pn; rocprofv3 --kernel-trace --pmc $set -d /tmp/pn -- python $R/tools/nl_modes.py "$mode" > /tmp/pn.log 2>&1
    python $R/tools/rocpd_pmc.py $(find /tmp/pn -name "*.db" | head -1) $OUT/set${i}_$tag.csv > /dev/null 2>&1
  done
done
python - <<PY
import csv, glob, os
for f in sorted(glob.glob("$OUT/set*.csv")):
    rows=[r for r in csv.DictReader(open(f)) if "nl_query_tiled" in r["kernel"]]
    print(os.path.basename(f), {r["counter"]: "%.3g" % float(r["per_launch"]) for r in rows})
PY

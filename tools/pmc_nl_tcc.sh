#!/bin/bash
# L2 -> memory write-request counters of the 40-Bohr matrix fill, with the calibration fill kernel of the same run as the reference.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_nl; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_TAG_STALL_sum" "TCC_WRITE_sum TCC_WRITEBACK_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pn; timeout 100 rocprofv3 --kernel-trace --pmc $set -d /tmp/pn -- python $R/tools/nl_modes.py "matrix + shifts" > /tmp/pn.log 2>&1 || { echo "pass $tag failed/timeout"; tail -3 /tmp/pn.log; continue; }
  python $R/tools/rocpd_pmc.py $(find /tmp/pn -name "*.db" | head -1) $OUT/tcc_$tag.csv > /dev/null 2>&1
done
python - <<PY
import csv, glob, os
for f in sorted(glob.glob("$OUT/tcc_*.csv")):
    for r in csv.DictReader(open(f)):
        if "nl_query_tiled" in r["kernel"] or "cal_fill" in r["kernel"]:
            print(os.path.basename(f), r["kernel"][:60].split("(")[-1] if False else ("nl_query_tiled" if "nl_query" in r["kernel"] else "cal_fill"), r["counter"], "%.4g" % float(r["per_launch"]), "avg_ns", r["avg_ns"])
PY

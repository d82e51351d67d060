#!/bin/bash
# PMC passes over config 4 for its two largest kernels (fp64 9 A tiled search, real-space erfc sum): where do the cycles go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/c4_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/prof_$tag
  BENCH_CALIB=0 rocprofv3 --kernel-trace --pmc $set -d /tmp/prof_$tag -- python $R/bench.py --workload c4 --processes 1 --steps 3 --warmup 1 --cpu-sample 0 > /tmp/prof_$tag.log 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/prof_$tag -name "*.db" | head -1) $OUT/pmc_$tag.csv 2>&1 | tail -2
done
python - "$OUT" <<PYEOF
import csv,sys,collections
out=sys.argv[1]
rows=[]
for f in ("pmc_SQ_WAVE_CYCLES.csv","pmc_SQ_INSTS_VALU.csv"):
    rows+=list(csv.DictReader(open(out+"/"+f)))
t=collections.defaultdict(dict)
for r in rows:
    for key in ("nl_query_tiled", "ewald_real_kernel", "pme_gather_box", "spread_box_kernel"):
        if key in r["kernel"]:
            t[key][r["counter"]]=float(r["per_launch"]); t[key]["us"]=float(r["avg_ns"])/1e3
for k,v in t.items():
    print(k, {a:(round(b/1e6,3) if a!="us" else round(b,1)) for a,b in sorted(v.items())})
PYEOF

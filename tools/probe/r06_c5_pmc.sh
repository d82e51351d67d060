#!/bin/bash
# SQ counters of config 5's kernels (128 x 2000-atom periodic boxes of 32 A, cutoff 21 A): what bounds the small-box search?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/c5_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/prof_$tag
  BENCH_CALIB=0 rocprofv3 --kernel-trace --pmc $set -d /tmp/prof_$tag -- python $R/bench.py --workload c5 --processes 1 --steps 2 --warmup 1 --cpu-sample 0 > /tmp/prof_$tag.log 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/prof_$tag -name "*.db" | head -1) $OUT/pmc_$tag.csv 2>&1 | tail -1
done
python - "$OUT" <<PYEOF
import csv,sys,collections
out=sys.argv[1]
rows=[]
for f in ("pmc_SQ_WAVE_CYCLES.csv","pmc_SQ_INSTS_VALU.csv","pmc_FETCH_SIZE.csv","pmc_WRITE_SIZE.csv"):
    rows+=list(csv.DictReader(open(out+"/"+f)))
t=collections.defaultdict(dict)
for r in rows:
    for key in ("nl_query_tiled_kernelIf", "nl_query_tiled_kernelId", "d3_energy_kernel", "d3_chain_kernel"):
        if key in r["kernel"]:
            t[key][r["counter"]]=float(r["per_launch"]); t[key]["us"]=float(r["avg_ns"])/1e3
for k,v in t.items():
    print(k, {a:(round(b/1e6,3) if a!="us" else round(b,1)) for a,b in sorted(v.items())})
PYEOF

#!/bin/bash
# A/B of the D3 chain records (NVALCHEMIOPS_D3_CHAIN_RECORDS=0 -> two gathers per neighbour) on the headline and on config 3, same box.
OUT=${1:-gpurun_out/chain_records_ab.log}; ROUNDS=${2:-2}
: > $OUT
py() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernels", {})
def g(n):
    r = k.get(n) or {}
    v = r.get("isolated_median_ms") or r.get("median_ms_timed_region")
    return round(v, 4) if v else None
print("step %.4f | energy %s chain %s" % (d["ms_per_step"], g("d3_energy"), g("d3_chain")))
PY
}
for r in $(seq $ROUNDS); do
  for v in 0 1; do
    NVALCHEMIOPS_D3_CHAIN_RECORDS=$v timeout 400 python bench.py --processes 1 --steps 50 --warmup 10 --cpu-sample 0 > /tmp/ab.json 2>/tmp/ab.err
    echo "headline records $v: $(py /tmp/ab.json)" | tee -a $OUT
    NVALCHEMIOPS_D3_CHAIN_RECORDS=$v timeout 300 python bench.py --workload c3 --steps 200 --warmup 20 --cpu-sample 0 > /tmp/ab.json 2>/tmp/ab.err
    echo "c3 records $v: $(py /tmp/ab.json)" | tee -a $OUT
  done
done

#!/bin/bash
# A/B of the trusted full-list form of the real-space sum (mi_ewald_real_listed) on config 4 and the headline:
#   trust 0 = per-entry symmetry checksums, trust 1 = counts + sampled mirror look-up (stride 64 / 0 = no sample / 1 = every row).
# Usage: bash tools/probe/ewald_trust_ab.sh <out.log> [rounds]
OUT=${1:-gpurun_out/ewald_trust_ab.log}; ROUNDS=${2:-3}
: > $OUT
py() { python - "$@" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {r["name"]: r for r in d.get("kernels", [])} if isinstance(d.get("kernels"), list) else d.get("kernels", {})
ew = k.get("ewald_real", {})
print(round(d["ms_per_step"], 4), round(ew.get("median_ms_timed_region", 0), 4), d["config"].get("e_pme", ""))
PY
}
for r in $(seq $ROUNDS); do
  for cfg in "0 64" "1 64" "1 0" "1 1"; do
    set -- $cfg
    NVALCHEMIOPS_EWALD_TRUST_FULL_LISTS=$1 NVALCHEMIOPS_NL_PACKED_VERIFY=$2 timeout 300 python bench.py --workload c4 --steps 200 --warmup 20 --cpu-sample 0 > /tmp/ab.json 2>/tmp/ab.err
    echo "c4 trust $1 verify $2: $(py /tmp/ab.json)" | tee -a $OUT
  done
done
for r in 1 2; do
  for t in 0 1; do
    NVALCHEMIOPS_EWALD_TRUST_FULL_LISTS=$t timeout 400 python bench.py --processes 1 --steps 50 --warmup 10 --cpu-sample 0 > /tmp/ab.json 2>/tmp/ab.err
    echo "headline trust $t: $(py /tmp/ab.json)" | tee -a $OUT
  done
done

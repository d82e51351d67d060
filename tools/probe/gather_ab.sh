#!/bin/bash
# A/B of the gather epilogue on config 4: NVALCHEMIOPS_GATHER=atom (per-atom kernel) | box (one tile per block) | default (super-tiles where eligible)
OUT=${1:-gpurun_out/gather_ab.log}; ROUNDS=${2:-3}
: > $OUT
for r in $(seq $ROUNDS); do for g in atom box super; do
  if [ $g = super ]; then unset NVALCHEMIOPS_GATHER; else export NVALCHEMIOPS_GATHER=$g; fi
  timeout 300 python bench.py --workload c4 --steps 200 --warmup 20 --cpu-sample 0 > /tmp/ab.json 2>/tmp/ab.err
  python - <<'PY' | tee -a $OUT
import json, os
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1]); k = d["kernels"]
print(os.environ.get("NVALCHEMIOPS_GATHER", "super"), round(d["ms_per_step"], 4), round(k["pme_gather_finish"]["median_ms_timed_region"], 4), d["config"]["e_pme"])
PY
done; done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s14
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
for w in c2 c3 c4; do timeout 300 python bench.py --workload $w > $OUT/bench_$w.json 2>> $OUT/err.log; echo "$w rc=$?"; done
timeout 400 python bench.py --workload c5 --steps 20 --warmup 3 > $OUT/bench_c5.json 2>> $OUT/err.log; echo "c5 rc=$?"
BENCH_GRAPH=1 timeout 300 python bench.py --workload c4 --cpu-sample 0 > $OUT/bench_c4_graph.json 2>> $OUT/err.log
python - <<'PY'
import json
for w in ("c2","c3","c4","c5","c4_graph"):
    try:
        d=json.load(open(f"gpurun_out/r05_s14/bench_{w}.json"))
        print(w, round(d["ms_per_step"],4), d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d.get("hip_graph"), (d["config"].get("list_buffers") or d["config"].get("d3_list_buffers")), {k: round(v["avg_ms_timed_region"],4) for k,v in d["kernels"].items()})
    except Exception as e: print(w, "failed", e)
PY

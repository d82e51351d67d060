#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s5
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
tail -3 $OUT/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_s5/bench_default.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print("processes", [(round(p["ms_per_step"], 3), p["list_fill_40bohr_in_step_ms"]) for p in d["processes"]["each"]])
print("buffers", d["config"].get("d3_list_buffers"))
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "achieved", "launch_ms", "frac_isolated", "moved_GBps")})
print("step_traffic", {k: v for k, v in d.get("step_traffic", {}).items() if k != "note"})
p = d.get("parity") or {}
print("parity", {k: v for k, v in p.items() if k not in ("bar", "oracle")})
print({k: (round(v['avg_ms_timed_region'],3), v['isolated_median_ms']) for k,v in d['kernels'].items()})
print({k: (round(v["ms"], 4), v["roofline"]["kernel"], round(v["roofline"]["frac"], 3)) for k, v in d["configs"].items() if "ms" in v})
PY

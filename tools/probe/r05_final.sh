#!/bin/bash
# Round 5 closing session: full GPU suite, smoke, then the round's profile collection (which starts with the default bench command).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_final
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
bash tools/profile_round.sh r05 > $OUT/profile.log 2>&1; echo "profile rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/profile_r05/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print("processes", [(round(p["ms_per_step"], 3), p["list_fill_40bohr_in_step_ms"]) for p in d["processes"]["each"]])
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "achieved", "launch_ms", "frac_isolated", "moved_GBps", "moved_frac_of_box_fill", "traffic")})
print("step_traffic", {k: v for k, v in d["step_traffic"].items() if k != "note"})
print({k: (round(v['avg_ms_timed_region'],3), v['isolated_median_ms'], v.get('traffic_bytes')) for k,v in d['kernels'].items()})
print({k: (round(v["ms"], 4), v["roofline"]["kernel"], round(v["roofline"]["frac"], 3)) for k, v in d["configs"].items() if "ms" in v})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["full_size"].get("all_cores", {}).get("value"))
PY

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s25
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_s25/bench.json"))
print(d["ms_per_step"], [(round(p["ms_per_step"], 3), p["list_fill_40bohr_in_step_ms"]) for p in d["processes"]["each"]])
print({k: d["roofline"].get(k) for k in ("kernel", "frac", "launch_ms", "frac_isolated", "moved_frac_of_box_fill")})
print(d["config"]["d3_list_buffers"]["trial_ms"])
print({k: (round(v['avg_ms_timed_region'],3), v['isolated_median_ms']) for k,v in d['kernels'].items()})
print({k: (round(v["ms"], 4), v["kernels_ms"]) for k, v in d["configs"].items() if "ms" in v})
PY

#!/bin/bash
# Round 5, GPU session 2: full parity suite, default bench line (with parity + configs), overlap schedules, NT companion stores.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s2
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_s2/bench_default.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print("processes", [(round(p["ms_per_step"], 3), p["list_fill_40bohr_in_step_ms"]) for p in d["processes"]["each"]])
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "achieved", "launch_ms", "frac_isolated")})
print("step_traffic", d.get("step_traffic"))
print("parity", d.get("parity"))
print("configs", json.dumps(d.get("configs"), indent=0)[:3000])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:200])
PY
for ov in 2 3 0; do
  timeout 300 python bench.py --processes 1 --steps 100 --cpu-sample 0 --overlap $ov > $OUT/bench_overlap$ov.json 2>> $OUT/bench.err
  python -c "
import json; d=json.load(open('$OUT/bench_overlap$ov.json')); print('overlap $ov', round(d['ms_per_step'],3), {k: round(v['avg_ms_timed_region'],3) for k,v in d['kernels'].items()})"
done
for v in base pknt base pknt; do
  cp $L/alt_$v.so $L/libnvalchemiops_hip.so
  timeout 300 python tools/probe/packed_ab.py --reps 20 > $OUT/packed_ab_$v.json 2>> $OUT/bench.err
  python -c "
import json; d=json.load(open('$OUT/packed_ab_$v.json')); r=d['round1_companion1']; print('$v', round(r['branch_ms_wall'],3), {k: v['median_ms'] for k,v in r.items() if isinstance(v, dict)})"
done
cp $L/alt_base.so $L/libnvalchemiops_hip.so

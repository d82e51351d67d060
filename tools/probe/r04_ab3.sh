#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
python -m pytest tests/test_autograd_gpu.py tests/test_compile_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_ab3.log
python bench.py --workload pme-train > gpurun_out/r04_bench_pme_train.json 2>> gpurun_out/r04_ab3.log
python - <<'PY' >> gpurun_out/r04_ab3.log
import json
d=json.load(open("gpurun_out/r04_bench_pme_train.json"))
for r in d["rows"]: print(r["spline_order"], r["loss"], "infer %.3f fwd %.3f bwd %.3f" % (r["forward_inference_ms"], r["forward_ms"], r["backward_ms"]))
PY
python bench.py > gpurun_out/r04_bench_default_b.json 2>> gpurun_out/r04_ab3.log
python - <<'PY' >> gpurun_out/r04_ab3.log
import json
d=json.load(open("gpurun_out/r04_bench_default_b.json"))
print("headline", round(d["ms_per_step"],4), d["processes"]["reported"], [ (round(p["ms_per_step"],3), round(p["list_fill_40bohr_isolated_median_ms"],3)) for p in d["processes"]["each"]], "cpu", round(d["cpu_baseline"]["value"]))
PY
cat gpurun_out/r04_ab3.log

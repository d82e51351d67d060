#!/bin/bash
# Round 5, GPU session 1: the three calls round 4 could not make + the packed-companion A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_s1
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
# 1. rocFFT drift probe: explicit embeds (shipped) and NULL embeds
timeout 300 python tools/probe/rocfft_drift_repro.py > $OUT/drift_explicit.log 2>&1; echo "drift explicit rc=$?"
NVALCHEMIOPS_FFT_LAYOUT=default timeout 300 python tools/probe/rocfft_drift_repro.py > $OUT/drift_default.log 2>&1; echo "drift default rc=$?"
grep -c "library plan [0-9.]*e-0[0-9] " $OUT/drift_explicit.log $OUT/drift_default.log
grep "(32, 8, 16)" $OUT/drift_explicit.log | head -4; grep "(32, 8, 16)" $OUT/drift_default.log | head -4
# 2. new tests of this round
timeout 900 python -m pytest tests/test_packed_companion_gpu.py tests/test_pme_gpu.py -q -x -m gpu -k "companion or failing_fft or bounded or fused_mesh_solve" > $OUT/pytest_new.log 2>&1; echo "pytest new rc=$?"
tail -15 $OUT/pytest_new.log
# 3. the experimental solve-under-autograd path
NVALCHEMIOPS_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_autograd_gpu.py -q -m gpu -k mesh_solve > $OUT/pytest_solve_autograd.log 2>&1; echo "pytest solve-autograd rc=$?"
tail -8 $OUT/pytest_solve_autograd.log
timeout 300 python bench.py --workload pme-train > $OUT/pme_train_plans.json 2> $OUT/pme_train.err; echo "pme-train plans rc=$?"
NVALCHEMIOPS_PME_SOLVE_AUTOGRAD=1 timeout 300 python bench.py --workload pme-train > $OUT/pme_train_solve.json 2>> $OUT/pme_train.err; echo "pme-train solve rc=$?"
head -c 1500 $OUT/pme_train_plans.json; echo; head -c 1500 $OUT/pme_train_solve.json; echo
# 4. packed companion: same-process A/B, then the headline step
timeout 600 python tools/probe/packed_ab.py > $OUT/packed_ab.json 2> $OUT/packed_ab.err; echo "packed_ab rc=$?"
cat $OUT/packed_ab.json | head -80
NVALCHEMIOPS_NL_PACKED=0 timeout 400 python bench.py --processes 1 --steps 100 --cpu-sample 0 > $OUT/bench_nopack.json 2> $OUT/bench.err; echo "bench nopack rc=$?"
timeout 400 python bench.py --processes 1 --steps 100 --cpu-sample 0 > $OUT/bench_pack.json 2>> $OUT/bench.err; echo "bench pack rc=$?"
python - <<'PY'
import json
for f in ("bench_nopack", "bench_pack"):
    try:
        d = json.load(open(f"gpurun_out/r05_s1/{f}.json"))
        print(f, d["ms_per_step"], {k: round(v["avg_ms_timed_region"], 3) for k, v in d.get("kernels", {}).items()})
    except Exception as e:
        print(f, "failed", e)
PY

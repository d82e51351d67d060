# A/B: the headline step with the PME k-space step on the fused mesh solve (whole-CU LDS blocks) vs on hipFFT plans + pme_convolve (small blocks
# that can share CUs with the dispersion passes), two-stream schedule, same box.
for s in auto 0 auto 0; do
  NVALCHEMIOPS_PME_MESH_SOLVE=$s timeout 300 python bench.py --processes 1 --cpu-sample 0 --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('solve=$s step', round(d['ms_per_step'],3), {n: round(v['avg_ms_timed_region'],3) for n,v in k.items()})"
done

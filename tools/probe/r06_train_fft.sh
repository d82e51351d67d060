# training step of particle_mesh_ewald on the 100k box: backward on the in-LDS transforms (default) vs on guarded hipFFT plans (NVALCHEMIOPS_FFT_LDS=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in 1 0 1 0; do
  NVALCHEMIOPS_FFT_LDS=$v python $R/bench.py --workload pme-train 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1])
for row in r['rows']: print('lds=$v', 'order', row['spline_order'], row['loss'], 'fwd', round(row['forward_ms'],4), 'bwd', round(row['backward_ms'],4), {k:v for k,v in row['kernel_median_ms'].items() if 'fft' in k})"
done

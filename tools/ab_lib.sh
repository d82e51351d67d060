L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
for r in 1 2 3; do for v in before after; do cp $L/alt_$v.so $L/libnvalchemiops_hip.so; timeout 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), d['kernel_ms_isolated'].get('d3_energy'), d['kernel_ms'].get('d3_energy'))"; done; done
cp $L/alt_after.so $L/libnvalchemiops_hip.so
timeout 300 python -m pytest tests/test_d3_gpu.py -q -x 2>&1 | tail -1

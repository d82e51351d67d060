# same-box A/B of two builds: lib/alt_before.so and lib/alt_after.so are copied over the library in turn (bench.py --processes 1, 3 rounds)
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
for r in 1 2 3; do for v in before after; do cp $L/alt_$v.so $L/libnvalchemiops_hip.so; timeout 200 python bench.py --processes 1 --steps 20 --warmup 5 --cpu-sample 0 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_isolated']; print('$v', round(d['ms_per_step'],3), k.get('d3_cn'), k.get('d3_energy'), k.get('d3_chain'))"; done; done
cp $L/alt_after.so $L/libnvalchemiops_hip.so

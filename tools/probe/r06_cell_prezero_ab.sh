# per-cell decision whether the 40-Bohr search pre-zeroes its shift rows (lib/alt_base.so = always, alt_cellpz.so = only where >= 50 % of the cell's candidates are un-shifted)
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
for r in 1 2; do for v in base cellpz; do cp $L/alt_$v.so $L/libnvalchemiops_hip.so
for w in c5 c3; do python bench.py --workload $w --processes 1 --steps 20 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']; print('$v $w', round(r['ms_per_step'],4), {n:round(v['avg_ms_timed_region'],3) for n,v in k.items() if n.startswith('nl_query') or n.startswith('d3_')})"; done
python bench.py --processes 1 --steps 60 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']; print('$v headline', round(r['ms_per_step'],4), round(k['nl_query_matrix_f32']['isolated_median_ms'],4))"
done; done
cp $L/alt_cellpz.so $L/libnvalchemiops_hip.so

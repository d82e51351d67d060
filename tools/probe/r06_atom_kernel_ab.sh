L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
for r in 1 2; do for v in old new; do cp $L/alt_$v.so $L/libnvalchemiops_hip.so
python bench.py --workload ref-nlist 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('$v ref-nlist', [round(x['median_ms'],4) for x in r['rows']])"
done; done
cp $L/alt_new.so $L/libnvalchemiops_hip.so
